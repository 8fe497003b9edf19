// mlp_bf16.hip -- the split-bf16 ("bf16x3") INFERENCE forward of FlexibleNeRFModel (nerf/models.py:233-258) for plans created
// with NERFHIP_PRECISION_BF16X3 (include/nerfhip.h).  Separate from, and never a substitute for, the fp32 kernels of
// mlp16.hip: those multiply exactly in fp32 (157 TFLOP/s peak); this one multiplies on v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s
// peak) with every operand split into two bf16 pieces and three MFMAs per product block,
//     x . w  ~  xh.wh + xh.wl + xl.wh          (fp32 accumulation; the dropped xl.wl term is ~2^-16 relative),
// so its own roofline is 2.5 PF / 3 fp32-equivalent FLOP/s and its products carry ~2^-16 instead of 2^-24 relative error.
//
// Shape (profiles/r03_split_bf16_mock.txt measured the loop before this kernel existed): a wave owns 32 samples; its
// activations live in registers as the B operands of the next layer -- the accumulator registers c = 8*(kb&1) + e of output
// tile kb>>1, converted to (hi, lo) pieces, ARE k-block kb of the next layer when the packed weights use the matching unit
// permutation nhb_unit (nh_plan.h) -- so nothing but the weights moves through memory between layers.  Weights stream
// L2 -> LDS by LDS-DMA in chunks of whole k-blocks, double buffered, one barrier per chunk; a layer's bias block rides in
// front of its first chunk.  256-wide nets: 4-wave workgroups, one wave per SIMD (144 accumulator + 128 operand registers);
// 128-wide nets: two such workgroups per CU.  No activation stash: inference only.
#include "nh_device.h"
#include "nh_mlp.h"

// ---- the piece format of this translation unit ------------------------------------------------------------------------------------
// The file is compiled twice: as is (bf16 pieces: the bf16x3 plans) and through mlp_f16.hip with NHB_F16 defined (IEEE fp16 pieces:
// the f16x3 plans, include/nerfhip.h).  Same loops, same images, same register layouts; what differs is the conversion, the MFMA,
// and -- fp16's 5-bit exponent, [2^-14, 2^16) with the low piece 2^-12 below the value -- block-floating-point bookkeeping, all in
// exact powers of two:
//   * packed weights (and biases) carry NHB_WS = 2^8: a weight's low piece is a normal fp16 number down to |w| = 2^-10;
//   * forward: every SAMPLE carries the exponent S of its current activations -- the operand pieces are those of h * 2^S, with S
//     chosen by the epilogue that produced h so that the sample's largest activation lands in [2^13, 2^14): no activation range is
//     out of reach and small activations keep both pieces (fp32-like relative precision).  A gemm starts its accumulators at
//     bias * 2^S, so they hold WS * 2^S * pre-activation; the epilogue finds the sample's maximum (one cross-half exchange), turns
//     it into the next S and applies the difference as the one multiply the weight scale costs anyway.  Encodings carry an exponent
//     of their own (from the sample's coordinates); a layer that reads both caps S at it and rescales the encoding pieces on the
//     fly (v_pk_mul_f16).  Raw outputs and stash rows are brought back to plain fp32 values on the way out;
//   * data gradient: the same bookkeeping -- the chain of a sample starts at its d(raw output) moved to [2^13, 2^14) and every
//     layer's d(pre-activation) is renormalised per sample (whatever the transposed layers amplify or damp); the images it stores
//     are plain fp32 values;
//   * what a sample's stored row holds is bounded by its exponent (pieces below 2^14, times 2^-s): the producers record, per
//     region, the largest such bound over their samples (`rmax` words: 256 + 14 - s; a wave maximum per gemm into the wave's LDS
//     slots, one atomicMax per wave and region at the end of the kernel), from which the fp16 weight-gradient kernel
//     (wgrad_f16.hip) takes the power of two it splits that region's values at.
#ifdef NHB_F16
typedef nh_f16 nh_pc;
typedef nh_f16x8 nh_pcx8;
#define nh_to_pc nh_to_f16
#define nh_from_pc nh_from_f16
#define nh_mfma_pc nh_mfma_f16
#define NHB_FMT "f16"
#define NHB_FN(stem) stem##_f16
#define NHB_KERNEL(stem) stem##_f16x3
constexpr float NHB_WS = NHB_F16_WSCALE;
constexpr int NHB_WS_LOG2 = 8;
constexpr bool NHB_IS_F16 = true;
#else
typedef nh_bf16 nh_pc;
typedef nh_bf16x8 nh_pcx8;
#define nh_to_pc nh_to_bf16
#define nh_from_pc nh_from_bf16
#define nh_mfma_pc nh_mfma_bf16
#define NHB_FMT "bf16"
#define NHB_FN(stem) stem##_bf16
#define NHB_KERNEL(stem) stem##_bf16x3
constexpr float NHB_WS = 1.0f;
constexpr int NHB_WS_LOG2 = 0;
constexpr bool NHB_IS_F16 = false;
#endif
static_assert(NHB_WS == (float)(1 << NHB_WS_LOG2), "weight scale");
constexpr float NHB_INV_WS = 1.0f / NHB_WS;
constexpr int NHB_TARGET_LOG2 = 13;  // a sample's largest operand value lands in [2^13, 2^14)
constexpr int NHB_NO_CAP = 100;
// the exponent given to a sample whose values are all zero (a dead layer; a padding sample of the last group): high enough that
// its bound (2^(14 - 60)) never sets a region's scale, low enough that a bias times 2^60 stays a float
constexpr int NHB_ZERO_EXP = 60;
// exponent for pieces of values whose largest magnitude has bit pattern `mb`, given that it currently carries 2^base_e
NH_DEVICE int exp_for(unsigned mb, int base_e) {
    return ((mb >> 23) & 255u) == 0u ? NHB_ZERO_EXP : base_e + nh_shift_to(mb, NHB_TARGET_LOG2);
}
constexpr int NHB_RM_LDS = 4 * NH_RMAX_WORDS * 4;  // LDS bytes behind the chunk buffers: a workgroup's four waves' region slots
// a raw network output from an accumulator that holds WS * 2^s * value
NH_DEVICE float nhb_raw(float acc, int s) { return NHB_IS_F16 ? acc * nh_pow2i(-NHB_WS_LOG2 - s) : acc; }

namespace {

#ifndef NHB_PACK_ONLY  // (mlp_f16.hip: the fp16-piece plans run on the kernels of mlp_f16w.hip; this file gives them the image packer)

#ifndef NHB_DMA_EVERY  // (A/B builds only) a wave issues one 1-KiB piece of the next chunk every so many blocks; 0: all at once
#define NHB_DMA_EVERY 1
#endif
#ifndef NHB_PREFETCH  // (A/B builds only) blocks of weight pieces in flight LDS -> registers ahead of the MFMAs
#define NHB_PREFETCH 2
#endif

template <int W>
struct BShape {
    static constexpr int TH = W / 32, KBH = W / 16, CHUNK = nhb_chunk_bytes(W), BUF = CHUNK + 2048, LDS_BYTES = 2 * BUF;
    static constexpr int WAVES_PER_SIMD = W >= 256 ? 1 : 2;
};

struct BCtx {
    char* lds;
    unsigned lds_addr;
    NhDmaSrc dma;
    int buf, lane, wave, h;
    unsigned* wrm;  // fp16 level-4 plans: this wave's NH_RMAX_WORDS region slots in LDS (behind the chunk buffers), else NULL
};
// the rows a gemm stored for region `ridx` came from pieces below 2^(NHB_TARGET + 1) at per-sample exponent s: note the bound
NH_DEVICE void note_region(const BCtx& cx, int ridx, int s) {
    const int e = 256 + 14 - s;
    const unsigned wm = nh_wave_max_u32((unsigned)(e < 1 ? 1 : (e > 511 ? 511 : e)));
    if (cx.lane == 0 && wm > cx.wrm[ridx]) cx.wrm[ridx] = wm;
}
// kernel prologue / epilogue of the region bookkeeping
NH_DEVICE void regions_begin(BCtx& cx, char* lds_tail, unsigned* rmax) {
    cx.wrm = rmax ? (unsigned*)lds_tail + cx.wave * NH_RMAX_WORDS : nullptr;
    if (cx.wrm) cx.wrm[cx.lane] = 0u;
}
NH_DEVICE void regions_end(const BCtx& cx, unsigned* rmax) {
    if (cx.wrm) {
        const unsigned v = cx.wrm[cx.lane];
        if (v != 0u) nh_atomic_max_u32(rmax + cx.lane, v);
    }
}

// `bytes` (a multiple of 1 KiB) of the image, from byte offset `src`, into chunk buffer b at byte offset dst_off: one 1-KiB
// piece per wave-instruction, pieces dealt round-robin to the four waves
template <int BUF>
NH_DEVICE void b_issue(const BCtx& cx, int64_t src, int bytes, int b, int dst_off) {
#ifdef NHB_EXP_NO_STREAM  // (diagnostic builds only, wrong results: what the kernel costs without the weight stream)
    const int pieces = (bytes >> 10) < 4 ? (bytes >> 10) : 4;
#else
    const int pieces = bytes >> 10;
#endif
    for (int p = cx.wave; p < pieces; p += 4)
        nh_dma16a(cx.dma, cx.lane * 16, (int)src + p * 1024, cx.lds_addr + (unsigned)(b * BUF + dst_off + p * 1024));
}

// one accumulator tile -> two k-blocks of the next layer's operand pieces: hi = bf16(v), lo = bf16(v - hi)
// (mul, fp16 pieces: the power of two that takes the accumulators to the scale the pieces are wanted at)
template <bool RELU>
NH_DEVICE void convert_tile(const f32x16& acc, nh_pcx8* oh, nh_pcx8* ol, float mul) {
#ifdef NHB_EXP_NO_EPI  // (diagnostic builds only, wrong results: what the kernel costs without the conversions)
    oh[0][0] = nh_to_pc(acc[0]);
    ol[1][0] = nh_to_pc(acc[8]);
    return;
#endif
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[half * 8 + j];
            if (NHB_IS_F16) v *= mul;
            if (RELU) v = nh_relu(v);
            const nh_pc hi = nh_to_pc(v);
            oh[half][j] = hi;
            ol[half][j] = nh_to_pc(v - nh_from_pc(hi));
        }
}

// accumulators (WS * 2^s_in * value) -> operand pieces of value * 2^s_out with the sample's largest magnitude moved to
// [2^13, 2^14) (never above `cap`): the renormalisation step of the data-gradient chain (the forward's sits in gemm_b's epilogue)
template <int NT>
NH_DEVICE int renorm_convert(const f32x16* acc, nh_pcx8* oh, nh_pcx8* ol, int s_in, int cap);

// bit pattern of the largest value the epilogue will convert (ReLU: of the positive ones; identity: of the magnitudes) over this
// lane's tiles AND those of the other lane half of its sample (the two halves hold different units of the same sample)
template <int NTE, bool RELU>
NH_DEVICE unsigned tile_max_bits(const f32x16* acc) {
    float m = 0.0f;
#pragma unroll
    for (int t = 0; t < NTE; ++t)
#pragma unroll
        for (int c = 0; c < 16; ++c) m = fmaxf(m, RELU ? acc[t][c] : fabsf(acc[t][c]));
    unsigned u;
    memcpy(&u, &m, 4);
    const unsigned o = (unsigned)nh_shfl_xor_i((int)u, 32);
    return u > o ? u : o;
}

template <int NT>
NH_DEVICE int renorm_convert(const f32x16* acc, nh_pcx8* oh, nh_pcx8* ol, int s_in, int cap) {
    float mul = NHB_INV_WS;
    int so = 0;
    if (NHB_IS_F16) {
        const unsigned mb = tile_max_bits<NT, false>(acc);
        const int base_e = NHB_WS_LOG2 + s_in;
        so = exp_for(mb, base_e);
        so = so < cap ? so : cap;
        so = so > base_e + 120 ? base_e + 120 : (so < base_e - 120 ? base_e - 120 : so);
        mul = nh_pow2i(so - base_e);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) convert_tile<false>(acc[t], oh + 2 * t, ol + 2 * t, mul);
    return so;
}

// EPI (0: none; 1: ReLU; 2: identity): the first NTE output tiles leave as the next layer's operand pieces oh / ol (k-blocks
// 2 t, 2 t + 1 from tile t; they may be the inputs themselves), converted after the layer's last chunk.  (Converting each tile of the last chunk as it
// completes, in the shadow of the next tile's MFMAs, measured 3 % SLOWER on MI355X: the second set of operand registers it
// needs takes the VGPR file to its limit -- profiles/r03_variant_ab.txt section 7.)
// acc[t] = bias + sum over NKA activation k-blocks (ah/al) and NKB encoding k-blocks (xh/xl) of this layer's image at byte
// offset `base`; while the last chunk is multiplied the first chunk of the next layer (next_base, next_first bytes) travels.
template <int NB>
NH_DEVICE void stash_mask_in(unsigned* tile16_mask, int s, int h, const nh_pcx8* vh);

// in_rows / in_mask (training launches, else NULL): the gemm stores ITS OWN activation inputs -- the previous layer's output
// as the operand pieces say it, hi + lo, i.e. exactly the values this layer consumes -- into that layer's stash region,
// one k-block (two 16-byte stores) every other block of its first chunk, under the MFMAs, and the ReLU bits of the same
// values in the data-gradient kernel's lane layout (as mlp16.hip: "every gemm stores its own input rows").
// ROW_SCALE (the fp16 data-gradient chain): the stored rows are the pieces' sum times row_scale -- the power of two that turns the
// sample's own scale into the launch's.
// DYN (the fp16 forward): the block-floating-point bookkeeping of the file header -- s_in: exponent of the hidden inputs (of the
// encoding inputs when there are no hidden ones), s_x: exponent the encoding pieces were made at (>= s_in: rescaled on the fly),
// cap: the largest exponent the outputs may get (the next layer's encoding exponent, or NHB_NO_CAP), *s_out: what they got.
template <int W, int NT, int NKA, int NKB, int EPI = 0, int NTE = 0, bool ROW_SCALE = false, bool DYN = false>
NH_DEVICE void gemm_b(BCtx& cx, const nh_pcx8* ah, const nh_pcx8* al, const nh_pcx8* xh, const nh_pcx8* xl, int64_t base,
                      int64_t next_base, int next_first, f32x16* acc, nh_pcx8* oh = nullptr, nh_pcx8* ol = nullptr,
                      float* in_rows = nullptr, unsigned* in_mask = nullptr, int s32 = 0, float row_scale = 1.0f, int s_in = 0,
                      int s_x = 0, int cap = NHB_NO_CAP, int* s_out = nullptr, int ridx = -1, float bias_mul = 0.0f,
                      bool use_bias_mul = false) {
    constexpr int NK = NKA + NKB, BUF = BShape<W>::BUF, CH = BShape<W>::CHUNK / (NT * 2048), NCH = (NK + CH - 1) / CH;
    static_assert(CH >= 1, "a k-block of every tile must fit one chunk buffer");
    int srow_next = 0;  // next input k-block whose rows go out
    auto store_step = [&](int kb) {
        float4 a4, b4;
#ifdef NHB_EXP_STASH_HI  // (diagnostic builds only, wrong results: what the hi + lo reconstruction costs)
        a4.x = nh_from_pc(ah[kb][0]), a4.y = nh_from_pc(ah[kb][1]), a4.z = nh_from_pc(ah[kb][2]), a4.w = nh_from_pc(ah[kb][3]);
        b4.x = nh_from_pc(ah[kb][4]), b4.y = nh_from_pc(ah[kb][5]), b4.z = nh_from_pc(ah[kb][6]), b4.w = nh_from_pc(ah[kb][7]);
#else
        a4.x = nh_from_pc(ah[kb][0]) + nh_from_pc(al[kb][0]);
        a4.y = nh_from_pc(ah[kb][1]) + nh_from_pc(al[kb][1]);
        a4.z = nh_from_pc(ah[kb][2]) + nh_from_pc(al[kb][2]);
        a4.w = nh_from_pc(ah[kb][3]) + nh_from_pc(al[kb][3]);
        b4.x = nh_from_pc(ah[kb][4]) + nh_from_pc(al[kb][4]);
        b4.y = nh_from_pc(ah[kb][5]) + nh_from_pc(al[kb][5]);
        b4.z = nh_from_pc(ah[kb][6]) + nh_from_pc(al[kb][6]);
        b4.w = nh_from_pc(ah[kb][7]) + nh_from_pc(al[kb][7]);
#endif
        if (ROW_SCALE) {
            a4.x *= row_scale, a4.y *= row_scale, a4.z *= row_scale, a4.w *= row_scale;
            b4.x *= row_scale, b4.y *= row_scale, b4.z *= row_scale, b4.w *= row_scale;
        }
#ifdef NHB_EXP_SWIZZLE_STASH  // (diagnostic builds only, wrong layout for the consumers: the 256-byte quarters of a sample's row
                              // XOR-ed with the sample index, so that one store instruction's 32 pieces spread over 16 L2 channels
                              // instead of 4 -- does the write path care?)
        float* const dst = in_rows + ((32 * (kb >> 1) + 16 * (kb & 1) + 4 * cx.h) ^ ((s32 & (NKA >= 16 ? 3 : 1)) << 6));
#else
        float* const dst = in_rows + 32 * (kb >> 1) + 16 * (kb & 1) + 4 * cx.h;  // units nhb_unit(kb, h, 0..3) and (kb, h, 4..7) = + 8
#endif
#ifdef NHB_EXP_NO_STASH_STORE  // (diagnostic builds only, wrong results: what the stores themselves cost)
        if (a4.x == 1.2345e-30f && b4.y == 5.4321e-30f) nh_store4(dst, a4.x, a4.y, a4.z, a4.w);
#elif defined(NHB_EXP_DENSE_STASH)  // (diagnostic builds only, wrong layout: the SAME bytes as [32-row tile][plane][slot][sample][16 B] --
        // every store instruction then writes eight whole 128-byte lines instead of 32 x 32 bytes at a 1-KiB stride)
        float* const tb = in_rows - s32 * (16 * NKA);
        nh_store4(tb + (((kb >> 1) * 2 + 0) * 4 + 2 * (kb & 1) + cx.h) * 128 + s32 * 4, a4.x, a4.y, a4.z, a4.w);
        nh_store4(tb + (((kb >> 1) * 2 + 1) * 4 + 2 * (kb & 1) + cx.h) * 128 + s32 * 4, b4.x, b4.y, b4.z, b4.w);
#else
        nh_store4(dst, a4.x, a4.y, a4.z, a4.w);
        nh_store4(dst + 8, b4.x, b4.y, b4.z, b4.w);
#endif
    };
    (void)srow_next;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        nh_wait_vmem();    // this wave's pieces of the current chunk have landed ...  (and its stash stores: vmcnt counts them --
                           // letting the youngest ones stay in flight, s_waitcnt vmcnt(2 NKA), measured no different)
        nh_block_sync();   // ... and everyone's; nobody still reads the other buffer
        // The next chunk (or the next layer's first one) goes to the other buffer WHILE this one is multiplied: its 1-KiB
        // pieces are dealt to the waves round-robin and each wave issues one piece every NHB_DMA_EVERY blocks, so that the
        // copy's LDS writes interleave with the MFMAs' LDS reads instead of arriving as one burst at the chunk's start.
        int64_t dsrc = 0;
        int dpieces = 0, ddst = 0;
        if (c + 1 < NCH) {
            const int nkb = NK - (c + 1) * CH < CH ? NK - (c + 1) * CH : CH;
            dsrc = base + 2048 + (int64_t)(c + 1) * CH * NT * 2048, dpieces = nkb * NT * 2, ddst = 2048;
        } else if (next_first > 0) {
            dsrc = next_base, dpieces = next_first >> 10, ddst = 0;
        }
#ifdef NHB_EXP_NO_STREAM
        dpieces = dpieces < 4 ? dpieces : 4;
#endif
        int dnext = cx.wave;  // this wave's next piece
        auto dma_step = [&]() {
            if (dnext < dpieces) {
                nh_dma16a(cx.dma, cx.lane * 16, (int)dsrc + dnext * 1024, cx.lds_addr + (unsigned)((cx.buf ^ 1) * BUF + ddst + dnext * 1024));
                dnext += 4;
            }
        };
#if NHB_DMA_EVERY == 0
        while (dnext < dpieces) dma_step();
#endif
        if (c == 0 && in_rows && NKA > 0) {
            // training launches: the copy pieces go out at once, ahead of this gemm's stash stores, and the mask words are
            // stored here, AFTER the chunk's wait (in front of it they were four HBM writes every layer had to wait for):
            // forward 5.32 -> 4.70 ms, data gradient 4.62 -> 4.02 ms per step
            while (dnext < dpieces) dma_step();
            if (in_mask) stash_mask_in<NKA>(in_mask, s32, cx.h, ah);
        }
        const char* const buf = cx.lds + cx.buf * BUF;
        if (c == 0) {  // the accumulators start at the bias of their rows: register 4 j + i of tile t holds row 32 t + 8 j + 4 h + i
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 b4 = *(const float4*)(buf + (32 * t + 8 * j + 4 * cx.h) * 4);
                    // (DYN: the products carry 2^s_in, so must the bias -- the multiply takes the place of the copy)
                    // (use_bias_mul, the data-gradient chain's head: the bias row holds fc_alpha's weights, the factor is the
                    // sample's d(sigma raw) at the inputs' exponent)
                    const bool scaled = DYN || use_bias_mul;
                    const float bsc = use_bias_mul ? bias_mul : (DYN ? nh_pow2i(s_in) : 1.0f);
                    acc[t][4 * j] = scaled ? b4.x * bsc : b4.x;
                    acc[t][4 * j + 1] = scaled ? b4.y * bsc : b4.y;
                    acc[t][4 * j + 2] = scaled ? b4.z * bsc : b4.z;
                    acc[t][4 * j + 3] = scaled ? b4.w * bsc : b4.w;
                }
        }
        const char* const wb = buf + 2048 + cx.lane * 16;
        // The (k-block, tile) blocks of this chunk, software-pipelined by hand: the weight pieces of block i + PF are read
        // from LDS before the MFMAs of block i issue, each side of a scheduling fence -- left to itself the compiler reads
        // every block into ONE register quad right in front of its MFMAs and waits for it (seen in the ISA: ds_read_b128,
        // s_waitcnt lgkmcnt(0), v_mfma, ... -- the matrix pipe idles for an LDS latency per block; 0.39 of the roofline).
        const int nkk = NK - c * CH < CH ? NK - c * CH : CH;  // k-blocks in this chunk
        const int nblk = nkk * NT;
        constexpr int PF = NHB_PREFETCH;
        constexpr int NBUF = PF + 1;
        nh_pcx8 wph[NBUF], wpl[NBUF];
        auto kk_of = [&](int i) { return i / NT; };
        auto t_of = [&](int i) { return i % NT; };
        auto load = [&](int i) {
            wph[i % NBUF] = *(const nh_pcx8*)(wb + ((kk_of(i) * NT + t_of(i)) * 2) * 1024);
            wpl[i % NBUF] = *(const nh_pcx8*)(wb + ((kk_of(i) * NT + t_of(i)) * 2 + 1) * 1024);
        };
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (i < nblk) load(i);
#pragma unroll
        for (int i = 0; i < CH * NT; ++i) {
            if (i < nblk) {
#if NHB_DMA_EVERY > 0
                if (i % NHB_DMA_EVERY == 0) dma_step();
#endif
#ifdef NHB_STORE_SPREAD  // (A/B builds only) a k-block's rows go out when its first tile is multiplied: the stores of a gemm
                          // spread over ALL its chunks instead of the first
                if (NKA > 0 && in_rows && t_of(i) == 0 && c * CH + kk_of(i) < NKA) store_step(c * CH + kk_of(i));
#else
                if (NKA > 0 && c == 0 && i % 2 == 0 && i / 2 < NKA) {
                    if (in_rows) store_step(i / 2);
                }
#endif
                if (i + PF < nblk) load(i + PF);
                nh_sched_fence();
                const int kb = c * CH + kk_of(i), t = t_of(i);
                nh_pcx8 bh = kb < NKA ? ah[kb < NKA ? kb : 0] : xh[kb >= NKA ? kb - NKA : 0];
                nh_pcx8 bl = kb < NKA ? al[kb < NKA ? kb : 0] : xl[kb >= NKA ? kb - NKA : 0];
#ifdef NHB_F16
                if (DYN && NKA > 0 && kb >= NKA) {  // encoding pieces made at 2^s_x, wanted at the hidden inputs' 2^s_in (<= s_x)
                    const float xf = nh_pow2i(s_in - s_x);
                    bh = nh_f16x8_scale(bh, xf);
                    bl = nh_f16x8_scale(bl, xf);
                }
#endif
                const nh_pcx8 wh = wph[i % NBUF], wl = wpl[i % NBUF];
                acc[t] = nh_mfma_pc(wl, bh, acc[t]);  // (the small terms first)
                acc[t] = nh_mfma_pc(wh, bl, acc[t]);
                acc[t] = nh_mfma_pc(wh, bh, acc[t]);
            }
        }
        while (dnext < dpieces) dma_step();  // (whatever the blocks did not cover: short chunks in front of long ones)
#ifndef NHB_STORE_SPREAD
        if (NKA > 0 && c == 0 && in_rows) {
#pragma unroll
            for (int kb = 0; kb < NKA; ++kb)
                if (kb >= (nblk + 1) / 2) store_step(kb);  // (first chunks with fewer than 2 NKA blocks: the rgb / fc_out gemms)
        }
#endif
        cx.buf ^= 1;
    }
    if (NHB_IS_F16 && cx.wrm && ridx >= 0 && in_rows && NKA > 0) note_region(cx, ridx, s_in);
    if (EPI != 0) {
        float mul = NHB_INV_WS;
        if (DYN) {
            // the accumulators hold WS * 2^s_in * value: move the sample's largest output to [2^13, 2^14) -- unless the next
            // layer's encodings sit lower -- and remember the exponent the pieces now carry
            const unsigned mb = tile_max_bits<NTE, EPI == 1>(acc);
            const int base_e = NHB_WS_LOG2 + s_in;
            int so = exp_for(mb, base_e);
            so = so < cap ? so : cap;
            so = so > base_e + 120 ? base_e + 120 : (so < base_e - 120 ? base_e - 120 : so);
            mul = nh_pow2i(so - base_e);
            *s_out = so;  // the pieces made below are those of value * 2^so
        }
#pragma unroll
        for (int t = 0; t < NTE; ++t) convert_tile<EPI == 1>(acc[t], oh + 2 * t, ol + 2 * t, mul);
    }
}

NH_DEVICE void put_pair(nh_pcx8& oh, nh_pcx8& ol, int e, float v) {
    const nh_pc hi = nh_to_pc(v);
    oh[e] = hi;
    ol[e] = nh_to_pc(v - nh_from_pc(hi));
}

NH_DEVICE float bsel3(int a, float x, float y, float z) { return a == 0 ? x : (a == 1 ? y : z); }

// eight fp32 slot values of one k-block -> operand pieces (of v * sc: fp16 pieces carry the sample's encoding exponent), and
// (training) -> this sample's row of the slot region (the plain values)
NH_DEVICE void put_block(nh_pcx8& oh, nh_pcx8& ol, const float* v, float* slot_row, float sc = 1.0f) {
#pragma unroll
    for (int e = 0; e < 8; ++e) put_pair(oh, ol, e, NHB_IS_F16 ? v[e] * sc : v[e]);
    if (slot_row) {
        float4 a4, b4;
        a4.x = v[0], a4.y = v[1], a4.z = v[2], a4.w = v[3];
        b4.x = v[4], b4.y = v[5], b4.z = v[6], b4.w = v[7];
        *(float4*)slot_row = a4;
        *(float4*)(slot_row + 4) = b4;
    }
}

// the encoding slots of lane half h (plan.cpp build_slot_map_b): slot 16 kb + 8 h + e; pair slot >> 1 = 3 f + axis.
// slot_row (training): this sample's row of the stash's slot region; the lane writes its slots 16 kb + 8 h .. + 7.
template <int NB>
NH_DEVICE void encode_b(nh_pcx8* oh, nh_pcx8* ol, float x, float y, float z, int h, const float* freqs, int Lf, float* slot_row, float sc) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pr = 8 * kb + 4 * h + q;
            const bool valid = pr < 3 * Lf;
            const int f = pr / 3, ax = pr - 3 * f;
            float s, c;
            nh_sincos(bsel3(ax, x, y, z) * freqs[f < 16 ? f : 15], &s, &c);
            float v0 = valid ? s : 0.0f, v1 = valid ? c : 0.0f;
            if (kb == NB - 1 && q == 2 && h == 1) v0 = x, v1 = y;  // slots NS-4, NS-3 (never a valid pair: 6 L <= NS - 4)
            if (kb == NB - 1 && q == 3 && h == 1) v0 = z, v1 = 0.0f;
            v[2 * q] = v0;
            v[2 * q + 1] = v1;
        }
        put_block(oh[kb], ol[kb], v, slot_row ? slot_row + 16 * kb + 8 * h : nullptr, sc);
    }
}
// the same slots gathered from a caller-encoded row (mode 0)
template <int NB>
NH_DEVICE void gather_b(nh_pcx8* oh, nh_pcx8* ol, const float* row, const signed char* col, int h, float* slot_row, float sc) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = (int)col[16 * kb + 8 * h + e];
            v[e] = c >= 0 ? row[c] : 0.0f;
        }
        put_block(oh[kb], ol[kb], v, slot_row ? slot_row + 16 * kb + 8 * h : nullptr, sc);
    }
}
// exponent for a sample's encodings (fp16 pieces): its largest magnitude `m` (both lane halves) to [2^13, 2^14)
NH_DEVICE int enc_exponent(float m) {
    unsigned u;
    memcpy(&u, &m, 4);
    const unsigned o = (unsigned)nh_shfl_xor_i((int)u, 32);
    return exp_for(u > o ? u : o, 0);
}
// largest magnitude among this lane's slots of a caller-encoded row
template <int NB>
NH_DEVICE float gather_max_b(const float* row, const signed char* col, int h) {
    float m = 0.0f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = (int)col[16 * kb + 8 * h + e];
            m = fmaxf(m, c >= 0 ? fabsf(row[c]) : 0.0f);
        }
    return m;
}

// ---- the training stash (NERFHIP_PRECISION_BF16X3_FWD): what the fp32 backward kernels read (nh_plan.h NhStashLayout) ------
// ReLU bits of a layer's NB k-blocks of operand pieces (value > 0 <=> its high piece > 0) in the data-gradient kernel's
// layout (mlp16.hip finish() / nh16_bitpos): its lane (sample & 15, g) holds unit 16 (r >> 2) + 4 g + (r & 3) in register r; of
// its n = 4 NB registers, r sits in word r >> 5 at bit min(32, n - 32 (r >> 5)) - 1 - (r & 31).  This lane's element e of
// k-block kb is unit 32 (kb >> 1) + 16 (kb & 1) + 8 (e >> 2) + 4 h + (e & 3) = register 4 kb + (e & 3) of the lane
// g = 2 (e >> 2) + h: it writes both lanes' words whole -- no exchange between lanes.
template <int NB>
NH_DEVICE void stash_mask_in(unsigned* tile16_mask, int s, int h, const nh_pcx8* vh) {
    constexpr int n = 4 * NB;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        unsigned w0 = 0u, w1 = 0u;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * kb + i, word = r >> 5;
                const int pos = ((n - 32 * word) < 32 ? (n - 32 * word) : 32) - 1 - (r & 31);
                const unsigned bit = nh_from_pc(vh[kb][4 * jb + i]) > 0.0f ? 1u : 0u;
                if (word == 0) w0 |= bit << pos;
                else w1 |= bit << pos;
            }
        unsigned* dst = tile16_mask + ((s & 15) + 16 * (2 * jb + h)) * 2;
        dst[0] = w0;
        dst[1] = w1;
    }
}

struct FwdBArgs {
    const float* packed;
    unsigned packed_bytes;
    NhPackedOffsets off;  // (32-bit word offsets)
    int L, skip;
    int64_t M, groups;  // sample points; 128-sample groups = ceil(M / 128)
    int mode;
    const float* x;
    int dx, dd;
    const float* rays;
    int ray_stride;
    const float* z;
    int S;
    signed char xcol[16 * NHB_XBLOCKS];
    signed char dcol[16 * NHB_DBLOCKS];
    float fx[16], fd[16];
    int Lx, Ld;
    float* out;
    float* stash;      // TRAIN launches: the activation stash of the fp32 backward kernels
    unsigned* rmax;    // fp16 level-4 plans: per-region maxima of what the stash holds (H[k] -> k, FEAT -> L), else NULL
    NhStashLayout sl;
    int64_t nt;        // 32-sample tiles of the launch (4 per 128-sample group)
};

// TRAIN (NERFHIP_PRECISION_BF16X3_FWD plans): the launch also writes the activation stash -- encoding slots, every layer's
// fp32 output rows as computed here, ReLU masks -- in the format k_mlp_dgrad16 / k_wgrad read
// (register budget: two waves per SIMD for the 128-wide inference kernel; the training one needs ~300 registers with its
// store addresses and runs one wave per SIMD like the 256-wide kernels rather than spill)
template <int W, bool VIEW, bool TRAIN>
NH_KERNEL void NH_LB(256, (TRAIN ? 1 : BShape<W>::WAVES_PER_SIMD)) NHB_KERNEL(k_mlp_fwd)(FwdBArgs a) {
    constexpr int TH = BShape<W>::TH, KBH = BShape<W>::KBH, BUF = BShape<W>::BUF, XB = NHB_XBLOCKS, DB = NHB_DBLOCKS;
    NH_DYN_LDS(lds_raw);
    BCtx cx;
    cx.lds = lds_raw;
    cx.lds_addr = nh_lds_addr((const float*)lds_raw);
    cx.dma = nh_dma_src(a.packed, a.packed_bytes);
    cx.buf = 0;
    cx.lane = nh_lane();
    cx.wave = nh_wave_in_block();
    cx.h = cx.lane >> 5;
    regions_begin(cx, lds_raw + BShape<W>::LDS_BYTES, (TRAIN && NHB_IS_F16) ? a.rmax : nullptr);
    const int h = cx.h;
    const NhPackedOffsets& po = a.off;
    auto first = [](int nk, int nt) { return nhb_first_bytes(nk, nt, W); };

    // the first weights travel while the encodings are formed
    b_issue<BUF>(cx, po.f_layer1 * 4, first(XB, TH), 0, 0);

    // Persistent workgroups: the launch holds as many workgroups as the chip runs at once (host: one or two per CU) and each
    // walks over its 128-sample groups -- a workgroup owns all of a CU's LDS, so with one group per workgroup the next
    // could only be dispatched after the previous one had drained (17 % of the SIMD time had no wave resident: rocprofv3
    // SQ_WAVE_CYCLES against the kernel's duration); here the last layer of a group already streams layer1 of the next.
    for (int64_t grp = blockIdx.x; grp < a.groups; grp += gridDim.x) {
    const bool again = grp + gridDim.x < a.groups;
    const int64_t m = grp * 128 + cx.wave * 32 + (cx.lane & 31);
    const bool valid = m < a.M;
    const int64_t mc = valid ? m : a.M - 1;
    // training: this sample's row of a stash region ([32-sample tile][sample][rows]; whole groups are written, clamped
    // samples included), and the mask words of its 16-sample tile ([tile][mask][64 lanes][2 words])
    const int s32 = cx.lane & 31;
    const int64_t tile32 = grp * 4 + cx.wave;
    auto srow = [&](const NhRegion& R, int rows) -> float* {
        return a.stash + (size_t)32 * (size_t)a.nt * (size_t)R.row_prefix + ((size_t)tile32 * 32 + (size_t)s32) * (size_t)rows;
    };
    auto smask = [&](int idx) -> unsigned* {
        return (unsigned*)(a.stash + (size_t)32 * (size_t)a.nt * (size_t)a.sl.total_rows) +
               ((size_t)(tile32 * 2 + (s32 >> 4)) * (size_t)a.sl.n_masks + (size_t)idx) * 128;
    };
    nh_pcx8 xh[XB], xl[XB];
    const int ray_i = a.mode == 0 ? 0 : (int)(mc / a.S);
    constexpr bool DYN = NHB_IS_F16;  // (fp16 pieces: per-sample exponents, file header; ex / ed: the encodings', s: the activations')
    constexpr bool RS = TRAIN && NHB_IS_F16;
    int ex = 0, ed = 0, s = 0;
    if (a.mode == 0) {
        const float* const row = a.x + (size_t)mc * (size_t)(a.dx + a.dd);
        if (DYN) ex = enc_exponent(gather_max_b<XB>(row, a.xcol, h));
        gather_b<XB>(xh, xl, row, a.xcol, h, TRAIN ? srow(a.sl.X, 16 * XB) : nullptr, nh_pow2i(ex));
    } else {
        const float* const rr = a.rays + (size_t)ray_i * a.ray_stride;
        const float zz = a.z[mc];
        // pts = ro + rd * z   (nerf/train_utils.py:67,107)
        const float px = rr[0] + rr[3] * zz, py = rr[1] + rr[4] * zz, pz = rr[2] + rr[5] * zz;
        if (DYN) ex = enc_exponent(fmaxf(fmaxf(fabsf(px), fabsf(py)), fmaxf(fabsf(pz), 1.0f)));  // (sines and cosines: <= 1)
        encode_b<XB>(xh, xl, px, py, pz, h, a.fx, a.Lx, TRAIN ? srow(a.sl.X, 16 * XB) : nullptr, nh_pow2i(ex));
    }
    s = ex;

    f32x16 acc[TH + 1];
    nh_pcx8 hh[KBH], hl[KBH];  // the current activations as operand pieces (a layer's output replaces them in place:
                                 // the conversion runs after the layer's last MFMA)
    {
        const bool more = a.L > 1;
        // no activation after layer1 (models.py:238)
        // (cap of the outputs' exponent: the next gemm's encodings -- layers_xyz[0] never is a skip layer; the head's are the
        // directions, whose exponent is not known yet: at most NHB_TARGET_LOG2, |viewdir| <= 1 ... taken when they are formed)
        gemm_b<W, TH, 0, XB, 2, TH, false, DYN>(cx, nullptr, nullptr, xh, xl, po.f_layer1 * 4, (more ? po.f_xyz[0] : po.f_head) * 4,
                                                more ? first(KBH, TH) : (VIEW ? first(KBH, TH + 1) : first(KBH, 1)), acc, hh, hl, nullptr,
                                                nullptr, 0, 1.0f, ex, ex, NHB_NO_CAP, &s);
    }
    for (int i = 0; i < a.L - 1; ++i) {
        const bool sk = (i % a.skip == 0) && i > 0;
        const bool more = i + 1 < a.L - 1;
        const bool nsk = more && ((i + 1) % a.skip == 0);
        const int64_t nxt = (more ? po.f_xyz[i + 1] : po.f_head) * 4;
        const int nfirst = more ? (nsk ? first(KBH + XB, TH) : first(KBH, TH)) : (VIEW ? first(KBH, TH + 1) : first(KBH, 1));
        // (training: the gemm stores its inputs H_i and their ReLU mask i - 1; H_0 = layer1's output has none)
        float* const in_rows = TRAIN ? srow(a.sl.H[i], W) : nullptr;
        unsigned* const in_mask = (TRAIN && i > 0) ? smask(i - 1) : nullptr;
        const int cap = nsk ? ex : NHB_NO_CAP;  // (a skip layer next: its hidden inputs must not sit above the encodings)
        const float rsc = RS ? nh_pow2i(-s) : 1.0f;  // (training, fp16 pieces: the stash rows are plain values)
        if (sk)
            gemm_b<W, TH, KBH, XB, 1, TH, RS, DYN>(cx, hh, hl, xh, xl, po.f_xyz[i] * 4, nxt, nfirst, acc, hh, hl, in_rows, in_mask, s32, rsc, s, ex,
                                                   cap, &s, TRAIN ? i : -1);
        else
            gemm_b<W, TH, KBH, 0, 1, TH, RS, DYN>(cx, hh, hl, nullptr, nullptr, po.f_xyz[i] * 4, nxt, nfirst, acc, hh, hl, in_rows, in_mask, s32, rsc,
                                                  s, s, cap, &s, TRAIN ? i : -1);
    }
    if (VIEW) {
        nh_pcx8 dh[DB], dl[DB];
        if (a.mode == 0) {
            const float* const row = a.x + (size_t)mc * (size_t)(a.dx + a.dd) + a.dx;
            if (DYN) ed = enc_exponent(gather_max_b<DB>(row, a.dcol, h));
            gather_b<DB>(dh, dl, row, a.dcol, h, TRAIN ? srow(a.sl.D, 16 * DB) : nullptr, nh_pow2i(ed));
        } else {
            const float* const rr = a.rays + (size_t)ray_i * a.ray_stride;
            if (DYN) ed = enc_exponent(fmaxf(fmaxf(fabsf(rr[8]), fabsf(rr[9])), fmaxf(fabsf(rr[10]), 1.0f)));
            encode_b<DB>(dh, dl, rr[8], rr[9], rr[10], h, a.fd, a.Ld, TRAIN ? srow(a.sl.D, 16 * DB) : nullptr, nh_pow2i(ed));
        }
        // tiles 0..TH-1: feat = relu(fc_feat(h)); tile TH row 0: fc_alpha(h), raw (models.py:248-249)
        // (training: each gemm stores its own inputs -- H_{L-1} and mask L - 2, FEAT and mask L - 1, DIRH and mask L)
        const int s_head = s;  // (the head's inputs: fc_alpha's raw row comes out at WS * 2^s_head)
        gemm_b<W, TH + 1, KBH, 0, 1, TH, RS, DYN>(cx, hh, hl, nullptr, nullptr, po.f_head * 4, po.f_dir * 4, first(KBH + DB, TH / 2), acc, hh, hl,
                                                  TRAIN ? srow(a.sl.H[a.L - 1], W) : nullptr, (TRAIN && a.L > 1) ? smask(a.L - 2) : nullptr, s32,
                                                  RS ? nh_pow2i(-s) : 1.0f, s, s, ed, &s, TRAIN ? a.L - 1 : -1);
        const float alpha = nhb_raw(acc[TH][0], s_head);
        gemm_b<W, TH / 2, KBH, DB, 1, TH / 2, RS, DYN>(cx, hh, hl, dh, dl, po.f_dir * 4, po.f_rgb * 4, first(KBH / 2, 1), acc, hh, hl,
                                                       TRAIN ? srow(a.sl.FEAT, W) : nullptr, TRAIN ? smask(a.L - 1) : nullptr, s32,
                                                       RS ? nh_pow2i(-s) : 1.0f, s, ed, NHB_NO_CAP, &s, TRAIN ? a.L : -1);
        gemm_b<W, 1, KBH / 2, 0, 0, 0, RS, DYN>(cx, hh, hl, nullptr, nullptr, po.f_rgb * 4, po.f_layer1 * 4, again ? first(XB, TH) : 0, acc, nullptr,
                                                nullptr, TRAIN ? srow(a.sl.DIRH, W / 2) : nullptr, TRAIN ? smask(a.L) : nullptr, s32,
                                                RS ? nh_pow2i(-s) : 1.0f, s, s);
        if (valid && h == 0) {
            float4 r4;
            r4.x = nhb_raw(acc[0][0], s);
            r4.y = nhb_raw(acc[0][1], s);
            r4.z = nhb_raw(acc[0][2], s);
            r4.w = alpha;
            *(float4*)(a.out + (size_t)m * 4) = r4;
        }
    } else {
        gemm_b<W, 1, KBH, 0, 0, 0, RS, DYN>(cx, hh, hl, nullptr, nullptr, po.f_head * 4, po.f_layer1 * 4, again ? first(XB, TH) : 0, acc, nullptr,
                                            nullptr, TRAIN ? srow(a.sl.H[a.L - 1], W) : nullptr, (TRAIN && a.L > 1) ? smask(a.L - 2) : nullptr, s32,
                                            RS ? nh_pow2i(-s) : 1.0f, s, s, NHB_NO_CAP, nullptr, TRAIN ? a.L - 1 : -1);  // fc_out (models.py:256)
        if (valid && h == 0) {
            float4 r4;
            r4.x = nhb_raw(acc[0][0], s);
            r4.y = nhb_raw(acc[0][1], s);
            r4.z = nhb_raw(acc[0][2], s);
            r4.w = nhb_raw(acc[0][3], s);
            *(float4*)(a.out + (size_t)m * 4) = r4;
        }
    }
    }  // (groups of this workgroup)
    regions_end(cx, a.rmax);
}

// ---- the data-gradient chain on the same loop (NERFHIP_PRECISION_BF16X3_FWD_DGRAD) -------------------------------------------
// dpre_{k-1} = relu'(H_{k-1}) * (W_{k-1}^T dpre_k), walked from d(raw output) down to layer1 (mlp16.hip k_mlp_dgrad16 is the
// fp32 original; the reference has no such function: it is autograd of nerf/models.py:233-258).  A wave owns 32 samples;
// d(pre-activation) lives in registers as operand pieces and every gemm stores its own input -- the d(pre-activation) image
// the weight-gradient kernel reads -- as the forward does with the activations.  ReLU masks: the words the forward kernels
// wrote for the fp32 data-gradient kernel's lanes (s & 15, g = 2 jb + h), jb = 0, 1 -- this lane's units.
struct DgradBArgs {
    const float* packed;
    unsigned packed_bytes;
    NhPackedOffsets off;
    int L;
    int64_t M, groups, nt;
    const float* g_out;
    const float* stash;
    NhStashLayout sl;
    float* grad;
    NhGradLayout gl;
    unsigned* rmax;  // fp16 level-4 plans: per-region maxima of the images written (P[k] -> k, PFEAT -> L, PDIR -> L + 1), else NULL
};

// zero the accumulators whose ReLU bit is 0: unit 32 t + 8 j + 4 h + i is register r = 4 (2 t + (j >> 1)) + i of lane g = 2 (j & 1) + h
template <int NT>
NH_DEVICE void gate_tiles(f32x16* acc, const unsigned* mw) {
    constexpr int n = 8 * NT;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * (2 * t + (j >> 1)) + i, word = r >> 5;
                const int pos = ((n - 32 * word) < 32 ? (n - 32 * word) : 32) - 1 - (r & 31);
                acc[t][4 * j + i] = nh_gate(acc[t][4 * j + i], mw[2 * (j & 1) + word], pos);
            }
}

template <int W, bool VIEW>
NH_KERNEL void NH_LB(256, 1) NHB_KERNEL(k_mlp_dgrad)(DgradBArgs a) {
    constexpr int TH = BShape<W>::TH, KBH = BShape<W>::KBH, BUF = BShape<W>::BUF;
    NH_DYN_LDS(lds_raw);
    BCtx cx;
    cx.lds = lds_raw;
    cx.lds_addr = nh_lds_addr((const float*)lds_raw);
    cx.dma = nh_dma_src(a.packed, a.packed_bytes);
    cx.buf = 0;
    cx.lane = nh_lane();
    cx.wave = nh_wave_in_block();
    cx.h = cx.lane >> 5;
    regions_begin(cx, lds_raw + BShape<W>::LDS_BYTES, NHB_IS_F16 ? a.rmax : nullptr);
    const int h = cx.h, L = a.L;
    const NhPackedOffsets& po = a.off;
    auto first = [](int nk, int nt) { return nhb_first_bytes(nk, nt, W); };
    const int64_t first_img = (VIEW ? po.b_rgb : po.b_head) * 4;
    const int first_bytes = VIEW ? first(1, TH / 2) : first(1, TH);
    b_issue<BUF>(cx, first_img, first_bytes, 0, 0);

    for (int64_t grp = blockIdx.x; grp < a.groups; grp += gridDim.x) {
        const bool again = grp + gridDim.x < a.groups;
        const int s32 = cx.lane & 31;
        const int64_t m = grp * 128 + cx.wave * 32 + s32;
        const int64_t tile32 = grp * 4 + cx.wave;
        float go[4] = {0.f, 0.f, 0.f, 0.f};  // d(raw output) of this lane's sample (zero beyond M: nothing flows)
        if (m < a.M) {
            const float4 t4 = *(const float4*)(a.g_out + (size_t)m * 4);
            go[0] = t4.x, go[1] = t4.y, go[2] = t4.z, go[3] = t4.w;
        }
        // fp16 pieces: the per-sample exponent of the chain (file header).  s: of the current d(pre-activation) pieces
        int s = 0;
        auto grow = [&](const NhRegion& R, int rows) -> float* {
            return a.grad + (size_t)32 * (size_t)a.nt * (size_t)R.row_prefix + ((size_t)tile32 * 32 + (size_t)s32) * (size_t)rows;
        };
        // ReLU mask `idx` of this lane's units: words [jb][0..1]
        auto get_mask = [&](int idx, unsigned* mw) {
            const unsigned* base = (const unsigned*)(a.stash + (size_t)32 * (size_t)a.nt * (size_t)a.sl.total_rows) +
                                   ((size_t)(tile32 * 2 + (s32 >> 4)) * (size_t)a.sl.n_masks + (size_t)idx) * 128;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const unsigned* p = base + ((s32 & 15) + 16 * (2 * jb + h)) * 2;
                mw[2 * jb] = p[0];
                mw[2 * jb + 1] = p[1];
            }
        };
        {  // POUT (32 rows): rows 0..2 d(rgb raw), row 3 d(sigma raw), the rest zero; lane half h writes rows 16 h .. 16 h + 15
            float* const pr = grow(a.gl.POUT, 32) + 16 * h;
            float4 z4, g4;
            z4.x = z4.y = z4.z = z4.w = 0.0f;
            g4.x = h == 0 ? go[0] : 0.0f, g4.y = h == 0 ? go[1] : 0.0f, g4.z = h == 0 ? go[2] : 0.0f, g4.w = h == 0 ? go[3] : 0.0f;
            *(float4*)pr = g4;
            *(float4*)(pr + 4) = z4;
            *(float4*)(pr + 8) = z4;
            *(float4*)(pr + 12) = z4;
        }
        f32x16 acc[TH];
        nh_pcx8 hh[KBH], hl[KBH];  // d(pre-activation) of the layer just finished, as operand pieces
        unsigned mw[4];
        nh_pcx8 d1h[1], d1l[1];    // the one k-block of d(raw output): elements 0..2 (0..3 without viewdirs) of lane half 0
        {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (h == 0) {
                v[0] = go[0], v[1] = go[1], v[2] = go[2];
                if (!VIEW) v[3] = go[3];
            }
            if (NHB_IS_F16) {  // (both lane halves of the sample agree: they loaded the same cotangent)
                const float mg = fmaxf(fmaxf(fabsf(go[0]), fabsf(go[1])), fmaxf(fabsf(go[2]), VIEW ? 0.0f : fabsf(go[3])));
                unsigned ub;
                memcpy(&ub, &mg, 4);
                s = exp_for(ub, 0);
            }
            put_block(d1h[0], d1l[0], v, nullptr, nh_pow2i(s));
        }
        if (VIEW) {
            gemm_b<W, TH / 2, 0, 1>(cx, nullptr, nullptr, d1h, d1l, po.b_rgb * 4, po.b_dir * 4, first(KBH / 2, TH), acc);
            get_mask(L, mw);  // DIRH
            gate_tiles<TH / 2>(acc, mw);
            s = renorm_convert<TH / 2>(acc, hh, hl, s, NHB_NO_CAP);
            gemm_b<W, TH, KBH / 2, 0, 0, 0, NHB_IS_F16>(cx, hh, hl, nullptr, nullptr, po.b_dir * 4, po.b_head * 4, first(KBH, TH), acc, nullptr,
                                                        nullptr, grow(a.gl.PDIR, W / 2), nullptr, 0, nh_pow2i(-s), s, s, NHB_NO_CAP, nullptr,
                                                        L + 1);
            get_mask(L - 1, mw);  // FEAT
            gate_tiles<TH>(acc, mw);
            s = renorm_convert<TH>(acc, hh, hl, s, NHB_NO_CAP);
            // d(sigma raw) enters through fc_alpha: the head image's bias row holds fc_alpha's weights (plan.cpp build_specs_b), and
            // the accumulators start at bias * d(sigma raw) * 2^s -- fp32, whatever its size next to the hidden inputs
            const bool last = L == 1;
            gemm_b<W, TH, KBH, 0, 0, 0, NHB_IS_F16>(cx, hh, hl, nullptr, nullptr, po.b_head * 4, last ? first_img : po.b_xyz[L > 1 ? L - 2 : 0] * 4,
                                                    last ? (again ? first_bytes : 0) : first(KBH, TH), acc, nullptr, nullptr, grow(a.gl.PFEAT, W),
                                                    nullptr, 0, nh_pow2i(-s), s, s, NHB_NO_CAP, nullptr, L, NHB_IS_F16 ? go[3] * nh_pow2i(s) : go[3], true);
        } else {
            const bool last = L == 1;
            gemm_b<W, TH, 0, 1>(cx, nullptr, nullptr, d1h, d1l, po.b_head * 4, last ? first_img : po.b_xyz[L > 1 ? L - 2 : 0] * 4,
                                last ? (again ? first_bytes : 0) : first(KBH, TH), acc);
        }
        // acc = W^T d(pre-activation) for H_{L-1}: gate by its ReLU mask L - 2 (H_0 = layer1's output has no activation)
        if (L > 1) {
            get_mask(L - 2, mw);
            gate_tiles<TH>(acc, mw);
        }
        s = renorm_convert<TH>(acc, hh, hl, s, NHB_NO_CAP);
        for (int k = L - 1; k >= 1; --k) {
            const bool last = k == 1;
            gemm_b<W, TH, KBH, 0, 0, 0, NHB_IS_F16>(cx, hh, hl, nullptr, nullptr, po.b_xyz[k - 1] * 4, last ? first_img : po.b_xyz[k >= 2 ? k - 2 : 0] * 4,
                                                    last ? (again ? first_bytes : 0) : first(KBH, TH), acc, nullptr, nullptr, grow(a.gl.P[k], W),
                                                    nullptr, 0, nh_pow2i(-s), s, s, NHB_NO_CAP, nullptr, k);
            if (k - 1 >= 1) {
                get_mask(k - 2, mw);  // H_{k-1}
                gate_tiles<TH>(acc, mw);
            }
            s = renorm_convert<TH>(acc, hh, hl, s, NHB_NO_CAP);
        }
        {  // d(pre-activation) of layer1: no gemm consumes it -- stored here (hi + lo, as every other image)
            if (NHB_IS_F16 && cx.wrm) note_region(cx, 0, s);
            float* const pr = grow(a.gl.P[0], W);
#pragma unroll
            for (int kb = 0; kb < KBH; ++kb) {
                float4 a4, b4;
                a4.x = nh_from_pc(hh[kb][0]) + nh_from_pc(hl[kb][0]);
                a4.y = nh_from_pc(hh[kb][1]) + nh_from_pc(hl[kb][1]);
                a4.z = nh_from_pc(hh[kb][2]) + nh_from_pc(hl[kb][2]);
                a4.w = nh_from_pc(hh[kb][3]) + nh_from_pc(hl[kb][3]);
                b4.x = nh_from_pc(hh[kb][4]) + nh_from_pc(hl[kb][4]);
                b4.y = nh_from_pc(hh[kb][5]) + nh_from_pc(hl[kb][5]);
                b4.z = nh_from_pc(hh[kb][6]) + nh_from_pc(hl[kb][6]);
                b4.w = nh_from_pc(hh[kb][7]) + nh_from_pc(hl[kb][7]);
                if (NHB_IS_F16) {
                    const float rs = nh_pow2i(-s);
                    a4.x *= rs, a4.y *= rs, a4.z *= rs, a4.w *= rs;
                    b4.x *= rs, b4.y *= rs, b4.z *= rs, b4.w *= rs;
                }
                float* const dst = pr + 32 * (kb >> 1) + 16 * (kb & 1) + 4 * h;
                *(float4*)dst = a4;
                *(float4*)(dst + 8) = b4;
            }
        }
    }
    regions_end(cx, a.rmax);
}

#endif  // NHB_PACK_ONLY

// ---- weight image --------------------------------------------------------------------------------------------------
struct PackBArgs {
    int n_layers;
    int64_t first;  // first word of the split-bf16 images (what lies in front is the fp32 image)
    int64_t base[2 * NH_MAX_LAYERS + 10];  // word offset of every layer image, ascending
};

NH_KERNEL void NHB_KERNEL(k_pack)(const float* __restrict__ params, const int32_t* __restrict__ table, int64_t n, PackBArgs la,
                             float* __restrict__ packed) {
    const int64_t i = la.first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int l = 0;
    while (l + 1 < la.n_layers && i >= la.base[l + 1]) ++l;
    const int64_t r = i - la.base[l];
    const int32_t s = table[i];
    const float v = (s >= 0 ? params[s] : 0.0f) * NHB_WS;  // (fp16 pieces: times 2^8, exact; every gemm takes it out again)
    if (r < 512) {  // bias word: joins accumulators of (scaled weights) x (scaled activations)
        packed[i] = v;
        return;
    }
    const int64_t w = r - 512, blk = w >> 9;  // (kb * nt + t), element lane * 8 + e inside it
    const int q = (int)(w & 511);
    nh_pc* const img = (nh_pc*)(packed + la.base[l] + 512);
    const nh_pc hi = nh_to_pc(v);
    img[(2 * blk) * 512 + q] = hi;
    img[(2 * blk + 1) * 512 + q] = nh_to_pc(v - nh_from_pc(hi));
}

#ifndef NHB_PACK_ONLY
// compute units of the current device (the emulator: 3, so that the CPU suite walks the persistent loop)
int b_compute_units() {
#ifndef NERFHIP_EMU
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        else
            cus = 256;
    }
    return cus;
#else
    return 3;
#endif
}

template <class K>
int b_lds_limit(K kern, int bytes) {
#ifndef NERFHIP_EMU
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        nh_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", bytes, hipGetErrorString(e));
        return NERFHIP_ERR_LAUNCH;
    }
#else
    (void)kern;
    (void)bytes;
#endif
    return NERFHIP_OK;
}
#endif  // NHB_PACK_ONLY

}  // namespace

#ifdef NHB_PACK_ONLY
int NHB_FN(nh_mlp_forward)(nerfhip_plan*, const float*, const NhMlpInput&, int64_t, float*, float*, nerfhip_stream_t) {
    nh_set_error("mlp_fwd: this build carries no one-wave-per-SIMD " NHB_FMT "x3 kernels (the plan's images must be in the geometry of mlp_f16w.hip)");
    return NERFHIP_ERR_UNSUPPORTED;
}
int NHB_FN(nh_mlp_dgrad)(nerfhip_plan*, const float*, const float*, int64_t, const float*, float*, unsigned*, nerfhip_stream_t) {
    nh_set_error("mlp_bwd: this build carries no one-wave-per-SIMD " NHB_FMT "x3 kernels (the plan's images must be in the geometry of mlp_f16w.hip)");
    return NERFHIP_ERR_UNSUPPORTED;
}
#else
int NHB_FN(nh_mlp_forward)(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                        nerfhip_stream_t stream) {
    NH_REQUIRE(M < ((int64_t)1 << 31), "mlp_fwd: at most 2^31 - 1 sample points per call (got %lld)", (long long)M);
    FwdBArgs a;
    memset(&a, 0, sizeof(a));
    a.packed = packed;
    a.packed_bytes = (unsigned)(p->packed_floats * 4);
    a.off = p->pob;
    a.L = p->L;
    a.skip = p->skip;
    a.M = M;
    a.stash = stash;
    // (fp16 plans whose large weight-gradient blocks run on the fp16 MFMAs: the stash's region maxima, zeroed here)
    a.rmax = (NHB_IS_F16 && stash && !p->bjobs.empty()) ? (unsigned*)(stash + nh_stash_floats(p, nh_ceil_div(M, 128) * 4)) : nullptr;
    if (a.rmax) {
        const int rc0 = nh_zero_words(a.rmax, NH_RMAX_WORDS, stream);
        if (rc0) return rc0;
    }
    a.sl = p->stash;
    a.nt = nh_ceil_div(M, 128) * 4;
    a.mode = in.mode;
    a.x = in.x;
    a.dx = p->Dx;
    a.dd = p->Dd;
    a.rays = in.rays;
    a.ray_stride = in.ray_stride;
    a.z = in.z;
    a.S = in.S;
    for (int s = 0; s < 16 * NHB_XBLOCKS; ++s) a.xcol[s] = (signed char)p->xyz_slot_b[s];
    for (int s = 0; s < 16 * NHB_DBLOCKS; ++s) a.dcol[s] = (signed char)p->dir_slot_b[s];
    for (int k = 0; k < 16; ++k) {
        a.fx[k] = p->freqs_xyz[k];
        a.fd[k] = p->freqs_dir[k];
    }
    a.Lx = p->cfg.num_encoding_fn_xyz;
    a.Ld = p->view ? p->cfg.num_encoding_fn_dir : 0;
    a.out = out;
    const int64_t groups = nh_ceil_div(M, 128);
    a.groups = groups;
    // as many workgroups as are resident at once (a workgroup per CU for the 256-wide nets -- its LDS --, two for 128-wide)
    int64_t resident = (int64_t)b_compute_units() * (p->W >= 256 ? 1 : 2);
#ifdef NHB_NOT_PERSISTENT  // (A/B builds only)
    resident = groups;
#endif
    const int64_t grid = groups < resident ? groups : resident;
    int rc = NERFHIP_OK;
#define NH_FWDB_T(WW, VV, TT)                                                                              \
    {                                                                                                      \
        rc = b_lds_limit(NHB_KERNEL(k_mlp_fwd)<WW, VV, TT>, BShape<WW>::LDS_BYTES + NHB_RM_LDS);                \
        if (rc) return rc;                                                                                 \
        NH_LAUNCH_NAMED("k_mlp_fwd_" NHB_FMT "x3<" #WW ", " #VV ", " #TT ">", (NHB_KERNEL(k_mlp_fwd)<WW, VV, TT>), grid, 256,  \
                        BShape<WW>::LDS_BYTES + NHB_RM_LDS, stream, a);                                    \
    }
#define NH_FWDB(WW, VV)              \
    {                                \
        if (stash)                   \
            NH_FWDB_T(WW, VV, true)  \
        else                         \
            NH_FWDB_T(WW, VV, false) \
    }
    if (p->W == 256 && p->view) NH_FWDB(256, true)
    else if (p->W == 256) NH_FWDB(256, false)
    else if (p->W == 128 && p->view) NH_FWDB(128, true)
    else if (p->W == 128) NH_FWDB(128, false)
    else {
        nh_set_error("mlp_fwd: no " NHB_FMT "x3 kernel for kernel width %d", p->W);
        return NERFHIP_ERR_UNSUPPORTED;
    }
#undef NH_FWDB
#undef NH_FWDB_T
    return nh_launch_status("mlp_fwd_" NHB_FMT "x3");
}

int NHB_FN(nh_mlp_dgrad)(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash, float* scratch,
                         unsigned* rmax, nerfhip_stream_t stream) {
    DgradBArgs d;
    memset(&d, 0, sizeof(d));
    d.packed = packed;
    d.packed_bytes = (unsigned)(p->packed_floats * 4);
    d.off = p->pob;
    d.L = p->L;
    d.M = M;
    d.groups = nh_ceil_div(M, 128);
    d.nt = d.groups * 4;
    d.g_out = g_out;
    d.stash = stash;
    d.sl = p->stash;
    d.grad = scratch;
    d.gl = p->grad;
    d.rmax = rmax;
    const int64_t resident = b_compute_units();  // (one wave per SIMD: one workgroup per CU also for the 128-wide nets)
    const int64_t grid = d.groups < resident ? d.groups : resident;
    int rc = NERFHIP_OK;
#define NH_BWDB(WW, VV)                                                                                    \
    {                                                                                                      \
        rc = b_lds_limit(NHB_KERNEL(k_mlp_dgrad)<WW, VV>, BShape<WW>::LDS_BYTES + NHB_RM_LDS);                  \
        if (rc) return rc;                                                                                 \
        NH_LAUNCH_NAMED("k_mlp_dgrad_" NHB_FMT "x3<" #WW ", " #VV ">", (NHB_KERNEL(k_mlp_dgrad)<WW, VV>), grid, 256,           \
                        BShape<WW>::LDS_BYTES + NHB_RM_LDS, stream, d);                                    \
    }
    if (p->W == 256 && p->view) NH_BWDB(256, true)
    else if (p->W == 256) NH_BWDB(256, false)
    else if (p->W == 128 && p->view) NH_BWDB(128, true)
    else if (p->W == 128) NH_BWDB(128, false)
    else {
        nh_set_error("mlp_bwd: no " NHB_FMT "x3 data-gradient kernel for kernel width %d", p->W);
        return NERFHIP_ERR_UNSUPPORTED;
    }
#undef NH_BWDB
    return nh_launch_status("mlp_dgrad_" NHB_FMT "x3");
}

#endif  // NHB_PACK_ONLY

// the split-precision layer images of a plan (behind its fp32 image, if it has one): gather, scale (fp16), split
int NHB_FN(nh_pack_pieces)(nerfhip_plan* plan, const float* params, const int32_t* table, float* packed, nerfhip_stream_t stream) {
    const int64_t n = plan->packed_floats, n32 = plan->packed32_floats;
    PackBArgs la;
    memset(&la, 0, sizeof(la));
    const NhPackedOffsets& o = plan->pob;
    int k = 0;
    la.base[k++] = o.f_layer1;
    for (int i = 0; i < plan->L - 1; ++i) la.base[k++] = o.f_xyz[i];
    la.base[k++] = o.f_head;
    if (plan->view) {
        la.base[k++] = o.f_dir;
        la.base[k++] = o.f_rgb;
    }
    if (nh_prec_level(plan->precision) >= 3) {  // (the order of plan.cpp for_each_spec_b)
        if (plan->view) {
            la.base[k++] = o.b_rgb;
            la.base[k++] = o.b_dir;
        }
        la.base[k++] = o.b_head;
        for (int i = 0; i < plan->L - 1; ++i) la.base[k++] = o.b_xyz[i];
    }
    la.n_layers = k;
    la.first = n32;
    NH_LAUNCH_NAMED("k_pack_" NHB_FMT "x3", NHB_KERNEL(k_pack), nh_ceil_div(n - n32, 256), 256, 0, stream, params, table, n, la, packed);
    return nh_launch_status("pack_weights_plan");
}

#ifndef NHB_F16
extern "C" int nerfhip_pack_weights_plan(nerfhip_plan_t plan, const float* params, const int32_t* table, float* packed,
                                         nerfhip_stream_t stream) {
    NH_REQUIRE(plan && params && table && packed, "pack_weights_plan: bad arguments");
    const int64_t n = plan->packed_floats;
    if (plan->precision == NERFHIP_PRECISION_FP32) return nerfhip_pack_weights(params, table, n, packed, stream);
    const int64_t n32 = plan->packed32_floats;  // (training-capable plans: the fp32 image in front -- a plain gather)
    if (n32 > 0) {
        const int rc = nerfhip_pack_weights(params, table, n32, packed, stream);
        if (rc) return rc;
    }
    return nh_prec_f16(plan->precision) ? nh_pack_pieces_f16(plan, params, table, packed, stream)
                                        : nh_pack_pieces_bf16(plan, params, table, packed, stream);
}
#endif

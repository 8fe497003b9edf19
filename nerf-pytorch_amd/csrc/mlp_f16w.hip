// mlp_f16w.hip -- forward and data-gradient chain of FlexibleNeRFModel (nerf/models.py:233-258) for the fp16-piece plans
// (NERFHIP_PRECISION_F16X3*, include/nerfhip.h) with TWO wavefronts per SIMD: the arithmetic of mlp_bf16.hip / mlp_f16.hip -- every
// operand split into two IEEE fp16 pieces, three MFMAs per product block, per-sample block floating point (that file's header) -- in
// the shape of mlp16.hip: a wave owns 16 samples and multiplies on v_mfma_f32_16x16x32_f16, an 8-wave workgroup owns a 128-sample
// group.  Why: the 32-sample waves of mlp_f16.hip need the whole register file of a SIMD (144 accumulator + 128 operand registers), so
// nothing runs while that one wave converts its accumulators, waits for LDS or queues stash stores (0.30-0.36 of its MFMA roofline
// in training); a 16-sample wave holds 68 + 64 and two of them share a SIMD -- one multiplies while the other
// converts or stores.
//
// Layout: lane l = (sample j = l & 15, group g = l >> 4); accumulator register c of output tile t holds unit 16 t + 4 g + c (the C/D
// layout of the instruction = the layout of mlp16.hip: ReLU mask words and stash rows are the fp32 kernels' own).  The B operand of
// a 32-deep k-block wants 8 k per lane, and k order is free as long as A and B agree: the packed weights use nhw_unit (nh_plan.h), under
// which the accumulators of tiles 2 kb and 2 kb + 1, converted to (hi, lo), ARE k-block kb of the next layer.  Images: per layer 2 KiB
// of fp32 biases, then per (k-block, tile) a 1-KiB block of high and a 1-KiB block of low pieces; the blocks stream L2 -> LDS by
// LDS-DMA in linear order, nhw_chunk_blocks of them per chunk buffer, double buffered, one barrier per chunk; copy pieces go out in
// the first half of a chunk, stash stores in its second half (mlp16.hip: a weight copy queued behind stash stores is what the
// store stream really costs).
#include "nh_device.h"
#include "nh_diag.h"
#include "nh_mlp.h"

namespace {

constexpr float WS = NHB_F16_WSCALE;  // the packed weights (and biases) carry 2^8
constexpr int WS_LOG2 = 8;
static_assert(WS == (float)(1 << WS_LOG2), "weight scale");
constexpr int TARGET_LOG2 = 13;  // a sample's largest operand value lands in [2^13, 2^14)
constexpr int NO_CAP = 100;
// Exponent of a sample whose values are all zero (a dead layer, a zero cotangent behind an opaque surface or in front of a white
// background, a padding sample of the last group): low enough that a bias times 2^60 stays a float.  The value is RESERVED for such
// samples -- a non-zero sample whose exponent comes out at exactly 60 takes 59 (one bit less headroom) -- because note_region must tell
// them apart: a zero sample sets no region bound at all.  (Round 4 let it set 2^(14 - 60) = 2^-46, "never a region's scale": true for
// every region training produces -- the soak's d(pre-activation) bounds are 2^-24 ... 2^-12 -- and wrong for a region whose largest
// value is below 2^-46: its values were then split 2^-46 too low and the weight gradient of cotangents below ~1e-20 lost bits, below
// ~1e-25 everything.  Found by tests/parity_cases.py case_f16x3_range_extremes on the emulator.)
constexpr int ZERO_EXP = 60;
// Every per-sample exponent stays inside [S_MIN, S_LIM]: the scales applied with it -- 2^s, 2^-s, 2^(-8 - s) -- are then exactly the
// powers of two nh_pow2i can represent (it clamps at 2^-126 / 2^127), so what a sample's exponent SAYS and what its values were multiplied
// by never part (nh_shift_to alone reaches 139 for magnitudes near 2^-126).  A sample whose largest value sits below
// 2^(13 - S_LIM) = 2^-97 keeps fewer piece bits instead -- it is zero to fp32's neighbours anyway.
// The lower end never binds: a float's magnitude is below 2^128, so its exponent to [2^13, 2^14) is at least 13 - 127 = S_MIN, and at
// S_MIN all three scales are still normal floats (2^-114, 2^114, 2^106).  (Round 5 clamped at -S_LIM: a sample whose largest value
// reached 2^124 was then scaled to 2^14 ... 2^17 -- above fp16's 65504, Inf pieces, NaN outputs)
constexpr int S_LIM = 110, S_MIN = TARGET_LOG2 - 127;
NH_DEVICE int clamp_exp(int s) { return s < S_MIN ? S_MIN : (s > S_LIM ? S_LIM : s); }
NH_DEVICE int not_zero_exp(int s) { return s == ZERO_EXP ? ZERO_EXP - 1 : s; }
NH_DEVICE int exp_for(unsigned mb, int base_e) {
    return ((mb >> 23) & 255u) == 0u ? ZERO_EXP : not_zero_exp(clamp_exp(base_e + nh_shift_to(mb, TARGET_LOG2)));
}
// ... a layer that also reads encodings caps the hidden exponent at theirs, and a renormalisation moves a sample by at most 2^120 in one
// step (the multiplier is a normal float), within the same limits; an all-zero sample keeps its reserved exponent
NH_DEVICE int step_exp(int so, int base_e, int cap) {
    if (so == ZERO_EXP) return so;
    so = so < cap ? so : cap;
    return not_zero_exp(clamp_exp(so > base_e + 120 ? base_e + 120 : (so < base_e - 120 ? base_e - 120 : so)));
}
// a raw network output from an accumulator that holds WS * 2^s * value
NH_DEVICE float raw_of(float acc, int s) { return acc * nh_pow2i(-WS_LOG2 - s); }

template <int W>
struct WShape {
    static constexpr int TW = W / 16, KB = W / 32, NW = 8;
    static constexpr int CB = nhw_chunk_blocks(W), CHUNK = CB * 2048, BUF = CHUNK + 2048, LDS_BYTES = 2 * BUF;
};
constexpr int RM_LDS = 8 * NH_RMAX_WORDS * 4;  // LDS bytes behind the chunk buffers: the eight waves' region slots
constexpr int XB = NHW_XBLOCKS, DB = NHW_DBLOCKS;

// ReLU bit of value r among the n values of an activation, 32 per word (mlp16.hip nh16_bitpos)
constexpr int w_bitpos(int r, int n) { return ((n - 32 * (r >> 5)) < 32 ? (n - 32 * (r >> 5)) : 32) - 1 - (r & 31); }

struct WCtx {
    char* lds;
    unsigned lds_addr;
    NhDmaSrc dma;
    int buf, lane, wave, g;
    bool landed;    // this wave has already waited for its copy pieces of the chunk about to be multiplied (gemm_w: the mid-chunk wait)
    unsigned* wrm;  // level-4 plans: this wave's NH_RMAX_WORDS region slots in LDS, else NULL
};
// the rows a gemm stored for region `ridx` came from pieces below 2^(TARGET + 1) at per-sample exponent s: note the bound
NH_DEVICE void note_region(const WCtx& cx, int ridx, int s) {
    const int e = s == ZERO_EXP ? 1 : 256 + 14 - s;  // (an all-zero sample bounds nothing: the smallest word)
    const unsigned wm = nh_wave_max_u32((unsigned)(e < 1 ? 1 : (e > 511 ? 511 : e)));
    if (cx.lane == 0 && wm > cx.wrm[ridx]) cx.wrm[ridx] = wm;
}
NH_DEVICE void regions_begin(WCtx& cx, char* lds_tail, unsigned* rmax) {
    cx.wrm = rmax ? (unsigned*)lds_tail + cx.wave * NH_RMAX_WORDS : nullptr;
    if (cx.wrm) cx.wrm[cx.lane] = 0u;
}
NH_DEVICE void regions_end(const WCtx& cx, unsigned* rmax) {
    if (cx.wrm) {
        const unsigned v = cx.wrm[cx.lane];
        if (v != 0u) nh_atomic_max_u32(rmax + cx.lane, v);
    }
}

// `bytes` (a multiple of 1 KiB) of the image, from byte offset `src`, into chunk buffer b at byte offset dst_off: one 1-KiB
// piece per wave-instruction, pieces dealt round-robin to the eight waves
template <int BUF>
NH_DEVICE void w_issue(const WCtx& cx, int64_t src, int bytes, int b, int dst_off) {
    const int pieces = bytes >> 10;
    for (int p = cx.wave; p < pieces; p += 8)
        nh_dma16a(cx.dma, cx.lane * 16, (int)src + p * 1024, cx.lds_addr + (unsigned)(b * BUF + dst_off + p * 1024));
}

// maximum over the four lane groups of a sample (they hold different units of it)
NH_DEVICE unsigned sample_max(unsigned u) {
    unsigned o = (unsigned)nh_shfl_xor_i((int)u, 16);
    u = u > o ? u : o;
    o = (unsigned)nh_shfl_xor_i((int)u, 32);
    return u > o ? u : o;
}

struct MaskW {
    unsigned w[2];
};

// two accumulator tiles (2 kb, 2 kb + 1) -> k-block kb of the next layer's operand pieces: hi = f16(v), lo = f16(v - hi), v = acc *
// mul; BITS: the ReLU bits of the VALUES (v > 0: the fp32 kernels' definition), shift-accumulated in register order
template <bool RELU, bool BITS>
NH_DEVICE void convert_pair(const f32x4& a0, const f32x4& a1, nh_f16x8& oh, nh_f16x8& ol, float mul, MaskW& bits, int r0) {
#ifdef NHW_EXP_NO_EPI  // (NH_DIAG builds only, wrong results: what the kernel costs without the conversions)
    oh[0] = nh_to_f16(a0[0] * mul);
    ol[4] = nh_to_f16(a1[0]);
    return;
#endif
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = (e < 4 ? a0[e & 3] : a1[e & 3]) * mul;
        if (RELU) v = nh_relu(v);
        const nh_f16 hi = nh_to_f16(v);
        oh[e] = hi;
        ol[e] = nh_to_f16(v - nh_from_f16(hi));
        if (BITS) {
            // the ReLU bit of the VALUE (v > 0; v is already >= 0 here), as the fp32 kernels record it -- not of its high piece: where a
            // cap holds a sample's exponent down to its encodings' (skip layers, layers_dir), activations below 2^-24 of the encodings
            // flush to zero pieces, and a bit taken from the piece would gate their unit's gradient off although the unit is active
            // (found by a scale fuzz on the emulator: weights x 1e-3 with zero biases lost fc_feat's / layer1's whole gradient)
            const int r = r0 + e;
            bits.w[r >> 5] = (bits.w[r >> 5] << 1) | nh_pos_bit(v);
        }
    }
}

// bit pattern of the largest value the epilogue will convert (ReLU: of the positive ones; identity: of the magnitudes) over this
// sample's units
template <int NTE, bool RELU>
NH_DEVICE unsigned tile_max_bits(const f32x4* acc) {
    float m = 0.0f;
#pragma unroll
    for (int t = 0; t < NTE; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) m = fmaxf(m, RELU ? acc[t][c] : fabsf(acc[t][c]));
    unsigned u;
    memcpy(&u, &m, 4);
    return sample_max(u);
}

// accumulators (WS * 2^s_in * value) -> operand pieces of value * 2^s_out with the sample's largest magnitude moved to
// [2^13, 2^14) (never above `cap`): the renormalisation step of the data-gradient chain
template <int NT>
NH_DEVICE int renorm_convert(const f32x4* acc, nh_f16x8* oh, nh_f16x8* ol, int s_in, int cap) {
    const int base_e = WS_LOG2 + s_in;
#ifdef NHW_EXP_NO_MAX  // (NH_DIAG builds only, wrong results)
    int so = base_e;
#else
    const unsigned mb = tile_max_bits<NT, false>(acc);
    int so = exp_for(mb, base_e);
#endif
    so = step_exp(so, base_e, cap);
    const float mul = nh_pow2i(so - base_e);
    MaskW none;
    none.w[0] = none.w[1] = 0u;
#pragma unroll
    for (int kb = 0; kb < NT / 2; ++kb) convert_pair<false, false>(acc[2 * kb], acc[2 * kb + 1], oh[kb], ol[kb], mul, none, 0);
    return so;
}

// zero the accumulators whose ReLU bit is 0: value r = 4 t + c of the n = 4 NT values
template <int NT>
NH_DEVICE void gate_tiles(f32x4* acc, const MaskW& mw) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = 4 * t + c;
            acc[t][c] = nh_gate(acc[t][c], mw.w[r >> 5], w_bitpos(r, 4 * NT));
        }
}

// global block index at which input row tile ts (16 rows: half a k-block) of a gemm with nts row tiles is stored: the tiles are dealt
// evenly to the chunks and go out in the SECOND half of their chunk (its first half carries the copy pieces of the next chunk)
#ifndef NHW_STORES_FIRST  // (A/B builds only) 1: the stores in the FIRST half of their chunk, the copy pieces in its second half
#define NHW_STORES_FIRST 0
#endif
constexpr int w_store_block(int ts, int nts, int nblk, int cb) {
    const int nch = (nblk + cb - 1) / cb;
    const int c = (ts * nch) / nts;
    const int b0 = (c * nts + nch - 1) / nch, b1 = ((c + 1) * nts + nch - 1) / nch;  // chunk c holds tiles [b0, b1)
    const int n_c = (nblk - c * cb) < cb ? (nblk - c * cb) : cb;
    const int half = n_c / 2, room = n_c - half;
    if (NHW_STORES_FIRST) return c * cb + ((ts - b0) * (half > 0 ? half : 1)) / (b1 - b0);
#ifdef NHW_STORES_SPREAD  // (A/B builds only) over the whole chunk
    return c * cb + ((ts - b0) * n_c) / (b1 - b0);
#endif
    return c * cb + half + ((ts - b0) * room) / (b1 - b0);
}

// acc[t] = bias + sum over NKA activation k-blocks (ah / al) and NKB encoding k-blocks (xh / xl) of this layer's image at byte offset
// `base`, t < NT; while the last chunk is multiplied the first chunk of the next layer (next_base, next_first bytes) travels.
// EPI (0: none; 1: ReLU; 2: identity): the first NTE output tiles leave as the next layer's operand pieces oh / ol (they may be the
// inputs themselves), converted after the layer's last block; BITS: their ReLU bits -> *bits_out.
// STORE, in_rows / in_mask (training launches; else false / NULL): the gemm stores ITS OWN activation inputs -- the previous layer's output as the
// operand pieces say it, hi + lo, times row_scale (the power of two that turns the sample's scale into plain values) -- one 16-byte
// store per row tile, and that layer's ReLU bits (in_bits) as one 8-byte store.
// DYN (the forward): the block-floating-point bookkeeping -- s_in: exponent of the hidden inputs (of the encoding inputs when there
// are no hidden ones), s_x: exponent the encoding pieces were made at (>= s_in: rescaled once per gemm), cap: the largest exponent
// the outputs may get, *s_out: what they got.
template <int W, int NT, int NKA, int NKB, int EPI = 0, int NTE = 0, bool DYN = false, bool BITS = false, bool STORE = false>
NH_DEVICE void gemm_w(WCtx& cx, const nh_f16x8* ah, const nh_f16x8* al, const nh_f16x8* xh, const nh_f16x8* xl, int64_t base,
                      int64_t next_base, int next_first, f32x4* acc, nh_f16x8* oh = nullptr, nh_f16x8* ol = nullptr,
                      float* in_rows = nullptr, unsigned* in_mask = nullptr, const MaskW* in_bits = nullptr, float row_scale = 1.0f,
                      int s_in = 0, int s_x = 0, int cap = NO_CAP, int* s_out = nullptr, MaskW* bits_out = nullptr, int ridx = -1,
                      float bias_mul = 0.0f, bool use_bias_mul = false) {
    constexpr int NK = NKA + NKB, NBLK = NK * NT, CB = WShape<W>::CB, BUF = WShape<W>::BUF, NCH = (NBLK + CB - 1) / CB;
    // A sample whose hidden inputs are ALL ZERO (a dead layer: ZERO_EXP, which no cap touches -- step_exp) carries any exponent equally
    // well: in a gemm that also reads encodings it takes theirs, so that the encoding pieces are not rescaled by 2^(60 - s_x) out of
    // fp16's range.  What the stored input rows say about their region is still "nothing" (s_note).
    const int s_note = s_in;
    if (DYN && NKA > 0 && NKB > 0 && s_in == ZERO_EXP) s_in = s_x;
    constexpr int NTS = STORE ? 2 * NKA : 0;  // input row tiles a training launch stores (STORE: in_rows is not NULL)
    // encoding pieces made at 2^s_x, wanted at the hidden inputs' 2^s_in (<= s_x): rescaled once (v_pk_mul_f16 by a power of two)
    constexpr bool RESCALE = DYN && NKA > 0 && NKB > 0;
    nh_f16x8 sxh[RESCALE ? NKB : 1], sxl[RESCALE ? NKB : 1];
    if (RESCALE) {
        const float xf = nh_pow2i(s_in - s_x);
#pragma unroll
        for (int k = 0; k < NKB; ++k) {
            sxh[k] = nh_f16x8_scale(xh[k], xf);
            sxl[k] = nh_f16x8_scale(xl[k], xf);
        }
    }
    auto store_tile = [&](int ts) {
        const int kb = ts >> 1, o = 4 * (ts & 1);
        float v0 = nh_from_f16(ah[kb][o]) + nh_from_f16(al[kb][o]);
        float v1 = nh_from_f16(ah[kb][o + 1]) + nh_from_f16(al[kb][o + 1]);
        float v2 = nh_from_f16(ah[kb][o + 2]) + nh_from_f16(al[kb][o + 2]);
        float v3 = nh_from_f16(ah[kb][o + 3]) + nh_from_f16(al[kb][o + 3]);
        v0 *= row_scale, v1 *= row_scale, v2 *= row_scale, v3 *= row_scale;
#ifdef NHW_EXP_NO_STORE  // (NH_DIAG builds only, wrong results: what the stores themselves cost)
        if (v0 != 1.2345e-30f || v1 != 5.4321e-30f) return;
#endif
        nh_store4(in_rows + 16 * ts + 4 * cx.g, v0, v1, v2, v3);  // units 16 ts + 4 g .. + 3
    };
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        // this wave's pieces of the current chunk have landed (and its stash stores: vmcnt counts them) ...
#ifdef NHW_MID_WAIT
        if (!cx.landed) nh_wait_vmem();
        cx.landed = false;
#else
        nh_wait_vmem();
#endif
#ifndef NHW_EXP_NO_BARRIER  // (NH_DIAG builds only, wrong results: what the chunk barriers cost)
        nh_block_sync();   // ... and everyone's; nobody still reads the other buffer
#endif
        // the next chunk (or the next layer's first one) goes to the other buffer WHILE this one is multiplied
        int64_t dsrc = 0;
        int dpieces = 0, ddst = 0;
        if (c + 1 < NCH) {
            const int nb = NBLK - (c + 1) * CB < CB ? NBLK - (c + 1) * CB : CB;
            dsrc = base + 2048 + (int64_t)(c + 1) * CB * 2048, dpieces = nb * 2, ddst = 2048;
        } else if (next_first > 0) {
            dsrc = next_base, dpieces = next_first >> 10, ddst = 0;
        }
#ifdef NHW_EXP_NO_STREAM  // (NH_DIAG builds only, wrong results: what the kernel costs without the weight stream)
        dpieces = dpieces < 8 ? dpieces : 8;
#endif
        int dnext = cx.wave;  // this wave's next piece
        auto dma_step = [&]() {
            if (dnext < dpieces) {
#ifdef NHW_EXP_SAME_SRC  // (NH_DIAG builds only, wrong results: every piece re-reads the image's first 8 KiB -- issue and LDS cost without the L2 traffic)
                nh_dma16a(cx.dma, cx.lane * 16, (dnext & 7) * 1024, cx.lds_addr + (unsigned)((cx.buf ^ 1) * BUF + ddst + dnext * 1024));
#else
                nh_dma16a(cx.dma, cx.lane * 16, (int)dsrc + dnext * 1024, cx.lds_addr + (unsigned)((cx.buf ^ 1) * BUF + ddst + dnext * 1024));
#endif
                dnext += 8;
            }
        };
        const char* const buf = cx.lds + cx.buf * BUF;
        if (c == 0) {
            if (in_mask && NKA > 0) {  // (after the chunk's wait: in front of it the layer would wait for this write)
                in_mask[0] = in_bits->w[0];
                in_mask[1] = in_bits->w[1];
            }
            // the accumulators start at the bias of their rows: register i of tile t holds row 16 t + 4 g + i
            // (DYN: the products carry 2^s_in, so must the bias; use_bias_mul, the data-gradient chain's head: the bias row holds
            // fc_alpha's weights, the factor is the sample's d(sigma raw) at the inputs' exponent)
            const bool scaled = DYN || use_bias_mul;
            const float bsc = use_bias_mul ? bias_mul : (DYN ? nh_pow2i(s_in) : 1.0f);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 b4 = *(const float4*)(buf + (16 * t + 4 * cx.g) * 4);
                acc[t][0] = scaled ? b4.x * bsc : b4.x;
                acc[t][1] = scaled ? b4.y * bsc : b4.y;
                acc[t][2] = scaled ? b4.z * bsc : b4.z;
                acc[t][3] = scaled ? b4.w * bsc : b4.w;
            }
        }
        const char* const wb = buf + 2048 + cx.lane * 16;
        // The blocks of this chunk in pairs, software-pipelined by hand: the weight pieces of the NEXT pair are read from LDS before
        // the MFMAs of this one issue, each side of a scheduling fence; the two blocks of a pair accumulate into different tiles, so
        // that no MFMA reads the accumulator its predecessor wrote.
        constexpr int nblk_c_max = CB;
        const int nblk = NBLK - c * CB < CB ? NBLK - c * CB : CB;  // blocks in this chunk
        nh_f16x8 wph[4], wpl[4];
        auto load = [&](int i) {
            wph[i & 3] = *(const nh_f16x8*)(wb + (2 * i) * 1024);
#ifdef NHW_EXP_HALF_LDS  // (NH_DIAG builds only, wrong results: half of the operand reads)
            wpl[i & 3] = wph[i & 3];
#else
            wpl[i & 3] = *(const nh_f16x8*)(wb + (2 * i + 1) * 1024);
#endif
        };
        if (0 < nblk) load(0);
        if (1 < nblk) load(1);
#pragma unroll
        for (int ip = 0; ip < nblk_c_max; ip += 2) {
            if (ip < nblk) {
                const bool two = ip + 1 < nblk;
                if (ip + 2 < nblk) load(ip + 2);
                if (ip + 3 < nblk) load(ip + 3);
                // one copy piece per pair: the copy is out before the chunk's second half ...
#ifdef NHW_DMA_BURST  // (A/B builds only) all of a wave's pieces at the chunk's start: forward +7 %, data gradient -2 %
                if (ip == 0)
                    while (dnext < dpieces) dma_step();
#elif defined(NHW_DMA_ONE)  // (A/B builds only) one piece per pair: forward +5 %, data gradient +4 %
                if (!(NHW_STORES_FIRST && NTS > 0) || ip >= nblk / 2) dma_step();
#else  // two pieces per pair: the copy is out after a quarter of the chunk
                dma_step();
                dma_step();
#endif
#ifdef NHW_MID_WAIT
                // Training launches, chunks of at least 20 blocks: the wave waits for its copy pieces HERE, half a chunk after it
                // issued them (2 per pair: at most 10 by pair 5) and before its first store of this chunk -- the next chunk's barrier
                // then needs no vmcnt wait, and the stores stay in flight across it (they are waited for here, half a chunk later)
                if (NTS > 0 && nblk >= 20 && ip == ((nblk / 2) & ~1)) {
                    while (dnext < dpieces) dma_step();
                    nh_wait_vmem();
                    cx.landed = true;
                }
#endif
                if (NTS > 0) {  // ... which carries the stores
#pragma unroll
                    for (int ts = 0; ts < NTS; ++ts) {
                        const int sb = w_store_block(ts, NTS, NBLK, CB);
                        if (sb == c * CB + ip || (two && sb == c * CB + ip + 1)) store_tile(ts);
                    }
                }
                nh_sched_fence();
                const int g0 = c * CB + ip, kb0 = g0 / NT, t0 = g0 % NT;
                const int g1 = two ? g0 + 1 : g0, kb1 = g1 / NT, t1 = g1 % NT;
                const nh_f16x8 b0h = kb0 < NKA ? ah[kb0 < NKA ? kb0 : 0] : (RESCALE ? sxh[kb0 >= NKA ? kb0 - NKA : 0] : xh[kb0 >= NKA ? kb0 - NKA : 0]);
                const nh_f16x8 b0l = kb0 < NKA ? al[kb0 < NKA ? kb0 : 0] : (RESCALE ? sxl[kb0 >= NKA ? kb0 - NKA : 0] : xl[kb0 >= NKA ? kb0 - NKA : 0]);
                const nh_f16x8 b1h = kb1 < NKA ? ah[kb1 < NKA ? kb1 : 0] : (RESCALE ? sxh[kb1 >= NKA ? kb1 - NKA : 0] : xh[kb1 >= NKA ? kb1 - NKA : 0]);
                const nh_f16x8 b1l = kb1 < NKA ? al[kb1 < NKA ? kb1 : 0] : (RESCALE ? sxl[kb1 >= NKA ? kb1 - NKA : 0] : xl[kb1 >= NKA ? kb1 - NKA : 0]);
                const nh_f16x8 w0h = wph[ip & 3], w0l = wpl[ip & 3], w1h = wph[(ip + 1) & 3], w1l = wpl[(ip + 1) & 3];
                acc[t0] = nh_mfma_f16_16(w0l, b0h, acc[t0]);  // (the small terms first)
                if (two) acc[t1] = nh_mfma_f16_16(w1l, b1h, acc[t1]);
                acc[t0] = nh_mfma_f16_16(w0h, b0l, acc[t0]);
                if (two) acc[t1] = nh_mfma_f16_16(w1h, b1l, acc[t1]);
                acc[t0] = nh_mfma_f16_16(w0h, b0h, acc[t0]);
                if (two) acc[t1] = nh_mfma_f16_16(w1h, b1h, acc[t1]);
            }
        }
        while (dnext < dpieces) dma_step();  // (whatever the pairs did not cover: short chunks in front of long ones)
        cx.buf ^= 1;
    }
    if (NTS > 0 && cx.wrm && ridx >= 0) note_region(cx, ridx, s_note);
    if (EPI != 0) {
        float mul = 1.0f / WS;
#ifdef NHW_EXP_NO_MAX  // (NH_DIAG builds only, wrong results: what the per-sample exponent search costs)
        if (DYN) *s_out = s_in;
        if (false) {
#else
        if (DYN) {
#endif
            // the accumulators hold WS * 2^s_in * value: move the sample's largest output to [2^13, 2^14) -- unless the next
            // layer's encodings sit lower -- and remember the exponent the pieces now carry
            // (collecting the maximum tile by tile under the last k-block's MFMAs instead measured 3.5 % SLOWER on the forward:
            // profiles/r04_f16w_ab.txt)
            const unsigned mb = tile_max_bits<NTE, EPI == 1>(acc);
            const int base_e = WS_LOG2 + s_in;
            int so = exp_for(mb, base_e);
            so = step_exp(so, base_e, cap);
            mul = nh_pow2i(so - base_e);
            *s_out = so;  // the pieces made below are those of value * 2^so
        }
        MaskW bits;
        bits.w[0] = bits.w[1] = 0u;
#pragma unroll
        for (int kb = 0; kb < NTE / 2; ++kb) convert_pair<EPI == 1, BITS>(acc[2 * kb], acc[2 * kb + 1], oh[kb], ol[kb], mul, bits, 8 * kb);
        if (BITS) *bits_out = bits;
    }
}

NH_DEVICE void put_pair(nh_f16x8& oh, nh_f16x8& ol, int e, float v) {
    const nh_f16 hi = nh_to_f16(v);
    oh[e] = hi;
    ol[e] = nh_to_f16(v - nh_from_f16(hi));
}
NH_DEVICE float wsel3(int a, float x, float y, float z) { return a == 0 ? x : (a == 1 ? y : z); }

// eight fp32 slot values of one k-block -> operand pieces of v * sc (the sample's encoding exponent), and (training) -> this
// sample's row of the slot region (the plain values)
NH_DEVICE void put_block(nh_f16x8& oh, nh_f16x8& ol, const float* v, float* slot_row, float sc) {
#pragma unroll
    for (int e = 0; e < 8; ++e) put_pair(oh, ol, e, v[e] * sc);
    if (slot_row) {
        nh_store4(slot_row, v[0], v[1], v[2], v[3]);
        nh_store4(slot_row + 4, v[4], v[5], v[6], v[7]);
    }
}

// the encoding slots of lane group g (plan.cpp build_slot_map_b): slot 32 kb + 8 g + e; pair slot >> 1 = 3 f + axis.
// slot_row (training): this sample's row of the stash's slot region; the lane writes its slots 32 kb + 8 g .. + 7.
template <int NB>
NH_DEVICE void encode_w(nh_f16x8* oh, nh_f16x8* ol, float x, float y, float z, int g, const float* freqs, int Lf, float* slot_row, float sc) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pr = 16 * kb + 4 * g + q;
            const bool valid = pr < 3 * Lf;
            const int f = pr / 3, ax = pr - 3 * f;
            float s, c;
            nh_sincos(wsel3(ax, x, y, z) * freqs[f < 16 ? f : 15], &s, &c);
            float v0 = valid ? s : 0.0f, v1 = valid ? c : 0.0f;
            if (kb == NB - 1 && q == 2 && g == 3) v0 = x, v1 = y;  // slots NS-4, NS-3 (never a valid pair: 6 L <= NS - 4)
            if (kb == NB - 1 && q == 3 && g == 3) v0 = z, v1 = 0.0f;
            v[2 * q] = v0;
            v[2 * q + 1] = v1;
        }
        put_block(oh[kb], ol[kb], v, slot_row ? slot_row + 32 * kb + 8 * g : nullptr, sc);
    }
}
// the same slots gathered from a caller-encoded row (mode 0)
template <int NB>
NH_DEVICE void gather_w(nh_f16x8* oh, nh_f16x8* ol, const float* row, const signed char* col, int g, float* slot_row, float sc) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = (int)col[32 * kb + 8 * g + e];
            v[e] = c >= 0 ? row[c] : 0.0f;
        }
        put_block(oh[kb], ol[kb], v, slot_row ? slot_row + 32 * kb + 8 * g : nullptr, sc);
    }
}
// exponent for a sample's encodings: its largest magnitude `m` (all four lane groups) to [2^13, 2^14)
NH_DEVICE int enc_exponent(float m) {
    unsigned u;
    memcpy(&u, &m, 4);
    return exp_for(sample_max(u), 0);
}
template <int NB>
NH_DEVICE float gather_max_w(const float* row, const signed char* col, int g) {
    float m = 0.0f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = (int)col[32 * kb + 8 * g + e];
            m = fmaxf(m, c >= 0 ? fabsf(row[c]) : 0.0f);
        }
    return m;
}

struct FwdWArgs {
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
    signed char xcol[32 * XB];
    signed char dcol[32 * DB];
    float fx[16], fd[16];
    int Lx, Ld;
    float* out;
    float* stash;      // TRAIN launches: the activation stash
    unsigned* rmax;    // level-4 plans: per-region maxima of what the stash holds (H[k] -> k, FEAT -> L), else NULL
    NhStashLayout sl;
    int64_t nt;        // 32-sample tiles of the launch (4 per 128-sample group)
    // a forward over a compaction list (the recomputing backward), or NULLs: slot c computes sample cidx[c] and writes that sample's
    // stash rows and ReLU masks AT SLOT c; cstats[NH_CSTAT_ACTIVE] slots carry a sample; `out` may be NULL
    const int* cidx;
    const int* cstats;
};

// TRAIN: the launch also writes the activation stash -- encoding slots, every layer's fp32 output rows (plain values), ReLU masks -- in
// the format k_mlp_dgrad16 / k_mlp_dgrad_f16x3w / k_wgrad / k_wgrad_f16x3 read
template <int W, bool VIEW, bool TRAIN>
NH_KERNEL void NH_LB(512, 2) k_mlp_fwd_f16x3w(FwdWArgs a) {
    constexpr int TW = WShape<W>::TW, KB = WShape<W>::KB, BUF = WShape<W>::BUF, CB = WShape<W>::CB;
    // (a launch over a list: its groups are the list's; a workgroup without one has nothing to stream)
    const bool listed = TRAIN && a.cidx;
    const int n_slots = listed ? nh_uload_i32(a.cstats, NH_CSTAT_ACTIVE) : 0;
    const int64_t groups = listed ? (int64_t)((n_slots + 127) >> 7) : a.groups;
    if ((int64_t)blockIdx.x >= groups) return;
    NH_DYN_LDS(lds_raw);
    WCtx cx;
    cx.lds = lds_raw;
    cx.lds_addr = nh_lds_addr((const float*)lds_raw);
    cx.dma = nh_dma_src(a.packed, a.packed_bytes);
    cx.buf = 0;
    cx.landed = false;
    cx.lane = nh_lane();
    cx.wave = nh_wave_in_block();
    cx.g = cx.lane >> 4;
    regions_begin(cx, lds_raw + WShape<W>::LDS_BYTES, TRAIN ? a.rmax : nullptr);
    const int g = cx.g, j = cx.lane & 15;
    const NhPackedOffsets& po = a.off;
    auto first = [](int nk, int nt) { return nhw_first_bytes(nk * nt, W); };
    (void)CB;

    // the first weights travel while the encodings are formed
    w_issue<BUF>(cx, po.f_layer1 * 4, first(XB, TW), 0, 0);

    // persistent workgroups (one per CU): each walks over its 128-sample groups; the last layer of a group already streams layer1 of
    // the next
    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const bool again = grp + gridDim.x < groups;
        const int64_t m = grp * 128 + cx.wave * 16 + j;
        bool valid = m < a.M;
        int64_t mc = valid ? m : a.M - 1;  // the sample this lane computes
        if (listed) {                      // (slot m of a list: its sample; padding slots compute sample cidx[m] = 0 and are never read)
            valid = m < (int64_t)n_slots;
            mc = (int64_t)a.cidx[m];
        }
        // training: this sample's row of a stash region ([32-sample tile][sample][rows]; whole groups are written, clamped samples
        // included), and this lane's mask words ([16-sample wave tile][mask][64 lanes][2 words])
        const int64_t tile32 = grp * 4 + (cx.wave >> 1);
        const int s32 = 16 * (cx.wave & 1) + j;
        auto srow = [&](const NhRegion& R, int rows) -> float* {
            return a.stash + (size_t)32 * (size_t)a.nt * (size_t)R.row_prefix + ((size_t)tile32 * 32 + (size_t)s32) * (size_t)rows;
        };
        auto smask = [&](int idx) -> unsigned* {
            return (unsigned*)(a.stash + (size_t)32 * (size_t)a.nt * (size_t)a.sl.total_rows) +
                   ((size_t)(grp * 8 + cx.wave) * (size_t)a.sl.n_masks + (size_t)idx) * 128 + (size_t)cx.lane * 2;
        };
        nh_f16x8 xh[XB], xl[XB];
        const int ray_i = a.mode == 0 ? 0 : (int)(mc / a.S);
        int ex = 0, ed = 0, s = 0;  // per-sample exponents: ex / ed the encodings', s the activations'
        if (a.mode == 0) {
            const float* const row = a.x + (size_t)mc * (size_t)(a.dx + a.dd);
            ex = enc_exponent(gather_max_w<XB>(row, a.xcol, g));
            gather_w<XB>(xh, xl, row, a.xcol, g, TRAIN ? srow(a.sl.X, 32 * XB) : nullptr, nh_pow2i(ex));
        } else {
            const float* const rr = a.rays + (size_t)ray_i * a.ray_stride;
            const float zz = a.z[mc];
            // pts = ro + rd * z   (nerf/train_utils.py:67,107)
            const float px = rr[0] + rr[3] * zz, py = rr[1] + rr[4] * zz, pz = rr[2] + rr[5] * zz;
            ex = enc_exponent(fmaxf(fmaxf(fabsf(px), fabsf(py)), fmaxf(fabsf(pz), 1.0f)));  // (sines and cosines: <= 1)
            encode_w<XB>(xh, xl, px, py, pz, g, a.fx, a.Lx, TRAIN ? srow(a.sl.X, 32 * XB) : nullptr, nh_pow2i(ex));
        }
        s = ex;
        // (the slot rows hold plain values, bounded by the sample's encoding exponent: the weight-gradient kernel's guest blocks)
        if (TRAIN && cx.wrm) note_region(cx, nh_rmax_x(a.L), ex);

        f32x4 acc[TW + 1];
        nh_f16x8 hh[KB], hl[KB];  // the current activations as operand pieces (a layer's output replaces them in place)
        MaskW bits;               // ReLU bits of the current activations (training)
        bits.w[0] = bits.w[1] = 0u;
        {
            const bool more = a.L > 1;
            // no activation after layer1 (models.py:238)
            gemm_w<W, TW, 0, XB, 2, TW, true, false>(cx, nullptr, nullptr, xh, xl, po.f_layer1 * 4, (more ? po.f_xyz[0] : po.f_head) * 4,
                                                     more ? first(KB, TW) : (VIEW ? first(KB, TW + 1) : first(KB, 1)), acc, hh, hl, nullptr,
                                                     nullptr, nullptr, 1.0f, ex, ex, NO_CAP, &s);
        }
        for (int i = 0; i < a.L - 1; ++i) {
            const bool sk = (i % a.skip == 0) && i > 0;
            const bool more = i + 1 < a.L - 1;
            const bool nsk = more && ((i + 1) % a.skip == 0);
            const int64_t nxt = (more ? po.f_xyz[i + 1] : po.f_head) * 4;
            const int nfirst = more ? (nsk ? first(KB + XB, TW) : first(KB, TW)) : (VIEW ? first(KB, TW + 1) : first(KB, 1));
            // (training: the gemm stores its inputs H_i and their ReLU mask i - 1; H_0 = layer1's output has none)
            float* const in_rows = TRAIN ? srow(a.sl.H[i], W) : nullptr;
            unsigned* const in_mask = (TRAIN && i > 0) ? smask(i - 1) : nullptr;
            const int cap = nsk ? ex : NO_CAP;  // (a skip layer next: its hidden inputs must not sit above the encodings)
            const float rsc = TRAIN ? nh_pow2i(-s) : 1.0f;  // (the stash rows are plain values)
            const MaskW in_bits = bits;
            if (sk)
                gemm_w<W, TW, KB, XB, 1, TW, true, TRAIN, TRAIN>(cx, hh, hl, xh, xl, po.f_xyz[i] * 4, nxt, nfirst, acc, hh, hl, in_rows, in_mask, &in_bits,
                                                          rsc, s, ex, cap, &s, &bits, TRAIN ? i : -1);
            else
                gemm_w<W, TW, KB, 0, 1, TW, true, TRAIN, TRAIN>(cx, hh, hl, nullptr, nullptr, po.f_xyz[i] * 4, nxt, nfirst, acc, hh, hl, in_rows, in_mask,
                                                         &in_bits, rsc, s, s, cap, &s, &bits, TRAIN ? i : -1);
        }
        if (VIEW) {
            nh_f16x8 dh[DB], dl[DB];
            if (a.mode == 0) {
                const float* const row = a.x + (size_t)mc * (size_t)(a.dx + a.dd) + a.dx;
                ed = enc_exponent(gather_max_w<DB>(row, a.dcol, g));
                gather_w<DB>(dh, dl, row, a.dcol, g, TRAIN ? srow(a.sl.D, 32 * DB) : nullptr, nh_pow2i(ed));
            } else {
                const float* const rr = a.rays + (size_t)ray_i * a.ray_stride;
                ed = enc_exponent(fmaxf(fmaxf(fabsf(rr[8]), fabsf(rr[9])), fmaxf(fabsf(rr[10]), 1.0f)));
                encode_w<DB>(dh, dl, rr[8], rr[9], rr[10], g, a.fd, a.Ld, TRAIN ? srow(a.sl.D, 32 * DB) : nullptr, nh_pow2i(ed));
            }
            if (TRAIN && cx.wrm) note_region(cx, nh_rmax_d(a.L), ed);
            // tiles 0..TW-1: feat = relu(fc_feat(h)); tile TW row 0: fc_alpha(h), raw (models.py:248-249)
            // (training: each gemm stores its own inputs -- H_{L-1} and mask L - 2, FEAT and mask L - 1, DIRH and mask L)
            const int s_head = s;  // (the head's inputs: fc_alpha's raw row comes out at WS * 2^s_head)
            MaskW in_bits = bits;
            gemm_w<W, TW + 1, KB, 0, 1, TW, true, TRAIN, TRAIN>(cx, hh, hl, nullptr, nullptr, po.f_head * 4, po.f_dir * 4, first(KB + DB, TW / 2), acc, hh, hl,
                                                         TRAIN ? srow(a.sl.H[a.L - 1], W) : nullptr, (TRAIN && a.L > 1) ? smask(a.L - 2) : nullptr,
                                                         &in_bits, TRAIN ? nh_pow2i(-s) : 1.0f, s, s, ed, &s, &bits, TRAIN ? a.L - 1 : -1);
            const float alpha = raw_of(acc[TW][0], s_head);
            in_bits = bits;
            gemm_w<W, TW / 2, KB, DB, 1, TW / 2, true, TRAIN, TRAIN>(cx, hh, hl, dh, dl, po.f_dir * 4, po.f_rgb * 4, first(KB / 2, 1), acc, hh, hl,
                                                              TRAIN ? srow(a.sl.FEAT, W) : nullptr, TRAIN ? smask(a.L - 1) : nullptr, &in_bits,
                                                              TRAIN ? nh_pow2i(-s) : 1.0f, s, ed, NO_CAP, &s, &bits, TRAIN ? a.L : -1);
            in_bits = bits;
            gemm_w<W, 1, KB / 2, 0, 0, 0, true, false, TRAIN>(cx, hh, hl, nullptr, nullptr, po.f_rgb * 4, po.f_layer1 * 4, again ? first(XB, TW) : 0, acc,
                                                       nullptr, nullptr, TRAIN ? srow(a.sl.DIRH, W / 2) : nullptr, TRAIN ? smask(a.L) : nullptr,
                                                       &in_bits, TRAIN ? nh_pow2i(-s) : 1.0f, s, s);
            if (valid && g == 0 && a.out) {
                float4 r4;
                r4.x = raw_of(acc[0][0], s);
                r4.y = raw_of(acc[0][1], s);
                r4.z = raw_of(acc[0][2], s);
                r4.w = alpha;
                *(float4*)(a.out + (size_t)m * 4) = r4;
            }
        } else {
            const MaskW in_bits = bits;
            gemm_w<W, 1, KB, 0, 0, 0, true, false, TRAIN>(cx, hh, hl, nullptr, nullptr, po.f_head * 4, po.f_layer1 * 4, again ? first(XB, TW) : 0, acc, nullptr,
                                                   nullptr, TRAIN ? srow(a.sl.H[a.L - 1], W) : nullptr,
                                                   (TRAIN && a.L > 1) ? smask(a.L - 2) : nullptr, &in_bits, TRAIN ? nh_pow2i(-s) : 1.0f, s, s, NO_CAP,
                                                   nullptr, nullptr, TRAIN ? a.L - 1 : -1);  // fc_out (models.py:256)
            if (valid && g == 0 && a.out) {
                float4 r4;
                r4.x = raw_of(acc[0][0], s);
                r4.y = raw_of(acc[0][1], s);
                r4.z = raw_of(acc[0][2], s);
                r4.w = raw_of(acc[0][3], s);
                *(float4*)(a.out + (size_t)m * 4) = r4;
            }
        }
    }  // (groups of this workgroup)
    regions_end(cx, a.rmax);
}

// ---- the data-gradient chain on the same loop ----------------------------------------------------------------------------------
// dpre_{k-1} = relu'(H_{k-1}) * (W_{k-1}^T dpre_k), walked from d(raw output) down to layer1 (mlp16.hip k_mlp_dgrad16 is the fp32
// original, mlp_bf16.hip k_mlp_dgrad the one-wave-per-SIMD form of this arithmetic).  d(pre-activation) lives in registers as operand
// pieces, renormalised per sample and layer; every gemm stores its own input -- the d(pre-activation) image the weight-gradient kernels
// read, plain fp32 values -- and the ReLU mask of the layer below is fetched BEFORE the gemm that needs it afterwards.
struct DgradWArgs {
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
    unsigned* rmax;  // level-4 plans: per-region maxima of the images written (P[k] -> k, PFEAT -> L, PDIR -> L + 1), else NULL
    // compacted backward (compact.hip), or NULLs: slot c of the launch is sample cidx[c] -- d(raw output) and the ReLU masks are
    // gathered by it, the d(pre-activation) images are written in slot order; cstats[NH_CSTAT_ACTIVE] slots carry a sample
    const int* cidx;
    const int* cstats;
    int cstash_listed;  // the stash is in list order too (a recomputed one): the ReLU masks of slot c are at slot c
};

template <int W, bool VIEW>
NH_KERNEL void NH_LB(512, 2) k_mlp_dgrad_f16x3w(DgradWArgs a) {
    constexpr int TW = WShape<W>::TW, KB = WShape<W>::KB, BUF = WShape<W>::BUF;
    // compacted: the launch's groups are those of the sample list; a workgroup without one has nothing to stream
    const int n_slots = a.cidx ? nh_uload_i32(a.cstats, NH_CSTAT_ACTIVE) : 0;
    const int64_t groups = a.cidx ? (int64_t)((n_slots + 127) >> 7) : a.groups;
    if ((int64_t)blockIdx.x >= groups) return;
    NH_DYN_LDS(lds_raw);
    WCtx cx;
    cx.lds = lds_raw;
    cx.lds_addr = nh_lds_addr((const float*)lds_raw);
    cx.dma = nh_dma_src(a.packed, a.packed_bytes);
    cx.buf = 0;
    cx.landed = false;
    cx.lane = nh_lane();
    cx.wave = nh_wave_in_block();
    cx.g = cx.lane >> 4;
    regions_begin(cx, lds_raw + WShape<W>::LDS_BYTES, a.rmax);
    const int g = cx.g, j = cx.lane & 15, L = a.L;
    const NhPackedOffsets& po = a.off;
    auto first = [](int nk, int nt) { return nhw_first_bytes(nk * nt, W); };
    const int64_t first_img = (VIEW ? po.b_rgb : po.b_head) * 4;
    const int first_bytes = VIEW ? first(1, TW / 2) : first(1, TW);
    w_issue<BUF>(cx, first_img, first_bytes, 0, 0);

    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const bool again = grp + gridDim.x < groups;
        int64_t m = grp * 128 + cx.wave * 16 + j;  // this lane's slot, then its sample
        bool valid = m < a.M;
        if (a.cidx) {
            valid = m < (int64_t)n_slots;
            m = valid ? (int64_t)a.cidx[m] : 0;
        }
        const int64_t tile32 = grp * 4 + (cx.wave >> 1);
        const int s32 = 16 * (cx.wave & 1) + j;
        float go[4] = {0.f, 0.f, 0.f, 0.f};  // d(raw output) of this lane's sample (zero beyond M / the list: nothing flows)
        if (valid) {
            const float4 t4 = *(const float4*)(a.g_out + (size_t)m * 4);
            go[0] = t4.x, go[1] = t4.y, go[2] = t4.z, go[3] = t4.w;
        }
        int s = 0;  // the per-sample exponent of the current d(pre-activation) pieces
        auto grow = [&](const NhRegion& R, int rows) -> float* {
            return a.grad + (size_t)32 * (size_t)a.nt * (size_t)R.row_prefix + ((size_t)tile32 * 32 + (size_t)s32) * (size_t)rows;
        };
        // ReLU mask `idx` of this lane's units: the forward wrote the words of wave tile (sample >> 4), lane 16 g + (sample & 15) --
        // this very lane's in the dense backward
        const bool mask_gather = a.cidx && !a.cstash_listed;
        const int64_t wave_tile = mask_gather ? (m >> 4) : grp * 8 + cx.wave;
        const int mask_lane = mask_gather ? 16 * g + (int)(m & 15) : cx.lane;
        auto get_mask = [&](int idx) -> MaskW {
            const unsigned* p = (const unsigned*)(a.stash + (size_t)32 * (size_t)a.nt * (size_t)a.sl.total_rows) +
                                ((size_t)wave_tile * (size_t)a.sl.n_masks + (size_t)idx) * 128 + (size_t)mask_lane * 2;
            MaskW mw;
            mw.w[0] = p[0];
            mw.w[1] = p[1];
            return mw;
        };
        {  // POUT (32 rows): rows 0..2 d(rgb raw), row 3 d(sigma raw), the rest zero; lane group g writes rows 8 g .. 8 g + 7
            float* const pr = grow(a.gl.POUT, 32) + 8 * g;
            const bool g0 = g == 0;
            nh_store4(pr, g0 ? go[0] : 0.0f, g0 ? go[1] : 0.0f, g0 ? go[2] : 0.0f, g0 ? go[3] : 0.0f);
            nh_store4(pr + 4, 0.0f, 0.0f, 0.0f, 0.0f);
            if (cx.wrm) {  // (the region's bound: all four cotangents -- fc_alpha's row rides on a large weight-gradient block)
                const float mp = fmaxf(fmaxf(fabsf(go[0]), fabsf(go[1])), fmaxf(fabsf(go[2]), fabsf(go[3])));
                unsigned up;
                memcpy(&up, &mp, 4);
                note_region(cx, nh_rmax_pout(L), exp_for(up, 0));
            }
        }
        f32x4 acc[TW];
        nh_f16x8 hh[KB], hl[KB];   // d(pre-activation) of the layer just finished, as operand pieces
        nh_f16x8 d1h[1], d1l[1];   // the one k-block of d(raw output): elements 0..2 (0..3 without viewdirs) of lane group 0
        {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (g == 0) {
                v[0] = go[0], v[1] = go[1], v[2] = go[2];
                if (!VIEW) v[3] = go[3];
            }
            // (all lane groups of the sample agree: they loaded the same cotangent)
            const float mg = fmaxf(fmaxf(fabsf(go[0]), fabsf(go[1])), fmaxf(fabsf(go[2]), VIEW ? 0.0f : fabsf(go[3])));
            unsigned ub;
            memcpy(&ub, &mg, 4);
            s = exp_for(ub, 0);
            put_block(d1h[0], d1l[0], v, nullptr, nh_pow2i(s));
        }
        MaskW mw;
        if (VIEW) {
            mw = get_mask(L);  // DIRH
            gemm_w<W, TW / 2, 0, 1>(cx, nullptr, nullptr, d1h, d1l, po.b_rgb * 4, po.b_dir * 4, first(KB / 2, TW), acc);
            gate_tiles<TW / 2>(acc, mw);
            s = renorm_convert<TW / 2>(acc, hh, hl, s, NO_CAP);
            mw = get_mask(L - 1);  // FEAT
            gemm_w<W, TW, KB / 2, 0, 0, 0, false, false, true>(cx, hh, hl, nullptr, nullptr, po.b_dir * 4, po.b_head * 4, first(KB, TW), acc, nullptr, nullptr,
                                     grow(a.gl.PDIR, W / 2), nullptr, nullptr, nh_pow2i(-s), s, s, NO_CAP, nullptr, nullptr, L + 1);
            gate_tiles<TW>(acc, mw);
            s = renorm_convert<TW>(acc, hh, hl, s, NO_CAP);
            // d(sigma raw) enters through fc_alpha: the head image's bias row holds fc_alpha's weights (plan.cpp build_specs_b), and
            // the accumulators start at bias * d(sigma raw) * 2^s -- fp32, whatever its size next to the hidden inputs
            const bool last = L == 1;
            if (L > 1) mw = get_mask(L - 2);  // H_{L-1}
            gemm_w<W, TW, KB, 0, 0, 0, false, false, true>(cx, hh, hl, nullptr, nullptr, po.b_head * 4, last ? first_img : po.b_xyz[L > 1 ? L - 2 : 0] * 4,
                                 last ? (again ? first_bytes : 0) : first(KB, TW), acc, nullptr, nullptr, grow(a.gl.PFEAT, W), nullptr, nullptr,
                                 nh_pow2i(-s), s, s, NO_CAP, nullptr, nullptr, L, go[3] * nh_pow2i(s), true);
        } else {
            const bool last = L == 1;
            if (L > 1) mw = get_mask(L - 2);  // H_{L-1}
            gemm_w<W, TW, 0, 1>(cx, nullptr, nullptr, d1h, d1l, po.b_head * 4, last ? first_img : po.b_xyz[L > 1 ? L - 2 : 0] * 4,
                                last ? (again ? first_bytes : 0) : first(KB, TW), acc);
        }
        // acc = W^T d(pre-activation) for H_{L-1}: gate by its ReLU mask L - 2 (H_0 = layer1's output has no activation)
        if (L > 1) gate_tiles<TW>(acc, mw);
        s = renorm_convert<TW>(acc, hh, hl, s, NO_CAP);
        for (int k = L - 1; k >= 1; --k) {
            const bool last = k == 1;
            if (k - 1 >= 1) mw = get_mask(k - 2);  // H_{k-1}
            gemm_w<W, TW, KB, 0, 0, 0, false, false, true>(cx, hh, hl, nullptr, nullptr, po.b_xyz[k - 1] * 4, last ? first_img : po.b_xyz[k >= 2 ? k - 2 : 0] * 4,
                                 last ? (again ? first_bytes : 0) : first(KB, TW), acc, nullptr, nullptr, grow(a.gl.P[k], W), nullptr, nullptr,
                                 nh_pow2i(-s), s, s, NO_CAP, nullptr, nullptr, k);
            if (k - 1 >= 1) gate_tiles<TW>(acc, mw);
            s = renorm_convert<TW>(acc, hh, hl, s, NO_CAP);
        }
        {  // d(pre-activation) of layer1: no gemm consumes it -- stored here (hi + lo, as every other image)
            if (cx.wrm) note_region(cx, 0, s);
            float* const pr = grow(a.gl.P[0], W);
            const float rs = nh_pow2i(-s);
#pragma unroll
            for (int ts = 0; ts < TW; ++ts) {
                const int kb = ts >> 1, o = 4 * (ts & 1);
                nh_store4(pr + 16 * ts + 4 * g, (nh_from_f16(hh[kb][o]) + nh_from_f16(hl[kb][o])) * rs,
                          (nh_from_f16(hh[kb][o + 1]) + nh_from_f16(hl[kb][o + 1])) * rs,
                          (nh_from_f16(hh[kb][o + 2]) + nh_from_f16(hl[kb][o + 2])) * rs,
                          (nh_from_f16(hh[kb][o + 3]) + nh_from_f16(hl[kb][o + 3])) * rs);
            }
        }
    }
    regions_end(cx, a.rmax);
}

// compute units of the current device (the emulator: 3, so that the CPU suite walks the persistent loop)
int w_compute_units() {
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
int w_lds_limit(K kern, int bytes) {
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

}  // namespace

int nh_mlp_forward_f16w(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                        nerfhip_stream_t stream, const NhCompact* list) {
    NH_REQUIRE(M < ((int64_t)1 << 31), "mlp_fwd: at most 2^31 - 1 sample points per call (got %lld)", (long long)M);
    NH_REQUIRE(out || (list && stash), "mlp_fwd: out is NULL");
    NH_REQUIRE(nh_prec_f16(p->precision), "mlp_fwd_f16w: not an fp16-piece plan");
    FwdWArgs a;
    memset(&a, 0, sizeof(a));
    a.packed = packed;
    a.packed_bytes = (unsigned)(p->packed_floats * 4);
    a.off = p->pob;
    a.L = p->L;
    a.skip = p->skip;
    a.M = M;
    a.stash = stash;
    // (plans whose large weight-gradient blocks run on the fp16 MFMAs: the stash's region maxima, zeroed here)
    a.rmax = (stash && !p->bjobs.empty()) ? (unsigned*)(stash + nh_stash_floats(p, nh_ceil_div(M, 128) * 4)) : nullptr;
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
    for (int s = 0; s < 32 * XB; ++s) a.xcol[s] = (signed char)p->xyz_slot_b[s];
    for (int s = 0; s < 32 * DB; ++s) a.dcol[s] = (signed char)p->dir_slot_b[s];
    for (int k = 0; k < 16; ++k) {
        a.fx[k] = p->freqs_xyz[k];
        a.fd[k] = p->freqs_dir[k];
    }
    a.Lx = p->cfg.num_encoding_fn_xyz;
    a.Ld = p->view ? p->cfg.num_encoding_fn_dir : 0;
    a.out = out;
    a.cidx = (list && stash) ? list->idx : nullptr;
    a.cstats = (list && stash) ? list->stats : nullptr;
    const int64_t groups = nh_ceil_div(M, 128);
    a.groups = groups;
    // as many workgroups as are resident at once: one 8-wave workgroup per CU (256-wide nets: its LDS; 128-wide: its 173-202 VGPRs)
    const int64_t resident = (int64_t)w_compute_units();
    const int64_t grid = groups < resident ? groups : resident;
    int rc = NERFHIP_OK;
#define NH_FWDW_T(WW, VV, TT)                                                                                     \
    {                                                                                                             \
        rc = w_lds_limit(k_mlp_fwd_f16x3w<WW, VV, TT>, WShape<WW>::LDS_BYTES + RM_LDS);                             \
        if (rc) return rc;                                                                                        \
        NH_LAUNCH_NAMED("k_mlp_fwd_f16x3w<" #WW ", " #VV ", " #TT ">", (k_mlp_fwd_f16x3w<WW, VV, TT>), grid, 512,      \
                        WShape<WW>::LDS_BYTES + RM_LDS, stream, a);                                               \
    }
#define NH_FWDW(WW, VV)              \
    {                                \
        if (stash)                   \
            NH_FWDW_T(WW, VV, true)  \
        else                         \
            NH_FWDW_T(WW, VV, false) \
    }
    if (p->W == 256 && p->view) NH_FWDW(256, true)
    else if (p->W == 256) NH_FWDW(256, false)
    else if (p->W == 128 && p->view) NH_FWDW(128, true)
    else if (p->W == 128) NH_FWDW(128, false)
    else if (p->W == 64 && p->view) NH_FWDW(64, true)
    else if (p->W == 64) NH_FWDW(64, false)
    else {
        nh_set_error("mlp_fwd: no f16x3 kernel for kernel width %d", p->W);
        return NERFHIP_ERR_UNSUPPORTED;
    }
#undef NH_FWDW
#undef NH_FWDW_T
    return nh_launch_status("mlp_fwd_f16w");
}

int nh_mlp_dgrad_f16w(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash, float* scratch,
                      unsigned* rmax, const NhCompact* cpt, nerfhip_stream_t stream) {
    NH_REQUIRE(nh_prec_f16(p->precision), "mlp_bwd_f16w: not an fp16-piece plan");
    DgradWArgs d;
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
    d.cidx = cpt ? cpt->idx : nullptr;
    d.cstats = cpt ? cpt->stats : nullptr;
    d.cstash_listed = (cpt && cpt->stash_in_list_order) ? 1 : 0;
    const int64_t resident = (int64_t)w_compute_units();
    const int64_t grid = d.groups < resident ? d.groups : resident;
    int rc = NERFHIP_OK;
#define NH_BWDW(WW, VV)                                                                                           \
    {                                                                                                             \
        rc = w_lds_limit(k_mlp_dgrad_f16x3w<WW, VV>, WShape<WW>::LDS_BYTES + RM_LDS);                               \
        if (rc) return rc;                                                                                        \
        NH_LAUNCH_NAMED("k_mlp_dgrad_f16x3w<" #WW ", " #VV ">", (k_mlp_dgrad_f16x3w<WW, VV>), grid, 512,               \
                        WShape<WW>::LDS_BYTES + RM_LDS, stream, d);                                               \
    }
    if (p->W == 256 && p->view) NH_BWDW(256, true)
    else if (p->W == 256) NH_BWDW(256, false)
    else if (p->W == 128 && p->view) NH_BWDW(128, true)
    else if (p->W == 128) NH_BWDW(128, false)
    else if (p->W == 64 && p->view) NH_BWDW(64, true)
    else if (p->W == 64) NH_BWDW(64, false)
    else {
        nh_set_error("mlp_bwd: no f16x3 data-gradient kernel for kernel width %d", p->W);
        return NERFHIP_ERR_UNSUPPORTED;
    }
#undef NH_BWDW
    return nh_launch_status("mlp_dgrad_f16w");
}

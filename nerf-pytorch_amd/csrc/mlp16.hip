// mlp16.hip -- FlexibleNeRFModel (nerf/models.py:185-256) forward and data-gradient chain on v_mfma_f32_16x16x4_f32
// (exact fp32: a k-ordered fmaf chain), TWO wavefronts per SIMD.
//
// Layout (nh_plan.h): lane l = (sample j = l & 15, k-group g = l >> 4); register r of an activation holds feature
// feat16(r,g) = 16*(r>>2) + 4*g + (r&3) -- the C/D layout of the instruction -- and k-step r of the next layer takes
// register r as its B operand, so activations never leave the register file: the (N*S, 90) encodings and (N*S, 256)
// hidden states the reference materialises (nerf/train_utils.py:8-25) do not exist.  Weights stream L2 -> LDS by LDS-DMA
// in K-chunks that cover all output tiles ([k-step][quad][lane][4]: one ds_read_b128 = the A operands of four 16-row
// tiles), double buffered, one barrier per chunk.  The stash / gradient images are [32-sample tile][sample][rows] (two
// waves fill one tile): what the weight-gradient kernel (wgrad.hip) reads.
//
// What the structure is tuned against (MI355X, scripts/loop_mock.hip, profiles/r02_loop_mock.txt): the bare chunk loop
// -- 4 ds_read_b128 per 16 MFMAs, two waves per SIMD -- runs at 99 % of the matrix pipe, 95.6 % with the chunk copy and
// a barrier every 8 k-steps, 98.0 % with 16-k-step chunks.  So: chunks as large as the LDS allows (two 64 KB buffers for
// 256-wide nets), the copy pieces (first half of a chunk) and the stash stores (second half: nh16_store_kstep) issued
// ONE PER K-STEP between the MFMA groups, never a burst in front of them, and every address of a layer computed once,
// before its first chunk.
#include <stdlib.h>

#include "nh_mlp.h"

namespace {

// Workgroup shape (measured, profiles/r01_mlp16_ab.txt): 256-wide nets -- ONE 8-wave workgroup per CU (two waves per
// SIMD; the weight stream is paid once per 128 samples); 128-wide nets (134 registers per wave, 52 KB of LDS) -- 4-wave
// workgroups, three per CU; 64-wide nets (config/llff.yml, pretrained/fern-lowres: hidden_size 64) -- 4-wave workgroups
// with 36 KB of LDS, four per CU.
#ifndef NH16_NARROW_NW  // (overridden by A/B builds only: scripts/build_variant.sh)
#define NH16_NARROW_NW 4
#define NH16_NARROW_CHUNK 6144
#endif
#ifndef NH16_WIDE_NW  // (A/B builds only; e.g. 4 waves and 8192-float chunks: two independent workgroups per CU)
#define NH16_WIDE_NW 8
#define NH16_WIDE_CHUNK 16384
#endif
template <int W>
struct Shape {
    // waves per workgroup.  512-wide nets (132 accumulator + 128 activation registers per 16 samples): ONE wave per SIMD
    // with the whole unified register file (accumulators in AGPRs), 4-wave workgroups, one per CU
    static constexpr int NW = W >= 512 ? 4 : (W >= 256 ? NH16_WIDE_NW : NH16_NARROW_NW);
    static constexpr int WAVES_PER_SIMD = W >= 512 ? 1 : 2;  // launch bound (minimum occupancy the registers must allow)
    static constexpr int MW = nh16_mask_words(W);            // 32-bit words of ReLU bits per lane and layer
};
// floats of one chunk buffer: 256-wide nets 16384 (16 k-steps x 4 quads; 2 x 64 KB + bias blocks = 132 KB: one
// workgroup per CU, which the 240-register waves allow anyway), 128-wide nets 6144 (8 k-steps x 3 quads; 52 KB -> 3 per CU),
// 64-wide nets 4096 (a whole 64 x 64 layer = 16 k-steps x 1 quad: one chunk per layer)
template <int W>
struct Lds {
    static constexpr int CHUNK_MAX = W >= 512 ? 16384 : (W >= 256 ? NH16_WIDE_CHUNK : (W >= 128 ? NH16_NARROW_CHUNK : 4096));
    static constexpr int BIAS = nh16_bias_floats(W);
    static constexpr int BYTES = (2 * CHUNK_MAX + 2 * BIAS) * 4;
    static constexpr int BYTES_ALL = BYTES + NH_CLK_LDS_BYTES;  // + the clock probe's stamps (nh_clk_begin)
};

#ifdef NH_PHASE_TIMING
__device__ unsigned long long g_phase16[16];
#define NH16_PH(i)                                \
    do {                                          \
        const unsigned long long _t = clock64();  \
        cx.ph[i] += _t - cx.last;                 \
        cx.last = _t;                             \
    } while (0)
#else
#define NH16_PH(i)
#endif

struct Ctx {
#ifdef NH_PHASE_TIMING
    unsigned long long ph[5], last;  // debug build: cycles per phase, accumulated per wave
#endif
    float* lds;
    int cmax;           // floats per chunk buffer (Lds<W>::CHUNK_MAX)
    int bfl;            // floats of a layer image's bias block (Lds<W>::BIAS)
    int nw;             // waves per workgroup
    unsigned lds_addr;  // LDS byte address of `lds`
    NhDmaSrc dma;  // descriptor over the whole packed image
    int buf, bbuf;  // chunk / bias buffer of the unit being consumed
    int wave, lane, g;
    // the copy in flight: up to two runs of 1-KiB pieces (a bias block, then a chunk); piece q is issued by wave q % nw
    int c_np0, c_np;        // pieces of run 0, pieces in total
    int c_src0, c_src1;     // byte offsets of the runs inside the packed image
    unsigned c_dst0, c_dst1;  // LDS byte addresses of the runs
    NH_MEMBER float* chunk(int b) const { return lds + b * cmax; }
    NH_MEMBER float* bias(int b) const { return lds + 2 * cmax + b * bfl; }
    NH_MEMBER void plan_copy(int64_t off0, int n0, float* dst0, int64_t off1, int n1, float* dst1) {
        c_np0 = n0 >> 8;
        c_np = c_np0 + (n1 >> 8);
        c_src0 = (int)off0 * 4;
        c_src1 = (int)off1 * 4;
        c_dst0 = lds_addr + (unsigned)((dst0 - lds) * 4);
        c_dst1 = lds_addr + (unsigned)((dst1 - lds) * 4);
    }
    // chunk `nfloats` at float offset `off` of the packed image -> dst
    NH_MEMBER void plan_chunk(int64_t off, int nfloats, float* dst) { plan_copy(off, nfloats, dst, 0, 0, dst); }
    // a layer's first unit: its bias block and chunk 0
    NH_MEMBER void plan_first(int64_t img_off, int first_floats, int b, int bb) {
        plan_copy(img_off, bfl, bias(bb), img_off + bfl, first_floats, chunk(b));
    }
    // this wave's j-th piece of the planned copy (one LDS-DMA instruction, or nothing)
    NH_MEMBER void issue(int j) const {
        int q = wave + j * nw;
#ifndef NERFHIP_EMU
        // (opaque to the optimiser: otherwise every piece's source / destination address of every chunk is hoisted out of
        // the layer loop as its own scalar register, and the kernel spills SGPRs into VGPR lanes)
        asm volatile("" : "+s"(q));
#endif
        if (q < c_np0)
            nh_dma16a(dma, lane * 16, c_src0 + q * 1024, c_dst0 + q * 1024);
        else if (q < c_np)
            nh_dma16a(dma, lane * 16, c_src1 + (q - c_np0) * 1024, c_dst1 + (q - c_np0) * 1024);
    }
    NH_MEMBER void issue_from(int j0) const {
        for (int j = j0; wave + j * nw < c_np; ++j) issue(j);
    }
};

// k-steps per chunk: the largest divisor of kr that fits (equal chunks: no ragged tail), unless that wastes more than
// half of the buffer
constexpr int nh16_kc(int kr, int kcm) {
    if (kr <= kcm) return kr;
    for (int d = kcm; 2 * d > kcm; --d)
        if (kr % d == 0) return d;
    return kcm;
}
template <int W, int KR, int T>
struct Geo {
    static constexpr int TQ = (T + 3) / 4;
    static constexpr int KCM = Lds<W>::CHUNK_MAX / (TQ * 256);  // k-steps that fit one chunk buffer
    static constexpr int KC = nh16_kc(KR, KCM);
    static constexpr int NCH = (KR + KC - 1) / KC;
    static constexpr int FIRST = KC * TQ * 256;  // floats of chunk 0
    static_assert(KC >= 1, "one k-step of this layer does not fit the chunk buffer");
};

// One linear layer for the 16 samples of this wavefront: acc[t] (16 rows x 16 samples) = W_t * in + bias_t, t < T.
// Precondition: the layer's first unit has been requested into chunk(buf) / bias(bbuf) (plan_first + issue_from(0)).
// While chunk c is multiplied, chunk c+1 -- or the first unit of the next layer (next_first > 0) -- travels to the
// other buffer, one piece per wave and k-step.
// `post` holds the global stores of the PREVIOUS layer's results: post.first() (masks, encoding slots) goes out with
// k-step 0, and its Post::NT row tiles (post.tile(t): one 16-byte store per lane) are spread evenly over the k-steps of
// this layer, so that they drain under the MFMAs instead of sitting in front of a vmcnt(0) (CDNA4's vmcnt counts
// stores too; the values stored are this layer's input registers, still live).
// ReLU bit of value r among the n = 4T values of a layer, 32 per word: bit position inside word r >> 5 (see finish())
NH_DEVICE constexpr int nh16_bitpos(int r, int n) { return ((n - 32 * (r >> 5)) < 32 ? (n - 32 * (r >> 5)) : 32) - 1 - (r & 31); }

// k-step at which row tile t (of nt) of the previous layer is stored.  The tiles are dealt evenly to the chunks and go
// out in the SECOND half of their chunk: its first half carries the copy pieces of the next chunk (one per k-step), and
// a weight copy queued behind stash stores is what the store stream really costs (scripts/loop_mock.hip F_DMA_ST*,
// profiles/r02_loop_mock.txt: copy alone 98.0 % of the pipe, stores alone 97.6 %, both with the stores spread over the
// chunk 93.2 %, copy pieces in the first half and stores in the second 97.8 %).
constexpr int nh16_store_kstep(int t, int nt, int kr, int kc, int nch) {
    if (kr != kc * nch) return (t * kr) / nt;  // (ragged chunks do not occur: nh16_kc)
    int c = 0;
    while (((c + 1) * nt) / nch <= t) ++c;  // chunk c holds tiles [c nt / nch, (c + 1) nt / nch)
    const int b0 = (c * nt) / nch, per = ((c + 1) * nt) / nch - b0, half = kc / 2, room = kc - half;
    return c * kc + half + ((t - b0) * room) / per;
}
// (GatePre processes group t at k-step 4 (t - 1), groups 0 and 1 at k-step 0: tile t must not be stored earlier)
constexpr bool nh16_stores_follow_pre(int nt, int kr, int kc, int nch) {
    for (int t = 0; t < nt; ++t)
        if (nh16_store_kstep(t, nt, kr, kc, nch) < (t <= 1 ? 0 : 4 * (t - 1))) return false;
    return true;
}
// the one-off stores of a layer (ReLU mask, encoding slots) go with the same k-step of its first chunk
constexpr int nh16_first_kstep(int kc) { return kc / 2; }

// `pre` finishes the INPUT registers just in time: group t (registers 4t..4t+3 of inA) is processed one group ahead of
// the k-step that first consumes it, in the shadow of the MFMAs (GatePre: the ReLU gate of the data-gradient chain --
// the epilogue between two layers is then a pure register renaming).
struct NoPre {
    static constexpr int N = 0;
    NH_MEMBER void group(int) const {}
};
// the ReLU bits of one activation: MW words per lane, value r at bit nh16_bitpos(r, n) of word r >> 5 (see finish())
template <int MW>
struct MaskBits {
    unsigned w[MW];
};
template <int N_, int MW = 2>
struct GatePre {
    static constexpr int N = N_;  // groups of four registers
    float* v;
    MaskBits<MW> m;  // stored ReLU bits of the 4 N values
    NH_MEMBER void group(int t) const {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = 4 * t + c;
            v[r] = nh_gate(v[r], m.w[r >> 5], nh16_bitpos(r, 4 * N));
        }
    }
};

template <int W, int KRA, int KRB, int T, class Post, class Pre = NoPre>
NH_DEVICE void gemm16(Ctx& cx, const float* inA, const float* inB, int64_t img_off, int64_t next_off, int next_first,
                      f32x4* acc, const Post& post, const Pre& pre = Pre()) {
    constexpr int KR = KRA + KRB, NT = Post::NT;
    static_assert(4 * Pre::N <= KRA, "pre-processed registers are inA's");
    static_assert(Pre::N == 0 || nh16_stores_follow_pre(NT, KR, Geo<W, KR, T>::KC, Geo<W, KR, T>::NCH),
                  "a row tile is stored before its registers were pre-processed");
    using G = Geo<W, KR, T>;
    constexpr int TQ = G::TQ, KC = G::KC, NCH = G::NCH;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int kc = (KR - c * KC) < KC ? (KR - c * KC) : KC;
        NH16_PH(4);  // [4] between gemms / chunks: epilogue, loop control
        nh_wait_vmem();
        NH16_PH(2);  // [2] s_waitcnt vmcnt(0)
        nh_block_sync();  // chunk c has landed for every wave; everybody is done with the other buffer
        NH16_PH(3);  // [3] s_barrier
        if (c + 1 < NCH) {
            const int kn = (KR - (c + 1) * KC) < KC ? (KR - (c + 1) * KC) : KC;
            cx.plan_chunk(img_off + Lds<W>::BIAS + (int64_t)(c + 1) * KC * TQ * 256, kn * TQ * 256, cx.chunk(cx.buf ^ 1));
        } else if (next_first > 0) {
            cx.plan_first(next_off, next_first, cx.buf ^ 1, cx.bbuf ^ 1);
        } else {
            cx.c_np0 = cx.c_np = 0;
        }
        if (c == 0) {
            const float* bp = cx.bias(cx.bbuf) + 4 * cx.g;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float4 b4 = *(const float4*)(bp + 16 * t);
                acc[t][0] = b4.x;
                acc[t][1] = b4.y;
                acc[t][2] = b4.z;
                acc[t][3] = b4.w;
            }
        }
        NH16_PH(0);  // [0] copy set-up + bias
        const float4* wp = (const float4*)cx.chunk(cx.buf) + cx.lane;
        // A operands run one k-step ahead of the MFMAs (two k-steps ahead measured no better and costs 16 registers)
        float4 a[2][TQ];
#pragma unroll
        for (int q = 0; q < TQ; ++q) a[0][q] = wp[q * 64];
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
            if (ks < kc) {
                if (ks + 1 < kc) {
#pragma unroll
                    for (int q = 0; q < TQ; ++q) a[(ks + 1) & 1][q] = wp[((ks + 1) * TQ + q) * 64];
                }
                const int r = c * KC + ks;
                if constexpr (Pre::N > 0) {
                    if (r == 0) {
                        pre.group(0);
                        if (Pre::N > 1) pre.group(1);
                    } else if ((r & 3) == 0 && (r >> 2) + 1 < Pre::N) {
                        pre.group((r >> 2) + 1);
                    }
                }
                cx.issue(ks);  // one copy piece ...
                if (r == nh16_first_kstep(KC)) post.first();  // ... and this k-step's share of the previous layer's stores
                if constexpr (NT > 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (nh16_store_kstep(t, NT, KR, KC, NCH) == r) post.tile(t);
                }
                nh_sched_fence();  // reads, copy and stores of this k-step are issued before its MFMAs
                const float b = r < KRA ? inA[r < KRA ? r : 0] : inB[r >= KRA ? r - KRA : 0];
            #pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float4& w = a[ks & 1][t >> 2];
                    const float av = (t & 3) == 0 ? w.x : ((t & 3) == 1 ? w.y : ((t & 3) == 2 ? w.z : w.w));
                    acc[t] = nh_mfma16(av, b, acc[t]);
                }
                        }
        }
        cx.issue_from(kc);  // a copy with more pieces per wave than this chunk had k-steps
        NH16_PH(1);  // [1] operand reads + MFMAs (+ copy pieces, stores)
        cx.buf ^= 1;
    }
    cx.bbuf ^= 1;
}

// epilogue: register r = 4t + c of the activation <- acc[t][c], gated by the stored ReLU mask (MASKED: data-gradient)
// and/or ReLU'd (RELU: forward); BITS: collect [v > 0] of the n = 4T values, 32 per word, by shift-accumulation --
// value r of a word lands at bit position (values in the word) - 1 - (r & 31) (no per-bit constants in registers).
template <int T, bool RELU, bool BITS, bool MASKED>
NH_DEVICE void finish(const f32x4* acc, float* act, unsigned* bits_out, const unsigned* mbits) {
#pragma unroll
    for (int r = 0; r < 4 * T; ++r) {
        float v = acc[r >> 2][r & 3];
        if (MASKED) v = nh_gate(v, mbits[r >> 5], nh16_bitpos(r, 4 * T));
        if (RELU) v = nh_relu(v);
        if (BITS) bits_out[r >> 5] = (bits_out[r >> 5] << 1) | nh_pos_bit(v);  // (v >= 0 here: masks are only taken after a ReLU)
        act[r] = v;
    }
}

// ---- the stores a gemm carries for the previous layer -------------------------------------------------------------------
// Addresses are (wave-uniform base pointer) + (32-bit per-lane byte offset) + immediate: the base lives in scalar
// registers, the lane offset is shared by all regions of the same row count, so a store costs no address arithmetic.
struct RowRef {
    char* base;    // wave-uniform; NULL: nothing is stored
    unsigned off;  // this lane's byte offset
    NH_MEMBER float* at(int float_index) const { return (float*)(base + (size_t)off) + float_index; }
};
// activation rows: tile t = rows feat16(4t.., g) = 16t + 4g .. +3 of this lane's sample, one 16-byte store; `mask`
// (optional) receives the ReLU bits of the activation first.
// NT > 16 (512 rows: 512-wide nets): tiles 16.. go to the activation's second 256-row region, `hi_bytes` further on
template <int NT_, int MW = 2>
struct RowsPost {
    static constexpr int NT = NT_;
    RowRef row;
    const float* act;
    RowRef mask;
    MaskBits<MW> b;
    size_t hi_bytes;
    NH_MEMBER void first() const {
        if (mask.base) {
            unsigned* p = (unsigned*)mask.at(0);
#pragma unroll
            for (int w = 0; w < MW; ++w) p[w] = b.w[w];
        }
    }
    NH_MEMBER void tile(int t) const {
        if (!row.base) return;
        float* dst = t < 16 ? row.at(16 * t) : (float*)((char*)row.at(16 * (t - 16)) + hi_bytes);
        nh_store4(dst, act[4 * t], act[4 * t + 1], act[4 * t + 2], act[4 * t + 3]);
    }
};
// encoding slots (register r of lane (j,g) is row g*KR + r; the lane offset already includes g*KR): stored with k-step 0
template <int KR>
struct SlotsPost {
    static constexpr int NT = 0;
    RowRef row;
    const float* e;
    NH_MEMBER void first() const {
        if (!row.base) return;
#pragma unroll
        for (int q = 0; q < KR / 4; ++q) nh_store4(row.at(4 * q), e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]);
    }
    NH_MEMBER void tile(int) const {}
};
// H_{L-1} rows + mask AND the direction-encoding slots (computed late: see k_mlp_fwd16)
template <int NT_, int KD, int MW = 2>
struct RowsAndSlotsPost {
    static constexpr int NT = RowsPost<NT_, MW>::NT;
    RowsPost<NT_, MW> rows;
    SlotsPost<KD> slots;
    NH_MEMBER void first() const {
        rows.first();
        slots.first();
    }
    NH_MEMBER void tile(int t) const { rows.tile(t); }
};
// d(raw output) rows of the data-gradient chain: 8 rows per lane group (rows 0..3 of group 0 carry the cotangent)
struct PoutPost {
    static constexpr int NT = 0;
    RowRef row;
    float v0, v1, v2, v3;
    NH_MEMBER void first() const {
        *(float4*)row.at(0) = make_float4(v0, v1, v2, v3);
        *(float4*)row.at(4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    NH_MEMBER void tile(int) const {}
};
// all rows of an activation at once (the last store of the data-gradient chain)
template <int T>
NH_DEVICE void store_rows(const RowRef& row, const float* act, size_t hi_bytes) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        float* dst = t < 16 ? row.at(16 * t) : (float*)((char*)row.at(16 * (t - 16)) + hi_bytes);
        nh_store4(dst, act[4 * t], act[4 * t + 1], act[4 * t + 2], act[4 * t + 3]);
    }
}

// wave-uniform base of a region's rows for the four 32-sample tiles of workgroup `wg` (two tiles for 4-wave workgroups)
NH_DEVICE char* region_wg_base(float* base, const NhRegion& R, int64_t nt, int64_t tile0) {
    return (char*)(base + (size_t)32 * (size_t)nt * (size_t)R.row_prefix + (size_t)tile0 * 32 * (size_t)R.rows);
}

NH_DEVICE float sel3(int a, float x, float y, float z) { return a == 0 ? x : (a == 1 ? y : z); }

// encoding registers of lane group g (slot layout: plan.cpp build_slot_map16)
template <int KR>
NH_DEVICE void encode_slots16(float* e, float x, float y, float z, int g, const float* freqs, int Lf) {
    constexpr int C = KR / 2, C3 = (KR - 3) / 2;
#pragma unroll
    for (int q = 0; q < C; ++q) {
        const int pr = g * C + q;
        const bool valid = pr < 3 * Lf && (g < 3 || q < C3);
        const int f = pr / 3, ax = pr - 3 * f;
        const float arg = sel3(ax, x, y, z) * freqs[f < 16 ? f : 15];
        float s, c;
        nh_sincos(arg, &s, &c);
        e[2 * q] = valid ? s : 0.0f;
        e[2 * q + 1] = valid ? c : 0.0f;
    }
    if (g == 3) {  // the raw coordinates ride in the last three registers of group 3
        e[KR - 3] = x;
        e[KR - 2] = y;
        e[KR - 1] = z;
    }
}

struct Fwd16Args {
    const float* packed;
    unsigned packed_bytes;
    NhPackedOffsets off;
    int L, skip;
    int64_t M, nt;
    int mode;
    const float* x;
    int dx, dd;
    const float* rays;
    int ray_stride;
    const float* z;
    int S;
    signed char xcol[4][NH16_KRX_EXT];  // (rows of KX / KD entries; reference columns < 100)
    signed char dcol[4][NH16_KRD_EXT];
    float fx[16], fd[16];
    int Lx, Ld;
    float* out;
    float* stash;
    NhStashLayout sl;
    unsigned long long* clk;  // shader-clock probe counters, or NULL (nh_prof_clock_slot)
    // a forward over a compaction list (the recomputing backward, mlp.hip nh_mlp_backward_recompute), or NULLs: slot c of the launch
    // computes sample cidx[c] and writes that sample's stash rows and ReLU masks AT SLOT c; cstats[NH_CSTAT_ACTIVE] slots carry a
    // sample; `out` may be NULL
    const int* cidx;
    const int* cstats;
};

// TRAIN: the launch writes the activation stash (rows, encoding slots, ReLU masks) for the backward kernels
// EXT: the extended encoding registers (num_encoding_fn_xyz > 10 or num_encoding_fn_dir > 4: nh_plan.h)
template <int W, bool VIEW, bool TRAIN, bool EXT = false>
NH_KERNEL void NH_LB(64 * Shape<W>::NW, Shape<W>::WAVES_PER_SIMD) k_mlp_fwd16(Fwd16Args a) {
    constexpr int KH = W / 4, TW = W / 16, KX = EXT ? NH16_KRX_EXT : NH16_KRX, KD = EXT ? NH16_KRD_EXT : NH16_KRD;
    constexpr int NW = Shape<W>::NW, MW = Shape<W>::MW;
    // (a launch over a list: a workgroup whose slots are all behind it has nothing to do)
    const int n_slots = (TRAIN && a.cidx) ? nh_uload_i32(a.cstats, NH_CSTAT_ACTIVE) : 0;
    if (TRAIN && a.cidx && (int64_t)blockIdx.x * (NW * 16) >= (int64_t)n_slots) return;
    NH_DYN_LDS(lds_raw);
    nh_clk_begin(a.clk, (unsigned long long*)(lds_raw + Lds<W>::BYTES));
    Ctx cx;
    cx.lds = (float*)lds_raw;
    cx.cmax = Lds<W>::CHUNK_MAX;
    cx.bfl = Lds<W>::BIAS;
    cx.nw = Shape<W>::NW;
    cx.lds_addr = nh_lds_addr(cx.lds);
    cx.dma = nh_dma_src(a.packed, a.packed_bytes);
    cx.buf = cx.bbuf = 0;
    cx.lane = nh_lane();
    cx.wave = nh_wave_in_block();
    cx.g = cx.lane >> 4;
#ifdef NH_PHASE_TIMING
    for (int q = 0; q < 5; ++q) cx.ph[q] = 0;
    cx.last = clock64();
#endif
    const int lane = cx.lane, g = cx.g, j = lane & 15, wave = cx.wave;
    const int64_t tile = (int64_t)blockIdx.x * (NW / 2) + (wave >> 1);  // 32-sample stash tile
    const int js = 16 * (wave & 1) + j;                         // this lane's sample inside it
    const int64_t m = tile * 32 + js;
    bool valid = m < a.M;
    int64_t mc = valid ? m : a.M - 1;  // the sample this lane computes
    if (TRAIN && a.cidx) {              // (slot m of a list: its sample; padding slots compute sample cidx[m] = 0 and are never read)
        valid = m < (int64_t)n_slots;
        mc = (int64_t)a.cidx[m];
    }
    const int m_i = (int)m;  // (the host checks M < 2^31)
    const NhPackedOffsets& po = a.off;

    // the first weights travel to LDS while the encodings are computed
    cx.plan_first(po.f_layer1, Geo<W, KX, TW>::FIRST, 0, 0);
    cx.issue_from(0);

    float ex[KX];
    // (only 32-bit row indices stay live across the hidden layers: the direction encoding re-forms its pointer later)
    const int xrow_i = (int)mc;                                 // row of x (mode 0)
    const int ray_i = a.mode == 0 ? 0 : (int)(mc / a.S);       // ray of this sample (mode 1)
    if (a.mode == 0) {
        const float* const xr = a.x + (size_t)xrow_i * (size_t)(a.dx + a.dd);
#pragma unroll
        for (int r = 0; r < KX; ++r) {
            const int c = a.xcol[g][r];
            ex[r] = c >= 0 ? xr[c] : 0.0f;
        }
    } else {
        const float* const rr = a.rays + (size_t)ray_i * a.ray_stride;
        const float zz = a.z[mc];
        // pts = ro + rd * z   (nerf/train_utils.py:67,107)
        const float px = rr[0] + rr[3] * zz, py = rr[1] + rr[4] * zz, pz = rr[2] + rr[5] * zz;
        encode_slots16<KX>(ex, px, py, pz, g, a.fx, a.Lx);
    }
    constexpr bool tr = TRAIN;
    // store addresses: wave-uniform region base of this workgroup's tiles + one lane offset per row count
    const int64_t tile0 = (int64_t)blockIdx.x * (NW / 2);
    const unsigned lrow = (unsigned)((wave >> 1) * 32 + js);  // this lane's sample row inside the workgroup's tiles
    auto sref = [&](const NhRegion& R, int rows, int first) -> RowRef {
        return RowRef{tr ? region_wg_base(a.stash, R, a.nt, tile0) : nullptr, (lrow * (unsigned)rows + (unsigned)first) * 4u};
    };
    constexpr int RW = W > 256 ? 256 : W;                        // rows of one stash region of a W-row activation
    const size_t hi = (size_t)32 * (size_t)a.nt * 256 * 4;       // bytes from a 512-row activation's first region to its second
    // ReLU masks for the data-gradient kernel: one bit per activation register, MW words per lane and layer,
    // [16-sample wave tile][mask][lane][MW words]
    char* const mask_base = tr ? (char*)((unsigned*)(a.stash + (size_t)32 * (size_t)a.nt * (size_t)a.sl.total_rows) +
                                         (size_t)((int64_t)blockIdx.x * NW + wave) * a.sl.n_masks * (64 * MW))
                               : nullptr;
    MaskBits<MW> bits;
    auto clear_bits = [&]() {
#pragma unroll
        for (int w = 0; w < MW; ++w) bits.w[w] = 0u;
    };
    clear_bits();
    auto mref = [&](int idx) -> RowRef {
        return RowRef{(tr && idx >= 0) ? mask_base + (size_t)idx * (256 * MW) : nullptr, (unsigned)lane * (4u * MW)};
    };

    float act[KH];
    f32x4 acc[TW + 1];
    {
        const bool more = a.L > 1;
        // layers_xyz[0] is never a skip layer (i > 0 is required); no activation after layer1 (models.py:238)
        gemm16<W, KX, 0, TW>(cx, ex, nullptr, po.f_layer1, more ? po.f_xyz[0] : po.f_head,
                             more ? Geo<W, KH, TW>::FIRST : (VIEW ? Geo<W, KH, TW + 1>::FIRST : Geo<W, KH, 1>::FIRST), acc,
                             SlotsPost<KX>{sref(a.sl.X, 4 * KX, g * KX), ex});
        finish<TW, false, false, false>(acc, act, bits.w, bits.w);
    }
    // every gemm stores its own input rows (= the previous layer's output) and that layer's ReLU mask
    for (int i = 0; i < a.L - 1; ++i) {
        const bool sk = (i % a.skip == 0) && i > 0;
        const bool more = i + 1 < a.L - 1;
        const bool nsk = more && ((i + 1) % a.skip == 0);
        const int64_t nxt = more ? po.f_xyz[i + 1] : po.f_head;
        const int nfirst = more ? (nsk ? Geo<W, KH + KX, TW>::FIRST : Geo<W, KH, TW>::FIRST)
                                : (VIEW ? Geo<W, KH, TW + 1>::FIRST : Geo<W, KH, 1>::FIRST);
        // H_i and its ReLU mask (none for H_0)
        const RowsPost<TW, MW> post{sref(a.sl.H[i], RW, 4 * g), act, mref(i - 1), bits, hi};
        if (sk)
            gemm16<W, KH, KX, TW>(cx, act, ex, po.f_xyz[i], nxt, nfirst, acc, post);
        else
            gemm16<W, KH, 0, TW>(cx, act, nullptr, po.f_xyz[i], nxt, nfirst, acc, post);
        clear_bits();
        finish<TW, true, TRAIN, false>(acc, act, bits.w, bits.w);
    }
    const RowsPost<TW, MW> post_last_hidden{sref(a.sl.H[a.L - 1], RW, 4 * g), act, mref(a.L - 2), bits, hi};  // H_{L-1}
    if (VIEW) {
        // The direction encoding is first needed by layers_dir: it is formed HERE, after the hidden layers, so that its
        // registers are not carried through them (its slots go to the stash with the head gemm's first k-step).
        float ed[KD];
        if (a.mode == 0) {
            const float* const xr = a.x + (size_t)xrow_i * (size_t)(a.dx + a.dd);
#pragma unroll
            for (int r = 0; r < KD; ++r) {
                const int c = (int)a.dcol[g][r];
                ed[r] = c >= 0 ? xr[a.dx + c] : 0.0f;
            }
        } else {
            const float* const rr = a.rays + (size_t)ray_i * a.ray_stride;
            encode_slots16<KD>(ed, rr[8], rr[9], rr[10], g, a.fd, a.Ld);
        }
        // tiles 0..TW-1: feat = relu(fc_feat(h)); tile TW row 0: fc_alpha(h), raw (models.py:248-249)
        gemm16<W, KH, 0, TW + 1>(cx, act, nullptr, po.f_head, po.f_dir, Geo<W, KH + KD, TW / 2>::FIRST, acc,
                                 RowsAndSlotsPost<TW, KD, MW>{post_last_hidden, SlotsPost<KD>{sref(a.sl.D, 4 * KD, g * KD), ed}});
        const float alpha = acc[TW][0];
        clear_bits();
        finish<TW, true, TRAIN, false>(acc, act, bits.w, bits.w);
        float dh[KH / 2];
        gemm16<W, KH, KD, TW / 2>(cx, act, ed, po.f_dir, po.f_rgb, Geo<W, KH / 2, 1>::FIRST, acc,
                                  RowsPost<TW, MW>{sref(a.sl.FEAT, RW, 4 * g), act, mref(a.L - 1), bits, hi});
        clear_bits();
        finish<TW / 2, true, TRAIN, false>(acc, dh, bits.w, bits.w);
        gemm16<W, KH / 2, 0, 1>(cx, dh, nullptr, po.f_rgb, 0, 0, acc,
                                RowsPost<TW / 2, MW>{sref(a.sl.DIRH, W / 2, 4 * g), dh, mref(a.L), bits, hi});
        if (valid && g == 0 && a.out) {
            float4 r4;
            r4.x = acc[0][0];
            r4.y = acc[0][1];
            r4.z = acc[0][2];
            r4.w = alpha;
            *(float4*)(a.out + (size_t)m_i * 4) = r4;
        }
    } else {
        gemm16<W, KH, 0, 1>(cx, act, nullptr, po.f_head, 0, 0, acc, post_last_hidden);
        if (valid && g == 0 && a.out) {
            float4 r4;
            r4.x = acc[0][0];
            r4.y = acc[0][1];
            r4.z = acc[0][2];
            r4.w = acc[0][3];
            *(float4*)(a.out + (size_t)m_i * 4) = r4;
        }
    }
    nh_clk_end((const unsigned long long*)(lds_raw + Lds<W>::BYTES));
#ifdef NH_PHASE_TIMING
    NH16_PH(4);
    if (lane == 0)
        for (int q = 0; q < 5; ++q) atomicAdd(&g_phase16[q], cx.ph[q]);
#endif
}

// ---- data-gradient chain ---------------------------------------------------------------------------------------------
struct Dgrad16Args {
    const float* packed;
    unsigned packed_bytes;
    NhPackedOffsets off;
    int L;
    int64_t M, nt;
    const float* g_out;
    const float* stash;
    NhStashLayout sl;
    float* grad;
    NhGradLayout gl;
    unsigned long long* clk;  // shader-clock probe counters, or NULL
    // compacted backward (compact.hip), or NULLs: slot c of the launch is sample cidx[c] -- d(raw output) and the ReLU masks are
    // gathered by it, the d(pre-activation) images are written in slot order; cstats[NH_CSTAT_ACTIVE] slots carry a sample
    const int* cidx;
    const int* cstats;
    int cstash_listed;  // the stash is in list order too (a recomputed one): the ReLU masks of slot c are at slot c
};

template <int W, bool VIEW>
NH_KERNEL void NH_LB(64 * Shape<W>::NW, Shape<W>::WAVES_PER_SIMD) k_mlp_dgrad16(Dgrad16Args a) {
    constexpr int KH = W / 4, TW = W / 16, NW = Shape<W>::NW, MW = Shape<W>::MW;
    // compacted: a workgroup whose slots are all behind the list has nothing to write (wgrad reads the first ceil(active / 32) tiles)
    const int n_slots = a.cidx ? nh_uload_i32(a.cstats, NH_CSTAT_ACTIVE) : 0;
    if (a.cidx && (int64_t)blockIdx.x * (NW * 16) >= (int64_t)n_slots) return;
    NH_DYN_LDS(lds_raw);
    nh_clk_begin(a.clk, (unsigned long long*)(lds_raw + Lds<W>::BYTES));
    Ctx cx;
    cx.lds = (float*)lds_raw;
    cx.cmax = Lds<W>::CHUNK_MAX;
    cx.bfl = Lds<W>::BIAS;
    cx.nw = Shape<W>::NW;
    cx.lds_addr = nh_lds_addr(cx.lds);
    cx.dma = nh_dma_src(a.packed, a.packed_bytes);
    cx.buf = cx.bbuf = 0;
    cx.lane = nh_lane();
    cx.wave = nh_wave_in_block();
    cx.g = cx.lane >> 4;
#ifdef NH_PHASE_TIMING
    for (int q = 0; q < 5; ++q) cx.ph[q] = 0;
    cx.last = clock64();
#endif
    const int lane = cx.lane, g = cx.g, j = lane & 15, wave = cx.wave;
    const int64_t tile = (int64_t)blockIdx.x * (NW / 2) + (wave >> 1);
    const int js = 16 * (wave & 1) + j;
    int64_t m = tile * 32 + js;  // this lane's slot, then its sample
    bool valid = m < a.M;
    if (a.cidx) {
        valid = m < (int64_t)n_slots;
        m = valid ? (int64_t)a.cidx[m] : 0;
    }
    const NhPackedOffsets& po = a.off;
    const int L = a.L;
    // the transposed images carry no bias: their 512-float bias block is all zero weights (index -1 -> 0.0f)
    if (VIEW)
        cx.plan_first(po.b_rgb, Geo<W, 1, TW / 2>::FIRST, 0, 0);
    else
        cx.plan_first(po.b_head, Geo<W, 1, TW>::FIRST, 0, 0);
    cx.issue_from(0);

    // d(raw output) of this lane's sample as four scalars (kept in registers: no address-taken aggregate)
    float go0 = 0.f, go1 = 0.f, go2 = 0.f, go3 = 0.f;
    if (valid) {
        const float4 t = *(const float4*)(a.g_out + (size_t)m * 4);
        go0 = t.x, go1 = t.y, go2 = t.z, go3 = t.w;
    }
    const int64_t tile0 = (int64_t)blockIdx.x * (NW / 2);
    const unsigned lrow = (unsigned)((wave >> 1) * 32 + js);
    auto gref = [&](const NhRegion& R, int rows, int first) -> RowRef {
        return RowRef{region_wg_base(a.grad, R, a.nt, tile0), (lrow * (unsigned)rows + (unsigned)first) * 4u};
    };
    constexpr int RW = W > 256 ? 256 : W;
    const size_t hi = (size_t)32 * (size_t)a.nt * 256 * 4;
    // POUT (32 rows): rows 0..2 d(rgb raw), row 3 d(sigma raw), rows 4..31 zero; group g writes rows 8g..8g+7
    const bool g0 = g == 0;
    const PoutPost store_pout{gref(a.gl.POUT, 32, 8 * g), g0 ? go0 : 0.f, g0 ? go1 : 0.f, g0 ? go2 : 0.f, g0 ? go3 : 0.f};
    // the forward wrote the mask words of wave tile (sample >> 4), lane 16 g + (sample & 15): this very lane's in the dense backward
    const bool mask_gather = a.cidx && !a.cstash_listed;
    const int64_t wave_tile = mask_gather ? (m >> 4) : ((int64_t)blockIdx.x * NW + wave);
    const int mask_lane = mask_gather ? 16 * g + (int)(m & 15) : lane;
    const char* const mask_base = (const char*)((const unsigned*)(a.stash + (size_t)32 * (size_t)a.nt * (size_t)a.sl.total_rows) +
                                                (size_t)wave_tile * a.sl.n_masks * (64 * MW));
    // ReLU masks: `mb` gates the d(pre-activation) currently held in registers -- applied by the NEXT gemm, group by
    // group, just before it consumes / stores them (GatePre) -- while `mn` is fetched for the one being accumulated
    MaskBits<MW> mb, mn;
    auto get_mask = [&](int idx) {
        const unsigned* p = (const unsigned*)(mask_base + (size_t)idx * (256 * MW) + (size_t)((unsigned)mask_lane * (4u * MW)));
#pragma unroll
        for (int w = 0; w < MW; ++w) mn.w[w] = p[w];
    };
    auto ones = [&]() {  // a layer without activation: nothing is gated
#pragma unroll
        for (int w = 0; w < MW; ++w) mn.w[w] = 0xFFFFFFFFu;
    };
    auto rotate = [&]() { mb = mn; };
    ones();
    rotate();
    f32x4 acc[TW];
    float dp[KH];  // d(pre-activation) of the layer just finished = B operand of the next transposed GEMM
    unsigned nobits[MW] = {};
    const MaskBits<MW> zero_bits = {};
    const RowRef none{nullptr, 0u};
    // every gemm stores its own input rows (= the d(pre-activation) the previous one produced), one tile per few k-steps
    if (VIEW) {
        // one k-step: group g carries d(rgb raw)[g]
        float d1[1];
        d1[0] = g == 0 ? go0 : (g == 1 ? go1 : (g == 2 ? go2 : 0.0f));
        get_mask(L);  // DIRH
        float dpd[KH / 2];
        gemm16<W, 1, 0, TW / 2>(cx, d1, nullptr, po.b_rgb, po.b_dir, Geo<W, KH / 2, TW>::FIRST, acc, store_pout);
        finish<TW / 2, false, false, false>(acc, dpd, nobits, nobits);
        rotate();
        get_mask(L - 1);  // FEAT
        gemm16<W, KH / 2, 0, TW>(cx, dpd, nullptr, po.b_dir, po.b_head, Geo<W, KH + 1, TW>::FIRST, acc,
                                 RowsPost<TW / 2, MW>{gref(a.gl.PDIR, W / 2, 4 * g), dpd, none, zero_bits, hi},
                                 GatePre<TW / 2, MW>{dpd, mb});
        finish<TW, false, false, false>(acc, dp, nobits, nobits);
        rotate();
        if (L > 1)
            get_mask(L - 2);  // H_{L-1}
        else
            ones();  // H_0 = layer1's output has no activation
        float da[1];
        da[0] = g == 0 ? go3 : 0.0f;  // d(sigma raw) enters through fc_alpha's row (k-step KH, group 0)
        gemm16<W, KH, 1, TW>(cx, dp, da, po.b_head, L > 1 ? po.b_xyz[L - 2] : 0, L > 1 ? Geo<W, KH, TW>::FIRST : 0, acc,
                             RowsPost<TW, MW>{gref(a.gl.PFEAT, RW, 4 * g), dp, none, zero_bits, hi}, GatePre<TW, MW>{dp, mb});
        finish<TW, false, false, false>(acc, dp, nobits, nobits);
        rotate();
    } else {
        float d1[1];
        d1[0] = g == 0 ? go0 : (g == 1 ? go1 : (g == 2 ? go2 : go3));
        if (L > 1)
            get_mask(L - 2);  // H_{L-1}
        else
            ones();
        gemm16<W, 1, 0, TW>(cx, d1, nullptr, po.b_head, L > 1 ? po.b_xyz[L - 2] : 0, L > 1 ? Geo<W, KH, TW>::FIRST : 0, acc,
                            store_pout);
        finish<TW, false, false, false>(acc, dp, nobits, nobits);
        rotate();
    }
    // dp = W^T d(pre-activation) for H_{L-1}, still to be gated by mb.  Walk down:
    // dpre_{k-1} = relu'(H_{k-1}) * (W_{k-1}^T dpre_k); H_0 = layer1 output has no activation (models.py:238).
    for (int k = L - 1; k >= 1; --k) {
        if (k - 1 >= 1)
            get_mask(k - 2);  // H_{k-1}
        else
            ones();
        gemm16<W, KH, 0, TW>(cx, dp, nullptr, po.b_xyz[k - 1], k >= 2 ? po.b_xyz[k - 2] : 0,
                             k >= 2 ? Geo<W, KH, TW>::FIRST : 0, acc, RowsPost<TW, MW>{gref(a.gl.P[k], RW, 4 * g), dp, none, zero_bits, hi},
                             GatePre<TW, MW>{dp, mb});
        finish<TW, false, false, false>(acc, dp, nobits, nobits);
        rotate();
    }
    // (mb is all ones here: d(pre-activation) of layer1 needs no gate)
    store_rows<TW>(gref(a.gl.P[0], RW, 4 * g), dp, hi);
    nh_clk_end((const unsigned long long*)(lds_raw + Lds<W>::BYTES));
#ifdef NH_PHASE_TIMING
    NH16_PH(4);
    if (lane == 0)
        for (int q = 0; q < 5; ++q) atomicAdd(&g_phase16[8 + q], cx.ph[q]);
#endif
}

template <class K>
int lds_limit(K kern, int bytes) {
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

#if defined(NH_PHASE_TIMING) && !defined(NH16_W512_TU) && !defined(NH16_EXT_TU)
extern "C" int nerfhip_debug_phases16(unsigned long long* host16, int reset) {
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_phase16), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase16), z, sizeof(z));
    }
    return 0;
}
#endif

// ---- host side ------------------------------------------------------------------------------------------------------
// The 512-wide instantiations (three kernels, ~2.5 minutes of compile time) live in their own translation unit,
// mlp16_w512.hip, which includes this file with NH16_W512_TU defined: it compiles the same templates for W = 512 only and
// exports nh_mlp16_forward_w512 / nh_mlp16_dgrad_w512; this unit holds the widths 64 / 128 / 256 and the dispatch.
// mlp16_ext.hip (NH16_EXT_TU) likewise holds the forward kernels with the extended encoding registers, every width.
namespace {

void fill_fwd_args(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash, Fwd16Args& a,
                   const NhCompact* list = nullptr) {
    memset(&a, 0, sizeof(a));
    a.packed = packed;
    a.packed_bytes = (unsigned)(p->packed_floats * 4);
    a.off = p->po;
    a.L = p->L;
    a.skip = p->skip;
    a.M = M;
    a.nt = nh_ceil_div(M, 128) * 4;
    a.mode = in.mode;
    a.x = in.x;
    a.dx = p->Dx;
    a.dd = p->Dd;
    a.rays = in.rays;
    a.ray_stride = in.ray_stride;
    a.z = in.z;
    a.S = in.S;
    for (int g = 0; g < 4; ++g) {
        for (int r = 0; r < NH16_KRX_EXT; ++r) a.xcol[g][r] = (signed char)p->xyz_col16[g][r];
        for (int r = 0; r < NH16_KRD_EXT; ++r) a.dcol[g][r] = (signed char)p->dir_col16[g][r];
    }
    for (int k = 0; k < 16; ++k) {
        a.fx[k] = p->freqs_xyz[k];
        a.fd[k] = p->freqs_dir[k];
    }
    a.Lx = p->cfg.num_encoding_fn_xyz;
    a.Ld = p->view ? p->cfg.num_encoding_fn_dir : 0;
    a.out = out;
    a.stash = stash;
    a.sl = p->stash;
    a.clk = nh_prof_clock_slot(NH_CLK_FWD);
    a.cidx = (list && stash) ? list->idx : nullptr;
    a.cstats = (list && stash) ? list->stats : nullptr;
}

void fill_dgrad_args(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash, float* scratch,
                     const NhCompact* cx, Dgrad16Args& d) {
    memset(&d, 0, sizeof(d));
    d.packed = packed;
    d.packed_bytes = (unsigned)(p->packed_floats * 4);
    d.off = p->po;
    d.L = p->L;
    d.M = M;
    d.nt = nh_ceil_div(M, 128) * 4;
    d.g_out = g_out;
    d.stash = stash;
    d.sl = p->stash;
    d.grad = scratch;
    d.gl = p->grad;
    d.clk = nh_prof_clock_slot(NH_CLK_DGRAD);
    d.cidx = cx ? cx->idx : nullptr;
    d.cstats = cx ? cx->stats : nullptr;
    d.cstash_listed = (cx && cx->stash_in_list_order) ? 1 : 0;
}

// whole 128-sample groups are launched: every stash tile is written
#define NH_FWD16_T(WW, VV, TT, EE)                                                                                    \
    {                                                                                                                 \
        rc = lds_limit(k_mlp_fwd16<WW, VV, TT, EE>, Lds<WW>::BYTES_ALL);                                              \
        if (rc) return rc;                                                                                            \
        NH_LAUNCH((k_mlp_fwd16<WW, VV, TT, EE>), groups * (8 / Shape<WW>::NW), 64 * Shape<WW>::NW, Lds<WW>::BYTES_ALL, stream, a); \
    }
#define NH_FWD16_E(WW, VV, EE)            \
    {                                     \
        if (stash)                        \
            NH_FWD16_T(WW, VV, true, EE)  \
        else                              \
            NH_FWD16_T(WW, VV, false, EE) \
    }
#define NH_FWD16(WW, VV) NH_FWD16_E(WW, VV, false)
#define NH_BWD16(WW, VV)                                                                                              \
    {                                                                                                                 \
        rc = lds_limit(k_mlp_dgrad16<WW, VV>, Lds<WW>::BYTES_ALL);                                                    \
        if (rc) return rc;                                                                                            \
        NH_LAUNCH((k_mlp_dgrad16<WW, VV>), groups * (8 / Shape<WW>::NW), 64 * Shape<WW>::NW, Lds<WW>::BYTES_ALL, stream, d); \
    }

}  // namespace

int nh_mlp16_forward_w512(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                          nerfhip_stream_t stream, const NhCompact* list);
int nh_mlp16_dgrad_w512(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash, float* scratch,
                        const NhCompact* cx, nerfhip_stream_t stream);
int nh_mlp16_forward_ext(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                         nerfhip_stream_t stream, const NhCompact* list);

#if defined(NH16_EXT_TU)
// every width with the extended encoding registers (mlp16_ext.hip)
int nh_mlp16_forward_ext(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                         nerfhip_stream_t stream, const NhCompact* list) {
    Fwd16Args a;
    fill_fwd_args(p, packed, in, M, out, stash, a, list);
    const int64_t groups = nh_ceil_div(M, 128);
    int rc = NERFHIP_OK;
    if (p->W == 512 && p->view) NH_FWD16_E(512, true, true)
    else if (p->W == 512) NH_FWD16_E(512, false, true)
    else if (p->W == 256 && p->view) NH_FWD16_E(256, true, true)
    else if (p->W == 256) NH_FWD16_E(256, false, true)
    else if (p->W == 128 && p->view) NH_FWD16_E(128, true, true)
    else if (p->W == 128) NH_FWD16_E(128, false, true)
    else if (p->view) NH_FWD16_E(64, true, true)
    else NH_FWD16_E(64, false, true)
    return nh_launch_status("mlp_fwd16");
}
#elif defined(NH16_W512_TU)
int nh_mlp16_forward_w512(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                          nerfhip_stream_t stream, const NhCompact* list) {
    Fwd16Args a;
    fill_fwd_args(p, packed, in, M, out, stash, a, list);
    const int64_t groups = nh_ceil_div(M, 128);
    int rc = NERFHIP_OK;
    if (p->view) NH_FWD16(512, true)
    else NH_FWD16(512, false)
    return nh_launch_status("mlp_fwd16");
}

int nh_mlp16_dgrad_w512(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash, float* scratch,
                        const NhCompact* cx, nerfhip_stream_t stream) {
    Dgrad16Args d;
    fill_dgrad_args(p, packed, g_out, M, stash, scratch, cx, d);
    const int64_t groups = nh_ceil_div(M, 128);
    int rc = NERFHIP_OK;
    if (p->view) NH_BWD16(512, true)
    else NH_BWD16(512, false)
    return nh_launch_status("mlp_dgrad16");
}
#else
int nh_mlp16_forward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                     nerfhip_stream_t stream, const NhCompact* list) {
    NH_REQUIRE(M < ((int64_t)1 << 31), "mlp_fwd: at most 2^31 - 1 sample points per call (got %lld)", (long long)M);
    NH_REQUIRE(out || (list && stash), "mlp_fwd: out is NULL");
    if (p->krx != NH16_KRX) return nh_mlp16_forward_ext(p, packed, in, M, out, stash, stream, list);
    if (p->W == 512) return nh_mlp16_forward_w512(p, packed, in, M, out, stash, stream, list);
    Fwd16Args a;
    fill_fwd_args(p, packed, in, M, out, stash, a, list);
    const int64_t groups = nh_ceil_div(M, 128);
    int rc = NERFHIP_OK;
    if (p->W == 256 && p->view) NH_FWD16(256, true)
    else if (p->W == 256) NH_FWD16(256, false)
    else if (p->W == 128 && p->view) NH_FWD16(128, true)
    else if (p->W == 128) NH_FWD16(128, false)
    else if (p->view) NH_FWD16(64, true)
    else NH_FWD16(64, false)
    return nh_launch_status("mlp_fwd16");
}

int nh_mlp16_dgrad(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash, float* scratch,
                   const NhCompact* cx, nerfhip_stream_t stream) {
    if (p->W == 512) return nh_mlp16_dgrad_w512(p, packed, g_out, M, stash, scratch, cx, stream);
    Dgrad16Args d;
    fill_dgrad_args(p, packed, g_out, M, stash, scratch, cx, d);
    const int64_t groups = nh_ceil_div(M, 128);
    int rc = NERFHIP_OK;
    if (p->W == 256 && p->view) NH_BWD16(256, true)
    else if (p->W == 256) NH_BWD16(256, false)
    else if (p->W == 128 && p->view) NH_BWD16(128, true)
    else if (p->W == 128) NH_BWD16(128, false)
    else if (p->view) NH_BWD16(64, true)
    else NH_BWD16(64, false)
    return nh_launch_status("mlp_dgrad16");
}
#endif
#undef NH_FWD16
#undef NH_FWD16_E
#undef NH_FWD16_T
#undef NH_BWD16

// mlp64r.hip -- the fused, stash-free backward of 64-wide nets (config/fern.yml, config/llff.yml: 4 x 64): persistent workgroups
// that keep the WHOLE net in LDS and, per 64 sample points, recompute the forward (nerf/models.py:233-256), run the data-gradient
// chain and sum the weight gradients (what autograd does behind train_nerf.py:259) -- no activation stash, no d(pre-activation)
// images, no separate weight-gradient kernel.  What reaches HBM is one partial gradient per workgroup (fixed-order reduction, no
// atomics: k_bwd64r_reduce).
//
// Why a different data flow for this width: a 4 x 64 net is 88 KB of weights -- it fits a CU's LDS, and a layer is 64 MFMAs per
// 16 samples.  The general kernels (mlp16.hip, wgrad.hip) stream every layer L2 -> LDS once per 64 samples behind a barrier, write
// 1.7 KB of stash and 1.5 KB of d(pre-activation) rows per sample and read both back: at this width they sit at 0.17-0.30 of the
// matrix pipe and 0.2-0.3 of HBM -- neither roof (profiles/r06_bench_line.json, labelled line fern_fp32).
//
// Layout vocabulary (nh_plan.h): a wave owns 16 sample points, lane l = (sample j = l & 15, k-group g = l >> 4), an activation of
// 64 units lives in 16 registers (register r: unit nh_feat16(r, g)) -- the C/D layout of v_mfma_f32_16x16x4_f32, so the forward and
// the transposed chain run register to register exactly as in mlp16.hip, with the A operands read from the resident image
// (nh_r64.h: one row-major, chunk-swizzled copy serves both orientations without bank conflicts).
//
// The weight gradient dW[out][in] = sum_samples dP[out][s] H[in][s] contracts over SAMPLES: both MFMA operands must have the unit on
// the lane's low bits and the sample on k -- the transpose of what the chain holds -- and its 88 KB of accumulators are 344 registers
// per lane: more than a wave that also carries the activations of every layer can hold.  So the workgroup's eight waves take TWO
// ROLES, one wave of each per SIMD:
//   chain waves (0..3): one 16-sample tile each per round -- encode, forward (registers only: X, H_0 .. H_{L-1}, FEAT, DIRH stay
//       live), then layer by layer the d(pre-activation) P; per step they write the step's operand blocks (16 units x 16 samples,
//       1 KiB each) into the exchange area, sample-major with an XOR swizzle (conflict-free 16-byte stores and 4-byte operand reads);
//   weight-gradient waves (4..7): wave v owns a quarter of every layer's 16 x 16 gradient tiles (25 tiles = 100 accumulator
//       registers for a 4-layer net, + 7 row sums for the biases) and accumulates them over the round's four tiles (k = 64 samples
//       per step), operands double buffered in registers it has to spare.
// The roles overlap: while the weight-gradient waves multiply step k, the chain waves compute the transposed layer of step k + 1 --
// the matrix pipe of a SIMD always has one wave of each kind to draw from, neither kind spills, and the first design's second
// barrier per step guards nothing but the hand-over.  (First design, all eight waves in both roles on 8 tiles per round: 256
// registers + 70..200 spilled, operands read just in time for want of registers, 0.54 of the fp32 MFMA peak at best.)
//
// One round (64 sample points; L layers).  Per step the chain waves do [free barrier | write operands | publish barrier], the
// weight-gradient waves [free barrier | publish barrier | multiply]:
//   encode | layer1 .. layers_dir (forward)
//   step a: POUT PDIR DIRH D      tiles: fc_rgb, layers_dir's direction columns        chain meanwhile: dFEAT
//   step b: FEAT                  tiles: layers_dir's hidden columns                   chain meanwhile: dH_{L-1}
//   step c: PFEAT H_{L-1}         tiles: fc_feat, fc_alpha                             chain meanwhile: dH_{L-2}
//   step d_k: P_{i+1} H_i         tiles: layers_xyz[i], i = L-2 .. 0                   chain meanwhile: dH_{i-1}
//   step e: P_0 X                 tiles: layer1                                        chain meanwhile: the next round's forward
// The samples may be a compaction list (compact.hip): slot c computes sample idx[c] -- a gather that costs nothing here, since the
// forward is recomputed from the rays anyway.
#include <stdlib.h>

#include "nh_mlp.h"
#include "nh_r64.h"

namespace {

constexpr int NWV = R64_WAVES;

struct Bwd64rArgs {
    const float* image;  // the plan's resident image inside the packed buffer (nh_r64.h)
    unsigned image_bytes;
    int64_t M;
    const float* rays;
    int ray_stride;
    const float* z;
    int S;
    float fx[16], fd[16];
    int Lx, Ld;
    const float* g_out;  // d(loss)/d(raw output) [M, 4]
    float* partial;      // [gridDim.x][r64_partial_floats(L)]
    const int* cidx;     // compaction list or NULL (dense: slot c is sample c)
    const int* cstats;
    const float* stash;  // k_bwd64r<L, true>: the register-image stash k_fwd64r<L, true> wrote (nh_r64.h), dense launches only
    unsigned long long* clk;
};

// NQ quads (four registers each) of this lane from / to a register-image tile (p = tile + quad * 256 + lane * 4)
template <int NQ>
NH_DEVICE void quads_load(float* dst, const float* p) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 t4 = *(const float4*)(p + q * 256);
        dst[4 * q] = t4.x, dst[4 * q + 1] = t4.y, dst[4 * q + 2] = t4.z, dst[4 * q + 3] = t4.w;
    }
}
template <int NQ>
NH_DEVICE void quads_store(float* p, const float* src) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) nh_store4(p + q * 256, src[4 * q], src[4 * q + 1], src[4 * q + 2], src[4 * q + 3]);
}

#ifdef NH_PHASE_TIMING  // (`make dbg` builds only: shader cycles per phase, summed over the waves of a role; scripts/r64_phases.py)
__device__ unsigned long long g_ph64[16];
#define PH_DECL unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last_ = clock64()
#define PH(i)                                   \
    do {                                        \
        const unsigned long long t_ = clock64(); \
        ph_[i] += t_ - ph_last_;                \
        ph_last_ = t_;                          \
    } while (0)
#define PH_FLUSH(base)                                                   \
    do {                                                                 \
        if (lane == 0)                                                   \
            for (int q_ = 0; q_ < 8; ++q_) atomicAdd(&g_ph64[(base) + q_], ph_[q_]); \
    } while (0)
#else
#define PH_DECL
#define PH(i)
#define PH_FLUSH(base)
#endif

NH_DEVICE float sel3(int a, float x, float y, float z) { return a == 0 ? x : (a == 1 ? y : z); }

// encoding registers of lane group g (slot layout: plan.cpp build_slot_map16; the same arithmetic as mlp16.hip encode_slots16).
// freqs: the 16 frequency bands, in LDS -- the band of a slot depends on the lane: as kernel arguments (scalar registers) each band is
// a 16-way select per lane, which the optimiser hoists out of the round loop as per-lane constants (band, axis, validity of every
// slot: ~100 registers, all spilled: the first build's round started with a dozen scratch reloads, each waited for); as an LDS table
// it is one ds_read per slot pair.  g arrives opaque (see the call) so that nothing here is a loop invariant.
template <int KR>
NH_DEVICE void encode_slots(float* e, float x, float y, float z, int g, const float* freqs, int Lf) {
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
    if (g == 3) {
        e[KR - 3] = x;
        e[KR - 2] = y;
        e[KR - 1] = z;
    }
}
// (a value the optimiser must treat as new: what is computed from it is not hoisted out of the enclosing loop)
NH_DEVICE int opaque(int v) {
#ifndef NERFHIP_EMU
    asm volatile("" : "+v"(v));
#endif
    return v;
}

#ifdef NH64_PAD_LAYOUT
#define FOFF(R, fx) (16 * (R) + (fx))
#else
#define FOFF(R, fx) ((16 * (R)) ^ (fx))
#endif
NH_DEVICE float pick4(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// acc[t] (16 rows x 16 samples) = bias rows 16 t + 4 g .. + 3
template <int T>
NH_DEVICE void bias_init(f32x4* acc, const float* bias_g) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float4 b4 = *(const float4*)(bias_g + 16 * t);
        acc[t][0] = b4.x, acc[t][1] = b4.y, acc[t][2] = b4.z, acc[t][3] = b4.w;
    }
}
template <int T>
NH_DEVICE void zero_acc(f32x4* acc) {
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.0f;
}

// forward: acc[t] += W[16 t + i][k cols] * in, KR k-registers (a multiple of 4).  w = (the 64-float half of) this lane's row i of
// the matrix; the chunk of k-registers 4 R .. 4 R + 3 of lane group g sits at float (16 R) ^ fx of a row = i (mod 16),
// fx = 16 (i >> 2) + 4 (g ^ (i & 3))   (= 4 ((4 R + g) ^ i): nh_r64.h).  Two output tiles at a time (their MFMAs alternate: no
// instruction waits for its predecessor), the A operands of the next pair read while these run: 4 x 4 operand registers in
// flight instead of the 8 x 4 of a whole-layer double buffer.
template <int KR, int T, int STRIDE>
NH_DEVICE void gemm_f(const float* w, int fx, const float* in, f32x4* acc) {
    static_assert(KR % 4 == 0 && T % 2 == 0, "four k-steps per 16-byte read, two tiles per step");
    constexpr int NS = (KR / 4) * (T / 2);  // steps: (R, tile pair), R-major
#ifndef NERFHIP_EMU
    asm volatile("" : "+v"(fx));  // (opaque: the four offsets (16 R) ^ fx are formed per call, not kept live across the round)
#endif
    float4 a[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) a[0][u] = *(const float4*)(w + 16 * u * STRIDE + FOFF(0, fx));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int R = s / (T / 2), tp = s % (T / 2);
        if (s + 1 < NS) {
            const int Rn = (s + 1) / (T / 2), tn = (s + 1) % (T / 2);
#pragma unroll
            for (int u = 0; u < 2; ++u) a[(s + 1) & 1][u] = *(const float4*)(w + 16 * (2 * tn + u) * STRIDE + FOFF(Rn, fx));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[2 * tp + u] = nh_mfma16(pick4(a[s & 1][u], c), in[4 * R + c], acc[2 * tp + u]);
        }
    }
}

// transposed: acc[t] (in units 16 t + i) += sum over out rows nh_feat16(r, g) of W[row][16 t + i] * dp[r].  wt = this lane's row 4 g
// of the matrix + (i & 3); row 16 R + 4 g + c holds column 16 t + i at dword ((16 t) ^ g16) + ((4 c) ^ i4) + (i & 3), g16 = 16 g,
// i4 = 4 (i >> 2)   (nh_r64.h)
template <int KR, int T, int STRIDE>
NH_DEVICE void gemm_t(const float* wt, int g16, int i4, const float* dp, f32x4* acc) {
#ifndef NERFHIP_EMU
    // (opaque to the optimiser: otherwise the sixteen lane offsets ((16 t) ^ g16) + ((4 c) ^ i4) are hoisted out of the round loop as
    // sixteen more live registers of a kernel that has none to spare)
    asm volatile("" : "+v"(g16), "+v"(i4));
#endif
#ifdef NH64_PAD_LAYOUT
    {
        float a[2][T];
#pragma unroll
        for (int t = 0; t < T; ++t) a[0][t] = wt[16 * t];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            if (r + 1 < KR) {
                const int row = 16 * ((r + 1) >> 2) + ((r + 1) & 3);
#pragma unroll
                for (int t = 0; t < T; ++t) a[(r + 1) & 1][t] = wt[row * STRIDE + 16 * t];
            }
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = nh_mfma16(a[r & 1][t], dp[r], acc[t]);
        }
        return;
    }
#endif
    int tg[T];
#pragma unroll
    for (int t = 0; t < T; ++t) tg[t] = (16 * t) ^ g16;
    float a[2][T];
#pragma unroll
    for (int t = 0; t < T; ++t) a[0][t] = wt[tg[t] + i4];
#pragma unroll
    for (int r = 0; r < KR; ++r) {
        if (r + 1 < KR) {
            const float* const pr = wt + (16 * ((r + 1) >> 2) + ((r + 1) & 3)) * STRIDE + ((4 * ((r + 1) & 3)) ^ i4);
#pragma unroll
            for (int t = 0; t < T; ++t) a[(r + 1) & 1][t] = pr[tg[t]];
        }
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = nh_mfma16(a[r & 1][t], dp[r], acc[t]);
    }
}

// NB gradient tiles that share their A block, over the round: acc[n] (16 out rows x 16 in units) += sum over the tiles' 16 samples
// of A[row][s] B_n[unit][s].  rq[q & 1] + 64 q = this lane's element of k-step q inside a block of the exchange area `ex`; aoff / boff = the A block's / the first B block's slot (floats; B blocks in
// consecutive slots).  Every read is (one of four pointers formed here) + a constant: formed from separate lane and block offsets,
// the compiler kept one address register per (tile, k-step, block) -- ~100 more live registers, all spilled.  The operands of the next
// tile are read while this one is multiplied.  rs += the A operands this lane saw (row sums = bias gradients; lane group g' holds the
// samples = g' mod 4)
template <int NB>
NH_DEVICE void unitN(const float* ex, const int* rq, int aoff, int boff, f32x4* acc, float& rs) {
    // (the four offsets are opaque INTEGERS, the base stays the LDS array: as opaque pointers the reads became flat loads)
    int oa[2] = {rq[0] + aoff, rq[1] + aoff}, ob[2] = {rq[0] + boff, rq[1] + boff};
#ifndef NERFHIP_EMU
    asm volatile("" : "+v"(oa[0]), "+v"(oa[1]), "+v"(ob[0]), "+v"(ob[1]));
#endif
    float av[2][4], bv[2][NB][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        av[0][q] = ex[oa[q & 1] + 64 * q];
#pragma unroll
        for (int n = 0; n < NB; ++n) bv[0][n][q] = ex[ob[q & 1] + 256 * n + 64 * q];
    }
#pragma unroll
    for (int t = 0; t < R64_TILES; ++t) {
        if (t + 1 < R64_TILES) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                av[(t + 1) & 1][q] = ex[oa[q & 1] + (t + 1) * R64_TILE_F + 64 * q];
#pragma unroll
                for (int n = 0; n < NB; ++n) bv[(t + 1) & 1][n][q] = ex[ob[q & 1] + (t + 1) * R64_TILE_F + 256 * n + 64 * q];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[n] = nh_mfma16(av[t & 1][q], bv[t & 1][n][q], acc[n]);
            rs += av[t & 1][q];
        }
    }
}

// this lane's four registers of a block -> slot `slot` of its wave's tile (pw = tile + j * 16 + 4 (g ^ ((j >> 1) & 3)): a block is
// [sample][16 units], the four 16-byte chunks of sample s swizzled with (s >> 1) & 3 -- a 16-byte store is served in groups of 8
// consecutive lanes over 32 banks: their eight chunks land on eight different bank quads; a unit's 4-byte reads (two groups of 32 lanes,
// k-groups {0, 1} / {2, 3}: samples of different parity, 16 units each) hit 32 different banks)
NH_DEVICE void ex_put(float* pw, int slot, float v0, float v1, float v2, float v3) {
    float4 v;
    v.x = v0, v.y = v1, v.z = v2, v.w = v3;
    *(float4*)(pw + slot * 256) = v;
}
template <int NB>
NH_DEVICE void ex_put_blocks(float* pw, int slot0, const float* regs) {
#pragma unroll
    for (int b = 0; b < NB; ++b) ex_put(pw, slot0 + b, regs[4 * b], regs[4 * b + 1], regs[4 * b + 2], regs[4 * b + 3]);
}

// v where the stored post-ReLU activation h is positive, else +0 (h >= +0: its bit pattern is non-zero iff h > 0)
NH_DEVICE float gate_pos(float v, float h) {
    unsigned u;
    memcpy(&u, &h, 4);
    return u != 0u ? v : 0.0f;
}

// ST: the stashed variant (mode 5).  The chain waves do not recompute the forward: X, D, H_0 .. H_{L-1}, FEAT, DIRH of their tile come from
// the register-image stash the training forward (k_fwd64r<L, true>) left, each array re-loaded for the NEXT round right behind its last
// use in this one -- the registers are the same, the loads have a whole round to land, and they arrive in the order the next round
// needs them (DIRH, D, d(raw) first, X last).  The weight-gradient waves prepare nothing.  Same arithmetic on the same values in the
// same order as the recomputing variant: the gradient is bit-identical.
template <int L, bool ST>
NH_KERNEL void NH_LB(64 * NWV, 2) k_bwd64r(Bwd64rArgs a) {
    constexpr R64Layout Y = r64_layout(L);
    constexpr int NU = r64_units(L), NB = r64_bias_regs(L), TL = R64_TILES;
    NH_DYN_LDS(lds_raw);
    float* const lds = (float*)lds_raw;
    // ST: layer1's weights (the image's last segment: only the forward reads them) and the hand-over area are not needed -- their room
    // holds a SECOND exchange buffer: the chain waves write step k + 1 into one while the weight-gradient waves multiply step k out of
    // the other, and a step has ONE barrier (publish) instead of two (free, publish): the chain never waits for its stores' turn.
    // `bo` = the float offset of the buffer of the current step (both roles toggle it after every step).  Operands that outlive their
    // step: POUT (slot 0 of step a's buffer) is read again in step c -- two steps later: the same buffer, whose slot 0 nobody writes in
    // between --, PDIR (step a) again in step b: the chain puts it into step b's buffer as well.
    constexpr int IMG_F = ST ? Y.res_floats : Y.image_floats;
    static_assert(!ST || Y.res_floats + 2 * R64_EX_F <= r64_lds_floats(L), "the second exchange buffer fits where layer1 and the hand-over area were");
    float* const ex = lds + IMG_F;
    float* const lfreq = lds + r64_lds_floats(L);  // 16 xyz + 16 direction frequency bands
    nh_clk_begin(a.clk, (unsigned long long*)(lds_raw + (r64_lds_floats(L) + 32) * 4));
    const int lane = nh_lane(), g = lane >> 4, j = lane & 15, wave = nh_wave_in_block();
    {
        // the whole image: 1-KiB pieces, wave w takes pieces w, w + 8, ...
        const unsigned lds_addr = nh_lds_addr(lds);
        const NhDmaSrc dma = nh_dma_src(a.image, a.image_bytes);
        for (int q = wave; q < IMG_F / 256; q += NWV) nh_dma16a(dma, lane * 16, q * 1024, lds_addr + (unsigned)q * 1024u);
        if (threadIdx.x < 32) lfreq[threadIdx.x] = threadIdx.x < 16 ? a.fx[threadIdx.x & 15] : a.fd[threadIdx.x & 15];
        nh_wait_vmem();
        nh_block_sync();
    }
    const int n_slots = a.cidx ? nh_uload_i32(a.cstats, NH_CSTAT_ACTIVE) : (int)a.M;
    const int rounds = (n_slots + 16 * TL - 1) / (16 * TL);
    // this lane's element of k-step q when it reads a block of the exchange area (sample s = 4 q + g: float
    // s * 16 + 4 ((j >> 2) ^ ((s >> 1) & 3)) + (j & 3) = 64 q + rq[q & 1])
    int rq[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) rq[q] = 16 * g + 4 * ((j >> 2) ^ ((2 * q + (g >> 1)) & 3)) + (j & 3);
    int bo = 0;  // (ST: the exchange buffer of the step at hand, 0 / R64_EX_F)

    if (wave >= TL) {
        // ================================ weight-gradient wave v: a quarter of every layer's gradient tiles ================================
        const int v = wave - TL;
        f32x4 U[NU];
        float rsum[NB];
#pragma unroll
        for (int u = 0; u < NU; ++u) U[u][0] = U[u][1] = U[u][2] = U[u][3] = 0.0f;
#pragma unroll
        for (int b = 0; b < NB; ++b) rsum[b] = 0.0f;
        // While the chain waves run their forward this wave has nothing to multiply: it prepares the inputs of ITS tile of the NEXT
        // round -- sample, depth, ray, both encodings (twelve sincosf per lane: 10 % of a round when the chain waves did it) -- and hands
        // them over through LDS.
        float* const pass = ex + R64_EX_F + v * R64_PASS_TILE_F + lane * 4;
        float Xn[NH16_KRX], Dn[NH16_KRD];
        auto prep = [&](int rd) {
            const int slot = rd * (16 * TL) + v * 16 + j;
            // (a list is padded with sample 0 up to a multiple of 128; a dense tail computes the last sample)
            const int m = a.cidx ? a.cidx[slot] : (slot < n_slots ? slot : (int)a.M - 1);
            const float* const rr = a.rays + (size_t)(m / a.S) * a.ray_stride;
            const float zz = a.z[m];
            // pts = ro + rd * z   (nerf/train_utils.py:67,107)
            const float px = rr[0] + rr[3] * zz, py = rr[1] + rr[4] * zz, pz = rr[2] + rr[5] * zz;
            encode_slots<NH16_KRX>(Xn, px, py, pz, opaque(g), lfreq, a.Lx);
            encode_slots<NH16_KRD>(Dn, rr[8], rr[9], rr[10], opaque(g), lfreq + 16, a.Ld);
        };
        auto hand_over = [&]() {
#pragma unroll
            for (int q = 0; q < 4; ++q) nh_store4(pass + q * 256, Xn[4 * q], Xn[4 * q + 1], Xn[4 * q + 2], Xn[4 * q + 3]);
#pragma unroll
            for (int q = 0; q < 2; ++q) nh_store4(pass + (4 + q) * 256, Dn[4 * q], Dn[4 * q + 1], Dn[4 * q + 2], Dn[4 * q + 3]);
        };
        if (!ST && (int)blockIdx.x < rounds) {
            prep((int)blockIdx.x);
            hand_over();
        }
        nh_block_sync();  // (the first round's hand-over is in LDS)
        PH_DECL;
        for (int round = (int)blockIdx.x; round < rounds; round += (int)gridDim.x) {
            const bool more = !ST && round + (int)gridDim.x < rounds;
            if (more) prep(round + (int)gridDim.x);
            PH(3);  // [11] preparing the next round
            // step a: slot 0 POUT, 1..2 PDIR, 3..4 DIRH, 5..6 D -- fc_rgb (waves 0, 1: POUT x DIRH block v),
            // layers_dir's direction columns (PDIR block v >> 1 x D block v & 1).  The chain waves read the hand-over area at the top
            // of their round, i.e. before they reach this barrier: the next round's may be written behind it.
            if (!ST) nh_block_sync();
            if (more) hand_over();
            nh_block_sync();
            PH(0);  // [8] waiting for step a (the chain waves' forward)
            if (v < 2) unitN<1>(ex, rq, bo, bo + (3 + v) * 256, &U[0], rsum[0]);
            unitN<1>(ex, rq, bo + (1 + (v >> 1)) * 256, bo + (5 + (v & 1)) * 256, &U[1], rsum[1]);
            if (ST) bo ^= R64_EX_F;
            PH(1);  // [9] multiplying
            // step b: slots 3..6 FEAT -- layers_dir's hidden columns (PDIR block v >> 1 x FEAT blocks 2 (v & 1), + 1)
            if (!ST) nh_block_sync();
            nh_block_sync();
            PH(2);  // [10] waiting for the steps b ..
            {
                float unused = 0.0f;
                unitN<2>(ex, rq, bo + (1 + (v >> 1)) * 256, bo + (3 + 2 * (v & 1)) * 256, &U[2], unused);
            }
            if (ST) bo ^= R64_EX_F;
            PH(1);
            // step c: slots 1..4 PFEAT, 5..8 H_{L-1} (POUT still in 0) -- fc_feat rows 16 v .., fc_alpha's units 16 v ..
            if (!ST) nh_block_sync();
            nh_block_sync();
            PH(2);
            unitN<4>(ex, rq, bo + (1 + v) * 256, bo + 5 * 256, &U[4], rsum[2]);
            {
                float unused = 0.0f;
                unitN<1>(ex, rq, bo, bo + (5 + v) * 256, &U[8], unused);
            }
            if (ST) bo ^= R64_EX_F;
            PH(1);
            // steps d_k: slots 1..4 P_{i+1}, 5..8 H_i -- layers_xyz[i] rows 16 v .., i = L-2-k; step e: P_0 and X -- layer1
#pragma unroll
            for (int k = 0; k < L; ++k) {
                if (!ST) nh_block_sync();
                nh_block_sync();
                PH(2);
                unitN<4>(ex, rq, bo + (1 + v) * 256, bo + 5 * 256, &U[9 + 4 * k], rsum[3 + k]);
                if (ST) bo ^= R64_EX_F;
                PH(1);
            }
        }
        PH_FLUSH(8);
        // this workgroup's partial: [weight-gradient wave][register][lane]
        float* const part = a.partial + (size_t)blockIdx.x * r64_partial_floats(L) + (size_t)v * r64_regs(L) * 64 + lane;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) part[(4 * u + c) * 64] = U[u][c];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) part[(4 * NU + b) * 64] = rsum[b];
    } else {
        // ================================ chain wave: tile `wave` of every round ================================
        // lane offsets into the image (floats; nh_r64.h): forward rows / transposed rows of the 64-column matrices and of layers_dir
#ifdef NH64_PAD_LAYOUT
        const int lf = j * R64_S, lt = 4 * g * R64_S + j, lfd = j * R64_SD, ltd = 4 * g * R64_SD + j;
        const int fx = 4 * g, g16 = 0, i4 = 0;
#else
        const int lf = j * R64_S, lt = 4 * g * R64_S + (j & 3), lfd = j * R64_SD, ltd = 4 * g * R64_SD + (j & 3);
        const int fx = 16 * (j >> 2) + 4 * (g ^ (j & 3)), g16 = 16 * g, i4 = 4 * (j >> 2);
#endif
        // where this lane writes a block of its tile
        float* const pw = ex + wave * R64_TILE_F + j * 16 + 4 * (g ^ ((j >> 1) & 3));
        const float* const pass = ex + R64_EX_F + wave * R64_PASS_TILE_F + lane * 4;
        // The sample of this lane's slot in round `rd` (a list entry: fetched one round ahead).  A list is padded with sample 0 up to a
        // multiple of 128; a dense tail computes the last sample: finite values times a zero cotangent
        int m_next = 0;
        auto fetch_sample = [&](int rd) {
            const int slot = rd * (16 * TL) + wave * 16 + j;
            m_next = a.cidx ? a.cidx[slot] : (slot < n_slots ? slot : (int)a.M - 1);
        };
        if ((int)blockIdx.x < rounds) fetch_sample((int)blockIdx.x);
        // ST: the tile of this wave in round `rd` inside the register-image stash (a tile behind the last one: the last one -- finite
        // values under zero cotangents), and the d(raw) row of this lane's sample there (slot = sample: ST launches are dense)
        constexpr int TS = r64_stash_tile_floats(L);
        const int64_t tiles_total = (a.M + 15) / 16;
        auto stash_tile = [&](int rd) {
            const int64_t t = (int64_t)rd * TL + wave;
            return a.stash + (size_t)(t < tiles_total ? t : tiles_total - 1) * TS + lane * 4;
        };
        float X[NH16_KRX], Dd[NH16_KRD], H[L][16], FEAT[16], DIRH[8];
        float gn0 = 0.f, gn1 = 0.f, gn2 = 0.f, gn3 = 0.f;  // (ST) d(raw) of the round to come
        auto fetch_go = [&](int rd) {
            const int slot = rd * (16 * TL) + wave * 16 + j;
            gn0 = gn1 = gn2 = gn3 = 0.f;
            if (slot < n_slots) {
                const float4 t4 = *(const float4*)(a.g_out + (size_t)slot * 4);
                gn0 = t4.x, gn1 = t4.y, gn2 = t4.z, gn3 = t4.w;
            }
        };
        if (ST && (int)blockIdx.x < rounds) {
            const float* const t0 = stash_tile((int)blockIdx.x);
            quads_load<2>(DIRH, t0 + r64_sq_dirh(L) * 256);
            quads_load<2>(Dd, t0 + R64_SQ_D * 256);
            fetch_go((int)blockIdx.x);
            quads_load<4>(FEAT, t0 + r64_sq_feat(L) * 256);
#pragma unroll
            for (int i = L - 1; i >= 0; --i) quads_load<4>(H[i], t0 + (R64_SQ_H + 4 * i) * 256);
            quads_load<4>(X, t0 + R64_SQ_X * 256);
        }
        nh_block_sync();  // (the first round's hand-over is in LDS)
        PH_DECL;
        for (int round = (int)blockIdx.x; round < rounds; round += (int)gridDim.x) {
            const bool valid = round * (16 * TL) + wave * 16 + j < n_slots;
            const int m = m_next;
            const bool st_more = ST && round + (int)gridDim.x < rounds;
            const float* const tn = st_more ? stash_tile(round + (int)gridDim.x) : nullptr;
            float go0 = 0.f, go1 = 0.f, go2 = 0.f, go3 = 0.f;
            f32x4 acc[4];
            if (ST) {
                go0 = gn0, go1 = gn1, go2 = gn2, go3 = gn3;
                PH(0);
            } else {
            // the encodings of this tile: prepared by the weight-gradient wave of this SIMD during the round before
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t4 = *(const float4*)(pass + q * 256);
                X[4 * q] = t4.x, X[4 * q + 1] = t4.y, X[4 * q + 2] = t4.z, X[4 * q + 3] = t4.w;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 t4 = *(const float4*)(pass + (4 + q) * 256);
                Dd[4 * q] = t4.x, Dd[4 * q + 1] = t4.y, Dd[4 * q + 2] = t4.z, Dd[4 * q + 3] = t4.w;
            }
            if (round + (int)gridDim.x < rounds) fetch_sample(round + (int)gridDim.x);
            // d(loss)/d(raw output) of this lane's sample: asked for here, needed when the backward starts
            if (valid) {
                const float4 t4 = *(const float4*)(a.g_out + (size_t)m * 4);
                go0 = t4.x, go1 = t4.y, go2 = t4.z, go3 = t4.w;
            }
            PH(0);  // [0] hand-over read
            // ---- forward, registers only (nerf/models.py:233-256); H[0] = layer1(x) has no activation (:238)
            bias_init<4>(acc, lds + Y.b_l1 + 4 * g);
            gemm_f<NH16_KRX, 4, R64_S>(lds + Y.l1 + lf, fx, X, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) H[0][r] = acc[r >> 2][r & 3];
#pragma unroll
            for (int i = 0; i < L - 1; ++i) {
                bias_init<4>(acc, lds + Y.b_xyz[i] + 4 * g);
                gemm_f<16, 4, R64_S>(lds + Y.xyz[i] + lf, fx, H[i], acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) H[i + 1][r] = nh_relu(acc[r >> 2][r & 3]);
            }
            bias_init<4>(acc, lds + Y.b_feat + 4 * g);
            gemm_f<16, 4, R64_S>(lds + Y.head + lf, fx, H[L - 1], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) FEAT[r] = nh_relu(acc[r >> 2][r & 3]);
            bias_init<2>(acc, lds + Y.b_dir + 4 * g);
            gemm_f<16, 2, R64_SD>(lds + Y.dir + lfd, fx, FEAT, acc);
            gemm_f<NH16_KRD, 2, R64_SD>(lds + Y.dir + 64 + lfd, fx, Dd, acc);
#pragma unroll
            for (int r = 0; r < 8; ++r) DIRH[r] = nh_relu(acc[r >> 2][r & 3]);
            }

            PH(1);  // [1] forward
            // ---- backward.  d(DIRH pre-activation) = relu'(DIRH) * fc_rgb^T d(rgb raw): ONE k-step, group g carries d(rgb raw)[g]
            float PDIR[8];
            {
                zero_acc<2>(acc);
                const float b = g == 0 ? go0 : (g == 1 ? go1 : (g == 2 ? go2 : 0.0f));
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = nh_mfma16(lds[Y.rgb + g * R64_SR + 16 * t + j], b, acc[t]);
#pragma unroll
                for (int r = 0; r < 8; ++r) PDIR[r] = gate_pos(acc[r >> 2][r & 3], DIRH[r]);
            }
            // step a: slot 0 POUT (rows 0..2 d(rgb raw), row 3 d(sigma raw)), 1..2 PDIR, 3..4 DIRH, 5..6 D
            PH(2);  // [2] transposed layers, gates
            if (!ST) nh_block_sync();  // (free: the weight-gradient waves are done with the step before)
            PH(3);  // [3] waiting for the weight-gradient waves
            ex_put(pw + bo, 0, g == 0 ? go0 : 0.f, g == 0 ? go1 : 0.f, g == 0 ? go2 : 0.f, g == 0 ? go3 : 0.f);
            ex_put_blocks<2>(pw + bo, 1, PDIR);
            ex_put_blocks<2>(pw + bo, 3, DIRH);
            ex_put_blocks<2>(pw + bo, 5, Dd);
            nh_block_sync(); if (ST) bo ^= R64_EX_F;  // (publish)
            if (st_more) {  // (their last use is behind them: the next round's, into the same registers)
                quads_load<2>(DIRH, tn + r64_sq_dirh(L) * 256);
                quads_load<2>(Dd, tn + R64_SQ_D * 256);
                fetch_go(round + (int)gridDim.x);
            }
            PH(4);  // [4] operand stores + publish barrier
            // d(FEAT pre-activation) = relu'(FEAT) * layers_dir[:, :64]^T PDIR
            float PFEAT[16];
            zero_acc<4>(acc);
            gemm_t<8, 4, R64_SD>(lds + Y.dir + ltd, g16, i4, PDIR, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) PFEAT[r] = gate_pos(acc[r >> 2][r & 3], FEAT[r]);
            // step b: slots 3..6 FEAT (PDIR stays in 1..2)
            PH(2);
            if (!ST) nh_block_sync();
            PH(3);
            if (ST) ex_put_blocks<2>(pw + bo, 1, PDIR);  // (this step's buffer is not step a's)
            ex_put_blocks<4>(pw + bo, 3, FEAT);
            nh_block_sync(); if (ST) bo ^= R64_EX_F;
            if (st_more) quads_load<4>(FEAT, tn + r64_sq_feat(L) * 256);
            PH(4);
            // dH_{L-1} = fc_feat^T PFEAT + fc_alpha^T d(sigma raw) (one more k-step: group 0 carries d(sigma raw))
            float P[2][16];
            {
                zero_acc<4>(acc);
                gemm_t<16, 4, R64_S>(lds + Y.head + lt, g16, i4, PFEAT, acc);
                const float b = g == 0 ? go3 : 0.0f;
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = nh_mfma16(lds[Y.head + 64 * R64_S + 16 * t + j], b, acc[t]);
#pragma unroll
                for (int r = 0; r < 16; ++r) P[0][r] = L > 1 ? gate_pos(acc[r >> 2][r & 3], H[L - 1][r]) : acc[r >> 2][r & 3];
            }
            // step c: slots 1..4 PFEAT, 5..8 H_{L-1} (POUT stays in 0)
            PH(2);
            if (!ST) nh_block_sync();
            PH(3);
            ex_put_blocks<4>(pw + bo, 1, PFEAT);
            ex_put_blocks<4>(pw + bo, 5, H[L - 1]);
            nh_block_sync(); if (ST) bo ^= R64_EX_F;
            if (st_more) quads_load<4>(H[L - 1], tn + (R64_SQ_H + 4 * (L - 1)) * 256);
            PH(4);
            // steps d_k: layers_xyz[i], i = L-2-k: first dH_i = layers_xyz[i]^T P_{i+1} (P_{i+1} = P[k & 1]), then slots 1..4 P_{i+1},
            // 5..8 H_i
#pragma unroll
            for (int k = 0; k < L - 1; ++k) {
                const int i = L - 2 - k;
                zero_acc<4>(acc);
                gemm_t<16, 4, R64_S>(lds + Y.xyz[i] + lt, g16, i4, P[k & 1], acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) P[(k + 1) & 1][r] = i >= 1 ? gate_pos(acc[r >> 2][r & 3], H[i][r]) : acc[r >> 2][r & 3];
                PH(2);
                if (!ST) nh_block_sync();
                PH(3);
                ex_put_blocks<4>(pw + bo, 1, P[k & 1]);
                ex_put_blocks<4>(pw + bo, 5, H[i]);
                nh_block_sync(); if (ST) bo ^= R64_EX_F;
                if (st_more) quads_load<4>(H[i], tn + (R64_SQ_H + 4 * i) * 256);
                PH(4);
            }
            // step e: layer1: slots 1..4 P_0, 5..8 X
            PH(2);
            if (!ST) nh_block_sync();
            PH(3);
            ex_put_blocks<4>(pw + bo, 1, P[(L - 1) & 1]);
            ex_put_blocks<4>(pw + bo, 5, X);
            nh_block_sync(); if (ST) bo ^= R64_EX_F;
            if (st_more) quads_load<4>(X, tn + R64_SQ_X * 256);
            PH(4);
        }
        PH_FLUSH(0);
    }
    nh_clk_end((const unsigned long long*)(lds_raw + (r64_lds_floats(L) + 32) * 4));
}

// ---- the persistent forward with the resident image: what the stash-free training forward and inference of these nets run ----------
// Sixteen waves per workgroup (four per SIMD: a wave needs ~100 registers without a backward to serve), one workgroup per CU, the image
// copied once; a wave walks over 16-sample tiles: encode, layer1 .. fc_rgb register to register, one 16-byte store per sample.  No
// barrier after the image has landed.  Same k order per output as k_mlp_fwd16 (bias first, k-steps in register order, four lane
// groups per MFMA): bit-identical results.
constexpr int NWF = 16;
struct Fwd64rArgs {
    const float* image;
    unsigned image_bytes;
    int64_t M;
    int mode;  // 0: encoded rows x [M, dx + dd]; 1: rays + depths
    const float* x;
    int dx, dd;
    signed char xcol[4][NH16_KRX], dcol[4][NH16_KRD];  // slot (g, r) -> column of x (mode 0), or -1
    const float* rays;
    int ray_stride;
    const float* z;
    int S;
    float fx[16], fd[16];
    int Lx, Ld;
    float* out;
    float* stash;  // k_fwd64r<L, true>: the register-image stash of the stashed fused backward (nh_r64.h), ceil(M / 16) tiles
    unsigned long long* clk;
};

// one output tile: acc += W[i][k cols] * in (16 dependent MFMAs per 4 k-registers are spread over ... one accumulator: short layers only)
template <int KR, int STRIDE>
NH_DEVICE void gemm_f1(const float* w, int fx, const float* in, f32x4& acc) {
#pragma unroll
    for (int R = 0; R < KR / 4; ++R) {
        const float4 a4 = *(const float4*)(w + FOFF(R, fx));
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = nh_mfma16(pick4(a4, c), in[4 * R + c], acc);
    }
}

// ST (the training forward of mode 5): every activation the backward's chain reads also goes to the register-image stash, one 1-KiB
// store per four registers, as it is produced.
template <int L, bool ST>
NH_KERNEL void NH_LB(64 * NWF, 4) k_fwd64r(Fwd64rArgs a) {
    constexpr R64Layout Y = r64_layout(L);
    NH_DYN_LDS(lds_raw);
    float* const lds = (float*)lds_raw;
    float* const lfreq = lds + Y.image_floats;
    nh_clk_begin(a.clk, (unsigned long long*)(lds_raw + (Y.image_floats + 32) * 4));
    const int lane = nh_lane(), g = lane >> 4, j = lane & 15, wave = nh_wave_in_block();
    {
        const unsigned lds_addr = nh_lds_addr(lds);
        const NhDmaSrc dma = nh_dma_src(a.image, a.image_bytes);
        for (int q = wave; q < Y.image_floats / 256; q += NWF) nh_dma16a(dma, lane * 16, q * 1024, lds_addr + (unsigned)q * 1024u);
        if (threadIdx.x < 32) lfreq[threadIdx.x] = threadIdx.x < 16 ? a.fx[threadIdx.x & 15] : a.fd[threadIdx.x & 15];
        nh_wait_vmem();
        nh_block_sync();
    }
#ifdef NH64_PAD_LAYOUT
    const int lf = j * R64_S, lfd = j * R64_SD, fx = 4 * g;
#else
    const int lf = j * R64_S, lfd = j * R64_SD, fx = 16 * (j >> 2) + 4 * (g ^ (j & 3));
#endif
    const int64_t tiles = (a.M + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * NWF + wave; tile < tiles; tile += (int64_t)gridDim.x * NWF) {
        const int64_t slot = tile * 16 + j;
        const bool valid = slot < a.M;
        const int m = (int)(valid ? slot : a.M - 1);
        float X[NH16_KRX];
        const float* const xr = a.mode == 0 ? a.x + (size_t)m * (size_t)(a.dx + a.dd) : nullptr;
        const float* const rr = a.mode == 0 ? nullptr : a.rays + (size_t)(m / a.S) * a.ray_stride;
        if (a.mode == 0) {
#pragma unroll
            for (int r = 0; r < NH16_KRX; ++r) {
                const int c = a.xcol[g][r];
                X[r] = c >= 0 ? xr[c] : 0.0f;
            }
        } else {
            const float zz = a.z[m];
            // pts = ro + rd * z   (nerf/train_utils.py:67,107)
            const float px = rr[0] + rr[3] * zz, py = rr[1] + rr[4] * zz, pz = rr[2] + rr[5] * zz;
            encode_slots<NH16_KRX>(X, px, py, pz, opaque(g), lfreq, a.Lx);
        }
        float* const st = ST ? a.stash + (size_t)tile * r64_stash_tile_floats(L) + lane * 4 : nullptr;
        if (ST) quads_store<4>(st + R64_SQ_X * 256, X);
        f32x4 acc[4];
        float Hc[16];
        bias_init<4>(acc, lds + Y.b_l1 + 4 * g);
        gemm_f<NH16_KRX, 4, R64_S>(lds + Y.l1 + lf, fx, X, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) Hc[r] = acc[r >> 2][r & 3];  // no activation after layer1 (models.py:238)
        if (ST) quads_store<4>(st + R64_SQ_H * 256, Hc);
#pragma unroll
        for (int i = 0; i < L - 1; ++i) {
            bias_init<4>(acc, lds + Y.b_xyz[i] + 4 * g);
            gemm_f<16, 4, R64_S>(lds + Y.xyz[i] + lf, fx, Hc, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) Hc[r] = nh_relu(acc[r >> 2][r & 3]);
            if (ST) quads_store<4>(st + (R64_SQ_H + 4 * (i + 1)) * 256, Hc);
        }
        // fc_alpha (row 64 of the head matrix: row 0 of a fifth tile; the other rows of that tile are whatever follows the matrix and
        // are never read back), then fc_feat
        f32x4 aa;
        bias_init<1>(&aa, lds + Y.b_alpha + 4 * g);
        gemm_f1<16, R64_S>(lds + Y.head + 64 * R64_S + lf, fx, Hc, aa);
        const float alpha = aa[0];
        bias_init<4>(acc, lds + Y.b_feat + 4 * g);
        gemm_f<16, 4, R64_S>(lds + Y.head + lf, fx, Hc, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) Hc[r] = nh_relu(acc[r >> 2][r & 3]);
        if (ST) quads_store<4>(st + r64_sq_feat(L) * 256, Hc);
        float Dd[NH16_KRD];
        if (a.mode == 0) {
#pragma unroll
            for (int r = 0; r < NH16_KRD; ++r) {
                const int c = (int)a.dcol[g][r];
                Dd[r] = c >= 0 ? xr[a.dx + c] : 0.0f;
            }
        } else {
            encode_slots<NH16_KRD>(Dd, rr[8], rr[9], rr[10], opaque(g), lfreq + 16, a.Ld);
        }
        if (ST) quads_store<2>(st + R64_SQ_D * 256, Dd);
        bias_init<2>(acc, lds + Y.b_dir + 4 * g);
        gemm_f<16, 2, R64_SD>(lds + Y.dir + lfd, fx, Hc, acc);
        gemm_f<NH16_KRD, 2, R64_SD>(lds + Y.dir + 64 + lfd, fx, Dd, acc);
        float dh[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) dh[r] = nh_relu(acc[r >> 2][r & 3]);
        if (ST) quads_store<2>(st + r64_sq_dirh(L) * 256, dh);
        // fc_rgb: 16 rows x 36 floats, not swizzled
        f32x4 ar;
        bias_init<1>(&ar, lds + Y.b_rgb + 4 * g);
#pragma unroll
        for (int R = 0; R < 2; ++R) {
            const float4 a4 = *(const float4*)(lds + Y.rgb + j * R64_SR + 16 * R + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c) ar = nh_mfma16(pick4(a4, c), dh[4 * R + c], ar);
        }
        if (valid && g == 0) {
            float4 r4;
            r4.x = ar[0], r4.y = ar[1], r4.z = ar[2], r4.w = alpha;
            *(float4*)(a.out + (size_t)m * 4) = r4;
        }
    }
    nh_clk_end((const unsigned long long*)(lds_raw + (Y.image_floats + 32) * 4));
}

// ---- reduction: g_params[e] = sum over the workgroups' partials, in workgroup order ------------------------------------------------------
struct Red64rArgs {
    const float* partial;
    float* g_params;
    int nwg, L, H, Dx, Dd;
    int64_t o_l1_w, o_l1_b, o_xyz_w[R64_MAX_LAYERS], o_xyz_b[R64_MAX_LAYERS], o_dir_w, o_dir_b, o_alpha_w, o_alpha_b, o_rgb_w, o_rgb_b,
        o_feat_w, o_feat_b;
    signed char xcol[4][NH16_KRX], dcol[4][NH16_KRD];  // slot (g, r) -> reference column, or -1
};

// One block per (weight-gradient wave, register) row of the partials: 64 elements x 4 slices of the workgroup range; slice s adds the partials of
// workgroups s, s + 4, ... in four interleaved running sums, the slices are combined through LDS in slice order -- a fixed order:
// bit-reproducible, no atomics.  Each element then decodes which parameter it is (none: padding, a wave that idles in that step).
NH_KERNEL void k_bwd64r_reduce(Red64rArgs a) {
    NH_SHARED float part[4][64];
    const int L = a.L, NU = r64_units(L), NR = r64_regs(L), H = a.H, H2 = H / 2, stride = R64_TILES * NR * 64;
    const int row = (int)blockIdx.x, lane = (int)threadIdx.x & 63, slice = (int)threadIdx.x >> 6;
    const int reg = row % NR, w = row / NR;
    {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        const float* p = a.partial + (size_t)row * 64 + lane;
        int q = slice;
        for (; q + 12 < a.nwg; q += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += p[(size_t)(q + 4 * u) * stride];
        }
        for (int u = 0; q < a.nwg; q += 4, ++u) s4[u & 3] += p[(size_t)q * stride];
        part[slice][lane] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    nh_block_sync();
    const float total = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const int fl = lane & 15, gq = lane >> 4;
    int64_t dst = -1;
    if (reg < 4 * NU) {
        const int u = reg >> 2, row16 = 4 * gq + (reg & 3);  // out row inside the A block, in unit inside the B block = fl
        if (u == 0) {  // fc_rgb (waves 0, 1)
            const int col = 16 * w + fl;
            if (w < 2 && row16 < 3 && col < H2) dst = a.o_rgb_w + (int64_t)row16 * H2 + col;
        } else if (u == 1) {  // layers_dir, direction columns: D block w & 1 = slot registers 4 b .. 4 b + 3 of group fl >> 2
            const int orow = 16 * (w >> 1) + row16, c = (int)a.dcol[fl >> 2][4 * (w & 1) + (fl & 3)];
            if (orow < H2 && c >= 0) dst = a.o_dir_w + (int64_t)orow * (H + a.Dd) + H + c;
        } else if (u < 4) {  // layers_dir, hidden columns
            const int orow = 16 * (w >> 1) + row16, col = 16 * (2 * (w & 1) + (u - 2)) + fl;
            if (orow < H2 && col < H) dst = a.o_dir_w + (int64_t)orow * (H + a.Dd) + col;
        } else if (u < 8) {  // fc_feat
            const int orow = 16 * w + row16, col = 16 * (u - 4) + fl;
            if (orow < H && col < H) dst = a.o_feat_w + (int64_t)orow * H + col;
        } else if (u == 8) {  // fc_alpha: row 3 of POUT
            const int col = 16 * w + fl;
            if (row16 == 3 && col < H) dst = a.o_alpha_w + col;
        } else if (u < 9 + 4 * (L - 1)) {  // layers_xyz[i]
            const int k = (u - 9) >> 2, i = L - 2 - k, orow = 16 * w + row16, col = 16 * ((u - 9) & 3) + fl;
            if (orow < H && col < H) dst = a.o_xyz_w[i] + (int64_t)orow * H + col;
        } else {  // layer1: X block b = slot registers 4 b .. 4 b + 3 of group fl >> 2
            const int b = (u - 9) & 3, orow = 16 * w + row16, c = (int)a.xcol[fl >> 2][4 * b + (fl & 3)];
            if (orow < H && c >= 0) dst = a.o_l1_w + (int64_t)orow * a.Dx + c;
        }
        if (dst >= 0 && slice == 0) a.g_params[dst] = total;
        return;
    }
    // row sums (the whole block is in this branch: `reg` is the block's): lane group g' holds the samples = g' mod 4 -- lanes 0..15
    // add the four groups
    nh_block_sync();
    if (slice == 0) part[0][lane] = total;
    nh_block_sync();
    if (gq != 0 || slice != 0) return;
    const int b = reg - 4 * NU;
    if (b == 0) {  // POUT: rows 0..2 fc_rgb's bias, row 3 fc_alpha's (wave 0's sums)
        if (w == 0) dst = fl < 3 ? a.o_rgb_b + fl : (fl == 3 ? a.o_alpha_b : -1);
    } else if (b == 1) {  // PDIR block w >> 1 (waves 0 and 2)
        if ((w & 1) == 0 && 16 * (w >> 1) + fl < H2) dst = a.o_dir_b + 16 * (w >> 1) + fl;
    } else {
        const int orow = 16 * w + fl;
        if (orow < H) {
            if (b == 2)
                dst = a.o_feat_b + orow;
            else if (b < L + 2)
                dst = a.o_xyz_b[L - 2 - (b - 3)] + orow;
            else
                dst = a.o_l1_b + orow;
        }
    }
    if (dst >= 0) a.g_params[dst] = (part[0][fl] + part[0][fl + 16]) + (part[0][fl + 32] + part[0][fl + 48]);
}

int compute_units() {
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
    return 3;  // (the CPU suite walks the persistent loop)
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

template <int L>
int launch_bwd(const Bwd64rArgs& a, int grid, nerfhip_stream_t stream) {
    const int bytes = (r64_lds_floats(L) + 32) * 4 + NH_CLK_LDS_BYTES;
    if (a.stash) {
        const int rc = lds_limit(k_bwd64r<L, true>, bytes);
        if (rc) return rc;
        NH_LAUNCH((k_bwd64r<L, true>), grid, 64 * NWV, bytes, stream, a);
        return nh_launch_status("bwd64r (stashed)");
    }
    const int rc = lds_limit(k_bwd64r<L, false>, bytes);
    if (rc) return rc;
    NH_LAUNCH((k_bwd64r<L, false>), grid, 64 * NWV, bytes, stream, a);
    return nh_launch_status("bwd64r");
}

}  // namespace

#ifdef NH_PHASE_TIMING
extern "C" int nerfhip_debug_phases64(unsigned long long* host16, int reset) {
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_ph64), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ph64), z, sizeof(z));
    }
    return 0;
}
#endif

template <int L>
static int launch_fwd(const Fwd64rArgs& a, int grid, nerfhip_stream_t stream) {
    const int bytes = (r64_layout(L).image_floats + 32) * 4 + NH_CLK_LDS_BYTES;
    if (a.stash) {
        const int rc = lds_limit(k_fwd64r<L, true>, bytes);
        if (rc) return rc;
        NH_LAUNCH((k_fwd64r<L, true>), grid, 64 * NWF, bytes, stream, a);
        return nh_launch_status("fwd64r (stashing)");
    }
    const int rc = lds_limit(k_fwd64r<L, false>, bytes);
    if (rc) return rc;
    NH_LAUNCH((k_fwd64r<L, false>), grid, 64 * NWF, bytes, stream, a);
    return nh_launch_status("fwd64r");
}

// The forward of an eligible plan without a stash (inference, the stash-free training forward of the fused modes): raw[M, 4]
// ... or, stash != NULL (the training forward of mode 5), with the register-image stash of nh_r64.h: ceil(M / 16) tiles
int nh_mlp64r_forward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                      nerfhip_stream_t stream) {
    NH_REQUIRE(nh_r64_eligible(p) && p->r64_off >= 0, "fwd64r: the plan has no resident image");
    NH_REQUIRE(!stash || nh_r64_stash_fits(p), "fwd64r: the register-image stash does not fit this plan's stash region");
    NH_REQUIRE(packed && out && M > 0 && M < ((int64_t)1 << 31), "fwd64r: bad arguments");
    Fwd64rArgs a;
    memset(&a, 0, sizeof(a));
    const R64Layout Y = r64_layout(p->L);
    a.image = packed + p->r64_off;
    a.image_bytes = (unsigned)(Y.image_floats * 4);
    a.M = M;
    a.mode = in.mode;
    a.x = in.x;
    a.dx = p->Dx;
    a.dd = p->Dd;
    for (int g = 0; g < 4; ++g) {
        for (int r = 0; r < NH16_KRX; ++r) a.xcol[g][r] = (signed char)p->xyz_col16[g][r];
        for (int r = 0; r < NH16_KRD; ++r) a.dcol[g][r] = (signed char)p->dir_col16[g][r];
    }
    a.rays = in.rays;
    a.ray_stride = in.ray_stride;
    a.z = in.z;
    a.S = in.S;
    for (int k = 0; k < 16; ++k) {
        a.fx[k] = p->freqs_xyz[k];
        a.fd[k] = p->freqs_dir[k];
    }
    a.Lx = p->cfg.num_encoding_fn_xyz;
    a.Ld = p->cfg.num_encoding_fn_dir;
    a.out = out;
    a.stash = stash;
    a.clk = nh_prof_clock_slot(NH_CLK_FWD);
    const int64_t wgs = nh_ceil_div(nh_ceil_div(M, 16), NWF);
    const int cus = compute_units();
    const int grid = (int)(wgs < cus ? wgs : cus);
    switch (p->L) {
        case 1: return launch_fwd<1>(a, grid, stream);
        case 2: return launch_fwd<2>(a, grid, stream);
        case 3: return launch_fwd<3>(a, grid, stream);
        default: return launch_fwd<4>(a, grid, stream);
    }
}

// workgroups of a fused backward over M sample points: one per compute unit, at most one per round
static int r64_grid(int64_t M) {
    const int64_t rounds = nh_ceil_div(M, 16 * R64_TILES);
    const int cus = compute_units();
    return (int)(rounds < cus ? (rounds < 1 ? 1 : rounds) : cus);
}

int64_t nh_mlp64r_partial_floats(const nerfhip_plan* p, int64_t M) {
    if (!nh_r64_eligible(p)) return 0;
    return (int64_t)r64_grid(M) * r64_partial_floats(p->L);
}

// The fused backward of an eligible plan (nh_r64_eligible) over the fused render's input: g_params = d(loss)/d(parameters) for
// d(loss)/d(raw output) = g_out.  cx: a compaction list built from g_out, or NULL (every sample).  partial: nh_mlp64r_partial_floats.
// stash: the register-image stash nh_mlp64r_forward left for these M sample points (then cx must be NULL: the stashed variant runs
// over every sample and recomputes nothing), or NULL (the forward is recomputed from `in`).
int nh_mlp64r_backward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, const float* g_out, int64_t M, float* partial,
                       float* g_params, const NhCompact* cx, const float* stash, nerfhip_stream_t stream) {
    NH_REQUIRE(nh_r64_eligible(p) && p->r64_off >= 0, "bwd64r: the plan has no resident image");
    NH_REQUIRE(!stash || (!cx && nh_r64_stash_fits(p)), "bwd64r: the stashed variant runs dense, over a stash that fits the plan's region");
    NH_REQUIRE(in.mode == 1 && in.rays && in.z && in.S > 0 && in.ray_stride >= 11 && p->freqs_set, "bwd64r: bad fused input");
    NH_REQUIRE(packed && g_out && partial && g_params && M > 0 && M < ((int64_t)1 << 31), "bwd64r: bad arguments");
    Bwd64rArgs a;
    memset(&a, 0, sizeof(a));
    const R64Layout Y = r64_layout(p->L);
    a.image = packed + p->r64_off;
    a.image_bytes = (unsigned)(Y.image_floats * 4);
    a.M = M;
    a.rays = in.rays;
    a.ray_stride = in.ray_stride;
    a.z = in.z;
    a.S = in.S;
    for (int k = 0; k < 16; ++k) {
        a.fx[k] = p->freqs_xyz[k];
        a.fd[k] = p->freqs_dir[k];
    }
    a.Lx = p->cfg.num_encoding_fn_xyz;
    a.Ld = p->cfg.num_encoding_fn_dir;
    a.g_out = g_out;
    a.partial = partial;
    a.cidx = cx ? cx->idx : nullptr;
    a.cstats = cx ? cx->stats : nullptr;
    a.stash = stash;
    a.clk = nh_prof_clock_slot(NH_CLK_DGRAD);
    const int grid = r64_grid(M);
    int rc = NERFHIP_OK;
    switch (p->L) {
        case 1: rc = launch_bwd<1>(a, grid, stream); break;
        case 2: rc = launch_bwd<2>(a, grid, stream); break;
        case 3: rc = launch_bwd<3>(a, grid, stream); break;
        default: rc = launch_bwd<4>(a, grid, stream); break;
    }
    if (rc) return rc;
    Red64rArgs r;
    memset(&r, 0, sizeof(r));
    r.partial = partial;
    r.g_params = g_params;
    r.nwg = grid;
    r.L = p->L;
    r.H = p->H;
    r.Dx = p->Dx;
    r.Dd = p->Dd;
    auto off = [p](int t) { return p->tensors[t].off; };
    r.o_l1_w = off(p->t_layer1_w), r.o_l1_b = off(p->t_layer1_b);
    for (int i = 0; i < p->L - 1; ++i) r.o_xyz_w[i] = off(p->t_xyz_w[i]), r.o_xyz_b[i] = off(p->t_xyz_b[i]);
    r.o_dir_w = off(p->t_dir_w), r.o_dir_b = off(p->t_dir_b), r.o_alpha_w = off(p->t_alpha_w), r.o_alpha_b = off(p->t_alpha_b);
    r.o_rgb_w = off(p->t_rgb_w), r.o_rgb_b = off(p->t_rgb_b), r.o_feat_w = off(p->t_feat_w), r.o_feat_b = off(p->t_feat_b);
    for (int g = 0; g < 4; ++g) {
        for (int k = 0; k < NH16_KRX; ++k) r.xcol[g][k] = (signed char)p->xyz_col16[g][k];
        for (int k = 0; k < NH16_KRD; ++k) r.dcol[g][k] = (signed char)p->dir_col16[g][k];
    }
    NH_LAUNCH(k_bwd64r_reduce, R64_TILES * r64_regs(p->L), 256, 0, stream, r);
    return nh_launch_status("bwd64r_reduce");
}

// mlp64r.hip -- the fused, stash-free backward of 64-wide nets (config/fern.yml, config/llff.yml: 4 x 64): persistent workgroups
// that keep the WHOLE net in LDS and, per 128 sample points, recompute the forward (nerf/models.py:233-256), run the data-gradient
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
// (nh_r64.h: one row-major copy serves both orientations).
//
// The weight gradient dW[out][in] = sum_samples dP[out][s] H[in][s] contracts over SAMPLES: both MFMA operands must have the unit on
// the lane's low bits and the sample on k -- the transpose of what the chain holds.  Every wave therefore writes its tile's
// operand blocks (16 units x 16 samples, 1 KiB) into the exchange area, sample-major with an XOR swizzle (conflict-free 16-byte
// writes, conflict-free 4-byte reads in operand order), and after a barrier every wave accumulates ITS share of the layer's
// 16 x 16 gradient tiles over all 8 tiles of the round (k = 128 samples): the 88 KB of gradient accumulators are spread over the
// 8 waves' registers (13 tiles = 52 registers each for a 4-layer net) instead of needing 344 registers in one wave.
//
// One round (128 sample points; L layers; two barriers per step):
//   copy layer1's weights into the exchange area | encode | layer1 .. layers_dir (forward, registers only)
//   step a: POUT PDIR DIRH D      units: fc_rgb, layers_dir's direction columns        then dFEAT
//   step b: FEAT                  units: layers_dir's hidden columns                   then dH_{L-1}
//   step c: PFEAT H_{L-1}         units: fc_feat, fc_alpha                             then dH_{L-2}
//   step d_k: P_{i+1} H_i         units: layers_xyz[i], i = L-2 .. 0                   then dH_{i-1}
//   step e: P_0 X                 units: layer1
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
    unsigned long long* clk;
};

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

#ifdef R64_PAD_LAYOUT
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
#ifdef R64_GEMMF_FULL
    {
        asm volatile("" : "+v"(fx));
        float4 a[2][T];
#pragma unroll
        for (int t = 0; t < T; ++t) a[0][t] = *(const float4*)(w + 16 * t * STRIDE + FOFF(0, fx));
#pragma unroll
        for (int R = 0; R < KR / 4; ++R) {
            if (R + 1 < KR / 4) {
#pragma unroll
                for (int t = 0; t < T; ++t) a[(R + 1) & 1][t] = *(const float4*)(w + 16 * t * STRIDE + FOFF(R + 1, fx));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t] = nh_mfma16(pick4(a[R & 1][t], c), in[4 * R + c], acc[t]);
            }
        }
        return;
    }
#endif
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
#ifdef R64_PAD_LAYOUT
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

// One gradient tile over the round: acc (16 out rows x 16 in units) += sum over the 8 tiles' 16 samples of A[row][s] B[unit][s];
// pa / pb = the blocks' slots in tile 0 of the exchange area; this lane's element of k-step q inside a block is 64 q + rq[q & 1].
// rs += the A operands this lane saw (row sums = bias gradients; lane group g' holds the samples = g' mod 4)
NH_DEVICE void unit1(const float* pa, const float* pb, const int* rq, f32x4& acc, float& rs) {
#pragma unroll
    for (int t = 0; t < NWV; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = pa[t * R64_TILE_F + 64 * q + rq[q & 1]], b = pb[t * R64_TILE_F + 64 * q + rq[q & 1]];
            acc = nh_mfma16(a, b, acc);
            rs += a;
        }
    }
}
// two tiles that share their A block (B blocks pb and pb + 256)
NH_DEVICE void unit2(const float* pa, const float* pb, const int* rq, f32x4& acc0, f32x4& acc1, float& rs) {
#pragma unroll
    for (int t = 0; t < NWV; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = t * R64_TILE_F + 64 * q + rq[q & 1];
            const float a = pa[e], b0 = pb[e], b1 = pb[e + 256];
            acc0 = nh_mfma16(a, b0, acc0);
            acc1 = nh_mfma16(a, b1, acc1);
            rs += a;
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

template <int L>
NH_KERNEL void NH_LB(64 * NWV, 2) k_bwd64r(Bwd64rArgs a) {
    constexpr R64Layout Y = r64_layout(L);
    constexpr int NU = r64_units(L), NB = r64_bias_regs(L);
    NH_DYN_LDS(lds_raw);
    float* const lds = (float*)lds_raw;
    float* const ex = lds + Y.res_floats;
    float* const lfreq = lds + r64_lds_floats(L);  // 16 xyz + 16 direction frequency bands
    nh_clk_begin(a.clk, (unsigned long long*)(lds_raw + (r64_lds_floats(L) + 32) * 4));
    if (threadIdx.x < 32) lfreq[threadIdx.x] = threadIdx.x < 16 ? a.fx[threadIdx.x & 15] : a.fd[threadIdx.x & 15];
    nh_block_sync();
    const int lane = nh_lane(), g = lane >> 4, j = lane & 15, wave = nh_wave_in_block();
    const unsigned lds_addr = nh_lds_addr(lds);
    const NhDmaSrc dma = nh_dma_src(a.image, a.image_bytes);

    // the resident segment: 1-KiB pieces, wave w takes pieces w, w + 8, ...
    for (int q = wave; q < Y.res_floats / 256; q += NWV) nh_dma16a(dma, lane * 16, q * 1024, lds_addr + (unsigned)q * 1024u);

    const int n_slots = a.cidx ? nh_uload_i32(a.cstats, NH_CSTAT_ACTIVE) : (int)a.M;
    const int rounds = (n_slots + 16 * NWV - 1) / (16 * NWV);

    // lane offsets into the image (floats; nh_r64.h): forward rows / transposed rows of the 64-column matrices and of layers_dir
#ifdef R64_PAD_LAYOUT
    const int lf = j * R64_S, lt = 4 * g * R64_S + j, lfd = j * R64_SD, ltd = 4 * g * R64_SD + j;
    const int fx = 4 * g, g16 = 0, i4 = 0;
#else
    const int lf = j * R64_S, lt = 4 * g * R64_S + (j & 3), lfd = j * R64_SD, ltd = 4 * g * R64_SD + (j & 3);
    const int fx = 16 * (j >> 2) + 4 * (g ^ (j & 3)), g16 = 16 * g, i4 = 4 * (j >> 2);
#endif
    // exchange area: where this lane writes a block of its wave's tile, and its element of k-step q when it reads one (sample
    // s = 4 q + g: float s * 16 + 4 ((j >> 2) ^ ((s >> 1) & 3)) + (j & 3) = 64 q + rq[q & 1])
    float* const pw = ex + wave * R64_TILE_F + j * 16 + 4 * (g ^ ((j >> 1) & 3));
    int rq[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) rq[q] = 16 * g + 4 * ((j >> 2) ^ ((2 * q + (g >> 1)) & 3)) + (j & 3);
    f32x4 U[NU];
    float rsum[NB];
#pragma unroll
    for (int u = 0; u < NU; ++u) U[u][0] = U[u][1] = U[u][2] = U[u][3] = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b) rsum[b] = 0.0f;

    // The sample of this lane's slot in round `rd` (a list entry: fetched one round ahead, during step e of the round before).
    // (a list is padded with sample 0 up to a multiple of 128; a dense tail computes the last sample: finite values times a zero
    // cotangent)
    int m_next = 0;
    auto fetch_sample = [&](int rd) {
        const int slot = rd * (16 * NWV) + wave * 16 + j;
        m_next = a.cidx ? a.cidx[slot] : (slot < n_slots ? slot : (int)a.M - 1);
    };
    if ((int)blockIdx.x < rounds) fetch_sample((int)blockIdx.x);
    for (int round = (int)blockIdx.x; round < rounds; round += (int)gridDim.x) {
        // ---- layer1's weights travel into the exchange area (everybody is done reading it: the barrier that closed step e) while
        // the encodings are computed
        for (int q = wave; q < r64_up(64 * R64_S, 256) / 256; q += NWV)
            nh_dma16a(dma, lane * 16, (Y.l1 + q * 256) * 4, lds_addr + (unsigned)(Y.res_floats + q * 256) * 4u);
        const bool valid = round * (16 * NWV) + wave * 16 + j < n_slots;
        const int m = m_next;
        const float* const rr = a.rays + (size_t)(m / a.S) * a.ray_stride;
        float X[NH16_KRX];
        {
            const float zz = a.z[m];
            // pts = ro + rd * z   (nerf/train_utils.py:67,107)
            const float px = rr[0] + rr[3] * zz, py = rr[1] + rr[4] * zz, pz = rr[2] + rr[5] * zz;
            encode_slots<NH16_KRX>(X, px, py, pz, opaque(g), lfreq, a.Lx);
        }
        nh_wait_vmem();
        nh_block_sync();

        // ---- forward, registers only (nerf/models.py:233-256); H[0] = layer1(x) has no activation (:238)
        f32x4 acc[4];
        float H[L][16];
        bias_init<4>(acc, lds + Y.b_l1 + 4 * g);
        gemm_f<NH16_KRX, 4, R64_S>(ex + lf, fx, X, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) H[0][r] = acc[r >> 2][r & 3];
        // d(loss)/d(raw output) of this lane's sample: asked for here, needed when the backward starts (behind the other layers)
        float go0 = 0.f, go1 = 0.f, go2 = 0.f, go3 = 0.f;
        if (valid) {
            const float4 t4 = *(const float4*)(a.g_out + (size_t)m * 4);
            go0 = t4.x, go1 = t4.y, go2 = t4.z, go3 = t4.w;
        }
#pragma unroll
        for (int i = 0; i < L - 1; ++i) {
            bias_init<4>(acc, lds + Y.b_xyz[i] + 4 * g);
            gemm_f<16, 4, R64_S>(lds + Y.xyz[i] + lf, fx, H[i], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) H[i + 1][r] = nh_relu(acc[r >> 2][r & 3]);
        }
        float FEAT[16], DIRH[8], Dd[NH16_KRD];
        bias_init<4>(acc, lds + Y.b_feat + 4 * g);
        gemm_f<16, 4, R64_S>(lds + Y.head + lf, fx, H[L - 1], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) FEAT[r] = nh_relu(acc[r >> 2][r & 3]);
        encode_slots<NH16_KRD>(Dd, rr[8], rr[9], rr[10], opaque(g), lfreq + 16, a.Ld);
        bias_init<2>(acc, lds + Y.b_dir + 4 * g);
        gemm_f<16, 2, R64_SD>(lds + Y.dir + lfd, fx, FEAT, acc);
        gemm_f<NH16_KRD, 2, R64_SD>(lds + Y.dir + 64 + lfd, fx, Dd, acc);
#pragma unroll
        for (int r = 0; r < 8; ++r) DIRH[r] = nh_relu(acc[r >> 2][r & 3]);

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
        nh_block_sync();  // every wave is done with layer1's weights: the exchange area may be written
        // step a: slot 0 POUT (rows 0..2 d(rgb raw), row 3 d(sigma raw)), 1..2 PDIR, 3..4 DIRH, 5..6 D
        ex_put(pw, 0, g == 0 ? go0 : 0.f, g == 0 ? go1 : 0.f, g == 0 ? go2 : 0.f, g == 0 ? go3 : 0.f);
        ex_put_blocks<2>(pw, 1, PDIR);
        ex_put_blocks<2>(pw, 3, DIRH);
        ex_put_blocks<2>(pw, 5, Dd);
        nh_block_sync();
        float PFEAT[16];
        {
            // fc_rgb: waves 0, 1 (POUT x DIRH block w); layers_dir direction columns: waves 2..5 (PDIR block x D block)
            if (wave < 2)
                unit1(ex, ex + (3 + wave) * 256, rq, U[0], rsum[0]);
            else if (wave < 6)
                unit1(ex + (1 + ((wave - 2) >> 1)) * 256, ex + (5 + ((wave - 2) & 1)) * 256, rq, U[0], rsum[0]);
            // d(FEAT pre-activation) = relu'(FEAT) * layers_dir[:, :64]^T PDIR
            zero_acc<4>(acc);
            gemm_t<8, 4, R64_SD>(lds + Y.dir + ltd, g16, i4, PDIR, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) PFEAT[r] = gate_pos(acc[r >> 2][r & 3], FEAT[r]);
        }
        nh_block_sync();
        // step b: slots 3..6 FEAT (PDIR stays in 1..2)
        ex_put_blocks<4>(pw, 3, FEAT);
        nh_block_sync();
        float P[2][16];
        {
            float unused = 0.0f;  // (layers_dir's bias was summed in step a)
            unit1(ex + (1 + (wave >> 2)) * 256, ex + (3 + (wave & 3)) * 256, rq, U[1], unused);
            // dH_{L-1} = fc_feat^T PFEAT + fc_alpha^T d(sigma raw) (one more k-step: group 0 carries d(sigma raw))
            zero_acc<4>(acc);
            gemm_t<16, 4, R64_S>(lds + Y.head + lt, g16, i4, PFEAT, acc);
            const float b = g == 0 ? go3 : 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = nh_mfma16(lds[Y.head + 64 * R64_S + 16 * t + j], b, acc[t]);
#pragma unroll
            for (int r = 0; r < 16; ++r) P[0][r] = L > 1 ? gate_pos(acc[r >> 2][r & 3], H[L - 1][r]) : acc[r >> 2][r & 3];
        }
        nh_block_sync();
        // step c: slots 1..4 PFEAT, 5..8 H_{L-1} (POUT stays in 0)
        ex_put_blocks<4>(pw, 1, PFEAT);
        ex_put_blocks<4>(pw, 5, H[L - 1]);
        nh_block_sync();
        {
            // fc_feat: wave w takes out rows 16 (w >> 1), in units 32 (w & 1) .. + 31; fc_alpha: waves 0..3 (one per SIMD), in units 16 w ..
            unit2(ex + (1 + (wave >> 1)) * 256, ex + (5 + 2 * (wave & 1)) * 256, rq, U[2], U[3], rsum[1]);
            if (wave < 4) {
                float unused = 0.0f;
                unit1(ex, ex + (5 + wave) * 256, rq, U[4], unused);
            }
            if (L > 1) {
                zero_acc<4>(acc);
                gemm_t<16, 4, R64_S>(lds + Y.xyz[L > 1 ? L - 2 : 0] + lt, g16, i4, P[0], acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) P[1][r] = L > 2 ? gate_pos(acc[r >> 2][r & 3], H[L > 2 ? L - 2 : 0][r]) : acc[r >> 2][r & 3];
            }
        }
        nh_block_sync();
        // steps d_k: layers_xyz[i], i = L-2-k: slots 1..4 P_{i+1}, 5..8 H_i
#pragma unroll
        for (int k = 0; k < L - 1; ++k) {
            const int i = L - 2 - k;
            ex_put_blocks<4>(pw, 1, P[k & 1]);
            ex_put_blocks<4>(pw, 5, H[i]);
            nh_block_sync();
            unit2(ex + (1 + (wave >> 1)) * 256, ex + (5 + 2 * (wave & 1)) * 256, rq, U[5 + 2 * k], U[6 + 2 * k], rsum[2 + k]);
            if (i >= 1) {  // dH_{i-1} = layers_xyz[i-1]^T P_i (P_i = the register set this step did not write)
                zero_acc<4>(acc);
                gemm_t<16, 4, R64_S>(lds + Y.xyz[i >= 1 ? i - 1 : 0] + lt, g16, i4, P[(k + 1) & 1], acc);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    P[k & 1][r] = i >= 2 ? gate_pos(acc[r >> 2][r & 3], H[i >= 2 ? i - 1 : 0][r]) : acc[r >> 2][r & 3];
            }
            nh_block_sync();
        }
        // step e: layer1: slots 1..4 P_0, 5..8 X
        ex_put_blocks<4>(pw, 1, P[(L - 1) & 1]);
        ex_put_blocks<4>(pw, 5, X);
        nh_block_sync();
        const bool more = round + (int)gridDim.x < rounds;
        if (more) fetch_sample(round + (int)gridDim.x);
        unit2(ex + (1 + (wave >> 1)) * 256, ex + (5 + 2 * (wave & 1)) * 256, rq, U[5 + 2 * (L - 1)], U[6 + 2 * (L - 1)], rsum[L + 1]);
        nh_block_sync();
    }
    nh_wait_vmem();
    // ---- this workgroup's partial: [wave][register][lane]
    float* const part = a.partial + (size_t)blockIdx.x * r64_partial_floats(L) + (size_t)wave * r64_regs(L) * 64 + lane;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) part[(4 * u + c) * 64] = U[u][c];
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) part[(4 * NU + b) * 64] = rsum[b];
    nh_clk_end((const unsigned long long*)(lds_raw + (r64_lds_floats(L) + 32) * 4));
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

// One block per (wave, register) row of the partials: 64 elements x 4 slices of the workgroup range; slice s adds the partials of
// workgroups s, s + 4, ... in four interleaved running sums, the slices are combined through LDS in slice order -- a fixed order:
// bit-reproducible, no atomics.  Each element then decodes which parameter it is (none: padding, a wave that idles in that step).
NH_KERNEL void k_bwd64r_reduce(Red64rArgs a) {
    NH_SHARED float part[4][64];
    const int L = a.L, NU = r64_units(L), NR = r64_regs(L), H = a.H, H2 = H / 2, stride = R64_WAVES * NR * 64;
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
        if (u == 0) {
            if (w < 2) {  // fc_rgb
                const int col = 16 * w + fl;
                if (row16 < 3 && col < H2) dst = a.o_rgb_w + (int64_t)row16 * H2 + col;
            } else if (w < 6) {  // layers_dir, direction columns: D block (w-2)&1 = slot registers 4 b .. 4 b + 3 of group fl >> 2
                const int orow = 16 * ((w - 2) >> 1) + row16, c = (int)a.dcol[fl >> 2][4 * ((w - 2) & 1) + (fl & 3)];
                if (orow < H2 && c >= 0) dst = a.o_dir_w + (int64_t)orow * (H + a.Dd) + H + c;
            }
        } else if (u == 1) {  // layers_dir, hidden columns
            const int orow = 16 * (w >> 2) + row16, col = 16 * (w & 3) + fl;
            if (orow < H2 && col < H) dst = a.o_dir_w + (int64_t)orow * (H + a.Dd) + col;
        } else if (u == 2 || u == 3) {  // fc_feat
            const int orow = 16 * (w >> 1) + row16, col = 16 * (2 * (w & 1) + (u - 2)) + fl;
            if (orow < H && col < H) dst = a.o_feat_w + (int64_t)orow * H + col;
        } else if (u == 4) {  // fc_alpha: row 3 of POUT (waves 0..3: in units 16 w ..)
            const int col = 16 * w + fl;
            if (w < 4 && row16 == 3 && col < H) dst = a.o_alpha_w + col;
        } else if (u < 5 + 2 * (L - 1)) {  // layers_xyz[i]
            const int k = (u - 5) >> 1, i = L - 2 - k, orow = 16 * (w >> 1) + row16, col = 16 * (2 * (w & 1) + ((u - 5) & 1)) + fl;
            if (orow < H && col < H) dst = a.o_xyz_w[i] + (int64_t)orow * H + col;
        } else {  // layer1: X block b = slot registers 4 b .. 4 b + 3 of group fl >> 2
            const int b = 2 * (w & 1) + ((u - 5) & 1), orow = 16 * (w >> 1) + row16, c = (int)a.xcol[fl >> 2][4 * b + (fl & 3)];
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
    if (b == 0) {
        if (w == 0)
            dst = fl < 3 ? a.o_rgb_b + fl : (fl == 3 ? a.o_alpha_b : -1);
        else if (w == 2 || w == 4)
            dst = 16 * ((w - 2) >> 1) + fl < H2 ? a.o_dir_b + 16 * ((w - 2) >> 1) + fl : -1;
    } else if ((w & 1) == 0) {
        const int orow = 16 * (w >> 1) + fl;
        if (orow < H) {
            if (b == 1)
                dst = a.o_feat_b + orow;
            else if (b < L + 1)
                dst = a.o_xyz_b[L - 2 - (b - 2)] + orow;
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
    const int rc = lds_limit(k_bwd64r<L>, bytes);
    if (rc) return rc;
    NH_LAUNCH((k_bwd64r<L>), grid, 64 * NWV, bytes, stream, a);
    return nh_launch_status("bwd64r");
}

}  // namespace

// workgroups of a fused backward over M sample points: one per compute unit, at most one per round
static int r64_grid(int64_t M) {
    const int64_t rounds = nh_ceil_div(M, 16 * NWV);
    const int cus = compute_units();
    return (int)(rounds < cus ? (rounds < 1 ? 1 : rounds) : cus);
}

int64_t nh_mlp64r_partial_floats(const nerfhip_plan* p, int64_t M) {
    if (!nh_r64_eligible(p)) return 0;
    return (int64_t)r64_grid(M) * r64_partial_floats(p->L);
}

// The fused backward of an eligible plan (nh_r64_eligible) over the fused render's input: g_params = d(loss)/d(parameters) for
// d(loss)/d(raw output) = g_out.  cx: a compaction list built from g_out, or NULL (every sample).  partial: nh_mlp64r_partial_floats.
int nh_mlp64r_backward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, const float* g_out, int64_t M, float* partial,
                       float* g_params, const NhCompact* cx, nerfhip_stream_t stream) {
    NH_REQUIRE(nh_r64_eligible(p) && p->r64_off >= 0, "bwd64r: the plan has no resident image");
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
    NH_LAUNCH(k_bwd64r_reduce, R64_WAVES * r64_regs(p->L), 256, 0, stream, r);
    return nh_launch_status("bwd64r_reduce");
}

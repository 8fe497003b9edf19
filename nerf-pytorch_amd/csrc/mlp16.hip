// mlp16.hip -- FlexibleNeRFModel (nerf/models.py:185-256) forward and data-gradient chain on v_mfma_f32_16x16x4_f32,
// TWO wavefronts per SIMD.
//
// Why a second shape: scripts/mfma_rate.hip (profiles/r01_mfma_issue_cost.txt) shows that with one wave per SIMD
// nothing overlaps a wave's own MFMAs -- every ds_read / VALU / VMEM instruction adds its issue time to the loop, which
// caps the 32x32x2 kernels of mlp.hip (400 registers per wave, one wave per SIMD) at 80-82 % of the matrix pipe; the
// same loop on 16x16x4 with two waves per SIMD measures 96.7 %.  With 16-sample waves an activation of F features takes
// F/4 registers, so a 256-wide layer needs 64 (in) + 64 (out) registers and two waves fit a SIMD.
//
// Layout (nh_plan.h, "v16"): lane l = (sample j = l & 15, k-group g = l >> 4); register r of an activation holds feature
// feat16(r,g) = 16*(r>>2) + 4*g + (r&3) -- the C/D layout of the instruction -- and k-step r of the next layer takes
// register r as its B operand, so activations never leave the register file (as in mlp.hip).  Weights stream L2 -> LDS by LDS-DMA in K-chunks that cover all output tiles ([k-step][quad][lane][4]:
// one ds_read_b128 = the A operands of four 16-row tiles), double buffered, one barrier per chunk.
// The stash / gradient images keep the [32-sample tile][sample][rows] format of mlp.hip (two waves fill one tile), so
// the weight-gradient kernel is shared.
#include <stdlib.h>

#include "nh_mlp.h"

namespace {

// Workgroup shape (measured, profiles/r01_mlp16_ab.txt): 256-wide nets -- ONE 8-wave workgroup per CU (two waves per
// SIMD; the weight stream is paid once per 128 samples: the LDS-DMA traffic of two independent workgroups cost more,
// 8 % of the pipe in scripts/mfma_rate.hip's pipeline mock, than their desynchronised epilogues won); 128-wide nets
// (134 registers per wave, 52 KB of LDS) -- 4-wave workgroups, three per CU.
template <int W>
struct Shape {
    static constexpr int NW = W >= 256 ? 8 : 4;  // waves per workgroup
};
// floats of the largest chunk: 256-wide nets 8192 (8 k-steps x 4 quads; 68 KB of LDS per workgroup -> 2 per CU),
// 128-wide nets 6144 (8 k-steps x 3 quads; 52 KB -> 3 per CU, their 130-register waves fit three to a SIMD)
template <int W>
struct Lds {
    static constexpr int CHUNK_MAX = W >= 256 ? 8192 : 6144;
    static constexpr int BYTES = (2 * CHUNK_MAX + 2 * NH16_BIAS_FLOATS) * 4;
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
    int nw;             // waves per workgroup
    unsigned lds_addr;  // LDS byte address of `lds`
    NhDmaSrc dma;  // descriptor over the whole packed image
    int buf, bbuf;  // chunk / bias buffer of the unit being consumed
    int wave, lane, g;
    NH_MEMBER float* chunk(int b) const { return lds + b * cmax; }
    NH_MEMBER float* bias(int b) const { return lds + 2 * cmax + b * NH16_BIAS_FLOATS; }
    // LDS-DMA of nfloats (a multiple of 256) from float offset `off` of the packed image; piece q by wave q % NW
    NH_MEMBER void copy(int64_t off, int nfloats, float* dst) const {
        const int np = nfloats >> 8;
        const int soff = (int)off * 4;
        const unsigned d = lds_addr + (unsigned)((dst - lds) * 4);
        for (int q = wave; q < np; q += nw) nh_dma16a(dma, lane * 16, soff + q * 1024, d + q * 1024);
    }
    // a layer's first unit: its bias block and chunk 0
    NH_MEMBER void copy_first(int64_t img_off, int first_floats, int b, int bb) const {
        copy(img_off, NH16_BIAS_FLOATS, bias(bb));
        copy(img_off + NH16_BIAS_FLOATS, first_floats, chunk(b));
    }
};

template <int KR, int T>
struct Geo {
    static constexpr int TQ = (T + 3) / 4;
    static constexpr int KC = TQ <= 1 ? 16 : (TQ <= 4 ? 8 : 4);
    static constexpr int NCH = (KR + KC - 1) / KC;
    static constexpr int FIRST = (KR < KC ? KR : KC) * TQ * 256;  // floats of chunk 0
    static constexpr int CHUNK = KC * TQ * 256;  // floats; the kernels check it against Lds<W>::CHUNK_MAX
};

// One linear layer for the 16 samples of this wavefront: acc[t] (16 rows x 16 samples) = W_t * in + bias_t, t < T.
// Precondition: the layer's first unit has been requested into chunk(buf) / bias(bbuf).  While chunk c is multiplied,
// chunk c+1 -- or the first unit of the next layer (next_first > 0) -- travels to the other buffer.
// `post(c, NCH)` holds the global stores of the PREVIOUS layer's results (stash rows, masks), cut into NCH shares: share
// c is issued right after the barrier that starts chunk c, so the stores are spread over the whole layer and drain
// under the MFMAs instead of sitting in front of a vmcnt(0) (CDNA4's vmcnt counts stores too; the values stored are
// this layer's input registers, still live).
template <int KRA, int KRB, int T, class Post>
NH_DEVICE void gemm16(Ctx& cx, const float* inA, const float* inB, int64_t img_off, int64_t next_off, int next_first,
                      f32x4* acc, Post&& post) {
    constexpr int KR = KRA + KRB;
    using G = Geo<KR, T>;
    constexpr int TQ = G::TQ, KC = G::KC, NCH = G::NCH;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        constexpr int dummy = 0;
        (void)dummy;
        const int kc = (KR - c * KC) < KC ? (KR - c * KC) : KC;
        NH16_PH(4);  // [4] between gemms / chunks: epilogue, stores, loop control
        nh_wait_vmem();
        NH16_PH(2);  // [2] s_waitcnt vmcnt(0)
        nh_block_sync();  // chunk c has landed for every wave; everybody is done with the other buffer
        NH16_PH(3);  // [3] s_barrier
        if (c + 1 < NCH) {
            const int kn = (KR - (c + 1) * KC) < KC ? (KR - (c + 1) * KC) : KC;
            cx.copy(img_off + NH16_BIAS_FLOATS + (int64_t)(c + 1) * KC * TQ * 256, kn * TQ * 256, cx.chunk(cx.buf ^ 1));
        } else if (next_first > 0) {
            cx.copy_first(next_off, next_first, cx.buf ^ 1, cx.bbuf ^ 1);
        }
        post(c, NCH);
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
        NH16_PH(0);  // [0] copy issue + previous layer's stores + bias
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
                nh_sched_fence();  // the operand reads of the next k-step are issued before this k-step's MFMAs
                const int r = c * KC + ks;
                const float b = r < KRA ? inA[r < KRA ? r : 0] : inB[r >= KRA ? r - KRA : 0];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float4& w = a[ks & 1][t >> 2];
                    const float av = (t & 3) == 0 ? w.x : ((t & 3) == 1 ? w.y : ((t & 3) == 2 ? w.z : w.w));
                    acc[t] = nh_mfma16(av, b, acc[t]);
                }
            }
        }
        NH16_PH(1);  // [1] operand reads + MFMAs
        cx.buf ^= 1;
    }
    cx.bbuf ^= 1;
}

// epilogue: register r = 4t + c of the activation <- acc[t][c], gated by the stored ReLU mask (data-gradient) and/or
// ReLU'd (forward); bit r of `bits_out` collects [v > 0]
template <int T>
NH_DEVICE void finish(const f32x4* acc, float* act, bool relu, unsigned* bits_out, bool want_bits, const unsigned* mbits,
                      bool masked) {
#pragma unroll
    for (int r = 0; r < 4 * T; ++r) {
        float v = acc[r >> 2][r & 3];
        if (masked) v = nh_gate(v, mbits[r >> 5], r & 31);
        if (relu) v = nh_relu(v);
        if (want_bits) bits_out[r >> 5] |= nh_pos_bit(v) << (r & 31);  // (v >= 0 here: masks are only taken after a ReLU)
        act[r] = v;
    }
}

// rows feat16(4t.., g) = 16t + 4g .. +3 of this lane's sample: one 16-byte store per tile; share c of nch (all: 0 of 1)
template <int T>
NH_DEVICE void store_rows(float* __restrict__ row, const float* act, int g, int c = 0, int nch = 1) {
    if (!row) return;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (t * nch / T != c) continue;
        float4 x;
        x.x = act[4 * t + 0];
        x.y = act[4 * t + 1];
        x.z = act[4 * t + 2];
        x.w = act[4 * t + 3];
        *(float4*)(row + 16 * t + 4 * g) = x;
    }
}
// encoding slots: register r of lane (j,g) is row g*KR + r
template <int KR>
NH_DEVICE void store_slots(float* __restrict__ row, const float* e, int g, int c = 0) {
    if (!row || c != 0) return;
#pragma unroll
    for (int q = 0; q < KR / 4; ++q) {
        float4 x;
        x.x = e[4 * q + 0];
        x.y = e[4 * q + 1];
        x.z = e[4 * q + 2];
        x.w = e[4 * q + 3];
        *(float4*)(row + g * KR + 4 * q) = x;
    }
}

NH_DEVICE float* region_row(float* base, const NhRegion& R, int64_t nt, int64_t tile, int js) {
    return base + (size_t)32 * (size_t)nt * (size_t)R.row_prefix + ((size_t)tile * 32 + (size_t)js) * (size_t)R.rows;
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
    short xcol[4][NH16_KRX];
    short dcol[4][NH16_KRD];
    float fx[16], fd[16];
    int Lx, Ld;
    float* out;
    float* stash;
    NhStashLayout sl;
};

template <int W, bool VIEW>
NH_KERNEL void NH_LB(64 * Shape<W>::NW, 2) k_mlp_fwd16(Fwd16Args a) {
    constexpr int KH = W / 4, TW = W / 16, KX = NH16_KRX, KD = NH16_KRD, NW = Shape<W>::NW;
    static_assert(Geo<KH, TW + 1>::CHUNK <= Lds<W>::CHUNK_MAX && Geo<KH, TW>::CHUNK <= Lds<W>::CHUNK_MAX &&
                      Geo<KH, TW / 2>::CHUNK <= Lds<W>::CHUNK_MAX && Geo<KH, 1>::CHUNK <= Lds<W>::CHUNK_MAX,
                  "chunk too large for the LDS buffer");
    NH_DYN_LDS(lds_raw);
    Ctx cx;
    cx.lds = (float*)lds_raw;
    cx.cmax = Lds<W>::CHUNK_MAX;
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
    const bool valid = m < a.M;
    const int64_t mc = valid ? m : a.M - 1;
    const NhPackedOffsets& po = a.off;

    // the first weights travel to LDS while the encodings are computed
    cx.copy_first(po.f_layer1, Geo<KX, TW>::FIRST, 0, 0);

    float ex[KX];
    float ed[KD];
    if (a.mode == 0) {
        const float* xr = a.x + (size_t)mc * (size_t)(a.dx + a.dd);
#pragma unroll
        for (int r = 0; r < KX; ++r) {
            const int c = a.xcol[g][r];
            ex[r] = c >= 0 ? xr[c] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < KD; ++r) {
            const int c = VIEW ? (int)a.dcol[g][r] : -1;
            ed[r] = c >= 0 ? xr[a.dx + c] : 0.0f;
        }
    } else {
        const int64_t ray = mc / a.S;
        const float* rr = a.rays + (size_t)ray * a.ray_stride;
        const float zz = a.z[mc];
        // pts = ro + rd * z   (nerf/train_utils.py:67,107)
        const float px = rr[0] + rr[3] * zz, py = rr[1] + rr[4] * zz, pz = rr[2] + rr[5] * zz;
        encode_slots16<KX>(ex, px, py, pz, g, a.fx, a.Lx);
        if (VIEW) {
            encode_slots16<KD>(ed, rr[8], rr[9], rr[10], g, a.fd, a.Ld);
        } else {
#pragma unroll
            for (int r = 0; r < KD; ++r) ed[r] = 0.0f;
        }
    }
    const bool tr = a.stash != nullptr;
    auto srow = [&](const NhRegion& R) -> float* { return tr ? region_row(a.stash, R, a.nt, tile, js) : nullptr; };

    unsigned bits[2] = {0u, 0u};
    // ReLU masks for the data-gradient kernel: 64 bits per lane per layer, [16-sample wave tile][mask][lane][2 words]
    auto put_mask = [&](int idx) {
        if (!tr || idx < 0) return;
        unsigned* p = (unsigned*)(a.stash + (size_t)32 * (size_t)a.nt * (size_t)a.sl.total_rows) +
                      ((size_t)(((int64_t)blockIdx.x * NW + wave) * a.sl.n_masks + idx) * 64 + lane) * 2;
        p[0] = bits[0];
        p[1] = bits[1];
    };

    float act[KH];
    f32x4 acc[TW + 1];
    {
        const bool more = a.L > 1;
        // layers_xyz[0] is never a skip layer (i > 0 is required); no activation after layer1 (models.py:238)
        gemm16<KX, 0, TW>(cx, ex, nullptr, po.f_layer1, more ? po.f_xyz[0] : po.f_head,
                          more ? Geo<KH, TW>::FIRST : (VIEW ? Geo<KH, TW + 1>::FIRST : Geo<KH, 1>::FIRST), acc, [&](int c, int) {
                              store_slots<KX>(srow(a.sl.X), ex, g, c);
                              if (VIEW) store_slots<KD>(srow(a.sl.D), ed, g, c);
                          });
        finish<TW>(acc, act, false, bits, false, bits, false);
    }
    // every gemm stores its own input rows (= the previous layer's output) and that layer's ReLU mask
    for (int i = 0; i < a.L - 1; ++i) {
        const bool sk = (i % a.skip == 0) && i > 0;
        const bool more = i + 1 < a.L - 1;
        const bool nsk = more && ((i + 1) % a.skip == 0);
        const int64_t nxt = more ? po.f_xyz[i + 1] : po.f_head;
        const int nfirst = more ? (nsk ? Geo<KH + KX, TW>::FIRST : Geo<KH, TW>::FIRST)
                                : (VIEW ? Geo<KH, TW + 1>::FIRST : Geo<KH, 1>::FIRST);
        auto post = [&](int c, int nch) {
            if (c == 0) put_mask(i - 1);  // H_i (none for H_0)
            store_rows<TW>(srow(a.sl.H[i]), act, g, c, nch);
        };
        if (sk)
            gemm16<KH, KX, TW>(cx, act, ex, po.f_xyz[i], nxt, nfirst, acc, post);
        else
            gemm16<KH, 0, TW>(cx, act, nullptr, po.f_xyz[i], nxt, nfirst, acc, post);
        bits[0] = bits[1] = 0u;
        finish<TW>(acc, act, true, bits, tr, bits, false);
    }
    auto post_last_hidden = [&](int c, int nch) {
        if (c == 0) put_mask(a.L - 2);  // H_{L-1}
        store_rows<TW>(srow(a.sl.H[a.L - 1]), act, g, c, nch);
    };
    if (VIEW) {
        // tiles 0..TW-1: feat = relu(fc_feat(h)); tile TW row 0: fc_alpha(h), raw (models.py:248-249)
        gemm16<KH, 0, TW + 1>(cx, act, nullptr, po.f_head, po.f_dir, Geo<KH + KD, TW / 2>::FIRST, acc, post_last_hidden);
        const float alpha = acc[TW][0];
        bits[0] = bits[1] = 0u;
        finish<TW>(acc, act, true, bits, tr, bits, false);
        float dh[KH / 2];
        gemm16<KH, KD, TW / 2>(cx, act, ed, po.f_dir, po.f_rgb, Geo<KH / 2, 1>::FIRST, acc, [&](int c, int nch) {
            if (c == 0) put_mask(a.L - 1);
            store_rows<TW>(srow(a.sl.FEAT), act, g, c, nch);
        });
        bits[0] = bits[1] = 0u;
        finish<TW / 2>(acc, dh, true, bits, tr, bits, false);
        gemm16<KH / 2, 0, 1>(cx, dh, nullptr, po.f_rgb, 0, 0, acc, [&](int c, int nch) {
            if (c == 0) put_mask(a.L);
            store_rows<TW / 2>(srow(a.sl.DIRH), dh, g, c, nch);
        });
        if (valid && g == 0) {
            float4 r4;
            r4.x = acc[0][0];
            r4.y = acc[0][1];
            r4.z = acc[0][2];
            r4.w = alpha;
            *(float4*)(a.out + (size_t)m * 4) = r4;
        }
    } else {
        gemm16<KH, 0, 1>(cx, act, nullptr, po.f_head, 0, 0, acc, post_last_hidden);
        if (valid && g == 0) {
            float4 r4;
            r4.x = acc[0][0];
            r4.y = acc[0][1];
            r4.z = acc[0][2];
            r4.w = acc[0][3];
            *(float4*)(a.out + (size_t)m * 4) = r4;
        }
    }
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
};

template <int W, bool VIEW>
NH_KERNEL void NH_LB(64 * Shape<W>::NW, 2) k_mlp_dgrad16(Dgrad16Args a) {
    constexpr int KH = W / 4, TW = W / 16, NW = Shape<W>::NW;
    NH_DYN_LDS(lds_raw);
    Ctx cx;
    cx.lds = (float*)lds_raw;
    cx.cmax = Lds<W>::CHUNK_MAX;
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
    const int64_t m = tile * 32 + js;
    const bool valid = m < a.M;
    const NhPackedOffsets& po = a.off;
    const int L = a.L;
    // the transposed images carry no bias: their 512-float bias block is all zero weights (index -1 -> 0.0f)
    if (VIEW)
        cx.copy_first(po.b_rgb, Geo<1, TW / 2>::FIRST, 0, 0);
    else
        cx.copy_first(po.b_head, Geo<1, TW>::FIRST, 0, 0);

    float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) go = *(const float4*)(a.g_out + (size_t)m * 4);
    auto grow = [&](const NhRegion& R) -> float* { return region_row(a.grad, R, a.nt, tile, js); };
    auto store_pout = [&](int c, int) {
        if (c != 0) return;
        // POUT (32 rows): rows 0..2 d(rgb raw), row 3 d(sigma raw), rows 4..31 zero; group g writes rows 8g..8g+7
        float* po_row = grow(a.gl.POUT) + 8 * g;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)(po_row + 0) = g == 0 ? go : z4;
        *(float4*)(po_row + 4) = z4;
    };
    unsigned mb[2];
    auto get_mask = [&](int idx) {
        const unsigned* p = (const unsigned*)(a.stash + (size_t)32 * (size_t)a.nt * (size_t)a.sl.total_rows) +
                            ((size_t)(((int64_t)blockIdx.x * NW + wave) * a.sl.n_masks + idx) * 64 + lane) * 2;
        mb[0] = p[0];
        mb[1] = p[1];
    };
    mb[0] = mb[1] = 0u;
    f32x4 acc[TW];
    float dp[KH];  // d(pre-activation) of the layer just finished = B operand of the next transposed GEMM
    unsigned nobits[2] = {0u, 0u};
    // every gemm stores its own input rows (= the d(pre-activation) the previous one produced) after its first barrier
    if (VIEW) {
        // one k-step: group g carries d(rgb raw)[g]
        float d1[1];
        d1[0] = g == 0 ? go.x : (g == 1 ? go.y : (g == 2 ? go.z : 0.0f));
        get_mask(L);  // DIRH
        float dpd[KH / 2];
        gemm16<1, 0, TW / 2>(cx, d1, nullptr, po.b_rgb, po.b_dir, Geo<KH / 2, TW>::FIRST, acc, store_pout);
        finish<TW / 2>(acc, dpd, false, nobits, false, mb, true);
        get_mask(L - 1);  // FEAT
        gemm16<KH / 2, 0, TW>(cx, dpd, nullptr, po.b_dir, po.b_head, Geo<KH + 1, TW>::FIRST, acc,
                              [&](int c, int nch) { store_rows<TW / 2>(grow(a.gl.PDIR), dpd, g, c, nch); });
        finish<TW>(acc, dp, false, nobits, false, mb, true);
        if (L > 1) get_mask(L - 2);  // H_{L-1}
        float da[1];
        da[0] = g == 0 ? go.w : 0.0f;  // d(sigma raw) enters through fc_alpha's row (k-step KH, group 0)
        gemm16<KH, 1, TW>(cx, dp, da, po.b_head, L > 1 ? po.b_xyz[L - 2] : 0, L > 1 ? Geo<KH, TW>::FIRST : 0, acc,
                          [&](int c, int nch) { store_rows<TW>(grow(a.gl.PFEAT), dp, g, c, nch); });
        finish<TW>(acc, dp, false, nobits, false, mb, L > 1);
    } else {
        float d1[1];
        d1[0] = g == 0 ? go.x : (g == 1 ? go.y : (g == 2 ? go.z : go.w));
        if (L > 1) get_mask(L - 2);  // H_{L-1}
        gemm16<1, 0, TW>(cx, d1, nullptr, po.b_head, L > 1 ? po.b_xyz[L - 2] : 0, L > 1 ? Geo<KH, TW>::FIRST : 0, acc,
                         store_pout);
        finish<TW>(acc, dp, false, nobits, false, mb, L > 1);
    }
    // dp = d(pre-activation of H_{L-1}).  Walk down: dpre_{k-1} = relu'(H_{k-1}) * (W_{k-1}^T dpre_k);
    // H_0 = layer1 output has no activation (models.py:238).
    for (int k = L - 1; k >= 1; --k) {
        const bool masked = k - 1 >= 1;
        if (masked) get_mask(k - 2);  // H_{k-1}
        gemm16<KH, 0, TW>(cx, dp, nullptr, po.b_xyz[k - 1], k >= 2 ? po.b_xyz[k - 2] : 0, k >= 2 ? Geo<KH, TW>::FIRST : 0, acc,
                          [&](int c, int nch) { store_rows<TW>(grow(a.gl.P[k]), dp, g, c, nch); });
        finish<TW>(acc, dp, false, nobits, false, mb, masked);
    }
    store_rows<TW>(grow(a.gl.P[0]), dp, g);
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

#ifdef NH_PHASE_TIMING
extern "C" int nerfhip_debug_phases16(unsigned long long* host16, int reset) {
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_phase16), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase16), z, sizeof(z));
    }
    return 0;
}
#endif

int nh_mlp16_forward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                     nerfhip_stream_t stream) {
    Fwd16Args a;
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
        for (int r = 0; r < NH16_KRX; ++r) a.xcol[g][r] = (short)p->xyz_col16[g][r];
        for (int r = 0; r < NH16_KRD; ++r) a.dcol[g][r] = (short)p->dir_col16[g][r];
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
    const int64_t groups = nh_ceil_div(M, 128);  // whole 128-sample groups: every stash tile is written
    int rc = NERFHIP_OK;
#define NH_FWD16(WW, VV)                                                              \
    {                                                                                 \
        rc = lds_limit(k_mlp_fwd16<WW, VV>, Lds<WW>::BYTES);                          \
        if (rc) return rc;                                                            \
        NH_LAUNCH((k_mlp_fwd16<WW, VV>), groups * (8 / Shape<WW>::NW), 64 * Shape<WW>::NW, Lds<WW>::BYTES, stream, a); \
    }
    if (p->W == 256 && p->view) NH_FWD16(256, true)
    else if (p->W == 256) NH_FWD16(256, false)
    else if (p->view) NH_FWD16(128, true)
    else NH_FWD16(128, false)
#undef NH_FWD16
    return nh_launch_status("mlp_fwd16");
}

int nh_mlp16_dgrad(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash, float* scratch,
                   nerfhip_stream_t stream) {
    Dgrad16Args d;
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
    const int64_t groups = nh_ceil_div(M, 128);
    int rc = NERFHIP_OK;
#define NH_BWD16(WW, VV)                                                              \
    {                                                                                 \
        rc = lds_limit(k_mlp_dgrad16<WW, VV>, Lds<WW>::BYTES);                        \
        if (rc) return rc;                                                            \
        NH_LAUNCH((k_mlp_dgrad16<WW, VV>), groups * (8 / Shape<WW>::NW), 64 * Shape<WW>::NW, Lds<WW>::BYTES, stream, d); \
    }
    if (p->W == 256 && p->view) NH_BWD16(256, true)
    else if (p->W == 256) NH_BWD16(256, false)
    else if (p->view) NH_BWD16(128, true)
    else NH_BWD16(128, false)
#undef NH_BWD16
    return nh_launch_status("mlp_dgrad16");
}

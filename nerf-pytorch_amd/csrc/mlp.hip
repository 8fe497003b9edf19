// mlp.hip -- FlexibleNeRFModel (nerf/models.py:185-256) forward and backward on fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32, a k-ordered fmaf chain).
//
// Forward / data-gradient kernels: a workgroup is 4 wavefronts (one per SIMD, <= 512 registers each); every
// wavefront owns 32 sample points for the whole network.  The computation is transposed, out^T = W * h^T: the
// weights are the MFMA A operand, the activations the B operand.  With that orientation the C/D register layout of
// one layer IS the B-operand layout of the next one, so activations never leave the register file -- no LDS
// round-trip, no HBM materialisation of the (N*S, 90) encodings or (N*S, 256) hidden states the reference creates
// (nerf/train_utils.py:8-25).  Weights stream L2 -> LDS in 32-output-row chunks by LDS-DMA (global_load_lds, issued
// when the previous chunk starts computing; double buffered, one barrier per chunk) and are broadcast to the 4
// wavefronts by ds_read_b128, prefetched two groups ahead of the MFMAs that consume them.
//
// Weight-gradient kernel: a split-K GEMM  dW[out,in] = sum_samples dpre[out][s] * act[in][s]  whose operands are
// the sample-major [tile][32 samples][rows] images the other two kernels write (one coalesced dword load = 32
// consecutive rows of one sample = one MFMA operand); each wavefront keeps up to a 128x128 patch of dW in 256
// accumulator registers and walks a contiguous range of sample tiles; a second kernel reduces the split-K partials
// in a fixed order (bit-reproducible) and scatters them into the reference parameter layout.
#include <stdlib.h>

#include "nh_mlp.h"

namespace {

template <int W>
struct Cfg {
    static constexpr int KH = W / 2;                    // registers of a hidden activation
    static constexpr int KRMAX = KH + NH_KRX;           // widest layer (skip layer)
    // floats of one LDS weight buffer: the widest chunk (KRMAX/4 + 1 pieces of 1 KiB) rounded up to 4 pieces per wave
    static constexpr int LB = ((KRMAX / 4 + 1 + 3) / 4) * 4 * 256;
    static constexpr int N4MAX = (LB / 4 + 255) / 256;  // float4 per thread to stage one chunk (register path)
    static constexpr int LDS_BYTES = 2 * LB * 4;
};

template <int N4MAX>
struct Stage {
    float4 v[N4MAX];   // register-staging path only
    NhDmaSrc dma;      // buffer descriptor over the whole packed-weight image
    const float* base; // its base pointer
    unsigned lds_addr; // LDS byte address of the weight buffers (DMA destinations are addresses, not pointers)
    const float* lds0; // the same as a pointer
#ifdef NH_PHASE_TIMING
    unsigned long long ph[6], last;  // debug build only: cycles per tile phase, accumulated per wave
#endif
};
#ifdef NH_PHASE_TIMING
__device__ unsigned long long g_phase[32];
#define NH_PH(i)                                   \
    do {                                           \
        const unsigned long long _t = clock64();   \
        st.ph[i] += _t - st.last;                  \
        st.last = _t;                              \
    } while (0)
#define NH_PH_INIT()                               \
    do {                                           \
        for (int _i = 0; _i < 6; ++_i) st.ph[_i] = 0; \
        st.last = clock64();                       \
    } while (0)
#define NH_PH_FLUSH(base)                                                                         \
    do {                                                                                          \
        if (lane == 0)                                                                            \
            for (int _i = 0; _i < 6; ++_i) atomicAdd(&g_phase[(base) + _i], st.ph[_i]);           \
    } while (0)
extern "C" int nerfhip_debug_phases(unsigned long long* host32, int reset) {
    (void)hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 32);
    if (reset) {
        unsigned long long z[32] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z));
    }
    return 0;
}
#else
#define NH_PH(i)
#define NH_PH_INIT()
#define NH_PH_FLUSH(base)
#endif

template <int N4MAX>
NH_DEVICE void stage_load(Stage<N4MAX>& s, const float* __restrict__ chunk, int n4) {
    const float4* p = (const float4*)chunk;
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int q = 0; q < N4MAX; ++q) {
        const int idx = tid + q * 256;
        if (idx < n4) s.v[q] = p[idx];
    }
}
template <int N4MAX>
NH_DEVICE void stage_store(const Stage<N4MAX>& s, float* ldsbuf, int n4) {
    float4* p = (float4*)ldsbuf;
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int q = 0; q < N4MAX; ++q) {
        const int idx = tid + q * 256;
        if (idx < n4) p[idx] = s.v[q];
    }
}
// LDS-DMA of one chunk: n4/64 pieces of 1 KiB, piece p by wave (p & 3)
// (every wave issues the same number of pieces; pieces past the end of the image are dropped by the descriptor)
template <int N4MAX>
NH_DEVICE void dma_issue(const Stage<N4MAX>& st, const float* __restrict__ chunk, int n4, float* ldsbuf, int wave, int lane) {
    const int qn = ((n4 >> 6) + 3) >> 2;
    const int soff = (int)(chunk - st.base) * 4;
    const unsigned dst = st.lds_addr + (unsigned)((ldsbuf - st.lds0) * 4);
    for (int q = 0; q < qn; ++q) nh_dma16a(st.dma, lane * 16, soff + (wave + 4 * q) * 1024, dst + (wave + 4 * q) * 1024);
}

NH_DEVICE int n4_of(int kr) { return kr * 16 + 64; }

// rows 32t .. 32t+31 of this lane's sample (its 16 registers res[16t..]) as four 16-byte stores
NH_DEVICE void store_tile_rows(float* __restrict__ st_row, const float* res, int t, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 x;
        x.x = res[16 * t + 4 * g + 0];
        x.y = res[16 * t + 4 * g + 1];
        x.z = res[16 * t + 4 * g + 2];
        x.w = res[16 * t + 4 * g + 3];
        *(float4*)(st_row + 32 * t + 8 * g + 4 * h) = x;
    }
}

// One linear layer for the 32 samples of this wavefront:  tile t (32 rows x 32 samples) = Wchunk_t * in + bias_t.
// Precondition: chunk 0 of this layer is in lds buffer `buf` and a barrier has been passed.  While tile t is being
// computed the next chunk (of this layer, or the first chunk of the next layer) travels to the other LDS buffer; it
// is published by the barrier that ends the tile.
// Epilogue of the first EPI tiles: v = acc, zeroed where mk <= 0 if `masked`, ReLU'd if `relu`; v becomes
// res[16t + c] (the next layer's B operand) and, if st_row != NULL, rows 32t.. of this lane's sample in a
// sample-major image -- four 16-byte stores issued right AFTER the barrier that ends the tile, so that they drain
// under the next tile's MFMAs instead of sitting in front of a vmcnt(0) (CDNA4's vmcnt counts stores too).
// Tiles >= EPI are returned raw in out[t - EPI].
// VMEM work of a tile -- the LDS-DMA pieces of the next chunk and the previous tile's four row stores -- is issued
// BETWEEN the MFMA groups: a VMEM instruction placed in front of an MFMA issues while the previous MFMA is still
// executing, so it costs no matrix-pipe time (issuing it all up front cost 17 % / 29 % of fwd / dgrad wave time --
// profiles/r01_phase_timing.txt).
// ReLU masks travel as bits: bit (r & 31) of word r >> 5 belongs to register r of the lane.  `bits_out` (forward)
// collects [v > 0] of the values this layer produces; `mbits` (data-gradient) gates them.
template <int W, int DMA, int KRA, int KRB, int TILES, int EPI>
NH_DEVICE void gemm_layer(const float* inA, const float* inB, const float* __restrict__ wl,
                          const float* __restrict__ next_chunk, int next_n4, float* lds, int& buf,
                          Stage<Cfg<W>::N4MAX>& st, f32x16* out, int lane, int wave, float* res, bool relu,
                          unsigned* bits_out, bool want_bits, const unsigned* mbits, bool masked, bool do_store,
                          float* __restrict__ st_row) {
    constexpr int KR = KRA + KRB;
    constexpr int CH = KR * 64 + 256;
    constexpr int NG = KR / 4;
    constexpr int QMAX = (Cfg<W>::LB / 256 + 3) / 4;  // most 1-KiB pieces one wave ever issues for a chunk
    static_assert(KR % 4 == 0, "KR must be a multiple of 4");
    static_assert(KR <= Cfg<W>::KRMAX, "KR too large for the LDS buffer");
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const float* nxt = (t + 1 < TILES) ? wl + (size_t)(t + 1) * CH : next_chunk;
        const int n4 = (t + 1 < TILES) ? CH / 4 : next_n4;
        float* other = lds + (buf ^ 1) * Cfg<W>::LB;
        // DMA: 0 = register staging, 1 = LDS-DMA + previous tile's stores issued here (before the MFMAs),
        //      2 = the same VMEM instructions spread between the MFMA groups
        if (nxt && DMA == 0) stage_load(st, nxt, n4);
        if (nxt && DMA == 1) dma_issue(st, nxt, n4, other, wave, lane);
        const bool st_prev = do_store && t >= 1 && t - 1 < EPI;  // rows of the previous tile still to store
        if (st_prev && DMA != 2) store_tile_rows(st_row, res, t - 1, h);
        const int qn = (nxt && DMA == 2) ? (((n4 >> 6) + 3) >> 2) : 0;  // pieces wave + 4q, q < qn (same for every wave)
        const int soff = nxt ? (int)(nxt - st.base) * 4 : 0;
        const bool st_mix = st_prev && DMA == 2;
        NH_PH(0);  // [0] between tiles / layers
        const float* cur = lds + buf * Cfg<W>::LB;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *(const float4*)(cur + KR * 64 + 8 * g + 4 * h);
            acc[4 * g + 0] = b4.x;
            acc[4 * g + 1] = b4.y;
            acc[4 * g + 2] = b4.z;
            acc[4 * g + 3] = b4.w;
        }
        const float4* w4 = (const float4*)cur + lane;
        float4 w0 = w4[0];
        float4 w1 = w4[(NG > 1 ? 1 : 0) * 64];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float4 w2 = w1;
            if (g + 2 < NG) w2 = w4[(g + 2) * 64];
            nh_sched_fence();  // the prefetch above is issued before these MFMAs; its data is used two groups later
            // this group's share of the tile's VMEM instructions
#pragma unroll
            // (256-wide nets: all pieces within the first half of the tile, so that the copy has half a tile to land
            //  before the vmcnt(0) -- measured +1 %; 128-wide nets: spread over the whole tile -- front-loading cost 14 %)
            for (int q = ((W >= 256 ? 2 : 1) * g * QMAX) / NG; q < ((W >= 256 ? 2 : 1) * (g + 1) * QMAX) / NG; ++q)
                if (q < qn && q < QMAX)
                    nh_dma16a(st.dma, lane * 16, soff + (wave + 4 * q) * 1024,
                              st.lds_addr + (unsigned)(((buf ^ 1) * Cfg<W>::LB + (wave + 4 * q) * 256) * 4));
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
                if (st_mix && g == (NG >= 8 ? 1 + k4 * (NG / 4) : 0)) {
                    float4 x;
                    x.x = res[16 * (t - 1) + 4 * k4 + 0];
                    x.y = res[16 * (t - 1) + 4 * k4 + 1];
                    x.z = res[16 * (t - 1) + 4 * k4 + 2];
                    x.w = res[16 * (t - 1) + 4 * k4 + 3];
                    *(float4*)(st_row + 32 * (t - 1) + 8 * k4 + 4 * h) = x;
                }
            const int r = 4 * g;
            acc = nh_mfma32(w0.x, (r + 0 < KRA) ? inA[r + 0] : inB[r + 0 - KRA], acc);
            acc = nh_mfma32(w0.y, (r + 1 < KRA) ? inA[r + 1] : inB[r + 1 - KRA], acc);
            acc = nh_mfma32(w0.z, (r + 2 < KRA) ? inA[r + 2] : inB[r + 2 - KRA], acc);
            acc = nh_mfma32(w0.w, (r + 3 < KRA) ? inA[r + 3] : inB[r + 3 - KRA], acc);
            w0 = w1;
            w1 = w2;
        }
        NH_PH(1);  // [1] bias + operand reads + MFMA issue (+ interleaved VMEM issue)
        if (t < EPI) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int r = 16 * t + c;
                float v = acc[c];
                if (masked) v = nh_gate(v, mbits[r >> 5], r & 31);
                if (relu) v = nh_relu(v);
                if (want_bits) bits_out[r >> 5] |= nh_pos_bit(v) << (r & 31);  // (v >= 0 here: masks are only taken after a ReLU)
                res[r] = v;
            }
        } else {
            out[t - EPI] = acc;
        }
        NH_PH(2);  // [2] epilogue (waits for the last MFMA)
        if (nxt) {
            if (DMA)
                nh_wait_vmem();
            else
                stage_store(st, other, n4);
        }
        NH_PH(3);  // [3] vmcnt(0): DMA + stores + mask loads
        nh_block_sync();
        NH_PH(4);  // [4] barrier
        buf ^= 1;
    }
    // the last epilogue tile: when raw tiles follow (EPI < TILES) it was stored at the start of tile EPI above
    if (do_store && EPI >= 1 && EPI == TILES) store_tile_rows(st_row, res, TILES - 1, h);
}

template <int W, int DMA>
NH_DEVICE void first_chunk(const float* __restrict__ chunk, int n4, float* lds, Stage<Cfg<W>::N4MAX>& st, int wave,
                           int lane) {
    if (DMA) {
        dma_issue(st, chunk, n4, lds, wave, lane);
        nh_wait_vmem();
    } else {
        stage_load(st, chunk, n4);
        stage_store(st, lds, n4);
    }
    nh_block_sync();
}

// ---- sample-major region images: element (tile, sample j, row) at base + ((tile*32 + j)*rows + row) -------------------
NH_DEVICE float* region_tile(float* base, const NhRegion& R, int64_t nt, int64_t tile) {
    return base + (size_t)32 * (size_t)nt * (size_t)R.row_prefix + (size_t)tile * (size_t)R.rows * 32;
}
NH_DEVICE const float* region_tile_c(const float* base, const NhRegion& R, int64_t nt, int64_t tile) {
    return base + (size_t)32 * (size_t)nt * (size_t)R.row_prefix + (size_t)tile * (size_t)R.rows * 32;
}
// hidden activations: registers 4q..4q+3 of lane (j,h) are rows feat(4q,h)..+3 of sample j
template <int N>
NH_DEVICE void store_feat_rows(float* __restrict__ tile_base, int rows, const float* v, int j, int h) {
    float* row = tile_base + (size_t)j * rows;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        float4 x;
        x.x = v[4 * q + 0];
        x.y = v[4 * q + 1];
        x.z = v[4 * q + 2];
        x.w = v[4 * q + 3];
        *(float4*)(row + nh_feat_of(q >> 2, 4 * (q & 3), h)) = x;
    }
}
template <int N>
NH_DEVICE void load_feat_rows(float* v, const float* __restrict__ tile_base, int rows, int j, int h) {
    const float* row = tile_base + (size_t)j * rows;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        const float4 x = *(const float4*)(row + nh_feat_of(q >> 2, 4 * (q & 3), h));
        v[4 * q + 0] = x.x;
        v[4 * q + 1] = x.y;
        v[4 * q + 2] = x.z;
        v[4 * q + 3] = x.w;
    }
}
// encoding slots: register r of lane (j,h) is row h*N + r
template <int N>
NH_DEVICE void store_slot_rows(float* __restrict__ tile_base, int rows, const float* v, int j, int h) {
    float* row = tile_base + (size_t)j * rows + h * N;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        float4 x;
        x.x = v[4 * q + 0];
        x.y = v[4 * q + 1];
        x.z = v[4 * q + 2];
        x.w = v[4 * q + 3];
        *(float4*)(row + 4 * q) = x;
    }
}

struct MlpFwdArgs {
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
    short xcol[2][NH_KRX];
    short dcol[2][NH_KRD];
    float fx[16], fd[16];
    int Lx, Ld, P0x, P0d;
    float* out;
    float* stash;
    NhStashLayout sl;
};

NH_DEVICE float sel3(int a, float x, float y, float z) { return a == 0 ? x : (a == 1 ? y : z); }

// encoding registers of one lane: slot layout documented in nh_plan.h / plan.cpp build_slot_map
template <int KR>
NH_DEVICE void encode_slots(float* e, float x, float y, float z, int h, const float* freqs, int Lf, int P0) {
    e[0] = h ? z : x;
    e[1] = h ? 0.0f : y;
#pragma unroll
    for (int q = 0; 3 + 2 * q < KR; ++q) {
        const int ph = q + P0;
        const int fh = ph / 3, ah = ph - 3 * fh;
        const bool valid = h ? (ph < 3 * Lf) : (q < P0);
        const float f_lo = freqs[q / 3];
        const float f_hi = freqs[fh < 16 ? fh : 15];
        const float c_lo = sel3(q % 3, x, y, z);
        const float c_hi = sel3(ah, x, y, z);
        const float arg = (h ? c_hi : c_lo) * (h ? f_hi : f_lo);
        float s, c;
        nh_sincos(arg, &s, &c);
        e[2 + 2 * q] = valid ? s : 0.0f;
        e[3 + 2 * q] = valid ? c : 0.0f;
    }
}

template <int W, bool VIEW, int DMA>
NH_KERNEL void NH_LB(256, 1) k_mlp_fwd(MlpFwdArgs a) {
    using C = Cfg<W>;
    constexpr int KH = C::KH, TW = W / 32;
    NH_DYN_LDS(lds_raw);
    float* lds = (float*)lds_raw;
    const int lane = nh_lane(), j = lane & 31, h = lane >> 5, wave = nh_wave_in_block();
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t m = tile * 32 + j;
    const bool valid = m < a.M;
    const int64_t mc = valid ? m : a.M - 1;

    Stage<C::N4MAX> st;
    st.base = a.packed;
    st.dma = nh_dma_src(a.packed, a.packed_bytes);
    st.lds0 = lds;
    st.lds_addr = nh_lds_addr(lds);
    NH_PH_INIT();
    // the first weight chunk travels to LDS while the encodings are computed
    if (DMA) dma_issue(st, a.packed + a.off.f_layer1, n4_of(NH_KRX), lds, wave, lane);

    float ex[NH_KRX];
    float ed[NH_KRD];
    if (a.mode == 0) {
        const float* xr = a.x + (size_t)mc * (size_t)(a.dx + a.dd);
#pragma unroll
        for (int r = 0; r < NH_KRX; ++r) {
            const int c = h ? a.xcol[1][r] : a.xcol[0][r];
            ex[r] = c >= 0 ? xr[c] : 0.0f;
        }
        if (VIEW) {
#pragma unroll
            for (int r = 0; r < NH_KRD; ++r) {
                const int c = h ? a.dcol[1][r] : a.dcol[0][r];
                ed[r] = c >= 0 ? xr[a.dx + c] : 0.0f;
            }
        }
    } else {
        const int64_t ray = mc / a.S;
        const float* rr = a.rays + (size_t)ray * a.ray_stride;
        const float zz = a.z[mc];
        // pts = ro + rd * z   (nerf/train_utils.py:67,107)
        const float px = rr[0] + rr[3] * zz, py = rr[1] + rr[4] * zz, pz = rr[2] + rr[5] * zz;
        encode_slots<NH_KRX>(ex, px, py, pz, h, a.fx, a.Lx, a.P0x);
        if (VIEW) encode_slots<NH_KRD>(ed, rr[8], rr[9], rr[10], h, a.fd, a.Ld, a.P0d);
    }
    if (!VIEW) {
#pragma unroll
        for (int r = 0; r < NH_KRD; ++r) ed[r] = 0.0f;
    }
    if (a.stash) {
        store_slot_rows<NH_KRX>(region_tile(a.stash, a.sl.X, a.nt, tile), a.sl.X.rows, ex, j, h);
        if (VIEW) store_slot_rows<NH_KRD>(region_tile(a.stash, a.sl.D, a.nt, tile), a.sl.D.rows, ed, j, h);
    }

    int buf = 0;
    const float* pk = a.packed;
    if (DMA) {
        nh_wait_vmem();
        nh_block_sync();
    } else {
        first_chunk<W, DMA>(pk + a.off.f_layer1, n4_of(NH_KRX), lds, st, wave, lane);
    }

    f32x16 o[1];  // raw tiles only: fc_alpha's tile / the rgb tile / fc_out
    float act[KH];
    float res[KH];
    unsigned bits[4];
    const bool tr = a.stash != nullptr;
    auto strow = [&](const NhRegion& R) -> float* {
        return tr ? region_tile(a.stash, R, a.nt, tile) + (size_t)j * R.rows : nullptr;
    };
    // ReLU masks for the data-gradient kernel: 128 bits per lane per layer, one coalesced 1-KiB record per wave
    auto put_mask = [&](int idx) {
        if (!tr) return;
        unsigned* p = (unsigned*)(a.stash + (size_t)32 * (size_t)a.nt * (size_t)a.sl.total_rows) +
                      ((size_t)(tile * a.sl.n_masks + idx) * 64 + lane) * 4;
        p[0] = bits[0];
        p[1] = bits[1];
        p[2] = bits[2];
        p[3] = bits[3];
    };
    {
        const bool more = a.L > 1;
        const float* nxt = pk + (more ? a.off.f_xyz[0] : a.off.f_head);
        const int nn4 = n4_of(KH);  // layers_xyz[0] is never a skip layer (i > 0 is required)
        // no activation after layer1 (models.py:238)
        gemm_layer<W, DMA, NH_KRX, 0, TW, TW>(ex, nullptr, pk + a.off.f_layer1, nxt, nn4, lds, buf, st, o, lane, wave, res,
                                              false, bits, false, bits, false, tr, strow(a.sl.H[0]));
#pragma unroll
        for (int r = 0; r < KH; ++r) act[r] = res[r];
    }
    for (int i = 0; i < a.L - 1; ++i) {
        const bool sk = (i % a.skip == 0) && i > 0;
        const bool more = i + 1 < a.L - 1;
        const bool nsk = more && ((i + 1) % a.skip == 0);
        const float* nxt = pk + (more ? a.off.f_xyz[i + 1] : a.off.f_head);
        const int nn4 = n4_of(KH + (nsk ? NH_KRX : 0));
        float* sr = strow(a.sl.H[i + 1]);
        bits[0] = bits[1] = bits[2] = bits[3] = 0u;
        if (sk)
            gemm_layer<W, DMA, KH, NH_KRX, TW, TW>(act, ex, pk + a.off.f_xyz[i], nxt, nn4, lds, buf, st, o, lane, wave, res,
                                                   true, bits, tr, bits, false, tr, sr);
        else
            gemm_layer<W, DMA, KH, 0, TW, TW>(act, nullptr, pk + a.off.f_xyz[i], nxt, nn4, lds, buf, st, o, lane, wave, res,
                                              true, bits, tr, bits, false, tr, sr);
        put_mask(i);  // H_{i+1}
#pragma unroll
        for (int r = 0; r < KH; ++r) act[r] = res[r];
    }
    if (VIEW) {
        // tiles 0..TW-1: feat = relu(fc_feat(h)); tile TW row 0: fc_alpha(h), raw (models.py:248-249)
        bits[0] = bits[1] = bits[2] = bits[3] = 0u;
        gemm_layer<W, DMA, KH, 0, TW + 1, TW>(act, nullptr, pk + a.off.f_head, pk + a.off.f_dir, n4_of(KH + NH_KRD), lds, buf,
                                              st, o, lane, wave, res, true, bits, tr, bits, false, tr, strow(a.sl.FEAT));
        put_mask(a.L - 1);
        const float alpha = o[0][0];
        float dh[KH / 2];
        bits[0] = bits[1] = bits[2] = bits[3] = 0u;
        gemm_layer<W, DMA, KH, NH_KRD, TW / 2, TW / 2>(res, ed, pk + a.off.f_dir, pk + a.off.f_rgb, n4_of(KH / 2), lds, buf, st,
                                                       o, lane, wave, dh, true, bits, tr, bits, false, tr, strow(a.sl.DIRH));
        put_mask(a.L);
        gemm_layer<W, DMA, KH / 2, 0, 1, 0>(dh, nullptr, pk + a.off.f_rgb, nullptr, 0, lds, buf, st, o, lane, wave, dh, false,
                                            bits, false, bits, false, false, nullptr);
        if (valid && h == 0) {
            float4 r4;
            r4.x = o[0][0];
            r4.y = o[0][1];
            r4.z = o[0][2];
            r4.w = alpha;
            *(float4*)(a.out + (size_t)m * 4) = r4;
        }
        NH_PH(5);
        NH_PH_FLUSH(0);
    } else {
        gemm_layer<W, DMA, KH, 0, 1, 0>(act, nullptr, pk + a.off.f_head, nullptr, 0, lds, buf, st, o, lane, wave, res, false,
                                        bits, false, bits, false, false, nullptr);
        if (valid && h == 0) {
            float4 r4;
            r4.x = o[0][0];
            r4.y = o[0][1];
            r4.z = o[0][2];
            r4.w = o[0][3];
            *(float4*)(a.out + (size_t)m * 4) = r4;
        }
    }
}

// ---- data-gradient chain ---------------------------------------------------------------------------------------------
struct DgradArgs {
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

template <int W, bool VIEW, int DMA>
NH_KERNEL void NH_LB(256, 1) k_mlp_dgrad(DgradArgs a) {
    using C = Cfg<W>;
    constexpr int KH = C::KH, TW = W / 32;
    NH_DYN_LDS(lds_raw);
    float* lds = (float*)lds_raw;
    const int lane = nh_lane(), j = lane & 31, h = lane >> 5, wave = nh_wave_in_block();
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t m = tile * 32 + j;
    const bool valid = m < a.M;
    float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) go = *(const float4*)(a.g_out + (size_t)m * 4);
    {
        // POUT: rows 0..2 d(rgb raw), row 3 d(sigma raw), rows 4..31 zero
        float* po = region_tile(a.grad, a.gl.POUT, a.nt, tile) + (size_t)j * 32 + h * 16;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)(po + 0) = h == 0 ? go : z4;
        *(float4*)(po + 4) = z4;
        *(float4*)(po + 8) = z4;
        *(float4*)(po + 12) = z4;
    }
    auto grow = [&](const NhRegion& R) -> float* { return region_tile(a.grad, R, a.nt, tile) + (size_t)j * R.rows; };
    Stage<C::N4MAX> st;
    st.base = a.packed;
    st.dma = nh_dma_src(a.packed, a.packed_bytes);
    st.lds0 = lds;
    st.lds_addr = nh_lds_addr(lds);
    NH_PH_INIT();
    int buf = 0;
    const float* pk = a.packed;
    f32x16 o[1];
    float dp[KH];  // d(pre-activation) of the layer just finished = B operand of the next transposed GEMM
    float res[KH];
    unsigned mb[4];  // ReLU mask bits (written by the forward kernel) of the layer being produced
    auto get_mask = [&](int idx) {
        const unsigned* p = (const unsigned*)(a.stash + (size_t)32 * (size_t)a.nt * (size_t)a.sl.total_rows) +
                            ((size_t)(tile * a.sl.n_masks + idx) * 64 + lane) * 4;
        mb[0] = p[0];
        mb[1] = p[1];
        mb[2] = p[2];
        mb[3] = p[3];
    };
    mb[0] = mb[1] = mb[2] = mb[3] = 0u;
    const int L = a.L;
    if (VIEW) {
        float d4[4];
        d4[0] = h == 0 ? go.x : 0.0f;
        d4[1] = h == 0 ? go.y : 0.0f;
        d4[2] = h == 0 ? go.z : 0.0f;
        d4[3] = 0.0f;
        get_mask(L);  // DIRH
        first_chunk<W, DMA>(pk + a.off.b_rgb, n4_of(4), lds, st, wave, lane);
        float dpd[KH / 2];
        gemm_layer<W, DMA, 4, 0, TW / 2, TW / 2>(d4, nullptr, pk + a.off.b_rgb, pk + a.off.b_dir, n4_of(KH / 2), lds, buf, st, o,
                                                 lane, wave, dpd, false, mb, false, mb, true, true, grow(a.gl.PDIR));
        get_mask(L - 1);  // FEAT
        gemm_layer<W, DMA, KH / 2, 0, TW, TW>(dpd, nullptr, pk + a.off.b_dir, pk + a.off.b_head, n4_of(KH + 4), lds, buf, st, o,
                                              lane, wave, dp, false, mb, false, mb, true, true, grow(a.gl.PFEAT));
        if (L > 1) get_mask(L - 2);  // H_{L-1}
        float da[4];
        da[0] = h == 0 ? go.w : 0.0f;
        da[1] = da[2] = da[3] = 0.0f;
        const float* nxt = L > 1 ? pk + a.off.b_xyz[L - 2] : nullptr;
        gemm_layer<W, DMA, KH, 4, TW, TW>(dp, da, pk + a.off.b_head, nxt, n4_of(KH), lds, buf, st, o, lane, wave, res, false,
                                          mb, false, mb, L > 1, true, grow(a.gl.P[L - 1]));
#pragma unroll
        for (int r = 0; r < KH; ++r) dp[r] = res[r];
    } else {
        float d4[4];
        d4[0] = h == 0 ? go.x : 0.0f;
        d4[1] = h == 0 ? go.y : 0.0f;
        d4[2] = h == 0 ? go.z : 0.0f;
        d4[3] = h == 0 ? go.w : 0.0f;
        if (L > 1) get_mask(L - 2);  // H_{L-1}
        first_chunk<W, DMA>(pk + a.off.b_head, n4_of(4), lds, st, wave, lane);
        const float* nxt = L > 1 ? pk + a.off.b_xyz[L - 2] : nullptr;
        gemm_layer<W, DMA, 4, 0, TW, TW>(d4, nullptr, pk + a.off.b_head, nxt, n4_of(KH), lds, buf, st, o, lane, wave, dp, false,
                                         mb, false, mb, L > 1, true, grow(a.gl.P[L - 1]));
    }
    // dp = d(pre-activation of H_{L-1}), already stored.  Walk down: dpre_{k-1} = relu'(H_{k-1}) * (W_{k-1}^T dpre_k);
    // H_0 = layer1 output has no activation (models.py:238).
    for (int k = L - 1; k >= 1; --k) {
        const bool masked = k - 1 >= 1;
        if (masked) get_mask(k - 2);  // H_{k-1}
        const float* nxt = k >= 2 ? pk + a.off.b_xyz[k - 2] : nullptr;
        gemm_layer<W, DMA, KH, 0, TW, TW>(dp, nullptr, pk + a.off.b_xyz[k - 1], nxt, n4_of(KH), lds, buf, st, o, lane, wave, res,
                                          false, mb, false, mb, masked, true, grow(a.gl.P[k - 1]));
#pragma unroll
        for (int r = 0; r < KH; ++r) dp[r] = res[r];
    }
    NH_PH(5);
    NH_PH_FLUSH(8);
}

// ---- weight gradients ----------------------------------------------------------------------------------------------
struct JobDev {
    int a_rows, a_prefix, a_tiles;
    int b_rows, b_prefix, b_row0, b_tiles;
    int wo, wi, po, pi;
    int r_lo, r_hi;
    int w_off, w_ld;
    int col_kind, col_base, col_count;
    int bias_off;
    int wg_start;
    int g;  // 32-sample tiles per LDS stage
};
constexpr int NH_JOBS_DEV = 32;
// weight-gradient kernel: 8 waves per workgroup (two per SIMD); two LDS stages of at most NH_WG_STAGE_FLOATS floats
// (+ slack for the operand prefetch that runs one k-step past the end of a stage)
constexpr int NH_WG_WAVES = 8;
constexpr int NH_WG_STAGE_FLOATS = 16384;
constexpr int NH_WG_LDS_BYTES = 2 * NH_WG_STAGE_FLOATS * 4 + 4096;
// floats of split-K partial per workgroup: 64 output tiles x 16 regs x 64 lanes, + 512 bias partials, + 128 for the
// timeline records of the 8 waves
constexpr int NH_PART = 65536 + 512 + 128;

struct WgradArgs {
    const float* stash;
    const float* grad;
    float* partial;
    float* g_params;
    int64_t nt;
    int njobs, total_wgs;
    JobDev jobs[NH_JOBS_DEV];
    short xslot[64];  // stash slot row -> reference column of the encoding, or -1
    short dslot[32];
};

// The weight-gradient GEMMs, dW[out, in] = sum over samples of dP[out][sample] * act[in][sample], as a split-K MFMA
// kernel over the sample-major images the forward / data-gradient kernels wrote.
//
// Measured on MI355X (scripts/mfma_rate.hip, profiles/r01_mfma_issue_cost.txt): with ONE wave per SIMD every
// instruction of the wave serialises with its MFMAs (v_add 4 cycles, ds_read ~12 per dword, an LDS-DMA ~14, a 64-cycle
// MFMA is not overlapped by anything of the same wave), which capped the previous 4-wave kernel at 82 % of the matrix
// pipe.  Here a workgroup is 8 waves = TWO per SIMD, each owning a patch of at most 8 accumulator tiles (128 AGPRs),
// so one wave's operand reads, bias sums and copy instructions issue underneath the other wave's MFMAs.
//
// A workgroup's operands for a run of G sample tiles are two CONTIGUOUS blocks of HBM ([G*32 samples][a_rows] of the
// gradient scratch, [G*32 samples][b_rows] of the stash: every job covers whole regions).  They are copied once per
// workgroup by LDS-DMA (1 KiB per instruction, no VGPRs), double buffered: stage n+1 streams in while stage n is
// multiplied.  Lane (i, k) reads its MFMA operands A[i][k] = lds[(2e + k) * rows + 32 * tile + i] with ds_read_b32,
// one k-step ahead of the MFMAs.
template <int PO, int PI>
struct WStep {
    float A[PO], B[PI];
};
template <int PO, int PI>
NH_DEVICE void wstep_load(WStep<PO, PI>& o, const float* pa, const float* pb) {
#pragma unroll
    for (int x = 0; x < PO; ++x) o.A[x] = pa[32 * x];
#pragma unroll
    for (int y = 0; y < PI; ++y) o.B[y] = pb[32 * y];
}
template <int PO, int PI, bool BIAS>
NH_DEVICE void wstep_mfma(const WStep<PO, PI>& o, f32x16 (&acc)[PO][PI], float (&bsum)[PO]) {
#pragma unroll
    for (int x = 0; x < PO; ++x) {
        if (BIAS) bsum[x] += o.A[x];
#pragma unroll
        for (int y = 0; y < PI; ++y) acc[x][y] = nh_mfma32(o.A[x], o.B[y], acc[x][y]);
    }
}

// One stage copy, cut into 1-KiB pieces (one DMA instruction each): block A (ntile * a_fl floats) -> stage[0 ..), block
// B (ntile * b_fl floats) -> stage[g * a_fl ..).  Wave w issues pieces w, w+8, ...; `issue(n)` emits the next n of
// them, so that the copy of stage n+1 is spread over the MFMA groups of stage n.  LDS destinations are byte addresses.
struct WStageDma {
    NhDmaSrc sa, sb;
    unsigned dst;  // LDS byte address of the stage
    int pa, ptot, boff, q, lane16;
    NH_MEMBER void init(const float* ga, const float* gb, int a_fl, int b_fl, int ntile, int g, unsigned stage_addr, int wave,
                        int lane) {
        sa = nh_dma_src(ga, (unsigned)(ntile * a_fl * 4));
        sb = nh_dma_src(gb, (unsigned)(ntile * b_fl * 4));
        dst = stage_addr;
        pa = ntile * a_fl / 256;
        ptot = pa + ntile * b_fl / 256;
        boff = (g * a_fl - pa * 256) * 4;  // piece q >= pa lands at stage + (g*a_fl + (q - pa)*256) floats
        q = wave;
        lane16 = lane * 16;
    }
    NH_MEMBER void issue(int n) {
        for (int c = 0; c < n && q < ptot; ++c, q += NH_WG_WAVES) {
            if (q < pa)
                nh_dma16a(sa, lane16, q * 1024, dst + q * 1024);
            else
                nh_dma16a(sb, lane16, (q - pa) * 1024, dst + boff + q * 1024);
        }
    }
};

// AR / BR: rows of the A / B region when known at compile time (0: read from the job) -- with constant strides the
// operand addresses of a whole stage are immediates of ONE base register.  BIAS: this wave also forms the bias
// gradient (row sums of A); only the waves of column iw == 0 do, the others skip the VALU adds.
template <int PO, int PI, int AR, int BR, bool BIAS>
NH_DEVICE void wgrad_body(const WgradArgs& a, const JobDev& jb, int64_t t0, int64_t t1, int ow, int iw, int wave, int lane,
                          int64_t wg, bool active, float* lds) {
    const int i = lane & 31, k = lane >> 5;
    f32x16 acc[PO][PI];
    float bsum[PO];
#pragma unroll
    for (int x = 0; x < PO; ++x) {
        bsum[x] = 0.0f;
#pragma unroll
        for (int y = 0; y < PI; ++y)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[x][y][c] = 0.0f;
    }
    const int ar = AR ? AR : jb.a_rows, br = BR ? BR : jb.b_rows;
    const int g = (AR && BR) ? NH_WG_STAGE_FLOATS / (32 * (AR + BR)) : jb.g;
    const int a_fl = 32 * ar, b_fl = 32 * br;
    const float* A0 = a.grad + (size_t)32 * (size_t)a.nt * (size_t)jb.a_prefix;
    const float* B0 = a.stash + (size_t)32 * (size_t)a.nt * (size_t)jb.b_prefix;
    const int nstage = (int)((t1 - t0 + g - 1) / g);
    const float* ga = A0 + (size_t)t0 * a_fl;  // stage n+1's blocks (running pointers: one 64-bit add per stage)
    const float* gb = B0 + (size_t)t0 * b_fl;
    int left = (int)(t1 - t0);                 // tiles not yet requested
    const unsigned lds_addr = nh_lds_addr(lds);
    WStageDma dma;
    dma.ptot = 0;
    if (nstage > 0) {
        const int nt0 = left < g ? left : g;
        dma.init(ga, gb, a_fl, b_fl, nt0, g, lds_addr, wave, lane);
        dma.issue(1 << 20);
        ga += (size_t)nt0 * a_fl, gb += (size_t)nt0 * b_fl, left -= nt0;
    }
    int ntile = (int)(t1 - t0) < g ? (int)(t1 - t0) : g;  // tiles of the stage being multiplied
    for (int n = 0; n < nstage; ++n) {
        const float* buf = lds + (n & 1) * NH_WG_STAGE_FLOATS;
        const float* pa = buf + k * ar + 32 * ow * PO + i;
        const float* pb = buf + g * a_fl + k * br + 32 * iw * PI + i;
        nh_wait_vmem();
        nh_block_sync();  // stage n has landed for every wave; everybody is done reading the other buffer
        WStep<PO, PI> c0, c1;
        if (active) wstep_load(c0, pa, pb);
        nh_sched_fence();  // first operand reads leave before the scalar set-up of the next copy
        const int ntn = left < g ? left : g;
        dma.ptot = 0;
        if (ntn > 0)
            dma.init(ga, gb, a_fl, b_fl, ntn, g, lds_addr + (unsigned)(((n + 1) & 1) * NH_WG_STAGE_FLOATS * 4), wave, lane);
        ga += (size_t)ntn * a_fl, gb += (size_t)ntn * b_fl, left -= ntn;
        if (active) {
            if (AR && BR && NH_WG_STAGE_FLOATS / (32 * (AR + BR)) == 1) {
                // one tile per stage, constant strides: 16 k-steps fully unrolled, every operand address an immediate
#pragma unroll
                for (int s = 0; s < 16; s += 2) {
                    wstep_load(c1, pa + (s + 1) * 2 * AR, pb + (s + 1) * 2 * BR);
                    dma.issue(1);
                    nh_sched_fence();
                    wstep_mfma<PO, PI, BIAS>(c0, acc, bsum);
                    wstep_load(c0, pa + (s + 2) * 2 * AR, pb + (s + 2) * 2 * BR);  // (last: one k-step past the stage, unused)
                    nh_sched_fence();
                    wstep_mfma<PO, PI, BIAS>(c1, acc, bsum);
                }
            } else {
                const int steps = 16 * ntile;  // k-steps of two samples each
                for (int s = 0; s < steps; s += 2) {
                    pa += 2 * ar, pb += 2 * br;
                    wstep_load(c1, pa, pb);
                    dma.issue(1);  // the next stage streams in underneath the MFMAs (at most 8 pieces per wave and stage)
                    nh_sched_fence();
                    wstep_mfma<PO, PI, BIAS>(c0, acc, bsum);
                    pa += 2 * ar, pb += 2 * br;
                    wstep_load(c0, pa, pb);  // the last one reads one k-step past the stage (slack / other block): unused
                    nh_sched_fence();
                    wstep_mfma<PO, PI, BIAS>(c1, acc, bsum);
                }
            }
        }
        dma.issue(1 << 20);  // idle waves, and whatever a short stage left over
        ntile = ntn;
    }
    if (!active) return;
    // split-K partial of this workgroup: output tile (a_t, b_t) of the job at [(a_t * b_tiles + b_t)][16 regs][64 lanes]
    float* part = a.partial + (size_t)wg * NH_PART;
#pragma unroll
    for (int x = 0; x < PO; ++x) {
        const int a_t = ow * PO + x;
        if (a_t >= jb.a_tiles) continue;
#pragma unroll
        for (int y = 0; y < PI; ++y) {
            const int b_t = iw * PI + y;
            if (b_t >= jb.b_tiles) continue;
            float* dst = part + (size_t)(a_t * jb.b_tiles + b_t) * 1024 + lane;
#pragma unroll
            for (int c = 0; c < 16; ++c) dst[c * 64] = acc[x][y][c];
        }
        if (BIAS) {  // bias gradient = row sums of A over this workgroup's samples
            const float tot = bsum[x] + nh_shfl_xor(bsum[x], 32);
            if (k == 0) part[65536 + a_t * 32 + i] = tot;
        }
    }
}

template <int PO, int PI>
NH_DEVICE void wgrad_dispatch(const WgradArgs& a, const JobDev& jb, int64_t t0, int64_t t1, int ow, int iw, int wave, int lane,
                              int64_t wg, bool active, float* lds) {
    const bool bias = iw == 0;  // (computed even when the job carries no bias tensor: the reduce kernel ignores it)
    if (PO == 4 && PI == 2 && jb.a_rows == 256 && jb.b_rows == 256) {  // the 256x256 jobs: 91 % of the 8x256 FLOPs
        if (bias)
            wgrad_body<PO, PI, 256, 256, true>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
        else
            wgrad_body<PO, PI, 256, 256, false>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
    } else if (bias) {
        wgrad_body<PO, PI, 0, 0, true>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
    } else {
        wgrad_body<PO, PI, 0, 0, false>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
    }
}

NH_KERNEL void NH_LB(64 * NH_WG_WAVES, 2) k_wgrad(WgradArgs a) {
    NH_DYN_LDS(smem);
    float* lds = (float*)smem;
    const unsigned long long t_begin = nh_wall_clock();
    const unsigned long long c_begin = nh_core_clock();
    const int64_t wg = blockIdx.x;
    int ji = 0;
    for (int q = 1; q < a.njobs; ++q)
        if ((int)wg >= a.jobs[q].wg_start) ji = q;
    const JobDev jb = a.jobs[ji];
    const int nks = (ji + 1 < a.njobs ? a.jobs[ji + 1].wg_start : a.total_wgs) - jb.wg_start;
    const int ks = (int)wg - jb.wg_start;
    const int64_t t0 = a.nt * ks / nks, t1 = a.nt * (ks + 1) / nks;
    const int lane = nh_lane(), wave = nh_wave_in_block();
    const bool active = wave < jb.wo * jb.wi;  // idle waves still copy and synchronise
    // wave -> patch (ow, iw); the column index is rotated by the row so that the bias-summing waves (iw == 0) of
    // different rows sit on different SIMDs (wave w runs on SIMD w % 4)
    const int ow = wave / jb.wi, iw = (wave % jb.wi + ow) % jb.wi;
    const int sel = jb.po * 8 + jb.pi;
    switch (sel) {
        case 4 * 8 + 2: wgrad_dispatch<4, 2>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 2 * 8 + 4: wgrad_dispatch<2, 4>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 4 * 8 + 1: wgrad_dispatch<4, 1>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 1 * 8 + 4: wgrad_dispatch<1, 4>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 2 * 8 + 2: wgrad_dispatch<2, 2>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 2 * 8 + 1: wgrad_dispatch<2, 1>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 1 * 8 + 2: wgrad_dispatch<1, 2>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        default: wgrad_dispatch<1, 1>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
    }
    if (lane == 0 && active) {  // timeline record (8 x u64 per wave): wall begin/end, job, K-slice, core-clock begin/end
        unsigned long long* dbg = (unsigned long long*)(a.partial + (size_t)wg * NH_PART + 65536 + 512) + wave * 8;
        dbg[0] = t_begin;
        dbg[1] = nh_wall_clock();
        dbg[2] = (unsigned long long)ji;
        dbg[3] = (unsigned long long)ks;
        dbg[4] = c_begin;
        dbg[5] = nh_core_clock();
    }
}

// fixed-order split-K reduction + scatter into the reference parameter layout
NH_KERNEL void k_wgrad_reduce(WgradArgs a) {
    const int ji = (int)(blockIdx.x >> 8);
    const JobDev jb = a.jobs[ji];
    const int local = (int)((blockIdx.x & 255u) * 256u + threadIdx.x);
    const int lane = local & 63, c = (local >> 6) & 15, tile = local >> 10;  // output tile (a_t, b_t) = a_t * b_tiles + b_t
    const int a_t = tile / jb.b_tiles, b_t = tile % jb.b_tiles;
    if (a_t >= jb.a_tiles) return;
    const int nks = (ji + 1 < a.njobs ? a.jobs[ji + 1].wg_start : a.total_wgs) - jb.wg_start;
    const int out_row = 32 * a_t + (c & 3) + 8 * (c >> 2) + 4 * (lane >> 5);
    const int in_row = 32 * b_t + (lane & 31);
    if (out_row >= jb.r_lo && out_row < jb.r_hi) {
        int col = -1;
        if (jb.col_kind == 0) {
            if (in_row < jb.col_count) col = jb.col_base + in_row;
        } else {
            const int cc = jb.col_kind == 1 ? (int)a.xslot[in_row] : (int)a.dslot[in_row];  // stash slot row -> column
            if (cc >= 0) col = jb.col_base + cc;
        }
        if (col >= 0) {
            // eight interleaved running sums (a fixed order: bit-reproducible) keep eight loads in flight per lane
            float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* p = a.partial + (size_t)jb.wg_start * NH_PART + ((size_t)tile * 16 + c) * 64 + lane;
            int q = 0;
            for (; q + 8 <= nks; q += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) sum[u] += p[(size_t)(q + u) * NH_PART];
            }
            for (; q < nks; ++q) sum[0] += p[(size_t)q * NH_PART];
            a.g_params[(size_t)jb.w_off + (size_t)(out_row - jb.r_lo) * jb.w_ld + col] =
                ((sum[0] + sum[1]) + (sum[2] + sum[3])) + ((sum[4] + sum[5]) + (sum[6] + sum[7]));
        }
    }
    if (jb.bias_off >= 0 && b_t == 0 && c == 0 && lane < 32) {
        const int brow = 32 * a_t + lane;
        if (brow >= jb.r_lo && brow < jb.r_hi) {
            float s = 0.0f;
            const float* p = a.partial + (size_t)jb.wg_start * NH_PART + 65536 + a_t * 32 + lane;
            for (int q = 0; q < nks; ++q) s += p[(size_t)q * NH_PART];
            a.g_params[(size_t)jb.bias_off + (brow - jb.r_lo)] = s;
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
constexpr int NH_WGRAD_TARGET_WGS = 1024;

// Split-K allocation: job j gets ks_j workgroups with ks_j proportional to its per-tile cost (every workgroup then
// runs for about the same time), and sum ks_j == NH_WGRAD_TARGET_WGS exactly (largest-remainder rounding) -- the grid
// is a whole number of rounds over the 256 CUs (one 4-wave workgroup per CU), with no straggler round.
void wgrad_schedule(const nerfhip_plan* p, int64_t nt, WgradArgs& w) {
    w.njobs = (int)p->jobs.size();
    int64_t cost[NH_JOBS_DEV];
    for (int q = 0; q < w.njobs; ++q) cost[q] = p->jobs[q].cost;
    if (const char* e = getenv("NERFHIP_WGRAD_COSTS")) {  // tuning aid: per-job costs measured by scripts/wgrad_timeline.py
        for (int q = 0; q < w.njobs && *e; ++q) {
            const long v = strtol(e, (char**)&e, 10);
            if (v > 0) cost[q] = v;
            if (*e == ',') ++e;
        }
    }
    int64_t total_cost = 0;
    for (int q = 0; q < w.njobs; ++q) total_cost += cost[q];
    int64_t ks[NH_JOBS_DEV], rem[NH_JOBS_DEV];
    int64_t used = 0;
    for (int q = 0; q < w.njobs; ++q) {
        const int64_t num = (int64_t)NH_WGRAD_TARGET_WGS * cost[q];
        ks[q] = num / total_cost;
        rem[q] = num % total_cost;
        if (ks[q] < 1) {
            ks[q] = 1;
            rem[q] = 0;
        }
        used += ks[q];
    }
    while (used < NH_WGRAD_TARGET_WGS) {  // hand out the remaining workgroups by largest remainder
        int best = 0;
        for (int q = 1; q < w.njobs; ++q)
            if (rem[q] > rem[best]) best = q;
        ks[best] += 1;
        rem[best] = -1;
        used += 1;
    }
    while (used > NH_WGRAD_TARGET_WGS) {  // (only if many jobs were lifted to 1) take from the largest
        int best = 0;
        for (int q = 1; q < w.njobs; ++q)
            if (ks[q] > ks[best]) best = q;
        if (ks[best] <= 1) break;
        ks[best] -= 1;
        used -= 1;
    }
    int start = 0;
    for (int q = 0; q < w.njobs; ++q) {
        const NhJob& j = p->jobs[q];
        if (ks[q] > nt) ks[q] = nt;
        JobDev& d = w.jobs[q];
        d.a_rows = j.a_region_rows;
        d.a_prefix = (int)j.a_row_prefix;
        d.a_tiles = j.a_tiles;
        d.b_rows = j.b_region_rows;
        d.b_prefix = (int)j.b_row_prefix;
        d.b_row0 = j.b_row0;
        d.b_tiles = j.b_tiles;
        d.wo = j.wo;
        d.wi = j.wi;
        d.po = j.po;
        d.pi = j.pi;
        d.r_lo = j.r_lo;
        d.r_hi = j.r_hi;
        d.w_off = (int)j.w_off;
        d.w_ld = j.w_ld;
        d.col_kind = j.col_kind;
        d.col_base = j.col_base;
        d.col_count = j.col_count;
        d.bias_off = (int)j.bias_off;
        d.wg_start = start;
        d.g = NH_WG_STAGE_FLOATS / (32 * (j.a_region_rows + j.b_region_rows));
        if (d.g < 1) d.g = 1;
        start += (int)ks[q];
    }
    w.total_wgs = start;
    for (int r = 0; r < 64; ++r) w.xslot[r] = (short)p->xyz_slot_col[r];
    for (int r = 0; r < 32; ++r) w.dslot[r] = (short)p->dir_slot_col[r];
}

template <class K>
int set_lds_limit(K kern, int bytes) {
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

// NERFHIP_STAGE selects how weight chunks reach LDS: reg (global -> VGPR -> ds_write), front (LDS-DMA issued at the
// start of a tile), mix (LDS-DMA pieces and row stores spread between the MFMA groups; default).
int use_dma() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("NERFHIP_STAGE");
        v = !e ? 2 : (e[0] == 'r' ? 0 : (e[0] == 'f' ? 1 : 2));  // reg | front | mix (default: measured fastest)
    }
    return v;
}

}  // namespace

int64_t nh_mlp_bwd_scratch_bytes(nerfhip_plan* p, int64_t M) {
    const int64_t nt = nh_ceil_div(M, 128) * 4;
    WgradArgs w;
    wgrad_schedule(p, nt > 0 ? nt : 1, w);
    return (nt * p->grad.total_rows * 32 + (int64_t)w.total_wgs * NH_PART) * (int64_t)sizeof(float);
}

int nh_mlp_forward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                   nerfhip_stream_t stream) {
    NH_REQUIRE(p && packed && out && M >= 0, "mlp_fwd: bad arguments");
    if (M == 0) return NERFHIP_OK;
    if (in.mode == 1) {
        NH_REQUIRE(p->freqs_set, "mlp_fwd: nerfhip_plan_set_freqs has not been called");
        NH_REQUIRE(in.rays && in.z && in.S > 0 && in.ray_stride >= (p->view ? 11 : 8), "mlp_fwd: bad fused input");
    } else {
        NH_REQUIRE(in.x, "mlp_fwd: x is NULL");
    }
    if (p->v16) return nh_mlp16_forward(p, packed, in, M, out, stash, stream);
    MlpFwdArgs a;
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
    for (int h = 0; h < 2; ++h) {
        for (int r = 0; r < NH_KRX; ++r) a.xcol[h][r] = (short)p->xyz_col[h][r];
        for (int r = 0; r < NH_KRD; ++r) a.dcol[h][r] = (short)p->dir_col[h][r];
    }
    for (int k = 0; k < 16; ++k) {
        a.fx[k] = p->freqs_xyz[k];
        a.fd[k] = p->freqs_dir[k];
    }
    a.Lx = p->cfg.num_encoding_fn_xyz;
    a.Ld = p->view ? p->cfg.num_encoding_fn_dir : 0;
    a.P0x = p->P0x;
    a.P0d = p->P0d;
    a.out = out;
    a.stash = stash;
    a.sl = p->stash;
    const int64_t grid = nh_ceil_div(M, 128);
    int rc = NERFHIP_OK;
#define NH_FWD_CASE(WW, VV, DD)                                                         \
    {                                                                                   \
        rc = set_lds_limit(k_mlp_fwd<WW, VV, DD>, Cfg<WW>::LDS_BYTES);                  \
        if (rc) return rc;                                                              \
        NH_LAUNCH((k_mlp_fwd<WW, VV, DD>), grid, 256, Cfg<WW>::LDS_BYTES, stream, a);   \
    }
    const int dma = use_dma();
#define NH_FWD_GEO(DD)                                      \
    if (p->W == 256 && p->view) NH_FWD_CASE(256, true, DD)  \
    else if (p->W == 256) NH_FWD_CASE(256, false, DD)       \
    else if (p->view) NH_FWD_CASE(128, true, DD)            \
    else NH_FWD_CASE(128, false, DD)
    if (dma == 0) { NH_FWD_GEO(0) } else if (dma == 2) { NH_FWD_GEO(2) } else { NH_FWD_GEO(1) }
#undef NH_FWD_GEO
#undef NH_FWD_CASE
    return nh_launch_status("mlp_fwd");
}

int nh_mlp_backward(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash,
                    float* scratch, int64_t scratch_bytes, float* g_params, nerfhip_stream_t stream) {
    NH_REQUIRE(p && packed && g_out && stash && scratch && g_params && M > 0, "mlp_bwd: bad arguments");
    NH_REQUIRE(scratch_bytes >= nh_mlp_bwd_scratch_bytes(p, M), "mlp_bwd: scratch too small (%lld < %lld)",
               (long long)scratch_bytes, (long long)nh_mlp_bwd_scratch_bytes(p, M));
    const int64_t nt = nh_ceil_div(M, 128) * 4;
    int rc = NERFHIP_OK;
    if (p->v16) {
        rc = nh_mlp16_dgrad(p, packed, g_out, M, stash, scratch, stream);
        if (rc) return rc;
    } else {
    DgradArgs d;
    memset(&d, 0, sizeof(d));
    d.packed = packed;
    d.packed_bytes = (unsigned)(p->packed_floats * 4);
    d.off = p->po;
    d.L = p->L;
    d.M = M;
    d.nt = nt;
    d.g_out = g_out;
    d.stash = stash;
    d.sl = p->stash;
    d.grad = scratch;
    d.gl = p->grad;
    const int64_t grid = nh_ceil_div(M, 128);
#define NH_BWD_CASE(WW, VV, DD)                                                           \
    {                                                                                     \
        rc = set_lds_limit(k_mlp_dgrad<WW, VV, DD>, Cfg<WW>::LDS_BYTES);                  \
        if (rc) return rc;                                                                \
        NH_LAUNCH((k_mlp_dgrad<WW, VV, DD>), grid, 256, Cfg<WW>::LDS_BYTES, stream, d);   \
    }
    const int dma = use_dma();
#define NH_BWD_GEO(DD)                                      \
    if (p->W == 256 && p->view) NH_BWD_CASE(256, true, DD)  \
    else if (p->W == 256) NH_BWD_CASE(256, false, DD)       \
    else if (p->view) NH_BWD_CASE(128, true, DD)            \
    else NH_BWD_CASE(128, false, DD)
    if (dma == 0) { NH_BWD_GEO(0) } else if (dma == 2) { NH_BWD_GEO(2) } else { NH_BWD_GEO(1) }
#undef NH_BWD_GEO
#undef NH_BWD_CASE
    rc = nh_launch_status("mlp_dgrad");
    if (rc) return rc;
    }

    WgradArgs w;
    memset(&w, 0, sizeof(w));
    wgrad_schedule(p, nt, w);
    w.stash = stash;
    w.grad = scratch;
    w.partial = scratch + (size_t)nt * (size_t)p->grad.total_rows * 32;
    w.g_params = g_params;
    w.nt = nt;
    for (int q = 0; q < w.njobs; ++q) {
        const NhJob& j = p->jobs[q];
        NH_REQUIRE(j.b_row0 == 0 && 32 * j.a_tiles == j.a_region_rows && 32 * j.b_tiles == j.b_region_rows &&
                       32 * (j.a_region_rows + j.b_region_rows) <= NH_WG_STAGE_FLOATS && j.a_tiles * j.b_tiles <= 64,
                   "wgrad: job %d does not cover whole regions", q);
    }
    rc = set_lds_limit(k_wgrad, NH_WG_LDS_BYTES);
    if (rc) return rc;
    NH_LAUNCH(k_wgrad, w.total_wgs, 64 * NH_WG_WAVES, NH_WG_LDS_BYTES, stream, w);
    rc = nh_launch_status("wgrad");
    if (rc) return rc;
    NH_LAUNCH(k_wgrad_reduce, w.njobs * 256, 256, 0, stream, w);
    return nh_launch_status("wgrad_reduce");
}

extern "C" int64_t nerfhip_plan_bwd_scratch_bytes(nerfhip_plan_t plan, int64_t m) {
    if (!plan || m < 0) return -1;
    return nh_mlp_bwd_scratch_bytes(plan, m);
}

extern "C" int nerfhip_mlp_fwd(nerfhip_plan_t plan, const float* packed, const float* x, int64_t m, float* out,
                               void* stash, nerfhip_stream_t stream) {
    NhMlpInput in;
    memset(&in, 0, sizeof(in));
    in.mode = 0;
    in.x = x;
    return nh_mlp_forward(plan, packed, in, m, out, (float*)stash, stream);
}

extern "C" int nerfhip_mlp_bwd(nerfhip_plan_t plan, const float* packed, const float* g_out, int64_t m,
                               const void* stash, void* scratch, int64_t scratch_bytes, float* g_params,
                               nerfhip_stream_t stream) {
    return nh_mlp_backward(plan, packed, g_out, m, (const float*)stash, (float*)scratch, scratch_bytes, g_params, stream);
}

// loop_mock.hip -- GPU-box microbenchmark (round 2): the inner loops of the three MLP kernels reduced to their
// LDS-operand reads + MFMAs + barriers, at the kernels' real occupancy (8-wave workgroups, two waves per SIMD, one
// workgroup per CU), to price structural variants before they are built into the product kernels.
//   F*: forward / data-gradient loop (v_mfma_f32_16x16x4_f32, 16 output tiles, A operands by ds_read_b128, chunks of KC k-steps)
//   W*: weight-gradient loop (v_mfma_f32_32x32x2_f32, 4x2 tile patch per wave, stages of 16 k-steps)
// Build:  hipcc --offload-arch=gfx950 -O3 scripts/loop_mock.hip -o scripts/loop_mock
// Output: per variant, efficiency = ideal MFMA cycles of a SIMD / measured shader cycles (s_memtime), and TFLOP/s by wall clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define DEV __device__ __forceinline__

DEV unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) void*)p; }
DEV void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct Dma {
    i32x4 r;
};
DEV Dma dma_src(const float* base, unsigned bytes) {
    Dma s;
    const unsigned long long b = (unsigned long long)base;
    s.r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    s.r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xFFFFu));
    s.r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    s.r[3] = 0x00020000;
    return s;
}
DEV void dma16(const Dma& s, int voff, int soff, unsigned lds_wave_addr) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(__builtin_amdgcn_readfirstlane((int)lds_wave_addr)), "v"(voff), "s"(s.r), "s"(__builtin_amdgcn_readfirstlane(soff))
        : "memory");
}

// ---- forward-like loop -------------------------------------------------------------------------------------------------
// VAR bits / values:
enum {
    F_BASE = 0,      // barrier at chunk start, A reads one k-step ahead (the product loop)
    F_NOBAR = 1,     // no barriers at all (upper bound for the barrier cost)
    F_MIDBAR = 2,    // ONE barrier per chunk, placed after k-step KC/2 (the "early barrier" of a 3-buffer ring)
    F_NOREAD = 3,    // no operand reads (pure MFMA at this occupancy)
    F_ILV = 4,       // the 4 reads of a k-step spread over its MFMAs (one per 4 MFMAs) instead of issued up front
    F_PRIO = 5,      // F_BASE + s_setprio 1 on waves 4..7
    F_AACC = 6,      // accumulators in AGPRs (asm MFMA)
    F_AOPS = 7,      // A operands read into AGPRs (asm ds_read + asm MFMA)
    F_DMA = 8,       // F_BASE + the chunk copy traffic (4 LDS-DMA pieces per wave and chunk into the other buffer)
    F_MIDBAR_DMA = 9,// F_MIDBAR + copy traffic into a third buffer
    F_AHEAD2 = 10,   // reads two k-steps ahead
    // F_DMA + the stash stores of the training forward: one 16-byte-per-lane store per wave every 4 k-steps, 128 KB per
    // workgroup and layer, a fresh region per layer (2 GB per launch)
    F_DMA_ST = 11,   //   product pattern: lane (sample j, group g) -> 16 B at sample*1 KB + (16 t + 4 g)*4: 16 segments of 64 B
    F_DMA_STC = 12,  //   the same bytes as ONE contiguous 1 KB block per instruction
    F_DMA_STNT = 13, //   product pattern, non-temporal
    F_DMA_ST1 = 14,  //   product pattern, but all layers of a workgroup overwrite ONE 128 KB region (no fresh HBM pages)
    F_DMA_ST_V4 = 15,  // product pattern; the chunk hand-over waits with vmcnt(4): for the copy pieces (all issued before the
                       // chunk's four stores), not for the write acknowledgements of those stores
    F_DMA_ST_LATE = 16,  // F_DMA_ST_V4 with the four stores at k-steps 8, 10, 12, 14
    F_ST_NODMA = 17,     // product stores, no copy traffic (wait + barrier kept)
    F_DMA_ST_PAIR = 18,  // two adjacent tiles (one full 128-byte line per sample row) stored back to back every 8 k-steps
    F_DMA_ST_SC01 = 19,  // product pattern, stores with sc0 sc1 (system scope: write-through)
    F_DMA_ST_SC1 = 20,   // ... sc1
    F_DMA_ST_SC01NT = 21,  // ... sc0 sc1 nt
    F_MIDBAR_DMA_ST = 22,  // F_MIDBAR_DMA (three-buffer ring) + product stores
    F_DMA_ST_END = 23,     // the four stores at k-steps 12..15
    F_DMA_ST_MID = 24,     // the four stores at k-steps 4, 6, 8, 10
    F_DMA_ST_LATE_SPREAD = 25,  // stores at 8, 10, 12, 14; the copy pieces one per k-step (k-steps 0..7) instead of a burst
};
constexpr bool f_is_dma(int v) { return v == F_DMA || (v >= F_DMA_ST && v != F_MIDBAR_DMA_ST); }
constexpr bool f_is_mid(int v) { return v == F_MIDBAR || v == F_MIDBAR_DMA || v == F_MIDBAR_DMA_ST; }

template <int VAR, int KC, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void kF(const float* __restrict__ gsrc, float* out, unsigned long long* cyc,
                                                     int nchunks, float* big) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TQ = 4, T = 16;
    constexpr int CH = KC * TQ * 256;  // floats per chunk buffer
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int i = threadIdx.x; i < 3 * CH; i += 64 * NW) lds[i] = (float)((i * 7) & 15) * 0.125f;
    __syncthreads();
    float act[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) act[r] = (float)((lane + r) & 7) * 0.25f;
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (VAR == F_PRIO) {
        if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
    }
    const Dma src = dma_src(gsrc, 1u << 20);
    const unsigned lbase = lds_addr_of(lds);
    int buf = 0;
    constexpr int NCH = 64 / KC;  // chunks per "layer" of 64 k-steps: unrolled, so that act[] is indexed statically
    const unsigned long long t0 = clock64();
    for (int layer = 0; layer < nchunks / NCH; ++layer) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (VAR == F_BASE || VAR == F_ILV || VAR == F_PRIO || VAR == F_AACC || VAR == F_AOPS || f_is_dma(VAR) ||
            VAR == F_AHEAD2) {
            if (VAR == F_DMA_ST_V4 || VAR == F_DMA_ST_LATE)
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (f_is_dma(VAR))
                wait_vm0();
            __builtin_amdgcn_s_barrier();
        }
        if (f_is_dma(VAR) && VAR != F_ST_NODMA && VAR != F_DMA_ST_LATE_SPREAD) {
            const int nb = buf ^ 1;
            for (int q = wave; q < KC * TQ; q += NW) dma16(src, lane * 16, (c & 3) * 65536 + q * 1024, lbase + nb * CH * 4 + q * 1024);
        }
        const float4* wp = (const float4*)(lds + buf * CH) + lane;
        if constexpr (VAR == F_AOPS) {
            // A operands live in AGPRs: asm reads, explicit waits
            f32x4 a0[TQ], a1[TQ];
            const unsigned ab = lbase + buf * CH * 4 + lane * 16;
#pragma unroll
            for (int q = 0; q < TQ; ++q) asm volatile("ds_read_b128 %0, %1" : "=a"(a0[q]) : "v"(ab + q * 1024));
#pragma unroll
            for (int ks = 0; ks < KC; ks += 2) {
#pragma unroll
                for (int q = 0; q < TQ; ++q)
                    asm volatile("ds_read_b128 %0, %1" : "=a"(a1[q]) : "v"(ab + ((ks + 1) * TQ + q) * 1024));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                FENCE();
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float b = act[(c * KC + ks) & 63];
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "a"(a0[t >> 2][t & 3]), "v"(b));
                }
                if (ks + 2 < KC) {
#pragma unroll
                    for (int q = 0; q < TQ; ++q)
                        asm volatile("ds_read_b128 %0, %1" : "=a"(a0[q]) : "v"(ab + ((ks + 2) * TQ + q) * 1024));
                    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                FENCE();
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float b = act[(c * KC + ks + 1) & 63];
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "a"(a1[t >> 2][t & 3]), "v"(b));
                }
            }
        } else {
            float4 a[3][TQ];
            constexpr int AH = VAR == F_AHEAD2 ? 2 : 1;
            if (VAR != F_NOREAD) {
#pragma unroll
                for (int d = 0; d < AH; ++d)
#pragma unroll
                    for (int q = 0; q < TQ; ++q) a[d][q] = wp[(d * TQ + q) * 64];
            } else {
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int q = 0; q < TQ; ++q) a[d][q] = make_float4(0.5f, 0.25f, 0.125f, 1.0f);
            }
#pragma unroll
            for (int ks = 0; ks < KC; ++ks) {
                if (VAR != F_NOREAD && VAR != F_ILV && ks + AH < KC) {
#pragma unroll
                    for (int q = 0; q < TQ; ++q) a[(ks + AH) % 3][q] = wp[((ks + AH) * TQ + q) * 64];
                }
                if constexpr (VAR == F_DMA_ST_LATE_SPREAD) {
                    const int q = wave + ks * NW;
                    if (q < KC * TQ) dma16(src, lane * 16, (c & 3) * 65536 + q * 1024, lbase + (buf ^ 1) * CH * 4 + q * 1024);
                }
                if constexpr (VAR >= F_DMA_ST) {
                    const int r = c * KC + ks;
                    if constexpr (VAR == F_DMA_ST_PAIR) {
                        if ((r & 7) == 0) {
                            const int j = lane & 15, g = lane >> 4;
                            char* blk = (char*)big + ((size_t)layer * gridDim.x + blockIdx.x) * (128u * 1024u);
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int t = (r >> 2) + u;
                                char* p = blk + ((wave >> 1) * 32 + 16 * (wave & 1) + j) * 1024 + (16 * t + 4 * g) * 4;
                                *(f32x4*)p = f32x4{act[4 * t], act[4 * t + 1], act[4 * t + 2], act[4 * t + 3]};
                            }
                        }
                    }
                    constexpr bool late = VAR == F_DMA_ST_LATE || VAR == F_DMA_ST_LATE_SPREAD;
                    const bool now = VAR == F_DMA_ST_PAIR  ? false
                                     : late                ? (ks >= KC / 2 && (ks & 1) == 0)
                                     : VAR == F_DMA_ST_END ? ks >= KC - 4
                                     : VAR == F_DMA_ST_MID ? (ks >= 4 && ks <= 10 && (ks & 1) == 0)
                                                           : (r & 3) == 0;
                    if (now) {
                        const int t = late                  ? (c * (KC / 4) + (ks - KC / 2) / 2) & 15
                                      : VAR == F_DMA_ST_END ? (c * 4 + ks - (KC - 4)) & 15
                                      : VAR == F_DMA_ST_MID ? (c * 4 + (ks - 4) / 2) & 15
                                                            : r >> 2;
                        const int j = lane & 15, g = lane >> 4;
                        const size_t region = VAR == F_DMA_ST1 ? (size_t)blockIdx.x : (size_t)layer * gridDim.x + blockIdx.x;
                        char* blk = (char*)big + region * (128u * 1024u);
                        char* p = VAR == F_DMA_STC ? blk + (t * NW + wave) * 1024 + lane * 16
                                                   : blk + ((wave >> 1) * 32 + 16 * (wave & 1) + j) * 1024 + (16 * t + 4 * g) * 4;
                        f32x4 v = {act[4 * t], act[4 * t + 1], act[4 * t + 2], act[4 * t + 3]};
                        if (VAR == F_DMA_STNT)
                            __builtin_nontemporal_store(v, (f32x4*)p);
                        else if (VAR == F_DMA_ST_SC01)
                            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
                        else if (VAR == F_DMA_ST_SC1)
                            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
                        else if (VAR == F_DMA_ST_SC01NT)
                            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
                        else
                            *(f32x4*)p = v;
                    }
                }
                FENCE();
                const float b = act[(c * KC + ks) & 63];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float4& w = a[ks % 3][t >> 2];
                    const float av = (t & 3) == 0 ? w.x : ((t & 3) == 1 ? w.y : ((t & 3) == 2 ? w.z : w.w));
                    if constexpr (VAR == F_AACC) {
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(av), "v"(b));
                    } else {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[t], 0, 0, 0);
                    }
                    if (VAR == F_ILV && (t & 3) == 3 && ks + 1 < KC) {
                        a[(ks + 1) % 3][t >> 2] = wp[((ks + 1) * TQ + (t >> 2)) * 64];
                        FENCE();
                    }
                }
                if (f_is_mid(VAR) && ks == KC / 2 - 1) {
                    FENCE();
                    if (VAR == F_MIDBAR_DMA || VAR == F_MIDBAR_DMA_ST) wait_vm0();
                    __builtin_amdgcn_s_barrier();
                    if (VAR == F_MIDBAR_DMA || VAR == F_MIDBAR_DMA_ST) {
                        const int nb = (buf + 2) % 3;
                        for (int q = wave; q < KC * TQ; q += NW)
                            dma16(src, lane * 16, (c & 3) * 65536 + q * 1024, lbase + nb * CH * 4 + q * 1024);
                    }
                    FENCE();
                }
            }
        }
        if (f_is_mid(VAR))
            buf = (buf + 1) % 3;
        else
            buf ^= 1;
    }
    }
    const unsigned long long t1 = clock64();
    wait_vm0();
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) r += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (r == 12345.678f) out[blockIdx.x * 64 * NW + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
}

// ---- weight-gradient-like loop -----------------------------------------------------------------------------------------
enum {
    W_BASE = 0,    // 6 x ds_read_b32 per k-step (4 A tiles + 2 B tiles), one k-step ahead, barrier per stage (the product loop)
    W_WIDE = 1,    // A: one ds_read_b128 (rows 4i..4i+3 = four tiles), B: one ds_read_b64 (rows 2i, 2i+1 = two tiles)
    W_PAIR = 2,    // W_WIDE, B image pair-interleaved: one ds_read_b128 = two tiles x two k-steps
    W_NOREAD = 3,  // pure MFMA
    W_NOBAR = 4,   // W_BASE without barriers
    W_WIDE_NOBAR = 5,
    W_BASE_DMA = 6,  // W_BASE + the stage copy (8 LDS-DMA pieces per wave and stage, spread over the k-steps)
    W_WIDE_DMA = 7,
    W_WIDE_MID_DMA = 8,  // W_WIDE + copy + the barrier moved to the middle of the stage (3-buffer ring)
    W_BIAS = 9,      // W_BASE + bias sums on waves 0 and 5 (4 v_add per k-step)
    W_WIDE_AOPS = 10,  // W_WIDE with AGPR operand destinations
    W_PAIR_DMA = 11,
    W_WIDE_DMA_HBM = 12,  // W_WIDE_DMA with the stage copy streaming fresh lines from HBM (a 4 GB window, 12 GB per launch)
                          // instead of re-reading 256 KB that live in L2: what k_wgrad really does
};

template <int VAR, int KS>
__global__ __launch_bounds__(512, 2) void kW(const float* __restrict__ gsrc, float* out, unsigned long long* cyc, int nstages,
                                             float* big) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PO = 4, PI = 2, ROWS = 512, ST = 2 * KS * ROWS;  // floats per stage: 2*KS samples x (256 + 256) rows
    constexpr int NPIECE = ST / 256;                                // 1-KiB copy pieces per stage
    constexpr bool MID = VAR == W_WIDE_MID_DMA;
    constexpr int NBUF = MID ? 3 : 2;
    const int lane = threadIdx.x & 63, i = lane & 31, k = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ow = wave >> 2, iw = wave & 3;  // 2 x 4 wave grid over the 8 x 8 output tiles
    for (int q = threadIdx.x; q < (NBUF * ST > 40960 ? 40960 : NBUF * ST); q += 512) lds[q] = (float)((q * 5) & 15) * 0.125f;
    __syncthreads();
    f32x16 acc[PO][PI];
#pragma unroll
    for (int x = 0; x < PO; ++x)
#pragma unroll
        for (int y = 0; y < PI; ++y)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[x][y][c] = 0.f;
    float bsum[PO] = {0.f, 0.f, 0.f, 0.f};
    const bool bias = VAR == W_BIAS && (wave == 0 || wave == 5);
    const Dma src = VAR == W_WIDE_DMA_HBM ? dma_src(big, 0xFFFFFFFFu) : dma_src(gsrc, 1u << 20);
    const unsigned lbase = lds_addr_of(lds);
    constexpr bool DMA = VAR == W_BASE_DMA || VAR == W_WIDE_DMA || VAR == W_WIDE_MID_DMA || VAR == W_PAIR_DMA || VAR == W_WIDE_DMA_HBM;
    constexpr bool WIDE = VAR == W_WIDE || VAR == W_WIDE_NOBAR || VAR == W_WIDE_DMA || VAR == W_WIDE_MID_DMA || VAR == W_WIDE_DMA_HBM;
    constexpr bool PAIR = VAR == W_PAIR || VAR == W_PAIR_DMA;
    constexpr bool NOBAR = VAR == W_NOBAR || VAR == W_WIDE_NOBAR || VAR == W_NOREAD;
    int buf = 0;
    const unsigned long long t0 = clock64();
    for (int n = 0; n < nstages; ++n) {
        const float* sb = lds + buf * ST;
        if (!NOBAR && !MID) {
            if (DMA) wait_vm0();
            __builtin_amdgcn_s_barrier();
        }
        int dq = wave;  // next copy piece of this wave
        const unsigned dst = lbase + (unsigned)(((buf + (MID ? 2 : 1)) % NBUF) * ST * 4);
        if constexpr (VAR == W_WIDE_AOPS) {
            f32x4 a0, a1;
            f32x2 b0, b1;
            const unsigned pa = lbase + (unsigned)((buf * ST + k * ROWS + 128 * ow + 4 * i) * 4);
            const unsigned pb = lbase + (unsigned)((buf * ST + k * ROWS + 256 + 64 * iw + 2 * i) * 4);
            asm volatile("ds_read_b128 %0, %1" : "=a"(a0) : "v"(pa));
            asm volatile("ds_read_b64 %0, %1" : "=a"(b0) : "v"(pb));
#pragma unroll
            for (int s = 0; s < KS; s += 2) {
                asm volatile("ds_read_b128 %0, %1" : "=a"(a1) : "v"(pa + (s + 1) * 2 * ROWS * 4));
                asm volatile("ds_read_b64 %0, %1" : "=a"(b1) : "v"(pb + (s + 1) * 2 * ROWS * 4));
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                FENCE();
#pragma unroll
                for (int x = 0; x < PO; ++x)
#pragma unroll
                    for (int y = 0; y < PI; ++y)
                        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[x][y]) : "a"(a0[x]), "a"(b0[y]));
                if (s + 2 < KS) {
                    asm volatile("ds_read_b128 %0, %1" : "=a"(a0) : "v"(pa + (s + 2) * 2 * ROWS * 4));
                    asm volatile("ds_read_b64 %0, %1" : "=a"(b0) : "v"(pb + (s + 2) * 2 * ROWS * 4));
                    asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                FENCE();
#pragma unroll
                for (int x = 0; x < PO; ++x)
#pragma unroll
                    for (int y = 0; y < PI; ++y)
                        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[x][y]) : "a"(a1[x]), "a"(b1[y]));
            }
        } else if constexpr (WIDE) {
            const float4* pa = (const float4*)(sb + k * ROWS + 128 * ow + 4 * i);
            const f32x2* pb = (const f32x2*)(sb + k * ROWS + 256 + 64 * iw + 2 * i);
            float4 a[2];
            f32x2 b[2];
            a[0] = pa[0];
            b[0] = pb[0];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
                    a[(s + 1) & 1] = pa[(s + 1) * 2 * ROWS / 4];
                    b[(s + 1) & 1] = pb[(s + 1) * 2 * ROWS / 2];
                }
                if (DMA && (s & 1) == 0 && dq < NPIECE) {
                    const unsigned so = VAR == W_WIDE_DMA_HBM
                                            ? (((unsigned)blockIdx.x * (unsigned)nstages + (unsigned)n) & 65535u) * 65536u
                                            : (unsigned)(n & 3) * 65536u;
                    dma16(src, lane * 16, (int)(so + dq * 1024), dst + dq * 1024);
                    dq += 8;
                }
                FENCE();
                const float4 av = a[s & 1];
                const f32x2 bv = b[s & 1];
                const float ax[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                for (int x = 0; x < PO; ++x)
#pragma unroll
                    for (int y = 0; y < PI; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[x], bv[y], acc[x][y], 0, 0, 0);
                if (MID && s == KS / 2 - 1) {
                    FENCE();
                    wait_vm0();
                    __builtin_amdgcn_s_barrier();
                    FENCE();
                }
            }
        } else if constexpr (PAIR) {
            const float4* pa = (const float4*)(sb + k * ROWS + 128 * ow + 4 * i);
            // B image pair-interleaved: [pair of samples][row][2]; lane (i,k) of k-step pair q reads pair 2q+k, rows 2i, 2i+1
            const float4* pb = (const float4*)(sb + 256 * 32 + k * 512 + 128 * iw + 4 * i);
            float4 a[2];
            float4 b[2];
            a[0] = pa[0];
            b[0] = pb[0];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
                    a[(s + 1) & 1] = pa[(s + 1) * 2 * ROWS / 4];
                    if (((s + 1) & 1) == 0) b[((s + 1) >> 1) & 1] = pb[((s + 1) >> 1) * 1024 / 4];
                }
                if (DMA && (s & 1) == 0 && dq < NPIECE) {
                    dma16(src, lane * 16, (n & 3) * 65536 + dq * 1024, dst + dq * 1024);
                    dq += 8;
                }
                FENCE();
                const float4 av = a[s & 1];
                const float4 bq = b[(s >> 1) & 1];
                const float ax[4] = {av.x, av.y, av.z, av.w};
                const float bx[2] = {(s & 1) ? bq.y : bq.x, (s & 1) ? bq.w : bq.z};
#pragma unroll
                for (int x = 0; x < PO; ++x)
#pragma unroll
                    for (int y = 0; y < PI; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[x], bx[y], acc[x][y], 0, 0, 0);
            }
        } else {
            const float* pa = sb + k * ROWS + 32 * ow * PO + i;
            const float* pb = sb + k * ROWS + 256 + 32 * iw * PI + i;
            float a[2][PO], b[2][PI];
            if (VAR != W_NOREAD) {
#pragma unroll
                for (int x = 0; x < PO; ++x) a[0][x] = pa[32 * x];
#pragma unroll
                for (int y = 0; y < PI; ++y) b[0][y] = pb[32 * y];
            } else {
#pragma unroll
                for (int d = 0; d < 2; ++d) {
#pragma unroll
                    for (int x = 0; x < PO; ++x) a[d][x] = 0.5f + x;
#pragma unroll
                    for (int y = 0; y < PI; ++y) b[d][y] = 0.25f + y;
                }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (VAR != W_NOREAD && s + 1 < KS) {
#pragma unroll
                    for (int x = 0; x < PO; ++x) a[(s + 1) & 1][x] = pa[(s + 1) * 2 * ROWS + 32 * x];
#pragma unroll
                    for (int y = 0; y < PI; ++y) b[(s + 1) & 1][y] = pb[(s + 1) * 2 * ROWS + 32 * y];
                }
                if (DMA && (s & 1) == 0 && dq < NPIECE) {
                    dma16(src, lane * 16, (n & 3) * 65536 + dq * 1024, dst + dq * 1024);
                    dq += 8;
                }
                FENCE();
#pragma unroll
                for (int x = 0; x < PO; ++x) {
                    if (bias) bsum[x] += a[s & 1][x];
#pragma unroll
                    for (int y = 0; y < PI; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][x], b[s & 1][y], acc[x][y], 0, 0, 0);
                }
            }
        }
        buf = (buf + 1) % NBUF;
    }
    const unsigned long long t1 = clock64();
    wait_vm0();
    float r = bsum[0] + bsum[1] + bsum[2] + bsum[3];
#pragma unroll
    for (int x = 0; x < PO; ++x)
#pragma unroll
        for (int y = 0; y < PI; ++y)
#pragma unroll
            for (int c = 0; c < 16; ++c) r += acc[x][y][c];
    if (r == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// ---- host --------------------------------------------------------------------------------------------------------------
#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);  \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

struct Bufs {
    float *src, *out, *big;
    unsigned long long* cyc;
};

template <class K>
void run(const char* name, K kern, int threads, int lds_bytes, int loops, double mfma_cycles_per_wave, int waves_per_simd,
         double flops_per_block, const Bufs& b) {
    const int grid = 256 * 4;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double best_ms = 1e30, eff = 0, effmin = 0;
    const int nw = threads / 64;
    std::vector<unsigned long long> h(grid * nw);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, 0, b.src, b.out, b.cyc, loops, b.big);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best_ms) {
            best_ms = ms;
            CK(hipMemcpy(h.data(), b.cyc, sizeof(unsigned long long) * grid * nw, hipMemcpyDeviceToHost));
            double sum = 0, mx = 0;
            for (int q = 0; q < grid * nw; ++q) {
                sum += (double)h[q];
                if ((double)h[q] > mx) mx = (double)h[q];
            }
            const double mean = sum / (grid * nw);
            eff = mfma_cycles_per_wave * waves_per_simd / mean;
            effmin = mfma_cycles_per_wave * waves_per_simd / mx;
        }
    }
    printf("%-34s eff(mean wave) %.4f  eff(slowest wave) %.4f  kernel %.3f ms  %.1f TFLOP/s\n", name, eff, effmin, best_ms,
           flops_per_block * grid / (best_ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv) {
    Bufs b;
    CK(hipMalloc(&b.src, 4 << 20));
    CK(hipMemset(b.src, 0, 4 << 20));
    CK(hipMalloc(&b.out, 1024 * 512 * 4));
    CK(hipMalloc(&b.cyc, 1024 * 16 * 8));
    CK(hipMalloc(&b.big, (size_t)1024 * 64 * 128 * 1024));  // 8 GB: 1024 workgroups x 64 layers x 128 KB
    const char* only = argc > 1 ? argv[1] : "";
    // F: KC = 8 -> 3 buffers of 32 KB (+ force one workgroup per CU with a 100 KB request); per chunk and wave 128 MFMAs
    {
        const int nch = 512;  // 512 chunks of 8 k-steps = 64 "layers" of 64 k-steps
        const double cyc = (double)nch * 8 * 16 * 32;
        const double fl = (double)nch * 8 * 16 * 8 * 2.0 * 16 * 16 * 4;
        const int L = 100 * 1024;
#define RF(V) \
    if (!*only || strstr(#V, only)) run(#V, kF<V, 8, 8>, 512, L, nch, cyc, 2, fl, b)
        RF(F_NOREAD);
        RF(F_BASE);
        RF(F_NOBAR);
        RF(F_MIDBAR);
        RF(F_ILV);
        RF(F_PRIO);
        RF(F_AACC);
        RF(F_AOPS);
        RF(F_AHEAD2);
        RF(F_DMA);
        RF(F_MIDBAR_DMA);
        RF(F_MIDBAR_DMA_ST);
        RF(F_DMA_ST);
#undef RF
        // chunks of 16 k-steps (two 64 KB buffers): half the barriers
        const int nch16 = 256;
        if (!*only || strstr("F16", only)) {
            run("F_BASE KC=16", kF<F_BASE, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA KC=16", kF<F_DMA, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA_ST KC=16 (product stores)", kF<F_DMA_ST, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA_STC KC=16 (contiguous 1 KB)", kF<F_DMA_STC, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA_STNT KC=16 (non-temporal)", kF<F_DMA_STNT, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA_ST1 KC=16 (one region)", kF<F_DMA_ST1, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA_ST_V4 KC=16 (wait vmcnt(4))", kF<F_DMA_ST_V4, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA_ST_LATE KC=16 (stores late)", kF<F_DMA_ST_LATE, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_ST_NODMA KC=16 (stores, no copy)", kF<F_ST_NODMA, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA_ST_PAIR KC=16 (full lines)", kF<F_DMA_ST_PAIR, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA_ST_SC01 KC=16", kF<F_DMA_ST_SC01, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA_ST_END KC=16 (stores 12..15)", kF<F_DMA_ST_END, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA_ST_MID KC=16 (stores 4..10)", kF<F_DMA_ST_MID, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
            run("F_DMA_ST_LATE_SPREAD KC=16", kF<F_DMA_ST_LATE_SPREAD, 16, 8>, 512, 140 * 1024, nch16, cyc, 2, fl, b);
        }
        // one wave per SIMD (4-wave workgroup, one per CU)
        if (!*only || strstr("F4W", only)) {
            const double fl4 = fl / 2;
            run("F_BASE 4 waves", kF<F_BASE, 8, 4>, 256, L, nch, cyc, 1, fl4, b);
            run("F_NOREAD 4 waves", kF<F_NOREAD, 8, 4>, 256, L, nch, cyc, 1, fl4, b);
            run("F_NOBAR 4 waves", kF<F_NOBAR, 8, 4>, 256, L, nch, cyc, 1, fl4, b);
        }
    }
    {
        const int nst = 192;
        const double cyc = (double)nst * 16 * 8 * 64;
        const double fl = (double)nst * 16 * 8 * 8 * 2.0 * 32 * 32 * 2;
#define RW(V, LDS) \
    if (!*only || strstr(#V, only)) run(#V, kW<V, 16>, 512, LDS, nst, cyc, 2, fl, b)
        const int L2 = 2 * 65536 + 4096, L3 = 160 * 1024;
        RW(W_NOREAD, L2);
        RW(W_BASE, L2);
        RW(W_NOBAR, L2);
        RW(W_WIDE, L2);
        RW(W_WIDE_NOBAR, L2);
        RW(W_PAIR, L2);
        RW(W_BIAS, L2);
        RW(W_WIDE_AOPS, L2);
        RW(W_BASE_DMA, L2);
        RW(W_WIDE_DMA, L2);
        RW(W_WIDE_DMA_HBM, L2);
        RW(W_PAIR_DMA, L2);
#undef RW
        // stages of 8 k-steps (32 KB): three buffers fit, the barrier sits in the middle of a stage
        if (!*only || strstr("W8", only)) {
            run("W_WIDE_DMA KS=8", kW<W_WIDE_DMA, 8>, 512, L3 / 2 + 20480, 2 * nst, cyc, 2, fl, b);
            run("W_WIDE_MID_DMA KS=8", kW<W_WIDE_MID_DMA, 8>, 512, 3 * 32768 + 8192, 2 * nst, cyc, 2, fl, b);
        }
    }
    return 0;
}

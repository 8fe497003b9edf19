// mfma_rate.hip -- GPU-box microbenchmark: sustained issue rate of the fp32 MFMA shapes, one wave per SIMD (the
// occupancy of the MLP kernels), as a function of how many accumulators rotate and of what else shares the loop.
// Build (here, cross-compiling):  hipcc --offload-arch=gfx950 -O3 scripts/mfma_rate.hip -o scripts/mfma_rate
// Output: shader-clock cycles per MFMA (s_memtime) -- ideal 64 for 32x32x2, 32 for 16x16x4 (= 157.3 TFLOP/s).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int NACC, int VALU, int LDS>
__global__ __launch_bounds__(256, 1) void k32(float* out, unsigned long long* cyc, int iters, float a0, float b0) {
    __shared__ float lds[4096];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int c = 0; c < 16; ++c) acc[i][c] = 0.f;
    float a[4] = {a0, a0 + 1, a0 + 2, a0 + 3}, b[4] = {b0, b0 + 1, b0 + 2, b0 + 3};
    float s[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    const float* lp = lds + (threadIdx.x & 63);
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = lp[64 * q + ((it & 7) << 8)];
                b[q] = lp[64 * q + 2048 + ((it & 7) << 8)];
            }
            FENCE();
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m >> 2], b[m & 3], acc[m % NACC], 0, 0, 0);
            if (VALU && (m & 3) == 0) s[m >> 2] += a[m >> 2];
        }
        FENCE();
    }
    unsigned long long t1 = clock64();
    float r = s[0] + s[1] + s[2] + s[3];
    for (int i = 0; i < NACC; ++i)
        for (int c = 0; c < 16; ++c) r += acc[i][c];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256, 1) void k16(float* out, unsigned long long* cyc, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int c = 0; c < 4; ++c) acc[i][c] = 0.f;
    float a[8], b[8];
    for (int q = 0; q < 8; ++q) a[q] = a0 + q, b[q] = b0 + q;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m)
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m >> 3], b[m & 7], acc[m % NACC], 0, 0, 0);
        FENCE();
    }
    unsigned long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < NACC; ++i)
        for (int c = 0; c < 4; ++c) r += acc[i][c];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}


// cost of ONE extra instruction of a given kind issued between MFMAs (single wave per SIMD): N of them per 16 MFMAs
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void kx(float* out, unsigned long long* cyc, int iters, float a0, float b0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int c = 0; c < 16; ++c) acc[i][c] = 0.f;
    float a[4] = {a0, a0 + 1, a0 + 2, a0 + 3}, b[4] = {b0, b0 + 1, b0 + 2, b0 + 3};
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x2 p2[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 15);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float* gp = out + (size_t)blockIdx.x * 4096 + threadIdx.x * 4;
    int soff = 0;
    if (KIND >= 20) {  // stagger the four waves of the workgroup so that their LDS reads do not collide
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        for (int q = 0; q < w; ++q) __builtin_amdgcn_s_sleep(1);  // 64 cycles each
    }
    constexpr int KK = KIND >= 20 ? KIND - 20 : KIND;
    float nx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 nx4[4] = {};
    f32x2 nx2[4] = {};
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const int ro = (it & 7) << 8;
#pragma unroll
        for (int q = 0; q < N; ++q) {
            if (KK == 1) s[q & 7] += a[q & 3];
            if (KK == 2) nx[q & 7] = lds[lane + 64 * q + ro];                                   // ds_read_b32
            if (KK == 3) nx4[q & 3] = *(const f32x4*)(lds + 4 * lane + 256 * q + ro);          // ds_read_b128
            if (KK == 4) nx2[q & 3] = *(const f32x2*)(lds + 2 * lane + 128 * q + ro);          // ds_read_b64
            if (KK == 5) nx[q & 7] = gp[q * 1024 + (it & 1)];                                   // global_load_dword
            if (KK == 10) {                                                                      // ds_read2_b32
                nx[(2 * q) & 7] = lds[lane + 128 * q + ro];
                nx[(2 * q + 1) & 7] = lds[lane + 128 * q + 32 + ro];
            }
            if (KK == 6) *(f32x4*)(gp + q * 1024) = f32x4{a[0], a[1], a[2], a[3]};            // global_store_dwordx4
            if (KK == 7) p2[q & 3] += f32x2{a[q & 3], b[q & 3]};                               // v_pk_add_f32
            if (KK == 8) soff = __builtin_amdgcn_readfirstlane(soff + q + it);                 // v_readfirstlane + salu
            if (KK == 9)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + q * 1024),
                                                 (__attribute__((address_space(3))) void*)(lds + 256 * q), 16, 0, 0);
        }
        FENCE();
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m >> 2], b[m & 3], acc[m], 0, 0, 0);
        FENCE();
        // loaded values are consumed AFTER the MFMAs (the prefetch pattern of the real kernels): one v_add per register
        if (KK == 2 || KK == 5 || KK == 10)
            for (int q = 0; q < (KK == 10 ? 2 * N : N) && q < 8; ++q) s[q] += nx[q];
        if (KK == 3)
            for (int q = 0; q < N && q < 4; ++q) s[q] += nx4[q][0] + nx4[q][3];
        if (KK == 4)
            for (int q = 0; q < N && q < 4; ++q) s[q] += nx2[q][0] + nx2[q][1];
        FENCE();
    }
    unsigned long long t1 = clock64();
    float r = (float)soff;
    for (int i = 0; i < 8; ++i) r += s[i];
    for (int i = 0; i < 4; ++i) r += p2[i][0] + p2[i][1];
    for (int i = 0; i < 16; ++i)
        for (int c = 0; c < 16; ++c) r += acc[i][c];
    if (r == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int N>
static void runx(const char* name) {
    const int grid = 256, iters = 20000;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)grid * 4096 * 4 + (1 << 20));
    hipMemset(out, 0, (size_t)grid * 4096 * 4 + (1 << 20));
    hipMalloc(&cyc, (size_t)grid * 4 * 8);
    hipFuncSetAttribute((const void*)kx<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((kx<KIND, N>), dim3(grid), dim3(256), 65536, 0, out, cyc, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)grid * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double per_iter = sum / h.size() / iters;
    printf("%-44s x%-2d per 16 MFMAs: %8.1f cycles / 16 MFMAs  -> %6.2f cycles per extra instruction\n", name, N, per_iter,
           N ? (per_iter - 1024.0) / N : 0.0);
    hipFree(out);
    hipFree(cyc);
}

// the same probes for (a) the 16x16x4 shape (half the accumulator traffic per FLOP) and (b) two waves per SIMD with
// 8 accumulator tiles each (what the 8-wave weight-gradient kernel does)
template <int KIND, int N, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void ky(float* out, unsigned long long* cyc, int iters, float a0, float b0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool SMALL = KIND >= 100;          // 16x16x4
    constexpr int K = SMALL ? KIND - 100 : KIND;
    constexpr int NT = THREADS == 512 ? 8 : 16;  // 32x32 accumulator tiles per wave
    f32x16 acc[NT];
    for (int i = 0; i < NT; ++i)
        for (int c = 0; c < 16; ++c) acc[i][c] = 0.f;
    float a[4] = {a0, a0 + 1, a0 + 2, a0 + 3}, b[4] = {b0, b0 + 1, b0 + 2, b0 + 3};
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < 8192; i += THREADS) lds[i] = (float)(i & 15);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const int ro = (it & 7) << 8;
#pragma unroll
        for (int q = 0; q < N; ++q) {
            if (K == 1) s[q & 7] += a[q & 3];
            if (K == 2) nx[q & 7] = lds[lane + 64 * q + ro];
        }
        FENCE();
        if (SMALL) {
#pragma unroll
            for (int m = 0; m < 4 * NT; ++m) {
                f32x4 t = {acc[m >> 2][4 * (m & 3)], acc[m >> 2][4 * (m & 3) + 1], acc[m >> 2][4 * (m & 3) + 2], acc[m >> 2][4 * (m & 3) + 3]};
                t = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m & 3], b[(m >> 2) & 3], t, 0, 0, 0);
                acc[m >> 2][4 * (m & 3)] = t[0], acc[m >> 2][4 * (m & 3) + 1] = t[1], acc[m >> 2][4 * (m & 3) + 2] = t[2],
                               acc[m >> 2][4 * (m & 3) + 3] = t[3];
            }
        } else {
#pragma unroll
            for (int m = 0; m < NT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m >> 2], b[m & 3], acc[m], 0, 0, 0);
        }
        FENCE();
        if (K == 2)
            for (int q = 0; q < N && q < 8; ++q) s[q] += nx[q];
        FENCE();
    }
    unsigned long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += s[i];
    for (int i = 0; i < NT; ++i)
        for (int c = 0; c < 16; ++c) r += acc[i][c];
    if (r == 12345.f) out[blockIdx.x * THREADS + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int N, int THREADS>
static void runy(const char* name) {
    const int grid = 256, iters = 20000, waves = THREADS / 64;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)grid * 4096 * 4);
    hipMalloc(&cyc, (size_t)grid * 8 * 8);
    hipMemset(cyc, 0, (size_t)grid * 8 * 8);
    hipFuncSetAttribute((const void*)ky<KIND, N, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((ky<KIND, N, THREADS>), dim3(grid), dim3(THREADS), 65536, 0, out, cyc, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)grid * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    int cnt = 0;
    for (int g = 0; g < grid; ++g)
        for (int w = 0; w < waves; ++w) sum += (double)h[(size_t)g * 8 + w], ++cnt;
    const double per_iter = sum / cnt / iters;  // per wave; the SIMD's MFMA work per iteration is 1024 cycles in every configuration
    printf("%-52s x%-2d: %8.1f cycles per 1024 MFMA-cycles -> %6.2f cycles per extra instruction\n", name, N, per_iter,
           N ? (per_iter - 1024.0) / (N * (waves / 4)) : 0.0);
    hipFree(out);
    hipFree(cyc);
}

// Mock of a 16x16x4-based MLP layer loop: each wave owns 16 samples, per k-step it reads 16 A-operand dwords from LDS
// (4 x ds_read_b128, prefetched one k-step ahead) and issues 16 MFMAs (one per 16-row output tile).  WAVES per workgroup
// = 4, 8 or 12 (1, 2, 3 per SIMD).  Ideal: 16 x 32 = 512 cycles per k-step and wave.
template <int WAVES, int SHAPE32>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void kmock(float* out, unsigned long long* cyc, int iters, float b0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 16384; i += 64 * WAVES) lds[i] = (float)(i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const f32x4* lp = (const f32x4*)lds + lane;
    f32x4 acc[16];
    f32x16 acc32[4];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int c = 0; c < 16; ++c) acc32[i][c] = 0.f;
    float b = b0;
    f32x4 c0[4], c1[4];
    for (int q = 0; q < 4; ++q) c0[q] = lp[64 * q];
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it += 2) {
        const int ro = ((it & 6) + 1) << 8;
        for (int q = 0; q < (SHAPE32 ? 1 : 4); ++q) c1[q] = lp[64 * q + ro];
        FENCE();
        if (SHAPE32) {  // today's shape: 4 MFMAs 32x32x2 (256 cycles) per ds_read_b128
            for (int m = 0; m < 4; ++m) acc32[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0[0][m], b, acc32[m], 0, 0, 0);
        } else {
            for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[m >> 2][m & 3], b, acc[m], 0, 0, 0);
        }
        FENCE();
        for (int q = 0; q < (SHAPE32 ? 1 : 4); ++q) c0[q] = lp[64 * q + ro + 256];
        FENCE();
        if (SHAPE32) {
            for (int m = 0; m < 4; ++m) acc32[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[0][m], b, acc32[m], 0, 0, 0);
        } else {
            for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[m >> 2][m & 3], b, acc[m], 0, 0, 0);
        }
        FENCE();
    }
    unsigned long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; ++i)
        for (int c = 0; c < 16; ++c) r += acc32[i][c];
    if (r == 12345.f) out[threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int WAVES, int SHAPE32>
static void runmock(const char* name) {
    const int grid = 256, iters = 40000;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1 << 20);
    hipMalloc(&cyc, (size_t)grid * 16 * 8);
    hipFuncSetAttribute((const void*)kmock<WAVES, SHAPE32>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kmock<WAVES, SHAPE32>), dim3(grid), dim3(64 * WAVES), 65536, 0, out, cyc, iters / 8, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kmock<WAVES, SHAPE32>), dim3(grid), dim3(64 * WAVES), 65536, 0, out, cyc, iters, 2.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_flop = SHAPE32 ? 4.0 * 4096.0 : 16.0 * 2048.0;
    const double tf = (double)grid * WAVES * iters * mfma_flop / (ms * 1e-3) / 1e12;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s = %5.1f %% of 157.3\n", name, ms, tf, tf / 1.573);
    hipFree(out);
    hipFree(cyc);
}

// the same mock with the A operands prefetched TWO k-steps ahead (3-deep register ring)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void kmock3(float* out, unsigned long long* cyc, int iters, float b0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 16384; i += 64 * WAVES) lds[i] = (float)(i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const f32x4* lp = (const f32x4*)lds + lane;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float b = b0;
    f32x4 c[3][4];
    for (int q = 0; q < 4; ++q) c[0][q] = lp[64 * q], c[1][q] = lp[64 * q + 256];
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int ro = (((it + u) & 7) + 2) << 8;
            for (int q = 0; q < 4; ++q) c[(u + 2) % 3][q] = lp[64 * q + ro];
            FENCE();
            for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(c[u][m >> 2][m & 3], b, acc[m], 0, 0, 0);
            FENCE();
        }
    }
    unsigned long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 12345.f) out[threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int WAVES>
static void runmock3(const char* name) {
    const int grid = 256, iters = 39999;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1 << 20);
    hipMalloc(&cyc, (size_t)grid * 16 * 8);
    hipFuncSetAttribute((const void*)kmock3<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kmock3<WAVES>), dim3(grid), dim3(64 * WAVES), 65536, 0, out, cyc, iters / 9, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kmock3<WAVES>), dim3(grid), dim3(64 * WAVES), 65536, 0, out, cyc, iters, 2.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)grid * WAVES * iters * 16.0 * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s = %5.1f %% of 157.3\n", name, ms, tf, tf / 1.573);
    hipFree(out);
    hipFree(cyc);
}

// Mock of the mlp16 layer pipeline, adding the real kernel's ingredients one at a time (FLAGS bits):
//   1: s_barrier + s_waitcnt vmcnt(0) every 8 k-steps (workgroups of 4 waves, two workgroups per CU)
//   2: 8 LDS-DMA pieces (1 KiB each) per wave and chunk, issued right after the barrier
//   4: layer epilogue every 64 k-steps: 64 x (v_max_i32 + v_min_u32 + v_lshl_or) VALU
//   8: 16 global_store_dwordx4 per layer, two per chunk
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void kpipe(float* out, const float* src, unsigned long long* cyc, int chunks, float b0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)(i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4* lp = (const f32x4*)lds + lane;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float act[64];
    for (int i = 0; i < 64; ++i) act[i] = b0 + i;
    unsigned bits[2] = {0, 0};
    float* gout = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 64;
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 rs;
    {
        const unsigned long long bb = (unsigned long long)src;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)bb);
        rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((bb >> 32) & 0xFFFFu));
        rs[2] = 1 << 22;
        rs[3] = 0x00020000;
    }
    unsigned long long t0 = clock64();
    for (int c = 0; c < chunks; ++c) {
        if (FLAGS & 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (FLAGS & 2) {
            for (int q = 0; q < 8; ++q)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(
                                 __builtin_amdgcn_readfirstlane((int)(ldsb + 32768 + ((c & 1) << 15) + (wave * 8 + q) * 1024))),
                             "v"(lane * 16), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(((c & 63) * 32 + wave * 8 + q) * 1024))
                             : "memory");
        }
        if (FLAGS & 8) {
            *(f32x4*)(gout + (c & 7) * 8) = f32x4{act[(c & 7) * 8], act[(c & 7) * 8 + 1], act[(c & 7) * 8 + 2], act[(c & 7) * 8 + 3]};
            *(f32x4*)(gout + (c & 7) * 8 + 4) = f32x4{act[(c & 7) * 8 + 4], act[(c & 7) * 8 + 5], act[(c & 7) * 8 + 6], act[(c & 7) * 8 + 7]};
        }
        f32x4 a[2][4];
        for (int q = 0; q < 4; ++q) a[0][q] = lp[64 * q];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 1 < 8)
                for (int q = 0; q < 4; ++q) a[(ks + 1) & 1][q] = lp[64 * q + ((ks + 1) << 8)];
            FENCE();
            const float b = act[(ks * 8) & 63];
            for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks & 1][m >> 2][m & 3], b, acc[m], 0, 0, 0);
            FENCE();
        }
        if ((FLAGS & 4) && (c & 7) == 7) {
            bits[0] = bits[1] = 0;
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                int iv = __float_as_int(acc[r >> 2][r & 3]);
                iv = iv > 0 ? iv : 0;
                const unsigned u = (unsigned)iv;
                bits[r >> 5] |= (u < 1u ? u : 1u) << (r & 31);
                act[r] = __int_as_float(iv) + 1.0f;
                acc[r >> 2][r & 3] = 0.0f;
            }
        }
    }
    unsigned long long t1 = clock64();
    float r = (float)(bits[0] + bits[1]);
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 64; ++i) r += act[i];
    if (r == 12345.f) out[threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int FLAGS>
static void runpipe(const char* name) {
    const int grid = 512 * 4, chunks = 75 * 8;  // 2 resident workgroups per CU x 4 rounds; 75 "layers" of 8 chunks
    float *out, *src;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)grid * 256 * 64 * 4);
    hipMalloc(&src, 8 << 20);
    hipMemset(src, 0, 8 << 20);
    hipMalloc(&cyc, (size_t)grid * 4 * 8);
    hipFuncSetAttribute((const void*)kpipe<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 69632);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kpipe<FLAGS>), dim3(grid), dim3(256), 69632, 0, out, src, cyc, chunks / 10, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kpipe<FLAGS>), dim3(grid), dim3(256), 69632, 0, out, src, cyc, chunks, 2.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)grid * 4 * chunks * 8 * 16 * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-72s %8.3f ms  %7.1f TFLOP/s = %5.1f %% of 157.3\n", name, ms, tf, tf / 1.573);
    hipFree(out);
    hipFree(src);
    hipFree(cyc);
}

// The same pipeline as ONE 8-wave workgroup per CU whose two halves (waves 0-3 / 4-7) run half a chunk out of phase:
// every wave meets a barrier each half chunk (4 k-steps); the copy of a chunk (4 pieces per wave: half the DMA traffic
// of two independent workgroups) is issued after every odd barrier into a 3-deep ring of LDS buffers.
template <int FLAGS>
__global__ __launch_bounds__(512, 2) void kpipe8(float* out, const float* src, unsigned long long* cyc, int chunks, float b0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 24576; i += 512) lds[i] = (float)(i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, phase = wave >> 2;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float act[64];
    for (int i = 0; i < 64; ++i) act[i] = b0 + i;
    unsigned bits[2] = {0, 0};
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 rs;
    {
        const unsigned long long bb = (unsigned long long)src;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)bb);
        rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((bb >> 32) & 0xFFFFu));
        rs[2] = 1 << 22;
        rs[3] = 0x00020000;
    }
    unsigned long long t0 = clock64();
    const int nslots = 2 * chunks + 1;
    for (int n = 0; n < nslots; ++n) {
        if (FLAGS & 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if ((FLAGS & 2) && (n & 1)) {
            const int cn = (n + 3) >> 1;
            for (int q = 0; q < 4; ++q)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(
                                 __builtin_amdgcn_readfirstlane((int)(ldsb + (cn % 3) * 32768 + (wave * 4 + q) * 1024))),
                             "v"(lane * 16), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(((cn & 63) * 32 + wave * 4 + q) * 1024))
                             : "memory");
        }
        const int hc = n - phase;  // this wave's half chunk
        if (hc < 0 || hc >= 2 * chunks) continue;
        const f32x4* lp = (const f32x4*)(lds + ((hc >> 1) % 3) * 8192 + (hc & 1) * 4096) + lane;
        f32x4 a[2][4];
        for (int q = 0; q < 4; ++q) a[0][q] = lp[64 * q];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4)
                for (int q = 0; q < 4; ++q) a[(ks + 1) & 1][q] = lp[64 * q + ((ks + 1) << 8)];
            FENCE();
            const float b = act[(ks * 8 + (hc & 1) * 32) & 63];
            for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks & 1][m >> 2][m & 3], b, acc[m], 0, 0, 0);
            FENCE();
        }
        if ((FLAGS & 4) && (hc & 15) == 15) {
            bits[0] = bits[1] = 0;
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                int iv = __float_as_int(acc[r >> 2][r & 3]);
                iv = iv > 0 ? iv : 0;
                const unsigned u = (unsigned)iv;
                bits[r >> 5] |= (u < 1u ? u : 1u) << (r & 31);
                act[r] = __int_as_float(iv) + 1.0f;
                acc[r >> 2][r & 3] = 0.0f;
            }
        }
    }
    unsigned long long t1 = clock64();
    float r = (float)(bits[0] + bits[1]);
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 64; ++i) r += act[i];
    if (r == 12345.f) out[threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int FLAGS>
static void runpipe8(const char* name) {
    const int grid = 256 * 4, chunks = 75 * 8;
    float *out, *src;
    unsigned long long* cyc;
    hipMalloc(&out, 1 << 20);
    hipMalloc(&src, 8 << 20);
    hipMemset(src, 0, 8 << 20);
    hipMalloc(&cyc, (size_t)grid * 8 * 8);
    hipFuncSetAttribute((const void*)kpipe8<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kpipe8<FLAGS>), dim3(grid), dim3(512), 98304, 0, out, src, cyc, chunks / 10, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kpipe8<FLAGS>), dim3(grid), dim3(512), 98304, 0, out, src, cyc, chunks, 2.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)grid * 8 * chunks * 8 * 16 * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-72s %8.3f ms  %7.1f TFLOP/s = %5.1f %% of 157.3\n", name, ms, tf, tf / 1.573);
    hipFree(out);
    hipFree(src);
    hipFree(cyc);
}

template <class K>
static void run(const char* name, K kern, int grid, int iters, int mfma_per_iter, double flop_per_mfma) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipMalloc(&cyc, (size_t)grid * 4 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, cyc, iters / 8, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f, 2.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (auto v : h) {
        sum += (double)v;
        if ((double)v > mx) mx = (double)v;
    }
    const double n = (double)iters * mfma_per_iter;
    const double tf = (double)grid * 4 * n * flop_per_mfma / (ms * 1e-3) / 1e12;
    printf("%-34s grid %5d  cycles/MFMA mean %7.2f max %7.2f   kernel %8.3f ms  %7.1f TFLOP/s  clock %.2f GHz\n", name, grid,
           sum / h.size() / n, mx / n, ms, tf, (sum / h.size()) / (ms * 1e6) * ((grid + 255) / 256 > 1 ? (grid / 256.0) : 1.0));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    const int it = 20000;
    for (int grid : {256, 1024}) {
        run("32x32x2 16 accumulators", k32<16, 0, 0>, grid, it, 16, 4096.0);
        run("32x32x2  8 accumulators", k32<8, 0, 0>, grid, it, 16, 4096.0);
        run("32x32x2  4 accumulators", k32<4, 0, 0>, grid, it, 16, 4096.0);
        run("32x32x2  2 accumulators", k32<2, 0, 0>, grid, it, 16, 4096.0);
        run("32x32x2  1 accumulator (chain)", k32<1, 0, 0>, grid, it, 16, 4096.0);
        run("32x32x2 16 acc + 4 v_add / 16", k32<16, 1, 0>, grid, it, 16, 4096.0);
        run("32x32x2 16 acc + 8 ds_read / 16", k32<16, 0, 1>, grid, it, 16, 4096.0);
        run("32x32x2 16 acc + v_add + ds_read", k32<16, 1, 1>, grid, it, 16, 4096.0);
        run("16x16x4 64 accumulators", k16<64>, grid, it / 4, 64, 2048.0);
        run("16x16x4 16 accumulators", k16<16>, grid, it / 4, 64, 2048.0);
        run("16x16x4  4 accumulators", k16<4>, grid, it / 4, 64, 2048.0);
        run("16x16x4  1 accumulator (chain)", k16<1>, grid, it / 4, 64, 2048.0);
    }
    runx<0, 0>("(nothing)");
    runx<1, 4>("v_add_f32");
    runx<1, 8>("v_add_f32");
    runx<7, 4>("v_pk_add_f32");
    runx<2, 8>("ds_read_b32 (prefetched) + 8 v_add");
    runx<10, 4>("ds_read2_b32 (prefetched) + 8 v_add");
    runx<4, 4>("ds_read_b64 (prefetched) + 8 v_add");
    runx<3, 2>("ds_read_b128 (prefetched) + 4 v_add");
    runx<3, 4>("ds_read_b128 (prefetched) + 8 v_add");
    runx<5, 4>("global_load_dword (prefetched) + 4 v_add");
    runx<6, 2>("global_store_dwordx4");
    runx<9, 2>("global_load_lds_dwordx4");
    runx<9, 4>("global_load_lds_dwordx4");
    runx<8, 8>("v_readfirstlane + s_add");
    runmock<4, 1>("mock layer loop: 32x32x2, 1 wave/SIMD, 1 ds_read_b128 / 4 MFMA (today)");
    runmock<8, 1>("mock layer loop: 32x32x2, 2 waves/SIMD (if registers allowed)");
    runmock<4, 0>("mock layer loop: 16x16x4, 1 wave/SIMD, 4 ds_read_b128 / 16 MFMA");
    runmock<8, 0>("mock layer loop: 16x16x4, 2 waves/SIMD");
    runmock<12, 0>("mock layer loop: 16x16x4, 3 waves/SIMD");
    runmock3<4>("mock layer loop: 16x16x4, 1 wave/SIMD, operands 2 k-steps ahead");
    runmock3<8>("mock layer loop: 16x16x4, 2 waves/SIMD, operands 2 k-steps ahead");
    runpipe<0>("mlp16 pipeline mock: MFMA + operand reads only (2 WGs of 4 waves per CU)");
    runpipe<1>("  + barrier and vmcnt(0) per chunk");
    runpipe<3>("  + barrier + 8 LDS-DMA pieces per wave and chunk");
    runpipe<7>("  + barrier + DMA + layer epilogue VALU");
    runpipe<15>("  + barrier + DMA + epilogue + stash stores (= training forward)");
    runpipe<11>("  + barrier + DMA + stash stores, no epilogue");
    runpipe8<0>("8-wave WG, halves half a chunk out of phase: MFMA + operand reads only");
    runpipe8<1>("  + barrier per half chunk");
    runpipe8<3>("  + barrier + 4 LDS-DMA pieces per wave and chunk (3-deep ring)");
    runpipe8<7>("  + barrier + DMA + layer epilogue VALU (= inference forward)");
    runx<23, 4>("STAGGERED ds_read_b128 (prefetched) + 8 v_add");
    runx<23, 2>("STAGGERED ds_read_b128 (prefetched) + 4 v_add");
    runx<22, 8>("STAGGERED ds_read_b32 (prefetched) + 8 v_add");
    runx<21, 8>("STAGGERED v_add_f32");
    runy<0, 0, 256>("1 wave/SIMD 32x32x2 (nothing)");
    runy<100, 0, 256>("1 wave/SIMD 16x16x4 (nothing)");
    runy<101, 8, 256>("1 wave/SIMD 16x16x4 + v_add_f32");
    runy<102, 8, 256>("1 wave/SIMD 16x16x4 + ds_read_b32 (prefetched) + v_add");
    runy<0, 0, 512>("2 waves/SIMD 32x32x2 (nothing)");
    runy<1, 4, 512>("2 waves/SIMD 32x32x2 + v_add_f32 (per wave)");
    runy<1, 8, 512>("2 waves/SIMD 32x32x2 + v_add_f32 (per wave)");
    runy<2, 4, 512>("2 waves/SIMD 32x32x2 + ds_read_b32 + v_add (per wave)");
    runy<2, 8, 512>("2 waves/SIMD 32x32x2 + ds_read_b32 + v_add (per wave)");
    runy<100, 0, 512>("2 waves/SIMD 16x16x4 (nothing)");
    runy<102, 8, 512>("2 waves/SIMD 16x16x4 + ds_read_b32 + v_add (per wave)");
    return 0;
}

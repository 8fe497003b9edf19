// f16w_loop_mock.hip -- GPU-box microbenchmark (NOT part of the library): what does the multiply loop of the two-waves-per-SIMD fp16
// kernels (csrc/mlp_f16w.hip) sustain on MI355X as a function of how many accumulator tiles a group of blocks interleaves?
//
// The product kernel multiplies a layer as 128 blocks (k-block, tile); per block two ds_read_b128 (the high and the low weight pieces)
// and three v_mfma_f32_16x16x32_f16 into the tile's accumulator (wl.bh, wh.bl, wh.bh).  It takes the blocks in PAIRS, the six MFMAs
// alternating between the pair's two tiles, so every MFMA reads the accumulator written ONE MFMA (16 issue cycles) earlier; the PMC
// pass of the bench line shows SQ_WAIT_INST_ANY at 0.40 of the wave cycles (profiles/r04_pmc_summary_f16x3_train.txt).  This mock runs
// that loop -- 8-wave workgroups, two waves per SIMD, one workgroup per CU, weights resident in LDS (no stream, no stash, no
// conversions: the diagnostic build `noall` of profiles/r04_f16w_ab.txt is the product's counterpart, 0.54 of the MFMA peak) -- with
// groups of G = 1, 2, 4, 8 blocks: the group's 3 G MFMAs go product by product over its G tiles (dependency distance G), the 2 G reads
// of the NEXT group in front of them.  Also: without the LDS reads (2 G blocks of random pieces held in registers: what the matrix pipe
// sustains on such operands, and what the dependency pattern alone costs) and with a barrier
// every 32 blocks (the product's chunk barrier).  Output: fraction of the 2.5 PF dense fp16 peak.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/f16w_loop_mock.hip -o scripts/f16w_loop_mock
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KB = 8, TW = 16, NBLK = KB * TW, CB = 32;  // a 256 x 256 layer: 8 k-blocks of 32, 16 tiles of 16; 32 blocks per chunk

// EPI: after every layer the 16 accumulator tiles become the next layer's operand pieces as in the product (per-sample maximum over
// the four lane groups, one multiply, ReLU, hi = f16(v), lo = f16(v - hi)); the accumulators restart at zero
// SHARE: every block read from LDS feeds SHARE accumulator tiles (with different register operands): 1 / SHARE of the operand reads per
// MFMA -- the inner loop a weight-stationary kernel would have (weight slices in registers, the activations as the LDS operand)
template <int G, bool READS, bool BARRIER, bool EPI = false, int SHARE = 1>
__global__ __launch_bounds__(512, 2) void k_loop(const char* __restrict__ wimg, float* __restrict__ out, int layers) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int p = threadIdx.x; p < CB * 2048 / 16; p += 512) ((float4*)lds)[p] = ((const float4*)wimg)[p];
    __syncthreads();
    f32x4 acc[TW];
    f16x8 bh[KB], bl[KB];
    for (int k = 0; k < KB; ++k)
        for (int j = 0; j < 8; ++j) {
            bh[k][j] = (_Float16)(37.0f * (float)(((lane * 7 + k * 13 + j * 29) & 255) - 100));
            bl[k][j] = (_Float16)(0.011f * (float)(((lane * 5 + 3 * k + j * 17) & 127) - 60));
        }
#pragma unroll
    for (int t = 0; t < TW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* const wb = lds + lane * 16;
    f16x8 wh[2 * G], wl[2 * G];
    if (!READS) {  // (random weight pieces held in registers: what the matrix pipe sustains on such operands without the LDS stream)
#pragma unroll
        for (int q = 0; q < 2 * G; ++q) {
            wh[q] = *(const f16x8*)(wb + (2 * q) * 1024);
            wl[q] = *(const f16x8*)(wb + (2 * q + 1) * 1024);
        }
    }
    for (int L = 0; L < layers; ++L) {
#pragma unroll
        for (int c = 0; c < NBLK / CB; ++c) {
            if (BARRIER) __syncthreads();
            auto load = [&](int i) {  // block i of the chunk -> buffer i % (2 G)
                if (READS) {
                    wh[i % (2 * G)] = *(const f16x8*)(wb + (2 * i) * 1024);
                    wl[i % (2 * G)] = *(const f16x8*)(wb + (2 * i + 1) * 1024);
                }
            };
#pragma unroll
            for (int j = 0; j < G; ++j) load(j);
#pragma unroll
            for (int i0 = 0; i0 < CB / SHARE; i0 += G) {
#pragma unroll
                for (int j = 0; j < G; ++j)
                    if (i0 + G + j < CB / SHARE) load(i0 + G + j);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int j = 0; j < G; ++j)
#pragma unroll
                        for (int sh = 0; sh < SHARE; ++sh) {
                            const int gi = c * CB + (i0 + j) * SHARE + sh, kb = (gi / TW + sh) % KB, t = gi % TW;
                            const f16x8 a = pr == 0 ? wl[(i0 + j) % (2 * G)] : wh[(i0 + j) % (2 * G)];
                            const f16x8 b = pr == 1 ? bl[kb] : bh[kb];
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
                        }
            }
        }
        if (EPI) {
            float m = 0.f;
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) m = fmaxf(m, acc[t][q]);
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            unsigned mb;
            __builtin_memcpy(&mb, &m, 4);
            const unsigned sb = ((mb >> 23) & 255u) == 0u ? (127u << 23) : ((unsigned)(127 + 13 + 127 - (int)((mb >> 23) & 255u)) << 23);
            float mul;
            __builtin_memcpy(&mul, &sb, 4);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = acc[2 * kb + (e >> 2)][e & 3] * mul;
                    v = v > 0.f ? v : 0.f;
                    const _Float16 hi = (_Float16)v;
                    bh[kb][e] = hi;
                    bl[kb][e] = (_Float16)(v - (float)hi);
                }
#pragma unroll
            for (int t = 0; t < TW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (!READS) {  // (keep the weight registers alive and varying without touching LDS)
#pragma unroll
            for (int q = 0; q < 2 * G; ++q) wh[q][0] += (_Float16)1.0f, wl[q][1] += (_Float16)1.0f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TW; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (EPI)  // (the accumulators were reset: what the last conversion left)
        for (int k = 0; k < KB; ++k) s += (float)bh[k][0] + (float)bl[k][7];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

template <int G, bool READS, bool BARRIER, bool EPI = false, int SHARE = 1>
void run(const char* what, const char* wimg, float* out, int grid, int layers) {
    hipFuncSetAttribute((const void*)k_loop<G, READS, BARRIER, EPI, SHARE>, hipFuncAttributeMaxDynamicSharedMemorySize, CB * 2048);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_loop<G, READS, BARRIER, EPI, SHARE>), dim3(grid), dim3(512), CB * 2048, 0, wimg, out, layers);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 3.0 * 16 * 16 * 32 * NBLK * (double)layers * 8.0 * grid;  // three MFMAs per block, 8 waves
    printf("%-72s %8.3f ms  %7.1f TF  %.3f of 2.5 PF\n", what, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15);
}

int main() {
    const int grid = 256, layers = 400;
    char* wimg;
    float* out;
    hipMalloc((void**)&wimg, CB * 2048);
    {  // weight pieces like a trained layer's: high pieces ~ N(0, 1/16) * 2^8, low pieces 2^-11 of that (random bits: realistic toggling --
       // constant data runs the matrix pipe at a higher clock, MI355X_MICROARCH.md "DVFS give-back")
        static _Float16 h[CB * 1024];
        unsigned st = 12345u;
        for (int b = 0; b < 2 * CB; ++b)
            for (int i = 0; i < 512; ++i) {
                st = st * 1664525u + 1013904223u;
                const float u = (float)((st >> 8) & 0xffff) / 65536.0f - 0.5f;
                h[b * 512 + i] = (_Float16)((b & 1) ? u * 0.03f : u * 64.0f);
            }
        hipMemcpy(wimg, h, CB * 2048, hipMemcpyHostToDevice);
    }
    hipMalloc((void**)&out, (size_t)grid * 512 * 4);
    printf("# %d layers of 256 x 256 on fp16 pieces (3 MFMAs per block), 16 samples per wave, two waves per SIMD, %d workgroups of 8 waves\n", layers, grid);
    run<1, true, false>("G = 1 (three dependent MFMAs back to back), LDS reads", wimg, out, grid, layers);
    run<2, true, false>("G = 2 (the product's pairs), LDS reads", wimg, out, grid, layers);
    run<4, true, false>("G = 4, LDS reads", wimg, out, grid, layers);
    run<8, true, false>("G = 8, LDS reads", wimg, out, grid, layers);
    run<2, true, true>("G = 2, LDS reads, barrier per 32 blocks", wimg, out, grid, layers);
    run<4, true, true>("G = 4, LDS reads, barrier per 32 blocks", wimg, out, grid, layers);
    run<2, true, true, true>("G = 2, LDS reads, barrier, per-layer conversion (maximum, ReLU, hi / lo)", wimg, out, grid, layers);
    run<4, true, true, true>("G = 4, LDS reads, barrier, per-layer conversion", wimg, out, grid, layers);
    run<2, true, true, false, 2>("G = 2, every LDS block feeds 2 tiles (half the reads per MFMA), barrier", wimg, out, grid, layers);
    run<4, true, true, false, 2>("G = 4, every LDS block feeds 2 tiles, barrier", wimg, out, grid, layers);
    run<2, true, true, false, 4>("G = 2, every LDS block feeds 4 tiles (a quarter of the reads), barrier", wimg, out, grid, layers);
    run<1, false, false>("G = 1, no LDS reads", wimg, out, grid, layers);
    run<2, false, false>("G = 2, no LDS reads", wimg, out, grid, layers);
    run<4, false, false>("G = 4, no LDS reads", wimg, out, grid, layers);
    return 0;
}

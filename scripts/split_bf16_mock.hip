// split_bf16_mock.hip -- GPU-box microbenchmark for VERDICT r2 item 6: what would the inner loop of a split-bf16
// (bf16x3 / bf16x6) version of the register-chained MLP layer sustain on MI355X?  NOT part of the library.
//
// The fp32 kernels (mlp16.hip) multiply on v_mfma_f32_16x16x4_f32 (64 FLOP/clk/SIMD = 157 TF); the only faster exact-
// enough route is v_mfma_f32_32x32x16_bf16 (1024 FLOP/clk/SIMD = 2.5 PF) on operands split into bf16 pieces:
// x.w ~ xh wh + xl wh + xh wl (bf16x3) [+ xl wl + xh w2 + x2 wh (bf16x6)], fp32 accumulation.  The same register chaining
// works: a 32-row output tile of a layer (16 accumulator registers per lane) is, after conversion, the B operand of two
// 16-deep k-blocks of the next layer when the packed weights use the matching feature permutation.  A wave then owns 32
// samples; one wave per SIMD (128 accumulator + 128 operand registers for a 256-wide layer).
//
// This mock runs that loop for a chain of 256 x 256 layers -- per (k-block, output tile): two ds_read_b128 (the hi and lo
// weight pieces) feeding NT MFMAs -- with the weights resident in LDS (variant 0) or streamed L2 -> LDS by LDS-DMA, double
// buffered, one barrier per chunk as in mlp16.hip (variant 1), with or without the per-layer conversion of the
// accumulators into bf16 hi/lo operand pieces.  Output: effective fp32-equivalent TFLOP/s (2 * 256 * 256 per sample and
// layer) and the fraction of the variant's own roofline (2.5 PF / NT).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/split_bf16_mock.hip -o scripts/split_bf16_mock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KB = 16, T = 8;           // 256-wide layer: 16 k-blocks of 16 features, 8 output tiles of 32 rows
constexpr int CH = 4;                   // k-blocks per LDS chunk: 4 x 8 tiles x (hi, lo) x 1 KiB = 64 KiB per buffer
constexpr int CHUNK_BYTES = CH * T * 2 * 1024;

__device__ __forceinline__ void dma16(const char* g, unsigned lds_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(size_t)lds_addr, 16, 0, 0);
}

// NT: MFMAs per (k-block, tile): 3 = bf16x3, 6 = bf16x6.  STREAM: weights streamed through LDS.  EPI: per-layer epilogue.
template <int NT, bool STREAM, bool EPI>
__global__ __launch_bounds__(256, 1) void k_chain(const char* __restrict__ wimg, float* __restrict__ out, int layers) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[T];
    bf16x8 bh[KB], bl[KB];
    for (int k = 0; k < KB; ++k)
        for (int j = 0; j < 8; ++j) {
            bh[k][j] = (__bf16)(0.01f * (float)((lane + k + j) & 15));
            bl[k][j] = (__bf16)(0.0001f * (float)((lane + 3 * k + j) & 7));
        }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
    // chunk 0 of layer 0
    for (int p = wave; p < CHUNK_BYTES / 1024; p += 4) {
        if (STREAM) dma16(wimg + (size_t)p * 1024 + lane * 16, lds0 + p * 1024);
        else *(float4*)(lds + p * 1024 + lane * 16) = *(const float4*)(wimg + (size_t)p * 1024 + lane * 16);
    }
    int buf = 0;
    for (int L = 0; L < layers; ++L) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[t][c] = 0.0f;
#pragma unroll
        for (int c = 0; c < KB / CH; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (STREAM) {  // next chunk -> the other buffer, spread over the waves (one 1-KiB piece per instruction)
                const int nxt = (L * (KB / CH) + c + 1) % (4 * (KB / CH));
                for (int p = wave; p < CHUNK_BYTES / 1024; p += 4)
                    dma16(wimg + (size_t)nxt * CHUNK_BYTES + (size_t)p * 1024 + lane * 16, lds0 + (buf ^ 1) * CHUNK_BYTES + p * 1024);
            }
            const char* base = lds + (STREAM ? buf * CHUNK_BYTES : 0) + lane * 16;
#pragma unroll
            for (int kk = 0; kk < CH; ++kk) {
                const int kb = c * CH + kk;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const bf16x8 wh = *(const bf16x8*)(base + ((kk * T + t) * 2 + 0) * 1024);
                    const bf16x8 wl = *(const bf16x8*)(base + ((kk * T + t) * 2 + 1) * 1024);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bh[kb], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bl[kb], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, bh[kb], acc[t], 0, 0, 0);
                    if (NT == 6) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, bl[kb], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bl[(kb + 1) & (KB - 1)], acc[t], 0, 0, 0);  // (third pieces)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, bh[(kb + 1) & (KB - 1)], acc[t], 0, 0, 0);
                    }
                }
            }
            if (STREAM) buf ^= 1;
        }
        if (EPI) {  // ReLU, then the accumulators become the next layer's operand pieces: hi = bf16(v), lo = bf16(v - hi)
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float v = acc[t][half * 8 + j];
                        v = v > 0.0f ? v : 0.0f;
                        const __bf16 h = (__bf16)v;
                        bh[2 * t + half][j] = h;
                        bl[2 * t + half][j] = (__bf16)(v - (float)h);
                    }
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) bh[2 * t][0] = (__bf16)acc[t][0];  // (keeps the chain dependent)
        }
    }
    float r = 0.f;
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < 16; ++c) r += acc[t][c];
    for (int k = 0; k < KB; ++k) r += (float)bh[k][0] + (float)bl[k][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NT, bool STREAM, bool EPI>
static void run(const char* name, const char* wimg, float* out, int grid, int layers) {
    const int lds = STREAM ? 2 * CHUNK_BYTES : CHUNK_BYTES;
    hipFuncSetAttribute((const void*)k_chain<NT, STREAM, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k_chain<NT, STREAM, EPI><<<grid, 256, lds>>>(wimg, out, layers);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) k_chain<NT, STREAM, EPI><<<grid, 256, lds>>>(wimg, out, layers);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    ms /= 3.f;
    const double samples = (double)grid * 4 * 32;
    const double tf = samples * layers * 2.0 * 256.0 * 256.0 / (ms * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %8.1f effective fp32-equivalent TFLOP/s  = %.3f of its own roofline (2500 / %d TF), %.2f x the fp32 "
           "MFMA peak (157.3 TF)\n", name, ms, tf, tf / (2500.0 / NT), NT, tf / 157.3);
}

int main() {
    const int grid = 256 * 8, layers = 64;
    char* wimg;
    float* out;
    const size_t wbytes = (size_t)4 * (KB / CH) * CHUNK_BYTES;  // four layers' worth of weight images (1 MiB: L2-resident)
    hipMalloc((void**)&wimg, wbytes);
    hipMemset(wimg, 0x3c, wbytes);
    hipMalloc((void**)&out, (size_t)grid * 256 * 4);
    printf("# chain of %d layers 256 x 256, 32 samples per wave, one wave per SIMD, %d workgroups of 4 waves\n", layers, grid);
    run<3, false, false>("bf16x3, weights resident in LDS, no epilogue", wimg, out, grid, layers);
    run<3, false, true>("bf16x3, weights resident in LDS, epilogue", wimg, out, grid, layers);
    run<3, true, false>("bf16x3, weights streamed L2->LDS, no epilogue", wimg, out, grid, layers);
    run<3, true, true>("bf16x3, weights streamed L2->LDS, epilogue", wimg, out, grid, layers);
    run<6, true, true>("bf16x6, weights streamed L2->LDS, epilogue", wimg, out, grid, layers);
    run<6, false, false>("bf16x6, weights resident in LDS, no epilogue", wimg, out, grid, layers);
    return 0;
}

// ws_loop_mock.hip -- GPU-box microbenchmark (NOT part of the library): what would a WEIGHT-STATIONARY layer loop of the fp16-piece
// forward sustain on MI355X?  ("rebuild the f16x3 forward / data-gradient inner loop weight-stationary; prototype
// first".)  This is the prototype of the LOOP STRUCTURE with everything that structure forces on a kernel, before any kernel is
// rewritten -- the way scripts/f16w_loop_mock.hip priced the product's structure in round 4 (0.60-0.65 of 2.5 PF on random pieces).
//
// Product (csrc/mlp_f16w.hip): a wave owns 16 samples, its activations live in registers as the B operand, the weights stream
// L2 -> LDS and every wave reads ALL of them: one 1-KiB ds_read_b128 per MFMA and a half, 2 MiB of LDS reads per layer and 128-sample
// group (171 of the LDS's 256 B/clk at the MFMA peak).  Waves are independent between chunk barriers, so one wave of a SIMD converts
// its accumulators while the other multiplies.
// Weight-stationary: wave w holds the weights of output tiles 2w, 2w + 1 (all 8 k-blocks, high and low pieces: 128 VGPRs) for a whole
// layer, loaded straight from L2 into registers (a k-block's registers are re-loaded for the NEXT layer as soon as its last MFMA has
// issued); the activations of the workgroup's 128 samples live in LDS as operand pieces ([sample group][k-block][hi | lo] x 1 KiB =
// 128 KiB, updated in place) and every wave reads all of them: one 2-KiB pair of reads per SIX MFMAs, 1 MiB per layer and group.
// What the structure forces: a layer's outputs are spread over the eight waves (32 units each), so
//   * the per-sample maximum of the block floating point (mlp_f16w.hip gemm_w EPI) needs an exchange through LDS, and
//   * nobody may overwrite the activations before everybody has read them: two barriers per layer, all eight waves in lockstep --
//     the conversions (maximum, multiply, ReLU, hi = f16(v), lo = f16(v - hi): ~10 VALU per value) are no longer hidden behind the
//     SIMD's other wave.  MODE 2 splits the 128 samples into two halves and interleaves the conversions of one half with the MFMAs
//     of the other inside each wave's own instruction stream (the only overlap the structure leaves).
// MODE 0: the multiply loop + register weight stream + the two barriers, no conversions (the counterpart of f16w_loop_mock's
//         "every LDS block feeds 2 tiles" rows, plus what those rows left out).   MODE 1: + lockstep conversions.   MODE 2: halves.
// Output: fraction of the 2.5 PF dense fp16 peak (3 MFMAs per product block counted, as everywhere).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ws_loop_mock.hip -o scripts/ws_loop_mock
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KB = 8, NSG = 8, NW = 8;                      // 256 inputs = 8 k-blocks of 32; 128 samples = 8 groups of 16; 8 waves
constexpr int ACT_BYTES = NSG * KB * 2048;                 // 128 KiB
constexpr int PMAX_BYTES = NSG * 16 * NW * 4;              // [sample group][sample][wave] partial maxima: 4 KiB
constexpr int LDS_BYTES = ACT_BYTES + PMAX_BYTES;
constexpr int IMG_BYTES = NW * KB * 2 * 2 * 1024;          // one layer: [wave][k-block][tile of the wave][hi | lo] x 1 KiB = 256 KiB
constexpr int NIMG = 4;                                    // distinct layer images (1 MiB: L2-resident, not L1-resident)

__device__ __forceinline__ void mfma6(f32x4& a0, f32x4& a1, const f16x8& wl0, const f16x8& wl1, const f16x8& wh0, const f16x8& wh1,
                                      const f16x8& xh, const f16x8& xl) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0, xh, a0, 0, 0, 0);  // (the small terms first, no MFMA reads its predecessor's result)
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl1, xh, a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, xl, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, xl, a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, xh, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, xh, a1, 0, 0, 0);
}

// this wave's partial maximum of sample group sg (its 32 units of each of the 16 samples) -> pmax[sg][sample][wave]
__device__ __forceinline__ void put_max(const f32x4& a0, const f32x4& a1, float* pmax, int sg, int lane, int wave) {
    float m = fmaxf(fmaxf(fmaxf(a0[0], a0[1]), fmaxf(a0[2], a0[3])), fmaxf(fmaxf(a1[0], a1[1]), fmaxf(a1[2], a1[3])));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (lane < 16) pmax[(sg * 16 + lane) * NW + wave] = m;
}
// the eight waves' partial maxima of this lane's sample -> the power of two that moves the sample's largest value to [2^13, 2^14);
// the wave's 8 values per lane -> ReLU, hi / lo pieces = k-block `wave` of the next layer's operand, written in place
__device__ __forceinline__ void convert_store(f32x4& a0, f32x4& a1, const float* pmax, char* act, int sg, int lane, int wave) {
    const float4 p0 = *(const float4*)(pmax + (sg * 16 + (lane & 15)) * NW), p1 = *(const float4*)(pmax + (sg * 16 + (lane & 15)) * NW + 4);
    const float m = fmaxf(fmaxf(fmaxf(p0.x, p0.y), fmaxf(p0.z, p0.w)), fmaxf(fmaxf(p1.x, p1.y), fmaxf(p1.z, p1.w)));
    unsigned mb;
    __builtin_memcpy(&mb, &m, 4);
    const unsigned sb = ((mb >> 23) & 255u) == 0u ? (127u << 23) : ((unsigned)(127 + 13 + 127 - (int)((mb >> 23) & 255u)) << 23);
    float mul;
    __builtin_memcpy(&mul, &sb, 4);
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = (e < 4 ? a0[e & 3] : a1[e & 3]) * mul;
        v = v > 0.f ? v : 0.f;
        const _Float16 hi = (_Float16)v;
        h[e] = hi;
        l[e] = (_Float16)(v - (float)hi);
    }
    *(f16x8*)(act + ((sg * KB + wave) * 2 + 0) * 1024 + lane * 16) = h;
    *(f16x8*)(act + ((sg * KB + wave) * 2 + 1) * 1024 + lane * 16) = l;
    a0 = f32x4{0.f, 0.f, 0.f, 0.f};
    a1 = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_ws(const char* __restrict__ wimg, float* __restrict__ out, int layers) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const act = lds;
    float* const pmax = (float*)(lds + ACT_BYTES);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // activations like the product's operand pieces: high pieces up to 2^13, low pieces 2^-11 of that (random-ish bits)
    for (int p = threadIdx.x; p < ACT_BYTES / 2; p += 512) {
        const int blk = p >> 9;
        const float u = (float)(((p * 2654435761u) >> 12) & 0xffff) / 65536.0f - 0.5f;
        ((_Float16*)act)[p] = (_Float16)((blk & 1) ? u * 4.0f : u * 8000.0f);
    }
    for (int p = threadIdx.x; p < PMAX_BYTES / 4; p += 512) pmax[p] = 1.0f;
    f16x8 wh[KB][2], wl[KB][2];
    auto load_w = [&](int layer, int kb) {  // this wave's slice of k-block kb of a layer image: 4 x 16 bytes per lane, from L2
        const char* const src = wimg + (size_t)(layer % NIMG) * IMG_BYTES + (size_t)((wave * KB + kb) * 4) * 1024 + lane * 16;
        wh[kb][0] = *(const f16x8*)(src);
        wl[kb][0] = *(const f16x8*)(src + 1024);
        wh[kb][1] = *(const f16x8*)(src + 2048);
        wl[kb][1] = *(const f16x8*)(src + 3072);
    };
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) load_w(0, kb);
    f32x4 acc[NSG][2];
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) acc[sg][0] = acc[sg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const char* const xb = act + lane * 16;
    // the MFMAs of sample groups [s0, s1) over all k-blocks; RELOAD: a k-block's weight registers are re-loaded for layer `nxt` once its
    // last MFMA of this pass has issued; epi(sg) (MODE 2): called once per k-block between the MFMAs -- the other half's conversions
    auto multiply = [&](int s0, int s1, bool reload, int nxt, auto&& epi) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int sg = s0; sg < s1; ++sg) {
                const f16x8 xh = *(const f16x8*)(xb + ((sg * KB + kb) * 2 + 0) * 1024);
                const f16x8 xl = *(const f16x8*)(xb + ((sg * KB + kb) * 2 + 1) * 1024);
                mfma6(acc[sg][0], acc[sg][1], wl[kb][0], wl[kb][1], wh[kb][0], wh[kb][1], xh, xl);
            }
            epi(kb);
            if (reload) load_w(nxt, kb);
        }
    };
    auto none = [](int) {};
    if (MODE != 2) {
        for (int L = 0; L < layers; ++L) {
            multiply(0, NSG, true, L + 1, none);
            if (MODE == 1) {
#pragma unroll
                for (int sg = 0; sg < NSG; ++sg) put_max(acc[sg][0], acc[sg][1], pmax, sg, lane, wave);
            }
            __syncthreads();  // everybody has read the activations (and the partial maxima are in place)
            if (MODE == 1) {
#pragma unroll
                for (int sg = 0; sg < NSG; ++sg) convert_store(acc[sg][0], acc[sg][1], pmax, act, sg, lane, wave);
            }
            __syncthreads();  // the next layer's operand pieces are in place
        }
    } else {
        constexpr int HS = NSG / 2;
        multiply(0, HS, false, 0, none);  // half A of layer 0
#pragma unroll
        for (int sg = 0; sg < HS; ++sg) put_max(acc[sg][0], acc[sg][1], pmax, sg, lane, wave);
        for (int L = 0; L < layers; ++L) {
            __syncthreads();  // half A of layer L multiplied by everybody, its partial maxima in place
            // half B of layer L, with half A's conversions dealt over its k-blocks (4 sample groups over 8 k-blocks: one every second)
            multiply(HS, NSG, true, L + 1, [&](int kb) {
                if ((kb & 1) == 0) convert_store(acc[kb >> 1][0], acc[kb >> 1][1], pmax, act, kb >> 1, lane, wave);
            });
#pragma unroll
            for (int sg = HS; sg < NSG; ++sg) put_max(acc[sg][0], acc[sg][1], pmax, sg, lane, wave);
            __syncthreads();  // half B multiplied, half A's next operand pieces in place
            multiply(0, HS, false, 0, [&](int kb) {  // half A of layer L + 1 (the weights were re-loaded during half B), with half B's conversions
                if ((kb & 1) == 0) convert_store(acc[HS + (kb >> 1)][0], acc[HS + (kb >> 1)][1], pmax, act, HS + (kb >> 1), lane, wave);
            });
#pragma unroll
            for (int sg = 0; sg < HS; ++sg) put_max(acc[sg][0], acc[sg][1], pmax, sg, lane, wave);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) s += acc[sg][0][0] + acc[sg][1][3];
    s += (float)wh[0][0][0] + (float)((const _Float16*)act)[threadIdx.x];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* what, const char* wimg, float* out, int grid, int layers) {
    hipFuncSetAttribute((const void*)k_ws<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_ws<MODE>), dim3(grid), dim3(512), LDS_BYTES, 0, wimg, out, layers);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const hipError_t err = hipGetLastError();
    // per layer and wave: 8 k-blocks x 8 sample groups x 6 MFMAs (MODE 2 multiplies one half-layer more at the start: counted)
    const double mf = (MODE == 2 ? (double)layers + 0.5 : (double)layers) * KB * NSG * 6.0;
    const double flops = 2.0 * 16 * 16 * 32 * mf * NW * grid;
    printf("%-96s %8.3f ms  %7.1f TF  %.3f of 2.5 PF  %s\n", what, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15,
           err == hipSuccess ? "" : hipGetErrorString(err));
}

int main() {
    const int grid = 256, layers = 400;
    char* wimg;
    float* out;
    hipMalloc((void**)&wimg, (size_t)NIMG * IMG_BYTES);
    {  // weight pieces like a trained layer's: high pieces ~ U * 2^6, low pieces 2^-11 of that (random bits: realistic toggling)
        static _Float16 h[NIMG * IMG_BYTES / 2];
        unsigned st = 12345u;
        for (int b = 0; b < NIMG * IMG_BYTES / 1024; ++b)
            for (int i = 0; i < 512; ++i) {
                st = st * 1664525u + 1013904223u;
                const float u = (float)((st >> 8) & 0xffff) / 65536.0f - 0.5f;
                h[b * 512 + i] = (_Float16)((b & 1) ? u * 0.03f : u * 64.0f);
            }
        hipMemcpy(wimg, h, (size_t)NIMG * IMG_BYTES, hipMemcpyHostToDevice);
    }
    hipMalloc((void**)&out, (size_t)grid * 512 * 4);
    printf("# weight-stationary loop: %d layers of 256 x 256 on fp16 pieces, 128 samples per 8-wave workgroup, %d workgroups (one per CU)\n", layers, grid);
    run<0>("MODE 0: multiply loop, weights L2 -> registers (rolling re-load), activations from LDS, 2 barriers per layer", wimg, out, grid, layers);
    run<1>("MODE 1: + per-sample maximum through LDS, ReLU, hi / lo conversion, in-place operand stores (all waves in lockstep)", wimg, out, grid, layers);
    run<2>("MODE 2: the same work, two sample halves: one half's conversions between the other half's MFMAs", wimg, out, grid, layers);
    return 0;
}

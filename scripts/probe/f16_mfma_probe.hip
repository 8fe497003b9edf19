// f16_mfma_probe.hip -- does v_mfma_f32_32x32x16_f16 honour fp16 subnormal INPUTS, and what do the f32 -> f16 conversions do
// at the bottom of the range?  (Decides whether the split-f16 kernels need to scale their low pieces.)
// hipcc --offload-arch=gfx950 -O2 scripts/probe/f16_mfma_probe.hip -o gpurun_out/f16_probe && gpurun_out/f16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k(const float* av, const float* bv, float* out, float* cvt_in, float* cvt_out, int ncvt) {
    const int l = threadIdx.x;
    for (int c = 0; c < 8; ++c) {  // case c: every A element = av[c], every B element = bv[c]: D = 16 * a * b
        h8 a, b;
        for (int e = 0; e < 8; ++e) a[e] = (_Float16)av[c], b[e] = (_Float16)bv[c];
        f16v acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (l == 0) out[c] = acc[0];
    }
    if (l < ncvt) {
        const float v = cvt_in[l];
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        cvt_out[3 * l] = (float)hi;
        cvt_out[3 * l + 1] = (float)lo;
        cvt_out[3 * l + 2] = v - ((float)hi + (float)lo);
    }
}

int main() {
    // a (as f16) x b (as f16): subnormal a values with normal b, and both subnormal-ish
    float av[8] = {ldexpf(1, -20), ldexpf(1, -24), ldexpf(3, -24), ldexpf(1, -14), ldexpf(1, -15), 1.0f, ldexpf(1, -20), 65504.f};
    float bv[8] = {ldexpf(1, 10), ldexpf(1, 12), 1.0f, 1.0f, 1.0f, ldexpf(1, -24), ldexpf(1, -20), 1.0f};
    float cin[8] = {0.1f, 0.0123456f, 3.14159265f, 1e-3f, 1e-5f, 250.123f, 6.1e-5f, 1e-7f};
    float *d_a, *d_b, *d_o, *d_ci, *d_co;
    hipMalloc(&d_a, 32); hipMalloc(&d_b, 32); hipMalloc(&d_o, 32); hipMalloc(&d_ci, 32); hipMalloc(&d_co, 96);
    hipMemcpy(d_a, av, 32, hipMemcpyHostToDevice); hipMemcpy(d_b, bv, 32, hipMemcpyHostToDevice);
    hipMemcpy(d_ci, cin, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_a, d_b, d_o, d_ci, d_co, 8);
    float o[8], co[24];
    hipMemcpy(o, d_o, 32, hipMemcpyDeviceToHost); hipMemcpy(co, d_co, 96, hipMemcpyDeviceToHost);
    for (int c = 0; c < 8; ++c) printf("mfma a=%g b=%g -> %g (exact 16ab = %g)\n", av[c], bv[c], o[c], 16.0 * (double)av[c] * (double)bv[c]);
    for (int c = 0; c < 8; ++c) printf("split v=%.9g hi=%.9g lo=%.9g resid=%.3g rel=%.3g\n", cin[c], co[3*c], co[3*c+1], co[3*c+2], fabs(co[3*c+2]) / fabs(cin[c]));
    return 0;
}

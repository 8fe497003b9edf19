#include <hip/hip_runtime.h>
__device__ __forceinline__ unsigned wave_max(unsigned v) {
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = v > t ? v : t;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__global__ void k(unsigned* out, const unsigned* in) {
    unsigned m = wave_max(in[threadIdx.x]);
    if (threadIdx.x == 0) out[0] = m;
}
int main() {
    unsigned h[64], *di, *d_o, o;
    for (int i = 0; i < 64; ++i) h[i] = (i * 37 + 11) % 101;
    h[29] = 5000; 
    hipMalloc(&di, 256); hipMalloc(&d_o, 4); hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    for (int t = 0; t < 64; t += 7) { unsigned hh[64]; for (int i = 0; i < 64; ++i) hh[i] = h[i]; hh[29] = 7; hh[t] = 9000 + t; hipMemcpy(di, hh, 256, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_o, di); hipMemcpy(&o, d_o, 4, hipMemcpyDeviceToHost); printf("max at lane %d -> %u\n", t, o); }
    return 0;
}

// nh_mlp.h -- internal host entry points of mlp.hip / mlp16.hip / wgrad.hip (used by the C ABI in mlp.hip and by fused.hip).
#pragma once
#include "nh_host.h"
#include "nh_plan.h"

struct NhMlpInput {
    int mode;          // 0: encoded rows x [M, dim_xyz+dim_dir];  1: rays [n, ray_stride] + depths z [n, S] (M = n*S)
    const float* x;
    const float* rays;
    int ray_stride;
    const float* z;
    int S;
};

int nh_mlp_forward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                   nerfhip_stream_t stream);
int nh_mlp_backward(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash,
                    float* scratch, int64_t scratch_bytes, float* g_params, nerfhip_stream_t stream);
int64_t nh_mlp_bwd_scratch_bytes(nerfhip_plan* p, int64_t M);

// mlp16.hip: the forward / data-gradient chain on v_mfma_f32_16x16x4_f32, two waves per SIMD
int nh_mlp16_forward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                     nerfhip_stream_t stream);
int nh_mlp16_dgrad(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash, float* scratch,
                   nerfhip_stream_t stream);

// wgrad.hip: split-K weight-gradient GEMMs over the stash / d(pre-activation) images (nt = 32-sample tiles) + reduction
int64_t nh_wgrad_partial_floats(nerfhip_plan* p, int64_t nt);
int nh_wgrad(nerfhip_plan* p, int64_t nt, const float* stash, const float* grad, float* partial, float* g_params,
             nerfhip_stream_t stream);

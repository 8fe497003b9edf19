// mlp.hip -- host side of the FlexibleNeRFModel kernels (nerf/models.py:185-256): argument checks and the launch
// sequences  forward = k_mlp_fwd16 (mlp16.hip),  backward = k_mlp_dgrad16 (mlp16.hip) -> k_wgrad -> k_wgrad_reduce
// (wgrad.hip), and the C-ABI entry points nerfhip_mlp_fwd / nerfhip_mlp_bwd.
#include <stdlib.h>

#include "nh_mlp.h"

// scratch of a backward over M sample points: the d(pre-activation) images the data-gradient kernel writes for the
// weight-gradient kernel, then the split-K partials
int64_t nh_mlp_bwd_scratch_bytes(nerfhip_plan* p, int64_t M) {
    const int64_t nt = nh_ceil_div(M, 128) * 4;
    return (nt * p->grad.total_rows * 32 + nh_wgrad_partial_floats(p, nt)) * (int64_t)sizeof(float);
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
    return nh_mlp16_forward(p, packed, in, M, out, stash, stream);
}

int nh_mlp_backward(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash,
                    float* scratch, int64_t scratch_bytes, float* g_params, nerfhip_stream_t stream) {
    NH_REQUIRE(p && packed && g_out && stash && scratch && g_params && M > 0, "mlp_bwd: bad arguments");
    NH_REQUIRE(scratch_bytes >= nh_mlp_bwd_scratch_bytes(p, M), "mlp_bwd: scratch too small (%lld < %lld)",
               (long long)scratch_bytes, (long long)nh_mlp_bwd_scratch_bytes(p, M));
    const int64_t nt = nh_ceil_div(M, 128) * 4;
    int rc = nh_mlp16_dgrad(p, packed, g_out, M, stash, scratch, stream);
    if (rc) return rc;
    return nh_wgrad(p, nt, stash, scratch, scratch + (size_t)nt * (size_t)p->grad.total_rows * 32, g_params, stream);
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

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

// The sample list of a compacted backward (compact.hip; nerfhip_plan_set_bwd_compaction): behind a backward scratch's region maxima.
// stats[NH_CSTAT_ACTIVE] = samples whose d(raw output) row is not all zero, stats[NH_CSTAT_TOTAL] = samples of the launch;
// idx[0 .. active): their sample indices, ascending, padded with sample 0 up to the next multiple of 128.
// idx == NULL in a kernel's arguments: the dense backward.
constexpr int NH_CSTAT_WORDS = 16, NH_CSTAT_ACTIVE = 0, NH_CSTAT_TOTAL = 1;
struct NhCompact {
    int* stats;
    int* counts;  // per 2048-sample block (k_compact_count)
    int* idx;
    // the stash the backward kernels read is ITSELF in list order (written by a forward over the list: nh_mlp_backward_recompute):
    // ReLU masks and activation rows of slot c sit at slot c -- nothing is gathered but d(raw output)
    bool stash_in_list_order;
};
int64_t nh_compact_ints(int64_t M);                  // 32-bit words of the area for M sample points
NhCompact nh_compact_view(int* area, int64_t M);
int nh_compact_build(const float* g_out, int64_t M, const NhCompact& c, nerfhip_stream_t stream);

int nh_mlp_forward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                   nerfhip_stream_t stream);
// the training forward of the fused render: writes into `stash` what THIS plan's backward mode will read there -- the general stash
// (modes 0, 1), nothing (the recomputing modes 2, 3, 4), the register-image stash of the stashed fused backward (mode 5)
int nh_mlp_forward_training(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                            nerfhip_stream_t stream);
int nh_mlp_backward(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash,
                    float* scratch, int64_t scratch_bytes, float* g_params, nerfhip_stream_t stream);
// plans with bwd_compact == 2 inside the fused render: the forward over `in` wrote no stash; lists the samples with a non-zero
// d(raw output) row, re-runs the forward for them (writing `stash` in list order) and differentiates those
// need_images: the caller reads the d(pre-activation) images afterwards (the ray gradient): a plan in a fused mode (3 / 4: mlp64r.hip
// leaves none) then runs as mode 2
int nh_mlp_backward_recompute(nerfhip_plan* p, const float* packed, const NhMlpInput& in, const float* g_out, int64_t M, float* stash,
                              float* scratch, int64_t scratch_bytes, float* g_params, bool need_images, nerfhip_stream_t stream);
int64_t nh_mlp_bwd_scratch_bytes(nerfhip_plan* p, int64_t M);
// a backward over M sample points of this plan re-runs its forward (bwd_compact == 2 and the launch compacts at all)
bool nh_mlp_recomputes(const nerfhip_plan* p, int64_t M);

// mlp16.hip: the forward / data-gradient chain on v_mfma_f32_16x16x4_f32, two waves per SIMD
// (list: a forward over the samples of a compaction list only -- slot c computes sample idx[c] and writes ITS stash rows / masks at
// slot c; `out` may then be NULL; else NULL)
int nh_mlp16_forward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                     nerfhip_stream_t stream, const NhCompact* list = nullptr);
// (cx: the compacted backward's sample list, or NULL -- here and in every backward kernel below)
int nh_mlp16_dgrad(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash, float* scratch,
                   const NhCompact* cx, nerfhip_stream_t stream);

// mlp64r.hip: the fused, stash-free backward of 64-wide nets with an LDS-resident image (nh_r64.h): forward recomputed, data
// gradient and weight gradient in one persistent kernel + a fixed-order reduction of one partial per workgroup
int64_t nh_mlp64r_partial_floats(const nerfhip_plan* p, int64_t M);
// ... and the persistent forward with the same image: raw[M, 4] without a stash (inference, the stash-free training forward)
// (stash: NULL, or -- mode 5 -- the register-image stash of nh_r64.h: written by the forward, read by the backward instead of
// recomputing; the backward then runs over every sample, cx == NULL)
int nh_mlp64r_forward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                      nerfhip_stream_t stream);
int nh_mlp64r_backward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, const float* g_out, int64_t M, float* partial,
                       float* g_params, const NhCompact* cx, const float* stash, nerfhip_stream_t stream);

// mlp_f16w.hip: forward (with / without stash) and data-gradient chain of the fp16-piece plans: two waves per SIMD, 16-sample waves on
// v_mfma_f32_16x16x32_f16; rmax (level-4 plans): NH_RMAX_WORDS zeroed device words for the region maxima, or NULL
int nh_mlp_forward_f16w(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                        nerfhip_stream_t stream, const NhCompact* list = nullptr);
int nh_mlp_dgrad_f16w(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash, float* scratch,
                      unsigned* rmax, const NhCompact* cx, nerfhip_stream_t stream);
// mlp.hip: n words of device memory to zero, on the stream
int nh_zero_words(unsigned* dev, int n, nerfhip_stream_t stream);
// pack_f16.hip: the fp16-piece layer images of a plan
int nh_pack_pieces_f16(nerfhip_plan* plan, const float* params, const int32_t* table, float* packed, nerfhip_stream_t stream);

// wgrad.hip: split-K weight-gradient GEMMs over the stash / d(pre-activation) images (nt = 32-sample tiles) + reduction
int64_t nh_wgrad_partial_floats(nerfhip_plan* p, int64_t nt);
// gscale: a device word with the bits of a maximum the images were scaled by (nh_gscale_of; the reduction multiplies by
// nh_gscale_inv), or NULL (every precision stores plain values today)
int nh_wgrad(nerfhip_plan* p, int64_t nt, const float* stash, const float* grad, float* partial, float* g_params,
             const unsigned* gscale, const NhCompact* cx, nerfhip_stream_t stream);

// wgrad_f16.hip: the large weight blocks of level-4 plans (plan->bjobs) on the fp16 MFMAs
int64_t nh_wgrad_x3_partial_floats(nerfhip_plan* p, int64_t nt);  // (-1 with an error message: a block list the schedule refuses)
// amax / bmax: the region maxima recorded by the data-gradient / forward launch that wrote `grad` / `stash`, or NULL
int nh_wgrad_f16(nerfhip_plan* p, int64_t nt, const float* stash, const float* grad, float* partial, float* g_params,
                 const unsigned* amax, const unsigned* bmax, const NhCompact* cx, nerfhip_stream_t stream);

// render.hip: compositing backward with the optional dL/d||rd|| output, and the gradient w.r.t. the packed rays
int nh_volume_render_bwd(const float* raw, const float* z, const float* rd, int rd_stride, int64_t n, int s, float noise_std,
                         const float* noise, uint64_t seed, uint32_t rng_stream, uint64_t ray_offset, int white_background,
                         const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_weights, float* g_raw,
                         float* g_norm, nerfhip_stream_t stream);
int nh_ray_grad(const float* rays, int stride, int64_t n, const float* z, int S, const float* g_x, int dx, int dd, int inc_x,
                int inc_d, int Lx, int Ld, const float* fx, const float* fd, const float* g_norm, float* g_rays, int accumulate,
                nerfhip_stream_t stream);

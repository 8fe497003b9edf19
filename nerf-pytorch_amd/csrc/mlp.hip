// mlp.hip -- host side of the FlexibleNeRFModel kernels (nerf/models.py:185-256): argument checks and the launch
// sequences  forward = k_mlp_fwd16 (mlp16.hip),  backward = k_mlp_dgrad16 (mlp16.hip) -> k_wgrad -> k_wgrad_reduce
// (wgrad.hip), and the C-ABI entry points nerfhip_mlp_fwd / nerfhip_mlp_bwd; plus the gradient w.r.t. the encoded input
// (nerfhip_mlp_bwd_input: off the hot path -- the render path never differentiates its encodings).
#include <stdlib.h>

#include "nh_mlp.h"
#include "nh_r64.h"

namespace {

// d(loss)/d(x) of FlexibleNeRFModel.forward (nerf/models.py:233-256): x = cat(xyz, dirs) enters layer1, every skip layer
// (xyz columns) and layers_dir[0] (direction columns), so
//   g_x[m, c] = sum over those layers  sum_u dpre[m, u] * W[u, col0 + c - out0]
// with dpre = the d(pre-activation) image the data-gradient kernel left in the backward scratch ([sample][rows]).
constexpr int NH_INGRAD_MAX_TERMS = 2 * (NH_MAX_LAYERS + 2);  // (512-wide nets: one term per 256-row half)
struct InGradTerm {
    int64_t a_prefix;  // region offset = 32 * nt * a_prefix floats
    int a_rows, nu;    // row count of the region, real units
    int64_t w_off;     // flat offset of the weight tensor
    int w_ld, col0;    // its column count, first column that multiplies x
    int out0, ncols;   // g_x columns [out0, out0 + ncols)
};
struct InGradArgs {
    InGradTerm t[NH_INGRAD_MAX_TERMS];
    int nterms;
    const float* scratch;
    const float* params;
    int64_t M, nt;
    int D;
    float* g_x;
    const unsigned* gscale;  // fp16 data-gradient chains: the images carry nh_gscale_of(*gscale); else NULL
    // compacted backward: image row r belongs to sample cidx[r], r < cstats[NH_CSTAT_ACTIVE]; the other samples' rows of g_x are
    // zero (k_zero_floats runs first)
    const int* cidx;
    const int* cstats;
};

NH_KERNEL void k_mlp_input_grad(InGradArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.M * a.D) return;
    const int64_t r = idx / a.D;  // row of the d(pre-activation) images
    const int c = (int)(idx - r * a.D);
    int64_t m = r;                // the sample it belongs to
    if (a.cidx) {
        if (r >= (int64_t)a.cstats[NH_CSTAT_ACTIVE]) return;
        m = a.cidx[r];
    }
    float s = 0.0f;
    for (int k = 0; k < a.nterms; ++k) {
        const InGradTerm& t = a.t[k];
        if (c < t.out0 || c >= t.out0 + t.ncols) continue;
        const float* dp = a.scratch + (size_t)32 * (size_t)a.nt * (size_t)t.a_prefix + (size_t)r * (size_t)t.a_rows;
        const float* w = a.params + t.w_off + t.col0 + (c - t.out0);
        for (int u = 0; u < t.nu; ++u) s = fmaf(dp[u], w[(size_t)u * t.w_ld], s);
    }
    a.g_x[m * a.D + c] = a.gscale ? s * nh_gscale_inv(*a.gscale) : s;
}

NH_KERNEL void k_zero_floats(float* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 0.0f;
}

NH_KERNEL void k_zero_words(unsigned* out, int n) {
    for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) out[i] = 0u;
}

}  // namespace

// scratch of a backward over M sample points: the d(pre-activation) images the data-gradient kernel writes for the
// weight-gradient kernel, then the split-K partials
// (... then NH_RMAX_WORDS words: the region maxima of the d(pre-activation) images, fp16 level-4 plans)
static int64_t gscale_word_offset(nerfhip_plan* p, int64_t nt) {
    return nt * p->grad.total_rows * 32 + nh_wgrad_partial_floats(p, nt) + nh_wgrad_x3_partial_floats(p, nt);
}
// (... then the sample list of a compacted backward, compact.hip: reserved whether or not the plan's option is on)
static int64_t compact_word_offset(nerfhip_plan* p, int64_t nt) { return gscale_word_offset(p, nt) + NH_RMAX_WORDS; }
// (... then, plans with a resident image: one partial gradient per workgroup of the fused backward, mlp64r.hip)
static int64_t fused_partial_offset(nerfhip_plan* p, int64_t nt) { return compact_word_offset(p, nt) + nh_compact_ints(nt * 32); }
int64_t nh_mlp_bwd_scratch_bytes(nerfhip_plan* p, int64_t M) {
    const int64_t nt = nh_ceil_div(M, 128) * 4;
    return (fused_partial_offset(p, nt) + nh_mlp64r_partial_floats(p, M)) * (int64_t)sizeof(float);
}
// A backward over M sample points runs compacted when the plan asks for it and a gathered row's byte offset inside a region (at most
// 256 rows of 4 bytes per sample) fits the 32-bit offset of a buffer instruction; otherwise it runs dense.
// (the fused modes 3 / 4 leave no d(pre-activation) images: whoever needs them -- the ray gradient, nerfhip_mlp_bwd with a stash of its
// caller -- gets mode 2's data flow)
static int image_mode(const nerfhip_plan* p) { return p->bwd_compact >= 3 ? 2 : p->bwd_compact; }
static bool compacts(const nerfhip_plan* p, int64_t M) { return image_mode(p) != 0 && nh_ceil_div(M, 128) * 128 * 1024 < ((int64_t)1 << 32); }
bool nh_mlp_recomputes(const nerfhip_plan* p, int64_t M) { return image_mode(p) == 2 && compacts(p, M) && nh_prec_level(p->precision) != 1; }

static int mlp_forward_any(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                           nerfhip_stream_t stream, const NhCompact* list);

int nh_mlp_forward(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                   nerfhip_stream_t stream) {
    NH_REQUIRE(out, "mlp_fwd: bad arguments");
    return mlp_forward_any(p, packed, in, M, out, stash, stream, nullptr);
}

static int mlp_forward_any(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                           nerfhip_stream_t stream, const NhCompact* list) {
    NH_REQUIRE(p && packed && M >= 0, "mlp_fwd: bad arguments");
    if (M == 0) return NERFHIP_OK;
    if (in.mode == 1) {
        NH_REQUIRE(p->freqs_set, "mlp_fwd: nerfhip_plan_set_freqs has not been called");
        NH_REQUIRE(in.rays && in.z && in.S > 0 && in.ray_stride >= (p->view ? 11 : 8), "mlp_fwd: bad fused input");
    } else {
        NH_REQUIRE(in.x, "mlp_fwd: x is NULL");
    }
    if (p->precision != NERFHIP_PRECISION_FP32) {
        NH_REQUIRE(!stash || nh_prec_level(p->precision) != 1,
                   "mlp_fwd: an f16x3 plan is inference-only (no activation stash, no backward; the _FWD / _FWD_DGRAD / _TRAIN plans train)");
        return nh_mlp_forward_f16w(p, packed, in, M, out, stash, stream, list);
    }
    // 64-wide nets with an LDS-resident image (nh_r64.h): the persistent forward whenever no stash is asked for
    if (p->r64_off >= 0 && !stash && !list) return nh_mlp64r_forward(p, packed, in, M, out, nullptr, stream);
    return nh_mlp16_forward(p, packed, in, M, out, stash, stream, list);
}

// mode 5 inside the fused render: fp32 plan with a resident image whose register-image stash fits the plan's stash region
static bool fused_stashed(const nerfhip_plan* p, int64_t M) {
    return p->bwd_compact == 5 && p->r64_off >= 0 && nh_r64_stash_fits(p) && nh_mlp_recomputes(p, M);
}

int nh_mlp_forward_training(nerfhip_plan* p, const float* packed, const NhMlpInput& in, int64_t M, float* out, float* stash,
                            nerfhip_stream_t stream) {
    NH_REQUIRE(p && out && stash, "mlp_fwd: bad arguments");
    if (M > 0 && fused_stashed(p, M)) return nh_mlp64r_forward(p, packed, in, M, out, stash, stream);
    return mlp_forward_any(p, packed, in, M, out, nh_mlp_recomputes(p, M) ? nullptr : stash, stream, nullptr);
}

// `recompute`: the forward of this launch wrote no stash (nh_mlp_recomputes); `in` names its input again
static int mlp_backward_any(nerfhip_plan* p, const float* packed, const NhMlpInput* recompute, const float* g_out, int64_t M, float* stash,
                            float* scratch, int64_t scratch_bytes, float* g_params, bool need_images, nerfhip_stream_t stream) {
    NH_REQUIRE(p && packed && g_out && stash && scratch && g_params && M > 0, "mlp_bwd: bad arguments");
    NH_REQUIRE(nh_prec_level(p->precision) != 1, "mlp_bwd: an f16x3 plan is inference-only");
    NH_REQUIRE(scratch_bytes >= nh_mlp_bwd_scratch_bytes(p, M), "mlp_bwd: scratch too small (%lld < %lld)",
               (long long)scratch_bytes, (long long)nh_mlp_bwd_scratch_bytes(p, M));
    const int64_t nt = nh_ceil_div(M, 128) * 4;
    const bool bdg = nh_prec_level(p->precision) >= 3;
    int rc = NERFHIP_OK;
    if (recompute && p->bwd_compact >= 3 && p->r64_off >= 0 && !need_images) {
        // the fused backward (mlp64r.hip): forward recomputed, data gradient and weight gradient in one kernel; mode 4 walks the list
        if (fused_stashed(p, M))  // mode 5: the forward left the register-image stash; nothing is recomputed, every sample is walked
            return nh_mlp64r_backward(p, packed, *recompute, g_out, M, scratch + fused_partial_offset(p, nt), g_params, nullptr, stash, stream);
        NhCompact lview;
        const NhCompact* lx = nullptr;
        if (p->bwd_compact == 4) {
            lview = nh_compact_view((int*)(scratch + compact_word_offset(p, nt)), nt * 32);
            rc = nh_compact_build(g_out, M, lview, stream);
            if (rc) return rc;
            lx = &lview;
        }
        return nh_mlp64r_backward(p, packed, *recompute, g_out, M, scratch + fused_partial_offset(p, nt), g_params, lx, nullptr, stream);
    }
    // compacted backward: list the samples whose d(raw output) row is not all zero; every kernel below then walks that list
    NhCompact cview;
    const NhCompact* cx = nullptr;
    if (compacts(p, M)) {
        cview = nh_compact_view((int*)(scratch + compact_word_offset(p, nt)), nt * 32);
        rc = nh_compact_build(g_out, M, cview, stream);
        if (rc) return rc;
        cx = &cview;
    }
    if (recompute) {
        // ... and the forward is run again for the listed samples only: their activation rows and ReLU masks, in list order (the
        // fp16-piece forward also records the stash's region maxima again, behind the stash as every training forward does)
        NH_REQUIRE(cx, "mlp_bwd: a recomputing backward needs the compacted list");
        rc = mlp_forward_any(p, packed, *recompute, M, nullptr, stash, stream, cx);
        if (rc) return rc;
        cview.stash_in_list_order = true;
    }
    // fp16 plans whose large weight-gradient blocks run on the fp16 MFMAs: the producers record per-region maxima (behind the
    // stash: the forward's; behind this scratch: the data-gradient launch's), from which k_wgrad_f16x3 takes its scales
    unsigned* amax = nullptr;
    const unsigned* bmax = nullptr;
    if (!p->bjobs.empty()) {
        amax = (unsigned*)(scratch + gscale_word_offset(p, nt));
        bmax = (const unsigned*)(stash + nh_stash_floats(p, nt));
        rc = nh_zero_words(amax, NH_RMAX_WORDS, stream);
        if (rc) return rc;
    }
    if (bdg)
        rc = nh_mlp_dgrad_f16w(p, packed, g_out, M, stash, scratch, amax, cx, stream);
    else
        rc = nh_mlp16_dgrad(p, packed, g_out, M, stash, scratch, cx, stream);
    if (rc) return rc;
    float* const partial = scratch + (size_t)nt * (size_t)p->grad.total_rows * 32;
    rc = nh_wgrad(p, nt, stash, scratch, partial, g_params, nullptr, cx, stream);
    if (rc) return rc;
    // (level 4: the large blocks, behind the fp32 kernel's partials)
    float* const partial_b = partial + nh_wgrad_partial_floats(p, nt);
    return nh_wgrad_f16(p, nt, stash, scratch, partial_b, g_params, amax, bmax, cx, stream);
}

int nh_mlp_backward(nerfhip_plan* p, const float* packed, const float* g_out, int64_t M, const float* stash,
                    float* scratch, int64_t scratch_bytes, float* g_params, nerfhip_stream_t stream) {
    return mlp_backward_any(p, packed, nullptr, g_out, M, (float*)stash, scratch, scratch_bytes, g_params, true, stream);
}

int nh_mlp_backward_recompute(nerfhip_plan* p, const float* packed, const NhMlpInput& in, const float* g_out, int64_t M, float* stash,
                              float* scratch, int64_t scratch_bytes, float* g_params, bool need_images, nerfhip_stream_t stream) {
    return mlp_backward_any(p, packed, &in, g_out, M, stash, scratch, scratch_bytes, g_params, need_images, stream);
}

extern "C" int nerfhip_plan_set_bwd_compaction(nerfhip_plan_t plan, int on) {
    NH_REQUIRE(plan, "plan_set_bwd_compaction: plan is NULL");
    NH_REQUIRE(on >= 0 && on <= 5, "plan_set_bwd_compaction: 0 (dense), 1 (compacted), 2 (compacted, the render path recomputes the stash), "
               "3 (the fused backward of 64-wide nets), 4 (the fused backward over the compacted list) or 5 (the fused backward over a "
               "register-image stash)");
    NH_REQUIRE(on < 3 || plan->r64_off >= 0, "plan_set_bwd_compaction: the fused backward (3, 4, 5) exists for fp32 plans of hidden_size <= 64 with view "
               "directions, at most 4 layers, no skip layer and num_encoding_fn_xyz <= 10 / num_encoding_fn_dir <= 4");
    NH_REQUIRE(on != 5 || nh_r64_stash_fits(plan), "plan_set_bwd_compaction: the register-image stash of mode 5 does not fit this plan's stash region");
    plan->bwd_compact = on;
    return NERFHIP_OK;
}
extern "C" int nerfhip_plan_bwd_compaction(nerfhip_plan_t plan) { return plan ? plan->bwd_compact : 0; }
extern "C" int64_t nerfhip_plan_bwd_stats_offset(nerfhip_plan_t plan, int64_t m) {
    if (!plan || m < 0) return -1;
    return compact_word_offset(plan, nh_ceil_div(m, 128) * 4) * (int64_t)sizeof(float);
}

int nh_zero_words(unsigned* dev, int n, nerfhip_stream_t stream) {
    NH_LAUNCH(k_zero_words, 1, 64, 0, stream, dev, n);
    return nh_launch_status("zero_words");
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

extern "C" int nerfhip_mlp_bwd_input(nerfhip_plan_t p, const float* params, int64_t m, const void* scratch, float* g_x,
                                     nerfhip_stream_t stream) {
    NH_REQUIRE(p && params && scratch && g_x && m > 0, "mlp_bwd_input: bad arguments");
    InGradArgs a;
    memset(&a, 0, sizeof(a));
    const int H = p->H;
    auto add = [&](const NhRegion& R, int nu, int tensor, int col0, int out0, int ncols) {
        // (units beyond 256 of a 512-wide net live in the second 256-row region of the activation)
        for (int u0 = 0; u0 < nu; u0 += 256) {
            InGradTerm& t = a.t[a.nterms++];
            t.a_prefix = R.row_prefix + u0;
            t.a_rows = R.rows;
            t.nu = nu - u0 < 256 ? nu - u0 : 256;
            t.w_ld = p->tensors[tensor].cols;
            t.w_off = p->tensors[tensor].off + (int64_t)u0 * t.w_ld;
            t.col0 = col0;
            t.out0 = out0;
            t.ncols = ncols;
        }
    };
    add(p->grad.P[0], H, p->t_layer1_w, 0, 0, p->Dx);
    for (int i = 0; i < p->L - 1; ++i)
        if (p->is_skip(i)) add(p->grad.P[i + 1], H, p->t_xyz_w[i], H, 0, p->Dx);
    if (p->view && p->Dd > 0) add(p->grad.PDIR, H / 2, p->t_dir_w, H, p->Dx, p->Dd);
    a.scratch = (const float*)scratch;
    a.params = params;
    a.M = m;
    a.nt = nh_ceil_div(m, 128) * 4;
    a.D = p->Dx + p->Dd;
    a.g_x = g_x;
    a.gscale = nullptr;  // (the d(pre-activation) images are plain values in every precision)
    if (compacts(p, m)) {  // (the backward that filled `scratch` ran compacted: its images are in list order)
        const NhCompact c = nh_compact_view((int*)const_cast<void*>(scratch) + compact_word_offset(p, a.nt), a.nt * 32);
        a.cidx = c.idx;
        a.cstats = c.stats;
        NH_LAUNCH(k_zero_floats, nh_ceil_div(m * a.D, 256), 256, 0, stream, g_x, m * a.D);
        const int rc = nh_launch_status("zero_floats");
        if (rc) return rc;
    }
    NH_LAUNCH(k_mlp_input_grad, nh_ceil_div(m * a.D, 256), 256, 0, stream, a);
    return nh_launch_status("mlp_input_grad");
}

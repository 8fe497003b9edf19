// fused.hip -- predict_and_render_radiance (nerf/train_utils.py:28-127) as a fixed pipeline of kernels on one stream:
//   stratified depths -> coarse MLP (encodes in registers) -> compositing -> inverse-CDF + merge -> fine MLP ->
//   compositing, and the matching backward.  Nothing of size (N*S, 90) or (N*S, 256) is materialised for inference;
//   a training forward additionally writes the activation stash the weight-gradient GEMM consumes.
#include "nh_mlp.h"

namespace {

struct Workspace {
    int64_t z_c, raw_c, w_c, z_f, raw_f, stash_c, stash_f, total;
    // backward buffers, one set per net: the coarse and the fine backward chains are independent and may run on
    // different streams (nerfhip_render_bwd_parts)
    int64_t g_raw_c, scratch_c, scratch_c_bytes, g_raw_f, scratch_f, scratch_f_bytes;
    int64_t gnorm_c, gnorm_f;  // dL/d||rd|| per ray of the two compositing backwards (read by the ray-gradient pass)
};

int64_t align_up(int64_t v) { return (v + 255) & ~(int64_t)255; }

// training: 0 = inference; 1 = training, one set of backward buffers per net (the two backward chains may then run on
// different streams: nerfhip_render_bwd_parts); 2 = training, the two nets SHARE one set of backward buffers (their
// backward chains must run one after the other on one stream -- what nerfhip_render_bwd does; a parts call says so with
// NERFHIP_PART_SHARED_BWD).  The forward's regions do not depend on that choice: they are laid out first.
Workspace layout(nerfhip_plan* pc, nerfhip_plan* pf, const nerfhip_render_cfg* cfg, int64_t n, int training) {
    Workspace w;
    memset(&w, 0, sizeof(w));
    const int64_t nc = cfg->num_coarse, nf = cfg->num_fine, sf = nc + nf;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off = align_up(off + bytes);
        return o;
    };
    w.z_c = take(n * nc * 4);
    w.raw_c = take(n * nc * 16);
    w.w_c = take(n * nc * 4);
    if (nf > 0) {
        w.z_f = take(n * sf * 4);
        w.raw_f = take(n * sf * 16);
    }
    if (training) {
        w.gnorm_c = take(n * 4);
        if (nf > 0) w.gnorm_f = take(n * 4);
        w.stash_c = take(nerfhip_plan_stash_bytes(pc, n * nc));
        if (nf > 0) w.stash_f = take(nerfhip_plan_stash_bytes(pf, n * sf));
        w.scratch_c_bytes = nh_mlp_bwd_scratch_bytes(pc, n * nc);
        if (nf > 0) w.scratch_f_bytes = nh_mlp_bwd_scratch_bytes(pf, n * sf);
        if (training == 2 && nf > 0) {
            w.g_raw_c = w.g_raw_f = take(n * sf * 16);
            w.scratch_c = w.scratch_f = take(w.scratch_c_bytes > w.scratch_f_bytes ? w.scratch_c_bytes : w.scratch_f_bytes);
        } else {
            w.g_raw_c = take(n * nc * 16);
            w.scratch_c = take(w.scratch_c_bytes);
            if (nf > 0) {
                w.g_raw_f = take(n * sf * 16);
                w.scratch_f = take(w.scratch_f_bytes);
            }
        }
    }
    w.total = off;
    return w;
}

int check_cfg(nerfhip_plan* pc, nerfhip_plan* pf, const nerfhip_render_cfg* cfg) {
    NH_REQUIRE(pc && cfg, "render: plan/cfg is NULL");
    NH_REQUIRE(cfg->num_coarse >= 3 || (cfg->num_coarse >= 1 && cfg->num_fine == 0), "render: num_coarse too small");
    NH_REQUIRE(cfg->num_fine >= 0, "render: num_fine < 0");
    NH_REQUIRE(cfg->num_fine == 0 || pf, "render: num_fine > 0 needs a fine plan");
    NH_REQUIRE(cfg->ray_stride >= (pc->view ? 11 : 8), "render: ray_stride too small for use_viewdirs");
    NH_REQUIRE(!pf || pf->view == pc->view, "render: coarse/fine use_viewdirs mismatch");
    return NERFHIP_OK;
}

}  // namespace

extern "C" int64_t nerfhip_render_workspace_bytes(nerfhip_plan_t plan_coarse, nerfhip_plan_t plan_fine,
                                                  const nerfhip_render_cfg* cfg, int64_t n_rays, int training) {
    if (check_cfg(plan_coarse, plan_fine, cfg) != NERFHIP_OK || n_rays < 0) return -1;
    return layout(plan_coarse, plan_fine, cfg, n_rays, training).total;
}

extern "C" int nerfhip_render_workspace_region(nerfhip_plan_t plan_coarse, nerfhip_plan_t plan_fine,
                                               const nerfhip_render_cfg* cfg, int64_t n_rays, int training, const char* name,
                                               int64_t* offset, int64_t* bytes) {
    int rc = check_cfg(plan_coarse, plan_fine, cfg);
    if (rc) return rc;
    NH_REQUIRE(name && offset && bytes && n_rays >= 0, "render_workspace_region: bad arguments");
    const Workspace w = layout(plan_coarse, plan_fine, cfg, n_rays, training);
    const int64_t nc = cfg->num_coarse, sf = nc + cfg->num_fine;
    struct {
        const char* name;
        int64_t off, bytes;
        bool fine;
    } regions[] = {{"z_coarse", w.z_c, n_rays * nc * 4, false},      {"raw_coarse", w.raw_c, n_rays * nc * 16, false},
                   {"weights_coarse", w.w_c, n_rays * nc * 4, false}, {"z_fine", w.z_f, n_rays * sf * 4, true},
                   {"raw_fine", w.raw_f, n_rays * sf * 16, true},
                   // (training layouts: each net's backward scratch -- nerfhip_plan_bwd_stats_offset points into it)
                   {"bwd_scratch_coarse", w.scratch_c, w.scratch_c_bytes, false},
                   {"bwd_scratch_fine", w.scratch_f, w.scratch_f_bytes, true}};
    for (const auto& r : regions)
        if (strcmp(name, r.name) == 0) {
            NH_REQUIRE(!r.fine || cfg->num_fine > 0, "render_workspace_region: %s needs num_fine > 0", name);
            NH_REQUIRE(training || strncmp(name, "bwd_", 4) != 0, "render_workspace_region: %s exists in a training layout only", name);
            *offset = r.off;
            *bytes = r.bytes;
            return NERFHIP_OK;
        }
    NH_REQUIRE(false, "render_workspace_region: unknown region '%s'", name);
}

extern "C" int nerfhip_render_fwd_parts(nerfhip_plan_t pc, nerfhip_plan_t pf, const nerfhip_render_cfg* cfg,
                                        const float* rays, int64_t n, const float* packed_c, const float* packed_f,
                                        const float* t_vals, const float* u_det, const nerfhip_render_rand* rnd,
                                        uint64_t seed, uint64_t ray_offset, const nerfhip_render_out* out, void* workspace,
                                        int64_t workspace_bytes, int training, int parts, nerfhip_stream_t stream) {
    int rc = check_cfg(pc, pf, cfg);
    if (rc) return rc;
    NH_REQUIRE(rays && packed_c && t_vals && out && workspace && n >= 0, "render_fwd: bad arguments");
    NH_REQUIRE(parts >= 1 && parts <= 3, "render_fwd: parts must be a combination of NERFHIP_PART_COARSE | NERFHIP_PART_FINE");
    NH_REQUIRE(cfg->num_fine == 0 || packed_f, "render_fwd: packed_fine is NULL");
    if (n == 0) return NERFHIP_OK;
    // (the regions a forward touches are the same in both training layouts; the size is checked against the layout the caller
    // names -- 1: one set of backward buffers per net, 2: shared -- so that an undersized workspace for the backward it intends
    // is refused here, not at backward time)
    NH_REQUIRE(training >= 0 && training <= 2, "render_fwd: training must be 0, 1 (two sets of backward buffers) or 2 (shared)");
    const Workspace w = layout(pc, pf, cfg, n, training);
    NH_REQUIRE(workspace_bytes >= w.total, "render_fwd: workspace too small (%lld < %lld)", (long long)workspace_bytes,
               (long long)w.total);
    char* ws = (char*)workspace;
    const int nc = cfg->num_coarse, nf = cfg->num_fine, sf = nc + nf, stride = cfg->ray_stride;
    nerfhip_render_rand none = {nullptr, nullptr, nullptr, nullptr};
    const nerfhip_render_rand* r = rnd ? rnd : &none;
    float* z_c = (float*)(ws + w.z_c);
    float* raw_c = (float*)(ws + w.raw_c);
    float* w_c = (float*)(ws + w.w_c);

    NhMlpInput in;
    memset(&in, 0, sizeof(in));
    in.mode = 1;
    in.rays = rays;
    in.ray_stride = stride;
    if (parts & NERFHIP_PART_COARSE) {
        rc = nerfhip_stratified_z(rays, stride, n, t_vals, nc, cfg->lindisp, cfg->perturb, r->t_rand, seed, ray_offset, z_c,
                                  stream);
        if (rc) return rc;
        in.z = z_c;
        in.S = nc;
        // (a plan whose backward recomputes the stash for the samples it keeps -- nerfhip_plan_set_bwd_compaction(plan, 2) -- runs the
        // stash-free forward here)
        rc = training ? nh_mlp_forward_training(pc, packed_c, in, n * nc, raw_c, (float*)(ws + w.stash_c), stream)
                      : nh_mlp_forward(pc, packed_c, in, n * nc, raw_c, nullptr, stream);
        if (rc) return rc;
        rc = nerfhip_volume_render_fwd(raw_c, z_c, rays + 3, stride, n, nc, cfg->noise_std, r->noise_coarse, seed, 1u,
                                       ray_offset, cfg->white_background, out->rgb_coarse, out->disp_coarse,
                                       out->acc_coarse, w_c, out->depth_coarse, stream);
        if (rc) return rc;
    }
    if (nf > 0 && (parts & NERFHIP_PART_FINE)) {
        float* z_f = (float*)(ws + w.z_f);
        float* raw_f = (float*)(ws + w.raw_f);
        const int det = cfg->perturb ? 0 : 1;  // det = (perturb == 0.0), nerf/train_utils.py:101
        NH_REQUIRE(!det || r->u || u_det, "render_fwd: perturb == 0 needs u_det");
        rc = nerfhip_hierarchical_z(z_c, w_c, n, nc, r->u, det, u_det, nf, seed, ray_offset, nullptr, z_f, stream);
        if (rc) return rc;
        in.z = z_f;
        in.S = sf;
        rc = training ? nh_mlp_forward_training(pf, packed_f, in, n * sf, raw_f, (float*)(ws + w.stash_f), stream)
                      : nh_mlp_forward(pf, packed_f, in, n * sf, raw_f, nullptr, stream);
        if (rc) return rc;
        rc = nerfhip_volume_render_fwd(raw_f, z_f, rays + 3, stride, n, sf, cfg->noise_std, r->noise_fine, seed, 3u,
                                       ray_offset, cfg->white_background, out->rgb_fine, out->disp_fine, out->acc_fine,
                                       nullptr, out->depth_fine, stream);
        if (rc) return rc;
    }
    return NERFHIP_OK;
}

extern "C" int nerfhip_render_fwd(nerfhip_plan_t pc, nerfhip_plan_t pf, const nerfhip_render_cfg* cfg, const float* rays,
                                  int64_t n, const float* packed_c, const float* packed_f, const float* t_vals,
                                  const float* u_det, const nerfhip_render_rand* rnd, uint64_t seed, uint64_t ray_offset,
                                  const nerfhip_render_out* out, void* workspace, int64_t workspace_bytes, int training,
                                  nerfhip_stream_t stream) {
    return nerfhip_render_fwd_parts(pc, pf, cfg, rays, n, packed_c, packed_f, t_vals, u_det, rnd, seed, ray_offset, out,
                                    workspace, workspace_bytes, training, NERFHIP_PART_COARSE | NERFHIP_PART_FINE, stream);
}

namespace {

// d(loss)/d(rays) of one net's pass: dL/d(encoded input) from the d(pre-activation) images its backward just left in the
// scratch (nerfhip_mlp_bwd_input), then the positional encoding's backward and pts = ro + rd * z (nh_ray_grad).
int ray_grad_of_pass(nerfhip_plan* p, const float* params, const float* rays, int stride, int64_t n, const float* z, int S,
                     const float* scratch, const float* g_norm, void* tmp, int64_t tmp_bytes, float* g_rays, int accumulate,
                     nerfhip_stream_t stream) {
    const int64_t M = n * S, need = M * (int64_t)(p->Dx + p->Dd) * 4;
    NH_REQUIRE(params && tmp && tmp_bytes >= need, "render_bwd: the ray gradient needs the flat parameters and %lld bytes of tmp",
               (long long)need);
    int rc = nerfhip_mlp_bwd_input(p, params, M, scratch, (float*)tmp, stream);
    if (rc) return rc;
    return nh_ray_grad(rays, stride, n, z, S, (const float*)tmp, p->Dx, p->Dd, p->cfg.include_input_xyz ? 1 : 0,
                       (p->view && p->cfg.include_input_dir) ? 1 : 0, p->cfg.num_encoding_fn_xyz,
                       p->view ? p->cfg.num_encoding_fn_dir : 0, p->freqs_xyz, p->freqs_dir, g_norm, g_rays, accumulate, stream);
}

}  // namespace

extern "C" int64_t nerfhip_render_bwd_rays_tmp_bytes(nerfhip_plan_t pc, nerfhip_plan_t pf, const nerfhip_render_cfg* cfg,
                                                     int64_t n) {
    if (check_cfg(pc, pf, cfg) != NERFHIP_OK || n < 0) return -1;
    int64_t b = n * cfg->num_coarse * (int64_t)(pc->Dx + pc->Dd) * 4;
    if (cfg->num_fine > 0) {
        const int64_t f = n * (cfg->num_coarse + cfg->num_fine) * (int64_t)(pf->Dx + pf->Dd) * 4;
        if (f > b) b = f;
    }
    return b;
}

extern "C" int nerfhip_render_bwd_rays(nerfhip_plan_t pc, nerfhip_plan_t pf, const nerfhip_render_cfg* cfg, const float* rays,
                                       int64_t n, const float* packed_c, const float* packed_f, const nerfhip_render_rand* rnd,
                                       uint64_t seed, uint64_t ray_offset, const nerfhip_render_cotangents* g, void* workspace,
                                       int64_t workspace_bytes, float* g_params_c, float* g_params_f, int parts,
                                       const float* params_c, const float* params_f, void* tmp, int64_t tmp_bytes,
                                       float* g_rays, nerfhip_stream_t stream) {
    int rc = check_cfg(pc, pf, cfg);
    if (rc) return rc;
    NH_REQUIRE(rays && packed_c && g && workspace && n > 0, "render_bwd: bad arguments");
    const int shared = parts & NERFHIP_PART_SHARED_BWD;
    parts &= ~NERFHIP_PART_SHARED_BWD;
    NH_REQUIRE(parts >= 1 && parts <= 3, "render_bwd: parts must be a combination of NERFHIP_PART_COARSE | NERFHIP_PART_FINE");
    const Workspace w = layout(pc, pf, cfg, n, shared ? 2 : 1);
    NH_REQUIRE(workspace_bytes >= w.total, "render_bwd: workspace too small (%lld < %lld)", (long long)workspace_bytes,
               (long long)w.total);
    char* ws = (char*)workspace;
    const int nc = cfg->num_coarse, nf = cfg->num_fine, sf = nc + nf, stride = cfg->ray_stride;
    nerfhip_render_rand none = {nullptr, nullptr, nullptr, nullptr};
    const nerfhip_render_rand* r = rnd ? rnd : &none;
    int wrote_rays = 0;  // the first pass that runs overwrites g_rays, the second accumulates
    if (nf > 0 && (parts & NERFHIP_PART_FINE)) {
        NH_REQUIRE(packed_f && g_params_f && (g->g_rgb_fine || g->g_acc_fine || g->g_depth_fine),
                   "render_bwd: fine arguments missing");
        float* g_raw = (float*)(ws + w.g_raw_f);
        rc = nh_volume_render_bwd((const float*)(ws + w.raw_f), (const float*)(ws + w.z_f), rays + 3, stride, n, sf,
                                  cfg->noise_std, r->noise_fine, seed, 3u, ray_offset, cfg->white_background, g->g_rgb_fine,
                                  g->g_depth_fine, g->g_acc_fine, nullptr, g_raw, g_rays ? (float*)(ws + w.gnorm_f) : nullptr, stream);
        if (rc) return rc;
        if (nh_mlp_recomputes(pf, n * sf)) {
            NhMlpInput in;
            memset(&in, 0, sizeof(in));
            in.mode = 1, in.rays = rays, in.ray_stride = stride, in.z = (const float*)(ws + w.z_f), in.S = sf;
            rc = nh_mlp_backward_recompute(pf, packed_f, in, g_raw, n * sf, (float*)(ws + w.stash_f), (float*)(ws + w.scratch_f),
                                           w.scratch_f_bytes, g_params_f, g_rays != nullptr, stream);
        } else {
            rc = nh_mlp_backward(pf, packed_f, g_raw, n * sf, (const float*)(ws + w.stash_f), (float*)(ws + w.scratch_f),
                                 w.scratch_f_bytes, g_params_f, stream);
        }
        if (rc) return rc;
        if (g_rays) {
            rc = ray_grad_of_pass(pf, params_f, rays, stride, n, (const float*)(ws + w.z_f), sf, (const float*)(ws + w.scratch_f),
                                  (const float*)(ws + w.gnorm_f), tmp, tmp_bytes, g_rays, wrote_rays, stream);
            if (rc) return rc;
            wrote_rays = 1;
        }
    }
    if (parts & NERFHIP_PART_COARSE) {
        NH_REQUIRE(g_params_c && (g->g_rgb_coarse || g->g_acc_coarse || g->g_depth_coarse),
                   "render_bwd: coarse arguments missing");
        float* g_raw = (float*)(ws + w.g_raw_c);
        rc = nh_volume_render_bwd((const float*)(ws + w.raw_c), (const float*)(ws + w.z_c), rays + 3, stride, n, nc,
                                  cfg->noise_std, r->noise_coarse, seed, 1u, ray_offset, cfg->white_background, g->g_rgb_coarse,
                                  g->g_depth_coarse, g->g_acc_coarse, nullptr, g_raw, g_rays ? (float*)(ws + w.gnorm_c) : nullptr,
                                  stream);
        if (rc) return rc;
        if (nh_mlp_recomputes(pc, n * nc)) {
            NhMlpInput in;
            memset(&in, 0, sizeof(in));
            in.mode = 1, in.rays = rays, in.ray_stride = stride, in.z = (const float*)(ws + w.z_c), in.S = nc;
            rc = nh_mlp_backward_recompute(pc, packed_c, in, g_raw, n * nc, (float*)(ws + w.stash_c), (float*)(ws + w.scratch_c),
                                           w.scratch_c_bytes, g_params_c, g_rays != nullptr, stream);
        } else {
            rc = nh_mlp_backward(pc, packed_c, g_raw, n * nc, (const float*)(ws + w.stash_c), (float*)(ws + w.scratch_c),
                                 w.scratch_c_bytes, g_params_c, stream);
        }
        if (rc) return rc;
        if (g_rays) {
            rc = ray_grad_of_pass(pc, params_c, rays, stride, n, (const float*)(ws + w.z_c), nc, (const float*)(ws + w.scratch_c),
                                  (const float*)(ws + w.gnorm_c), tmp, tmp_bytes, g_rays, wrote_rays, stream);
            if (rc) return rc;
            wrote_rays = 1;
        }
    }
    return NERFHIP_OK;
}

extern "C" int nerfhip_render_bwd_parts(nerfhip_plan_t pc, nerfhip_plan_t pf, const nerfhip_render_cfg* cfg,
                                        const float* rays, int64_t n, const float* packed_c, const float* packed_f,
                                        const nerfhip_render_rand* rnd, uint64_t seed, uint64_t ray_offset,
                                        const nerfhip_render_cotangents* g, void* workspace, int64_t workspace_bytes,
                                        float* g_params_c, float* g_params_f, int parts, nerfhip_stream_t stream) {
    return nerfhip_render_bwd_rays(pc, pf, cfg, rays, n, packed_c, packed_f, rnd, seed, ray_offset, g, workspace, workspace_bytes,
                                   g_params_c, g_params_f, parts, nullptr, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int nerfhip_render_bwd(nerfhip_plan_t pc, nerfhip_plan_t pf, const nerfhip_render_cfg* cfg, const float* rays,
                                  int64_t n, const float* packed_c, const float* packed_f, const nerfhip_render_rand* rnd,
                                  uint64_t seed, uint64_t ray_offset, const float* g_rgb_c, const float* g_rgb_f,
                                  void* workspace, int64_t workspace_bytes, float* g_params_c, float* g_params_f,
                                  nerfhip_stream_t stream) {
    NH_REQUIRE(g_rgb_c && g_params_c, "render_bwd: bad arguments");
    nerfhip_render_cotangents g = {g_rgb_c, nullptr, nullptr, g_rgb_f, nullptr, nullptr};
    return nerfhip_render_bwd_parts(pc, pf, cfg, rays, n, packed_c, packed_f, rnd, seed, ray_offset, &g, workspace,
                                    workspace_bytes, g_params_c, g_params_f,
                                    NERFHIP_PART_COARSE | NERFHIP_PART_FINE | NERFHIP_PART_SHARED_BWD, stream);
}

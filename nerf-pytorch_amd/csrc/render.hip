// render.hip -- sigma/alpha compositing (volume_render_radiance_field, nerf/volume_rendering_utils.py:6-53) forward
// and its closed-form backward.  One wavefront owns one ray; samples are walked in chunks of 64 (one per lane) with
// a wave-level fp64 product scan carrying the transmittance across chunks, so any samples-per-ray count works and
// all global accesses are coalesced.
#include "nh_host.h"

struct RenderSample {
    float alpha, b, e, dist, sig_in;
};

// per-sample quantities of lines :17-39 of the reference
NH_DEVICE RenderSample nh_render_sample(const float* __restrict__ raw, const float* __restrict__ z, int64_t ray, int s,
                                        int i, float norm, float noise_std, const float* __restrict__ noise,
                                        uint64_t seed, uint32_t rng_stream, uint64_t ray_offset) {
    RenderSample r;
    int64_t g = ray * s + i;
    float zi = z[g];
    float d = (i + 1 < s) ? (z[g + 1] - zi) : 1e10f;
    r.dist = d * norm;
    float nz = 0.0f;
    if (noise_std > 0.0f) {
        float dr = noise ? noise[g] : nh_rand_normal(seed, rng_stream, (ray_offset + (uint64_t)ray) * (uint64_t)s + i);
        nz = dr * noise_std;
    }
    r.sig_in = raw[g * 4 + 3] + nz;
    float sigma = r.sig_in > 0.0f ? r.sig_in : 0.0f;
    if (r.sig_in != r.sig_in) sigma = r.sig_in;  // relu propagates NaN
    r.e = expf(-sigma * r.dist);
    r.alpha = 1.0f - r.e;
    r.b = 1.0f - r.alpha + 1e-10f;
    return r;
}

NH_DEVICE float nh_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

NH_KERNEL void k_volume_render_fwd(const float* __restrict__ raw, const float* __restrict__ z,
                                   const float* __restrict__ rd, int rd_stride, int64_t n, int s, float noise_std,
                                   const float* __restrict__ noise, uint64_t seed, uint32_t rng_stream,
                                   uint64_t ray_offset, int white, float* __restrict__ rgb, float* __restrict__ disp,
                                   float* __restrict__ acc, float* __restrict__ weights, float* __restrict__ depth) {
    const int64_t ray = blockIdx.x;
    const int lane = nh_lane();
    const float dx = rd[ray * rd_stride], dy = rd[ray * rd_stride + 1], dz = rd[ray * rd_stride + 2];
    const float norm = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));  // == torch CPU norm(p=2), bit-exact
    double carry = 1.0;
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
    for (int base = 0; base < s; base += 64) {
        const int i = base + lane;
        const bool valid = i < s;
        RenderSample q;
        q.alpha = 0.f;
        q.b = 1.f;
        if (valid) q = nh_render_sample(raw, z, ray, s, i, norm, noise_std, noise, seed, rng_stream, ray_offset);
        double p = (double)q.b;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            double o = nh_shfl_up_d(p, d);
            if (lane >= d) p *= o;
        }
        double excl = nh_shfl_up_d(p, 1);
        if (lane == 0) excl = 1.0;
        const float T = (float)(carry * excl);
        carry *= nh_shfl_d(p, 63);
        if (valid) {
            const float w = q.alpha * T;
            const int64_t g = ray * s + i;
            if (weights) weights[g] = w;
            sr += w * nh_sigmoid(raw[g * 4 + 0]);
            sg += w * nh_sigmoid(raw[g * 4 + 1]);
            sb += w * nh_sigmoid(raw[g * 4 + 2]);
            sd += w * z[g];
            sa += w;
        }
    }
    sr = nh_wave_sum(sr);
    sg = nh_wave_sum(sg);
    sb = nh_wave_sum(sb);
    sd = nh_wave_sum(sd);
    sa = nh_wave_sum(sa);
    if (lane == 0) {
        if (white) {
            float bg = 1.0f - sa;
            sr = sr + bg;
            sg = sg + bg;
            sb = sb + bg;
        }
        if (rgb) {
            rgb[ray * 3 + 0] = sr;
            rgb[ray * 3 + 1] = sg;
            rgb[ray * 3 + 2] = sb;
        }
        if (depth) depth[ray] = sd;
        if (acc) acc[ray] = sa;
        if (disp) {
            float q = sd / sa;                                      // 0/0 -> NaN, as in the reference
            float m = (q != q) ? q : (q > 1e-10f ? q : 1e-10f);     // torch.max propagates NaN
            disp[ray] = 1.0f / m;
        }
    }
}

extern "C" int nerfhip_volume_render_fwd(const float* raw, const float* z, const float* rd, int rd_stride, int64_t n,
                                         int s, float noise_std, const float* noise, uint64_t seed, uint32_t rng_stream,
                                         uint64_t ray_offset, int white_background, float* rgb, float* disp, float* acc,
                                         float* weights, float* depth, nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(raw && z && rd && rd_stride >= 3 && n >= 0 && s > 0, "volume_render_fwd: bad arguments");
    NH_LAUNCH(k_volume_render_fwd, n, 64, 0, stream, raw, z, rd, rd_stride, n, s, noise_std, noise, seed, rng_stream,
              ray_offset, white_background, rgb, disp, acc, weights, depth);
    return nh_launch_status("volume_render_fwd");
}

// Backward.  With c = sigmoid(raw[:3]), a = alpha, b = 1 - a + 1e-10, T = exclusive cumprod(b),
// w = a T and G_i = dL/dw_i:   dL/da_i = G_i T_i - (sum_{k>i} G_k w_k) / b_i ;
// dL/draw_i[3] = dL/da_i * dist_i * exp(-sigma_i dist_i) * [raw_i[3] + noise_i > 0] ;  dL/draw_i[c] = g_c w_i c (1-c).
NH_KERNEL void k_volume_render_bwd(const float* __restrict__ raw, const float* __restrict__ z,
                                   const float* __restrict__ rd, int rd_stride, int64_t n, int s, float noise_std,
                                   const float* __restrict__ noise, uint64_t seed, uint32_t rng_stream,
                                   uint64_t ray_offset, int white, const float* __restrict__ g_rgb,
                                   const float* __restrict__ g_depth, const float* __restrict__ g_acc,
                                   const float* __restrict__ g_weights, float* __restrict__ g_raw,
                                   float* __restrict__ g_norm) {
    NH_DYN_LDS(lds_raw);
    float* sT = (float*)lds_raw;  // [s] transmittance
    const int64_t ray = blockIdx.x;
    const int lane = nh_lane();
    const float dx = rd[ray * rd_stride], dy = rd[ray * rd_stride + 1], dz = rd[ray * rd_stride + 2];
    const float norm = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));  // == torch CPU norm(p=2), bit-exact
    // pass 1: transmittance
    double carry = 1.0;
    for (int base = 0; base < s; base += 64) {
        const int i = base + lane;
        const bool valid = i < s;
        RenderSample q;
        q.b = 1.f;
        if (valid) q = nh_render_sample(raw, z, ray, s, i, norm, noise_std, noise, seed, rng_stream, ray_offset);
        double p = (double)q.b;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            double o = nh_shfl_up_d(p, d);
            if (lane >= d) p *= o;
        }
        double excl = nh_shfl_up_d(p, 1);
        if (lane == 0) excl = 1.0;
        if (valid) sT[i] = (float)(carry * excl);
        carry *= nh_shfl_d(p, 63);
    }
    nh_block_sync();
    const float gr = g_rgb ? g_rgb[ray * 3 + 0] : 0.f;
    const float gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.f;
    const float gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.f;
    const float gd = g_depth ? g_depth[ray] : 0.f;
    const float ga = (g_acc ? g_acc[ray] : 0.f) - (white ? ((gr + gg) + gb) : 0.f);
    // pass 2: walk the chunks back to front carrying the suffix sum of G_k w_k
    double suffix = 0.0;
    float gn = 0.0f;  // dL/d||rd|| (optional output): dists_i = (z_{i+1} - z_i) * ||rd||  (volume_rendering_utils.py:24)
    const int nchunks = (s + 63) / 64;
    for (int ch = nchunks - 1; ch >= 0; --ch) {
        const int i = ch * 64 + lane;
        const bool valid = i < s;
        RenderSample q;
        q.alpha = 0.f;
        q.b = 1.f;
        q.e = 0.f;
        q.dist = 0.f;
        q.sig_in = 0.f;
        float T = 0.f, w = 0.f, G = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
        if (valid) {
            q = nh_render_sample(raw, z, ray, s, i, norm, noise_std, noise, seed, rng_stream, ray_offset);
            const int64_t g = ray * s + i;
            T = sT[i];
            w = q.alpha * T;
            cr = nh_sigmoid(raw[g * 4 + 0]);
            cg = nh_sigmoid(raw[g * 4 + 1]);
            cb = nh_sigmoid(raw[g * 4 + 2]);
            G = ((gr * cr + gg * cg) + gb * cb) + gd * z[g] + ga + (g_weights ? g_weights[g] : 0.f);
        }
        double v = (double)G * (double)w;
        double incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            double o = nh_shfl_down_d(incl, d);
            if (lane + d < 64) incl += o;
        }
        const double after = suffix + (incl - v);  // sum over k > i
        suffix += nh_shfl_d(incl, 0);
        if (valid) {
            const int64_t g = ray * s + i;
            const float da = G * T - (float)after / q.b;
            const float mask = q.sig_in > 0.0f ? 1.0f : 0.0f;
            float4 o;
            o.x = gr * w * cr * (1.0f - cr);
            o.y = gg * w * cg * (1.0f - cg);
            o.z = gb * w * cb * (1.0f - cb);
            o.w = mask != 0.0f ? da * (q.dist * q.e) : 0.0f;
            *(float4*)(g_raw + g * 4) = o;
            // dL/d dist_i = dL/da_i * sigma_i * exp(-sigma_i dist_i); d dist_i / d||rd|| = dist_i / ||rd||.  (The 1e10 tail:
            // sigma > 0 gives exp(-inf) = 0, sigma == 0 is masked.)
            if (mask != 0.0f && q.e > 0.0f) gn += da * (q.sig_in * q.e) * (q.dist / norm);
        }
    }
    if (g_norm) {
        gn = nh_wave_sum(gn);
        if (lane == 0) g_norm[ray] = gn;
    }
}

// g_norm (optional, dev [n]): additionally dL/d||rd|| of every ray -- the path through dists (the fused render's gradient
// w.r.t. the ray directions)
int nh_volume_render_bwd(const float* raw, const float* z, const float* rd, int rd_stride, int64_t n, int s, float noise_std,
                         const float* noise, uint64_t seed, uint32_t rng_stream, uint64_t ray_offset, int white_background,
                         const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_weights, float* g_raw,
                         float* g_norm, nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(raw && z && rd && g_raw && rd_stride >= 3 && n >= 0 && s > 0, "volume_render_bwd: bad arguments");
    NH_REQUIRE(s <= 8192, "volume_render_bwd: at most 8192 samples per ray");
    NH_LAUNCH(k_volume_render_bwd, n, 64, (size_t)s * sizeof(float), stream, raw, z, rd, rd_stride, n, s, noise_std,
              noise, seed, rng_stream, ray_offset, white_background, g_rgb, g_depth, g_acc, g_weights, g_raw, g_norm);
    return nh_launch_status("volume_render_bwd");
}

extern "C" int nerfhip_volume_render_bwd(const float* raw, const float* z, const float* rd, int rd_stride, int64_t n,
                                         int s, float noise_std, const float* noise, uint64_t seed, uint32_t rng_stream,
                                         uint64_t ray_offset, int white_background, const float* g_rgb,
                                         const float* g_depth, const float* g_acc, const float* g_weights, float* g_raw,
                                         nerfhip_stream_t stream) {
    return nh_volume_render_bwd(raw, z, rd, rd_stride, n, s, noise_std, noise, seed, rng_stream, ray_offset, white_background,
                                g_rgb, g_depth, g_acc, g_weights, g_raw, nullptr, stream);
}

// ---- gradient w.r.t. the rays (nerf/train_utils.py:67,107: pts = ro + rd * z is differentiable under autograd) -----------
// One wavefront per ray; lane l walks samples l, l + 64, ...  g_x: dev [n*S, dx + dd] = dL/d(encoded input) of the MLP in
// the reference's column layout [x | sin(f0 x) | cos(f0 x) | sin(f1 x) | ...] (nerfhip_mlp_bwd_input).  The positional
// encoding's backward (nerf/nerf_helpers.py:130-157): dL/dp_c = g[c] + sum_f f * (cos(f p_c) g_sin[f][c] - sin(f p_c) g_cos[f][c]).
// g_rays [n, stride]: columns 0..2 (origin) += sum_s dL/dpts, 3..5 (direction) += sum_s z_s dL/dpts + g_norm * rd / ||rd||,
// 8..10 (viewdirs) += sum_s dL/d(dir); near / far (6, 7) carry no gradient here (the depths are treated as constants).
struct RayGradArgs {
    const float* rays;
    int stride;
    int64_t n;
    const float* z;
    int S;
    const float* g_x;
    int dx, dd, inc_x, inc_d, Lx, Ld;
    float fx[16], fd[16];
    const float* g_norm;
    float* g_rays;
    int accumulate;
};

NH_DEVICE float nh_posenc_bwd(const float* __restrict__ g, float v, int c, int inc, int L, const float* freqs) {
    float s = inc ? g[c] : 0.0f;
    const float* q = g + (inc ? 3 : 0);
    for (int f = 0; f < L; ++f) {
        float sn, cs;
        nh_sincos(v * freqs[f], &sn, &cs);
        s += freqs[f] * (cs * q[6 * f + c] - sn * q[6 * f + 3 + c]);
    }
    return s;
}

NH_KERNEL void k_ray_grad(RayGradArgs a) {
    const int64_t ray = blockIdx.x;
    const int lane = nh_lane();
    const float* rr = a.rays + ray * a.stride;
    const float o[3] = {rr[0], rr[1], rr[2]}, d[3] = {rr[3], rr[4], rr[5]};
    float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f};
    for (int i = lane; i < a.S; i += 64) {
        const int64_t m = ray * a.S + i;
        const float zz = a.z[m];
        const float* g = a.g_x + m * (int64_t)(a.dx + a.dd);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gp = nh_posenc_bwd(g, o[c] + d[c] * zz, c, a.inc_x, a.Lx, a.fx);
            go[c] += gp;
            gd[c] += gp * zz;
            if (a.dd > 0) gv[c] += nh_posenc_bwd(g + a.dx, rr[8 + c], c, a.inc_d, a.Ld, a.fd);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        go[c] = nh_wave_sum(go[c]);
        gd[c] = nh_wave_sum(gd[c]);
        gv[c] = nh_wave_sum(gv[c]);
    }
    if (lane == 0) {
        float* out = a.g_rays + ray * a.stride;
        const float norm = sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
        const float gn = a.g_norm ? a.g_norm[ray] : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float dirterm = gd[c] + (norm > 0.0f ? gn * d[c] / norm : 0.0f);
            if (a.accumulate) {
                out[c] += go[c];
                out[3 + c] += dirterm;
                if (a.dd > 0) out[8 + c] += gv[c];
            } else {
                out[c] = go[c];
                out[3 + c] = dirterm;
                if (a.dd > 0) out[8 + c] = gv[c];
            }
        }
        if (!a.accumulate) {
            out[6] = 0.0f;
            out[7] = 0.0f;
            for (int c = (a.dd > 0 ? 11 : 8); c < a.stride; ++c) out[c] = 0.0f;
        }
    }
}

int nh_ray_grad(const float* rays, int stride, int64_t n, const float* z, int S, const float* g_x, int dx, int dd, int inc_x,
                int inc_d, int Lx, int Ld, const float* fx, const float* fd, const float* g_norm, float* g_rays, int accumulate,
                nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;
    NH_REQUIRE(rays && z && g_x && g_rays && S > 0 && stride >= (dd > 0 ? 11 : 8) && Lx <= 16 && Ld <= 16, "ray_grad: bad arguments");
    RayGradArgs a;
    memset(&a, 0, sizeof(a));
    a.rays = rays;
    a.stride = stride;
    a.n = n;
    a.z = z;
    a.S = S;
    a.g_x = g_x;
    a.dx = dx;
    a.dd = dd;
    a.inc_x = inc_x;
    a.inc_d = inc_d;
    a.Lx = Lx;
    a.Ld = Ld;
    for (int k = 0; k < 16; ++k) {
        a.fx[k] = fx[k];
        a.fd[k] = fd[k];
    }
    a.g_norm = g_norm;
    a.g_rays = g_rays;
    a.accumulate = accumulate;
    NH_LAUNCH(k_ray_grad, n, 64, 0, stream, a);
    return nh_launch_status("ray_grad");
}

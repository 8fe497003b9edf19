// elementwise.hip -- the HBM-bound, per-ray / per-sample pieces of the path: ray generation, NDC, ray packing,
// positional encoding (stand-alone form), stratified depths, exclusive cumprod, RNG fill, weight gather,
// MSE loss and Adam.  All arithmetic is fp32 with the reference's operation order (-ffp-contract=off).
#include <stdarg.h>
#include <stdio.h>

#include "nh_diag.h"
#include "nh_host.h"
#include "nh_rays.h"

// ---- error string ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void nh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* nerfhip_last_error(void) { return g_err; }
extern "C" int nerfhip_version(void) {
#ifdef NH_DIAG  // (make variant: an A/B or diagnostic build -- nh_diag.h; the Python package refuses it)
    return 104 + NH_DIAG_VERSION_FLAG;
#else
    return 104;  // (102: round 6 -- the compacted backward's three entry points; 103: + the fused backward modes 3 / 4 of 64-wide nets; 104: + mode 5)
#endif
}
extern "C" int nerfhip_is_emulated(void) {
#ifdef NERFHIP_EMU
    return 1;
#else
    return 0;
#endif
}

// ---- per-kernel profiling (HIP events on the launch stream) ---------------------------------------------------------
#include <map>
#include <string>
#include <vector>
namespace {
bool g_prof_on = false;
#ifndef NERFHIP_EMU
struct ProfRec {
    std::string name;
    hipEvent_t a, b;
};
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;  // events of reported launches, reused: creating a pair per launch costs ~7 us of host time
hipEvent_t prof_event() {
    hipEvent_t e = nullptr;
    if (!g_prof_pool.empty()) {
        e = g_prof_pool.back();
        g_prof_pool.pop_back();
    } else {
        (void)hipEventCreate(&e);
    }
    return e;
}
unsigned long long* g_clk_dev = nullptr;  // [NH_CLK_KERNELS][4]: shader cycles, 100 MHz ticks, workgroups, unused
#endif
}  // namespace
unsigned long long* nh_prof_clock_slot(int kind) {
#ifndef NERFHIP_EMU
    return (g_prof_on && g_clk_dev) ? g_clk_dev + 4 * kind : nullptr;
#else
    (void)kind;
    return nullptr;
#endif
}
void nh_prof_events(const char* name, void** start, void** stop) {
    *start = *stop = nullptr;
#ifndef NERFHIP_EMU
    if (!g_prof_on) return;
    ProfRec r;
    r.name = name;
    r.a = prof_event();
    r.b = prof_event();
    if (!r.a || !r.b) {  // (one of the pair could not be created: the other goes back to the pool, the launch stays untimed)
        if (r.a) g_prof_pool.push_back(r.a);
        if (r.b) g_prof_pool.push_back(r.b);
        return;
    }
    *start = (void*)r.a;
    *stop = (void*)r.b;
    g_prof.push_back(std::move(r));
#else
    (void)name;
#endif
}
extern "C" int nerfhip_profile_enable(int on) {
#ifndef NERFHIP_EMU
    if (on && !g_clk_dev) {  // 96 bytes of device counters, private to the library, kept for the life of the process
        if (hipMalloc((void**)&g_clk_dev, sizeof(unsigned long long) * 4 * NH_CLK_KERNELS) != hipSuccess) g_clk_dev = nullptr;
        if (g_clk_dev) (void)hipMemset(g_clk_dev, 0, sizeof(unsigned long long) * 4 * NH_CLK_KERNELS);
    }
#endif
    g_prof_on = on != 0;
    return NERFHIP_OK;
}
// Pre-creates the event pairs of `launches` launches, so that none is created inside a timed region.
extern "C" int nerfhip_profile_reserve(int64_t launches) {
    NH_REQUIRE(launches >= 0 && launches <= (1 << 20), "profile_reserve: bad arguments");
#ifndef NERFHIP_EMU
    while ((int64_t)g_prof_pool.size() < 2 * launches) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) {
            nh_set_error("profile_reserve: hipEventCreate failed");
            return NERFHIP_ERR_LAUNCH;
        }
        g_prof_pool.push_back(e);
    }
    g_prof.reserve(g_prof.size() + (size_t)launches);
#endif
    return NERFHIP_OK;
}
// Shader clock the MLP kernels ran at while profiling was enabled: out[3 * k + {0,1,2}] = shader-clock cycles, 100 MHz
// reference ticks and workgroups summed over all workgroups of kernel k (0 forward, 1 data gradient, 2 weight gradient).
// Waits for the device, then clears the counters.
extern "C" int nerfhip_profile_clocks(uint64_t* out9) {
    NH_REQUIRE(out9, "profile_clocks: bad arguments");
    for (int i = 0; i < 3 * NH_CLK_KERNELS; ++i) out9[i] = 0;
#ifndef NERFHIP_EMU
    if (!g_clk_dev) return NERFHIP_OK;
    unsigned long long host[4 * NH_CLK_KERNELS];
    if (hipMemcpy(host, g_clk_dev, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) {
        nh_set_error("profile_clocks: copy failed");
        return NERFHIP_ERR_LAUNCH;
    }
    (void)hipMemset(g_clk_dev, 0, sizeof(host));
    for (int k = 0; k < NH_CLK_KERNELS; ++k)
        for (int c = 0; c < 3; ++c) out9[3 * k + c] = host[4 * k + c];
#endif
    return NERFHIP_OK;
}
// Waits for the recorded events, writes "kernel_name launches total_ms\n" lines into buf and clears the records.
extern "C" int nerfhip_profile_report(char* buf, int64_t cap) {
    NH_REQUIRE(buf && cap > 0, "profile_report: bad arguments");
    buf[0] = 0;
#ifndef NERFHIP_EMU
    std::map<std::string, std::pair<int64_t, double>> acc;
    for (ProfRec& r : g_prof) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
        auto& e = acc[r.name];
        e.first += 1;
        e.second += ms;
    }
    g_prof.clear();
    int64_t used = 0;
    for (auto& kv : acc) {
        int w = snprintf(buf + used, (size_t)(cap - used), "%s %lld %.6f\n", kv.first.c_str(), (long long)kv.second.first,
                         kv.second.second);
        if (w < 0 || used + w >= cap) break;
        used += w;
    }
#endif
    return NERFHIP_OK;
}

// ---- K1 get_ray_bundle (nerf/nerf_helpers.py:67-110) ----------------------------------------------------------------
NH_KERNEL void k_ray_bundle(int height, int width, float focal, const float* __restrict__ c2w, int ld,
                            const int64_t* __restrict__ pixels, int64_t n, float* __restrict__ ro,
                            float* __restrict__ rd) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int64_t p = pixels ? pixels[idx] : idx;
    float o[3], d[3];
    nh_pinhole_ray(height, width, focal, c2w, ld, p / width, p % width, o, d);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rd[idx * 3 + c] = d[c];
        ro[idx * 3 + c] = o[c];
    }
}

extern "C" int nerfhip_ray_bundle(int height, int width, float focal, const float* c2w, int c2w_ld,
                                  const int64_t* pixels, int64_t n, float* ray_origins, float* ray_directions,
                                  nerfhip_stream_t stream) {
    NH_REQUIRE(height > 0 && width > 0 && c2w && ray_origins && ray_directions && c2w_ld >= 4,
               "ray_bundle: bad arguments");
    NH_REQUIRE(pixels || n == (int64_t)height * width, "ray_bundle: n must be height*width when pixels is NULL");
    if (n == 0) return NERFHIP_OK;
    NH_LAUNCH(k_ray_bundle, nh_ceil_div(n, 256), 256, 0, stream, height, width, focal, c2w, c2w_ld, pixels, n,
              ray_origins, ray_directions);
    return nh_launch_status("ray_bundle");
}

// ---- ndc_rays (nerf/nerf_helpers.py:170-197) -------------------------------------------------------------------------
NH_KERNEL void k_ndc_rays(NhNdc k, const float* __restrict__ ro, const float* __restrict__ rd, int64_t n,
                          float* __restrict__ oo, float* __restrict__ od) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[3] = {ro[i * 3], ro[i * 3 + 1], ro[i * 3 + 2]};
    float d[3] = {rd[i * 3], rd[i * 3 + 1], rd[i * 3 + 2]};
    nh_ndc_ray(k, o, d);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        oo[i * 3 + c] = o[c];
        od[i * 3 + c] = d[c];
    }
}

extern "C" int nerfhip_ndc_rays(float near, float cw, float ch, float two_near, float neg_two_near,
                                const float* rays_o, const float* rays_d, int64_t n, float* out_o, float* out_d,
                                nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(rays_o && rays_d && out_o && out_d && n >= 0, "ndc_rays: bad arguments");
    NhNdc k = {near, cw, ch, two_near, neg_two_near};
    NH_LAUNCH(k_ndc_rays, nh_ceil_div(n, 256), 256, 0, stream, k, rays_o, rays_d, n, out_o, out_d);
    return nh_launch_status("ndc_rays");
}

// Vector-Jacobian product of ndc_rays: the chain rule through the statements of nh_ndc_ray in reverse order, which is what
// autograd does to nerf/nerf_helpers.py:170-197 (t = -(near + oz)/dz; p = o + t d; the six outputs are rational in p, d).
NH_KERNEL void k_ndc_rays_bwd(NhNdc k, const float* __restrict__ ro, const float* __restrict__ rd,
                              const float* __restrict__ g_oo, const float* __restrict__ g_od, int64_t n,
                              float* __restrict__ g_ro, float* __restrict__ g_rd) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ox = ro[i * 3], oy = ro[i * 3 + 1], oz = ro[i * 3 + 2];
    const float dx = rd[i * 3], dy = rd[i * 3 + 1], dz = rd[i * 3 + 2];
    const float gO0 = g_oo[i * 3], gO1 = g_oo[i * 3 + 1], gO2 = g_oo[i * 3 + 2];
    const float gD0 = g_od[i * 3], gD1 = g_od[i * 3 + 1], gD2 = g_od[i * 3 + 2];
    const float t = -(k.near + oz) / dz;
    const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
    const float ipz = 1.0f / pz, idz = 1.0f / dz;
    // outputs -> p, d
    const float ax = k.cw * (gO0 - gD0), ay = k.ch * (gO1 - gD1);  // d/d(px/pz), d/d(py/pz)
    float gpx = ax * ipz, gpy = ay * ipz;
    float gpz = -(ax * px + ay * py + k.two_near * gO2 + k.neg_two_near * gD2) * ipz * ipz;
    float gdx = k.cw * gD0 * idz, gdy = k.ch * gD1 * idz;
    float gdz = -(k.cw * gD0 * dx + k.ch * gD1 * dy) * idz * idz;
    // p = o + t d
    const float gt = gpx * dx + gpy * dy + gpz * dz;
    gdx += t * gpx;
    gdy += t * gpy;
    gdz += t * gpz;
    // t = -(near + oz) / dz
    gpz += -gt * idz;                         // (g wrt oz: through p and through t)
    gdz += gt * (k.near + oz) * idz * idz;
    g_ro[i * 3] = gpx;
    g_ro[i * 3 + 1] = gpy;
    g_ro[i * 3 + 2] = gpz;
    g_rd[i * 3] = gdx;
    g_rd[i * 3 + 1] = gdy;
    g_rd[i * 3 + 2] = gdz;
}

extern "C" int nerfhip_ndc_rays_bwd(float near, float cw, float ch, float two_near, float neg_two_near, const float* rays_o,
                                    const float* rays_d, const float* g_out_o, const float* g_out_d, int64_t n, float* g_rays_o,
                                    float* g_rays_d, nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;
    NH_REQUIRE(rays_o && rays_d && g_out_o && g_out_d && g_rays_o && g_rays_d && n >= 0, "ndc_rays_bwd: bad arguments");
    NhNdc k = {near, cw, ch, two_near, neg_two_near};
    NH_LAUNCH(k_ndc_rays_bwd, nh_ceil_div(n, 256), 256, 0, stream, k, rays_o, rays_d, g_out_o, g_out_d, n, g_rays_o, g_rays_d);
    return nh_launch_status("ndc_rays_bwd");
}

// ---- viewdirs + ray packing (nerf/train_utils.py:143-168) ------------------------------------------------------------
NH_KERNEL void k_pack_rays(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ vsrc,
                           float near, float far, int64_t n, float* __restrict__ rays) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[3] = {ro[i * 3], ro[i * 3 + 1], ro[i * 3 + 2]};
    float d[3] = {rd[i * 3], rd[i * 3 + 1], rd[i * 3 + 2]};
    if (vsrc) {
        const float v[3] = {vsrc[i * 3], vsrc[i * 3 + 1], vsrc[i * 3 + 2]};
        nh_write_ray_row(rays + i * 11, o, d, near, far, v);
    } else {
        nh_write_ray_row(rays + i * 8, o, d, near, far, nullptr);
    }
}

extern "C" int nerfhip_pack_rays(const float* rays_o, const float* rays_d, const float* viewdir_src, float near,
                                 float far, int64_t n, float* rays_out, nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(rays_o && rays_d && rays_out && n >= 0, "pack_rays: bad arguments");
    NH_LAUNCH(k_pack_rays, nh_ceil_div(n, 256), 256, 0, stream, rays_o, rays_d, viewdir_src, near, far, n, rays_out);
    return nh_launch_status("pack_rays");
}

// ---- K3 positional_encoding (nerf/nerf_helpers.py:113-157) -----------------------------------------------------------
NH_KERNEL void k_posenc(const float* __restrict__ x, int64_t m, int d, const float* __restrict__ freqs, int nf,
                        int include_input, float* __restrict__ out) {
    int dout = d * (include_input + 2 * nf);
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * dout) return;
    int64_t row = idx / dout;
    int e = (int)(idx % dout);
    int blk = e / d, c = e % d;
    float v = x[row * d + c];
    float r;
    if (include_input && blk == 0) {
        r = v;
    } else {
        int b = blk - include_input;
        float arg = v * freqs[b >> 1];
        r = (b & 1) ? cosf(arg) : sinf(arg);
    }
    out[idx] = r;
}

extern "C" int nerfhip_positional_encoding(const float* x, int64_t m, int d, const float* freqs, int num_freqs,
                                           int include_input, float* out, nerfhip_stream_t stream) {
    if (m == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(x && out && m >= 0 && d > 0 && num_freqs >= 0 && (num_freqs == 0 || freqs),
               "positional_encoding: bad arguments");
    include_input = include_input ? 1 : 0;
    int64_t total = m * d * (include_input + 2 * num_freqs);
    if (total == 0) return NERFHIP_OK;
    NH_LAUNCH(k_posenc, nh_ceil_div(total, 256), 256, 0, stream, x, m, d, freqs, num_freqs, include_input, out);
    return nh_launch_status("positional_encoding");
}

// ---- K2 stratified depths (nerf/train_utils.py:38-65) ----------------------------------------------------------------
NH_DEVICE float nh_depth_at(float near, float far, float t, int lindisp) {
    if (!lindisp) return near * (1.0f - t) + far * t;
    return 1.0f / (1.0f / near * (1.0f - t) + 1.0f / far * t);
}

NH_KERNEL void k_stratified_z(const float* __restrict__ rays, int stride, int64_t n, const float* __restrict__ t_vals,
                              int nc, int lindisp, int perturb, const float* __restrict__ t_rand, uint64_t seed,
                              uint64_t ray_offset, float* __restrict__ z_out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * nc) return;
    int64_t r = idx / nc;
    int s = (int)(idx % nc);
    float near = rays[r * stride + 6], far = rays[r * stride + 7];
    float z = nh_depth_at(near, far, t_vals[s], lindisp);
    if (perturb) {
        float lower = z, upper = z;
        if (s > 0) lower = 0.5f * (z + nh_depth_at(near, far, t_vals[s - 1], lindisp));
        if (s < nc - 1) upper = 0.5f * (nh_depth_at(near, far, t_vals[s + 1], lindisp) + z);
        float tr = t_rand ? t_rand[idx] : nh_rand_uniform(seed, 0u, (ray_offset + (uint64_t)r) * (uint64_t)nc + s);
        z = lower + (upper - lower) * tr;
    }
    z_out[idx] = z;
}

extern "C" int nerfhip_stratified_z(const float* rays, int ray_stride, int64_t n, const float* t_vals, int nc,
                                    int lindisp, int perturb, const float* t_rand, uint64_t seed, uint64_t ray_offset,
                                    float* z_out, nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(rays && t_vals && z_out && ray_stride >= 8 && nc > 0 && n >= 0, "stratified_z: bad arguments");
    NH_LAUNCH(k_stratified_z, nh_ceil_div(n * nc, 256), 256, 0, stream, rays, ray_stride, n, t_vals, nc, lindisp,
              perturb, t_rand, seed, ray_offset, z_out);
    return nh_launch_status("stratified_z");
}

// ---- cumprod_exclusive (nerf/nerf_helpers.py:43-64) ------------------------------------------------------------------
// One wavefront per row.  torch's CPU cumprod accumulates fp32 rows in fp64 and rounds each prefix to fp32; the
// wave scan below forms the same fp64 prefix products (association differs only in the last fp64 bits).
NH_KERNEL void k_cumprod_exclusive(const float* __restrict__ x, int64_t rows, int cols, float* __restrict__ out) {
    int64_t row = blockIdx.x;
    int lane = nh_lane();
    double carry = 1.0;
    for (int base = 0; base < cols; base += 64) {
        int c = base + lane;
        double p = (c < cols) ? (double)x[row * cols + c] : 1.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            double o = nh_shfl_up_d(p, d);
            if (lane >= d) p *= o;
        }
        double excl = nh_shfl_up_d(p, 1);
        if (lane == 0) excl = 1.0;
        if (c < cols) out[row * cols + c] = (float)(carry * excl);
        carry *= nh_shfl_d(p, 63);
    }
}

extern "C" int nerfhip_cumprod_exclusive(const float* x, int64_t rows, int cols, float* out, nerfhip_stream_t stream) {
    if (rows == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(x && out && rows >= 0 && cols > 0, "cumprod_exclusive: bad arguments");
    NH_LAUNCH(k_cumprod_exclusive, rows, 64, 0, stream, x, rows, cols, out);
    return nh_launch_status("cumprod_exclusive");
}

// Backward of the exclusive cumulative product (what autograd computes for nerf/nerf_helpers.py:43-64: cumprod, roll,
// overwrite column 0): with y_i = prod_{k<i} x_k,  dL/dx_j = sum_{i>j} g_i * prod_{k<i, k != j} x_k -- formed without a
// division, so rows containing zeros are exact.  One wavefront per row, lane = column j (strided), fp64 running product.
NH_KERNEL void k_cumprod_exclusive_bwd(const float* __restrict__ x, const float* __restrict__ y,
                                       const float* __restrict__ g, int64_t rows, int cols, float* __restrict__ gx) {
    const int64_t row = blockIdx.x;
    const float* xr = x + row * cols;
    const float* gr = g + row * cols;
    for (int j = nh_lane(); j < cols; j += 64) {
        double t = (double)y[row * cols + j];  // prod_{k<j} x_k
        double acc = 0.0;
        for (int i = j + 1; i < cols; ++i) {
            acc += (double)gr[i] * t;
            t *= (double)xr[i];
        }
        gx[row * cols + j] = (float)acc;
    }
}

extern "C" int nerfhip_cumprod_exclusive_bwd(const float* x, const float* y, const float* g_y, int64_t rows, int cols,
                                             float* g_x, nerfhip_stream_t stream) {
    if (rows == 0) return NERFHIP_OK;
    NH_REQUIRE(x && y && g_y && g_x && rows >= 0 && cols > 0, "cumprod_exclusive_bwd: bad arguments");
    NH_LAUNCH(k_cumprod_exclusive_bwd, rows, 64, 0, stream, x, y, g_y, rows, cols, g_x);
    return nh_launch_status("cumprod_exclusive_bwd");
}

// ---- RNG fill ----------------------------------------------------------------------------------------------------
NH_KERNEL void k_rng_fill(int kind, uint64_t seed, uint32_t stream_id, uint64_t first, int64_t n,
                          float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t e = first + (uint64_t)i;
    out[i] = kind == NERFHIP_RNG_NORMAL ? nh_rand_normal(seed, stream_id, e) : nh_rand_uniform(seed, stream_id, e);
}

extern "C" int nerfhip_rng_fill(int kind, uint64_t seed, uint32_t stream_id, uint64_t first, int64_t n, float* out,
                                nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(out && n >= 0 && (kind == NERFHIP_RNG_UNIFORM || kind == NERFHIP_RNG_NORMAL), "rng_fill: bad arguments");
    NH_LAUNCH(k_rng_fill, nh_ceil_div(n, 256), 256, 0, stream, kind, seed, stream_id, first, n, out);
    return nh_launch_status("rng_fill");
}

// ---- weight gather ---------------------------------------------------------------------------------------------------
NH_KERNEL void k_pack_weights(const float* __restrict__ params, const int32_t* __restrict__ table, int64_t n,
                              float* __restrict__ packed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t s = table[i];
    packed[i] = s >= 0 ? params[s] : 0.0f;
}

extern "C" int nerfhip_pack_weights(const float* params, const int32_t* table, int64_t n, float* packed,
                                    nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(params && table && packed && n >= 0, "pack_weights: bad arguments");
    NH_LAUNCH(k_pack_weights, nh_ceil_div(n, 256), 256, 0, stream, params, table, n, packed);
    return nh_launch_status("pack_weights");
}

// ---- loss (train_nerf.py:244-258) ------------------------------------------------------------------------------------
// One workgroup; deterministic fp64 tree.  loss_out = {mse(rgb_c, tgt), mse(rgb_f, tgt), sum}; cotangents are
// 2 (rgb - tgt) / (3 n) * grad_scale.
NH_KERNEL void k_mse_loss(const float* __restrict__ rc, const float* __restrict__ rf, const float* __restrict__ tgt,
                          int tstride, int64_t n, float grad_scale, float* __restrict__ gc, float* __restrict__ gf,
                          float* __restrict__ loss_out) {
    NH_SHARED double red[2][16];
    double sc = 0.0, sf = 0.0;
    const float k = 2.0f / (float)(3 * n) * grad_scale;
    for (int64_t i = threadIdx.x; i < n * 3; i += blockDim.x) {
        int64_t r = i / 3;
        int c = (int)(i % 3);
        float t = tgt[r * tstride + c];
        float dc = rc[i] - t;
        sc += (double)dc * (double)dc;
        if (gc) gc[i] = dc * k;
        if (rf) {
            float df = rf[i] - t;
            sf += (double)df * (double)df;
            if (gf) gf[i] = df * k;
        }
    }
    sc = nh_wave_sum_d(sc);
    sf = nh_wave_sum_d(sf);
    int w = nh_wave_in_block();
    if (nh_lane() == 0) {
        red[0][w] = sc;
        red[1][w] = sf;
    }
    nh_block_sync();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        int nw = (int)(blockDim.x >> 6);
        for (int i = 0; i < nw; ++i) {
            a += red[0][i];
            b += red[1][i];
        }
        float la = (float)(a / (double)(3 * n)), lb = (float)(b / (double)(3 * n));
        loss_out[0] = la;
        loss_out[1] = lb;
        loss_out[2] = la + lb;
    }
}

extern "C" int nerfhip_mse_loss_fwd_bwd(const float* rgb_coarse, const float* rgb_fine, const float* target,
                                        int target_stride, int64_t n, float grad_scale, float* g_rgb_coarse,
                                        float* g_rgb_fine, float* loss_out, nerfhip_stream_t stream) {
    NH_REQUIRE(rgb_coarse && target && loss_out && n > 0 && target_stride >= 3, "mse_loss: bad arguments");
    NH_LAUNCH(k_mse_loss, 1, 1024, 0, stream, rgb_coarse, rgb_fine, target, target_stride, n, grad_scale, g_rgb_coarse,
              g_rgb_fine, loss_out);
    return nh_launch_status("mse_loss");
}

// ---- Adam (torch.optim.Adam single-tensor step; train_nerf.py:141-143,261) -------------------------------------------
NH_KERNEL void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                      int64_t n, float one_minus_b1, float b2, float one_minus_b2, float step_size, float bc2_sqrt,
                      float eps, float grad_scale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * grad_scale;
    float mi = m[i];
    mi = mi + one_minus_b1 * (gi - mi);  // exp_avg.lerp_(grad, 1 - beta1)
    float vi = v[i] * b2 + one_minus_b2 * (gi * gi);
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

extern "C" int nerfhip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 float lr, float beta1, float beta2, float eps, int64_t step, float grad_scale,
                                 nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(params && grads && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "adam_step: bad arguments");
    double bc1 = 1.0 - pow((double)beta1, (double)step);
    double bc2 = 1.0 - pow((double)beta2, (double)step);
    float step_size = (float)((double)lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    NH_LAUNCH(k_adam, nh_ceil_div(n, 256), 256, 0, stream, params, grads, exp_avg, exp_avg_sq, n,
              (float)(1.0 - (double)beta1), beta2, (float)(1.0 - (double)beta2), step_size, bc2_sqrt, eps, grad_scale);
    return nh_launch_status("adam_step");
}

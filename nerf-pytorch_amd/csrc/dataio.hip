// dataio.hip -- the stages either side of the render path:
//   * training-ray selection: draw N distinct pixels, generate ONLY those rays, pack them, gather their target
//     colours (train_nerf.py:210-227, cached branch :175-194) -- one launch instead of a host permutation of the
//     whole image, a full-image get_ray_bundle and three fancy-index gathers;
//   * 8-bit output: cast_to_image / cast_to_disparity_image (eval_nerf.py:23-36).
// Everything here is HBM/latency-bound byte and index work: one thread per ray / pixel, coalesced rows.
#include "nh_host.h"
#include "nh_rays.h"

// ---- keyed permutation of [0, population) ----------------------------------------------------------------------------
// The reference draws np.random.choice(population, N, replace=False): a uniformly random N-subset in random order.
// Here: positions first..first+n-1 of a keyed pseudo-random permutation of [0, population) -- a 6-round balanced
// Feistel network over the smallest even number of bits covering the population, cycle-walked back into range
// (a bijection by construction, so the indices are distinct; ranks take disjoint position ranges of the SAME
// permutation).  Round keys: Philox4x32-10 of (seed, step), streams 4 and 5.
struct NhPerm {
    uint32_t k[6];
    uint32_t half, mask;
    uint64_t population;
};
NH_DEVICE NhPerm nh_perm_key(uint64_t seed, uint64_t step, uint64_t population) {
    NhPerm p;
    nh_u4 a = nh_philox(seed, step, 4u);
    nh_u4 b = nh_philox(seed, step, 5u);
    p.k[0] = a.x, p.k[1] = a.y, p.k[2] = a.z, p.k[3] = a.w, p.k[4] = b.x, p.k[5] = b.y;
    uint32_t bits = 2;
    while (bits < 64 && ((uint64_t)1 << bits) < population) bits += 2;
    p.half = bits / 2;
    p.mask = (uint32_t)(((uint64_t)1 << p.half) - 1);
    p.population = population;
    return p;
}
NH_DEVICE uint64_t nh_perm_at(const NhPerm& p, uint64_t x) {
    do {
        uint32_t l = (uint32_t)(x >> p.half), r = (uint32_t)x & p.mask;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            uint32_t h = r ^ p.k[i];
            h *= 0x85EBCA6Bu;
            h ^= h >> 13;
            h *= 0xC2B2AE35u;
            h ^= h >> 16;
            uint32_t t = l ^ (h & p.mask);
            l = r;
            r = t;
        }
        x = ((uint64_t)l << p.half) | r;
    } while (x >= p.population);
    return x;
}

NH_KERNEL void k_select_indices(uint64_t seed, uint64_t step, uint64_t population, int64_t first, int64_t n,
                                int64_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    NhPerm p = nh_perm_key(seed, step, population);
    out[i] = (int64_t)nh_perm_at(p, (uint64_t)(first + i));
}

extern "C" int nerfhip_select_indices(uint64_t seed, uint64_t step, int64_t population, int64_t first, int64_t n,
                                      int64_t* out, nerfhip_stream_t stream) {
    NH_REQUIRE(n >= 0 && first >= 0 && population >= 0 && population <= ((int64_t)1 << 32) && (n == 0 || out),
               "select_indices: bad arguments");
    NH_REQUIRE(first + n <= population, "select_indices: first + n exceeds the population (sampling is without replacement)");
    if (n == 0) return NERFHIP_OK;
    NH_LAUNCH(k_select_indices, nh_ceil_div(n, 256), 256, 0, stream, seed, step, (uint64_t)population, first, n, out);
    return nh_launch_status("select_indices");
}

// ---- fused selection -> rays -> packed rows + target gather ----------------------------------------------------------
// c2w != NULL: image branch (rays generated for the selected pixels only); else cached branch (rows of the stored
// ray bundle).  The rest is run_one_iter_of_nerf's prologue (train_utils.py:143-168): viewdirs from the pre-NDC
// directions, optional ndc_rays with near = 1.0, [o d near far viewdirs] rows.
NH_KERNEL void k_select_rays(nerfhip_select_cfg s, NhNdc ndc, const float* __restrict__ c2w, int ld,
                             const float* __restrict__ cached_o, const float* __restrict__ cached_d,
                             const float* __restrict__ targets, uint64_t population,
                             const int64_t* __restrict__ inds_in, int64_t n, float* __restrict__ rays,
                             float* __restrict__ target_out, int64_t* __restrict__ inds_out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t k;
    if (inds_in) {
        k = inds_in[i];
    } else {
        NhPerm p = nh_perm_key(s.seed, s.step, population);
        k = (int64_t)nh_perm_at(p, (uint64_t)(s.first + i));
    }
    if (inds_out) inds_out[i] = k;
    float o[3], d[3], v[3];
    int64_t pix;  // row-major position of the target pixel
    if (c2w) {
        // coords = stack(meshgrid_xy(arange(H), arange(W)), -1).reshape(-1, 2): entry k is (k % H, k / H), used as
        // (row, col) -- train_nerf.py:214-225
        int64_t row = k % s.height, col = k / s.height;
        nh_pinhole_ray(s.height, s.width, s.focal, c2w, ld, row, col, o, d);
        pix = row * s.width + col;
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c] = cached_o[k * 3 + c];
            d[c] = cached_d[k * 3 + c];
        }
        pix = k;
    }
    v[0] = d[0], v[1] = d[1], v[2] = d[2];
    if (s.ndc) nh_ndc_ray(ndc, o, d);
    nh_write_ray_row(rays + i * (s.use_viewdirs ? 11 : 8), o, d, s.near, s.far, s.use_viewdirs ? v : nullptr);
    if (targets) {
        for (int c = 0; c < s.channels; ++c) target_out[i * s.channels + c] = targets[pix * s.channels + c];
    }
}

static int select_launch(const nerfhip_select_cfg* cfg, const float* c2w, int ld, const float* co, const float* cd,
                         const float* targets, int64_t population, const int64_t* inds, int64_t n, float* rays,
                         float* target_out, int64_t* inds_out, nerfhip_stream_t stream, const char* what) {
    NH_REQUIRE(cfg && n >= 0 && (n == 0 || rays), "%s: bad arguments", what);
    NH_REQUIRE(!targets || (target_out && cfg->channels >= 1 && cfg->channels <= 4), "%s: bad target arguments", what);
    NH_REQUIRE(population >= 0 && population <= ((int64_t)1 << 32), "%s: bad population", what);
    NH_REQUIRE(inds || (cfg->first >= 0 && cfg->first + n <= population),
               "%s: first + n exceeds the population (sampling is without replacement)", what);
    if (n == 0) return NERFHIP_OK;
    NhNdc ndc = {cfg->ndc_near, cfg->ndc_cw, cfg->ndc_ch, cfg->ndc_two_near, cfg->ndc_neg_two_near};
    NH_LAUNCH(k_select_rays, nh_ceil_div(n, 256), 256, 0, stream, *cfg, ndc, c2w, ld, co, cd, targets,
              (uint64_t)population, inds, n, rays, target_out, inds_out);
    return nh_launch_status(what);
}

extern "C" int nerfhip_select_rays(const nerfhip_select_cfg* cfg, const float* c2w, int c2w_ld, const float* image,
                                   const int64_t* select_inds, int64_t n, float* rays, float* target,
                                   int64_t* inds_out, nerfhip_stream_t stream) {
    NH_REQUIRE(cfg && c2w && c2w_ld >= 4 && cfg->height > 0 && cfg->width > 0, "select_rays: bad arguments");
    return select_launch(cfg, c2w, c2w_ld, nullptr, nullptr, image, (int64_t)cfg->height * cfg->width, select_inds, n,
                         rays, target, inds_out, stream, "select_rays");
}

extern "C" int nerfhip_select_cached_rays(const nerfhip_select_cfg* cfg, const float* ray_origins,
                                          const float* ray_directions, const float* targets, int64_t population,
                                          const int64_t* select_inds, int64_t n, float* rays, float* target,
                                          int64_t* inds_out, nerfhip_stream_t stream) {
    NH_REQUIRE(cfg && ray_origins && ray_directions, "select_cached_rays: bad arguments");
    return select_launch(cfg, nullptr, 0, ray_origins, ray_directions, targets, population, select_inds, n, rays, target,
                         inds_out, stream, "select_cached_rays");
}

// ---- 8-bit output stage (eval_nerf.py:23-36) ---------------------------------------------------------------------------
// float -> uint8 exactly as the reference's host conversion behaves on x86-64: truncate towards zero to a 32-bit
// integer, keep the low byte (NaN / out-of-range -> the "integer indefinite" 0x80000000 -> 0).
NH_DEVICE uint8_t nh_to_u8(float v) {
    int32_t q;
    if (v != v || v >= 2147483648.0f || v < -2147483648.0f)
        q = (int32_t)0x80000000u;
    else
        q = (int32_t)v;
    return (uint8_t)((uint32_t)q & 0xFFu);
}

// cast_to_image: torchvision ToPILImage of a float CHW tensor = pic.mul(255).byte() -> HWC bytes.  in: [pixels,
// in_channels] (the first 3 are used, as the caller slices rgb[..., :3]); out: [pixels, 3] uint8.
NH_KERNEL void k_cast_to_image(const float* __restrict__ in, int in_channels, int64_t pixels, uint8_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pixels * 3) return;
    int64_t p = i / 3;
    int c = (int)(i - p * 3);
    out[i] = nh_to_u8(in[p * in_channels + c] * 255.0f);
}

extern "C" int nerfhip_cast_to_image(const float* rgb, int in_channels, int64_t pixels, uint8_t* out,
                                     nerfhip_stream_t stream) {
    NH_REQUIRE(pixels >= 0 && in_channels >= 3 && (pixels == 0 || (rgb && out)), "cast_to_image: bad arguments");
    if (pixels == 0) return NERFHIP_OK;
    NH_LAUNCH(k_cast_to_image, nh_ceil_div(pixels * 3, 256), 256, 0, stream, rgb, in_channels, pixels, out);
    return nh_launch_status("cast_to_image");
}

// cast_to_disparity_image: (t - min) / (max - min), clamp(0,1) * 255, astype(uint8).  torch's min()/max() propagate
// NaN, so a single NaN pixel (acc == 0) turns the whole reference image into zeros -- reproduced.
// Pass 1: one workgroup reduces min / max / any-NaN into scratch[0..2]; pass 2: the map.
NH_KERNEL void k_disparity_range(const float* __restrict__ t, int64_t n, float* __restrict__ scratch) {
    NH_SHARED float s_lo[16], s_hi[16], s_nan[16];
    float lo = INFINITY, hi = -INFINITY, bad = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        float v = t[i];
        if (v != v) {
            bad = 1.0f;
        } else {
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        lo = fminf(lo, nh_shfl_xor(lo, m));
        hi = fmaxf(hi, nh_shfl_xor(hi, m));
        bad = fmaxf(bad, nh_shfl_xor(bad, m));
    }
    int wave = nh_wave_in_block(), lane = nh_lane();
    if (lane == 0) s_lo[wave] = lo, s_hi[wave] = hi, s_nan[wave] = bad;
    nh_block_sync();
    if (threadIdx.x == 0) {
        int waves = (int)(blockDim.x >> 6);
        for (int w = 1; w < waves; ++w) {
            lo = fminf(lo, s_lo[w]);
            hi = fmaxf(hi, s_hi[w]);
            bad = fmaxf(bad, s_nan[w]);
        }
        scratch[0] = lo;
        scratch[1] = hi;
        scratch[2] = bad;
    }
}
NH_KERNEL void k_disparity_image(const float* __restrict__ t, int64_t n, const float* __restrict__ scratch,
                                 uint8_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float lo = scratch[0], hi = scratch[1];
    if (scratch[2] != 0.0f) lo = hi = NAN;
    float v = (t[i] - lo) / (hi - lo);
    v = v != v ? v : fminf(fmaxf(v, 0.0f), 1.0f);  // torch.clamp keeps NaN
    out[i] = nh_to_u8(v * 255.0f);
}

extern "C" int nerfhip_cast_to_disparity_image(const float* disparity, int64_t pixels, float* scratch3, uint8_t* out,
                                               nerfhip_stream_t stream) {
    NH_REQUIRE(pixels >= 0 && (pixels == 0 || (disparity && out && scratch3)), "cast_to_disparity_image: bad arguments");
    if (pixels == 0) return NERFHIP_OK;
    NH_LAUNCH(k_disparity_range, 1, 1024, 0, stream, disparity, pixels, scratch3);
    NH_LAUNCH(k_disparity_image, nh_ceil_div(pixels, 256), 256, 0, stream, disparity, pixels, scratch3, out);
    return nh_launch_status("cast_to_disparity_image");
}

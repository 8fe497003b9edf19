// sample.hip -- inverse-CDF importance sampling (sample_pdf_2, nerf/nerf_helpers.py:260-302, including the
// torchsearchsorted call at :288) and the hierarchical merge of predict_and_render_radiance
// (nerf/train_utils.py:96-105).  One wavefront per ray; the ray's CDF and bin edges live in LDS.
//
// Accumulation order is part of the contract: the weight sum is accumulated sequentially in fp64
// and rounded to fp32; the CDF is a sequential fp64 running sum whose every prefix is rounded to fp32 -- what
// torch's CPU cumsum does for fp32 rows.  searchsorted(side="right") is an exact upper-bound binary search, so the
// indices are bit-exact for a given (cdf, u).
#include "nh_host.h"

NH_DEVICE int nh_next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// mode 0: bins [n,nb], weights [n,nb-1] as given.  mode 1 (hierarchical): zc [n,nc], wfull [n,nc]:
// bins = mids of zc (nb = nc-1), weights = wfull[1:-1].
NH_KERNEL void k_sample_pdf(int mode, const float* __restrict__ bins_in, const float* __restrict__ w_in, int64_t n,
                            int nb, const float* __restrict__ u, int det, const float* __restrict__ u_det, int nf,
                            uint64_t seed, uint64_t ray_offset, float* __restrict__ samples, int64_t* __restrict__ inds,
                            float* __restrict__ cdf_out, float* __restrict__ z_fine, int sortp) {
    NH_DYN_LDS(lds_raw);
    float* s_cdf = (float*)lds_raw;  // [nb]
    float* s_bins = s_cdf + nb;      // [nb]
    float* s_sort = s_bins + nb;     // [sortp] (hierarchical mode only)
    const int64_t ray = blockIdx.x;
    const int lane = nh_lane();
    const int nc = nb + 1;  // mode 1: coarse samples per ray

    for (int i = lane; i < nb; i += 64) {
        if (mode == 1) {
            const float* zr = bins_in + ray * nc;
            s_bins[i] = 0.5f * (zr[i + 1] + zr[i]);
            if (i + 1 < nb) s_cdf[i + 1] = w_in[ray * nc + 1 + i] + 1e-5f;
        } else {
            s_bins[i] = bins_in[ray * nb + i];
            if (i + 1 < nb) s_cdf[i + 1] = w_in[ray * (nb - 1) + i] + 1e-5f;
        }
    }
    if (lane == 0) s_cdf[0] = 0.0f;
    nh_block_sync();
    if (lane == 0) {
        double tot = 0.0;
        for (int i = 1; i < nb; ++i) tot += (double)s_cdf[i];
        const float sum = (float)tot;
        double run = 0.0;
        for (int i = 1; i < nb; ++i) {
            const float pdf = s_cdf[i] / sum;
            run += (double)pdf;
            s_cdf[i] = (float)run;
        }
    }
    nh_block_sync();
    if (cdf_out)
        for (int i = lane; i < nb; i += 64) cdf_out[ray * nb + i] = s_cdf[i];

    for (int k = lane; k < nf; k += 64) {
        float uu;
        if (u)
            uu = u[ray * nf + k];
        else if (det)
            uu = u_det[k];
        else
            uu = nh_rand_uniform(seed, 2u, (ray_offset + (uint64_t)ray) * (uint64_t)nf + k);
        int lo = 0, hi = nb;  // first index with cdf[idx] > u  == searchsorted(..., side="right")
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (s_cdf[mid] <= uu)
                lo = mid + 1;
            else
                hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < nb - 1 ? lo : nb - 1;
        const float c0 = s_cdf[below], c1 = s_cdf[above];
        const float b0 = s_bins[below], b1 = s_bins[above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (uu - c0) / denom;
        const float smp = b0 + t * (b1 - b0);
        if (samples) samples[ray * nf + k] = smp;
        if (inds) inds[ray * nf + k] = (int64_t)lo;
        if (mode == 1) s_sort[nc + k] = smp;
    }
    if (mode == 1) {
        for (int i = lane; i < nc; i += 64) s_sort[i] = bins_in[ray * nc + i];
        for (int i = nc + nf + lane; i < sortp; i += 64) s_sort[i] = INFINITY;
        nh_block_sync();
        // bitonic sort, ascending (values only, like torch.sort(...)[0] at train_utils.py:105)
        for (int k = 2; k <= sortp; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = lane; i < sortp; i += 64) {
                    const int p = i ^ j;
                    if (p > i) {
                        const float a = s_sort[i], b = s_sort[p];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) {
                            s_sort[i] = b;
                            s_sort[p] = a;
                        }
                    }
                }
                nh_block_sync();
            }
        }
        const int tot = nc + nf;
        for (int i = lane; i < tot; i += 64) z_fine[ray * tot + i] = s_sort[i];
    }
}

extern "C" int nerfhip_sample_pdf(const float* bins, const float* weights, int64_t n, int nbins, const float* u, int det,
                                  const float* u_det, int nf, uint64_t seed, uint64_t ray_offset, float* samples,
                                  int64_t* inds, float* cdf, nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(bins && weights && n >= 0 && nbins >= 2 && nf > 0, "sample_pdf: bad arguments");
    NH_REQUIRE(nbins <= 4096, "sample_pdf: at most 4096 bins");
    NH_REQUIRE(u || !det || u_det, "sample_pdf: det=1 needs u_det (linspace(0,1,nf))");
    size_t lds = (size_t)2 * nbins * sizeof(float);
    NH_LAUNCH(k_sample_pdf, n, 64, lds, stream, 0, bins, weights, n, nbins, u, det, u_det, nf, seed, ray_offset, samples,
              inds, cdf, (float*)nullptr, 0);
    return nh_launch_status("sample_pdf");
}

extern "C" int nerfhip_hierarchical_z(const float* z_coarse, const float* weights, int64_t n, int nc, const float* u,
                                      int det, const float* u_det, int nf, uint64_t seed, uint64_t ray_offset,
                                      float* z_samples, float* z_fine, nerfhip_stream_t stream) {
    if (n == 0) return NERFHIP_OK;  // empty input: nothing to launch, pointers may be NULL
    NH_REQUIRE(z_coarse && weights && z_fine && n >= 0 && nc >= 3 && nf > 0, "hierarchical_z: bad arguments");
    NH_REQUIRE(nc + nf <= 8192, "hierarchical_z: at most 8192 samples per ray");
    NH_REQUIRE(u || !det || u_det, "hierarchical_z: det=1 needs u_det (linspace(0,1,nf))");
    int sortp = 1;
    while (sortp < nc + nf) sortp <<= 1;
    size_t lds = (size_t)(2 * (nc - 1) + sortp) * sizeof(float);
    NH_LAUNCH(k_sample_pdf, n, 64, lds, stream, 1, z_coarse, weights, n, nc - 1, u, det, u_det, nf, seed, ray_offset,
              z_samples, (int64_t*)nullptr, (float*)nullptr, z_fine, sortp);
    return nh_launch_status("hierarchical_z");
}

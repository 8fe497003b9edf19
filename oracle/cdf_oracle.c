/* cdf_oracle.c -- TEST INFRASTRUCTURE.  Plain-C restatement of the inverse-CDF sampler with a DECLARED accumulation
 * order, for the "inverse-CDF sample indices bit-exact" part of the parity contract.
 *
 * Follows sample_pdf_2 (nerf/nerf_helpers.py:260-302) and the torchsearchsorted call at :288 (third-party,
 * un-pinned git dependency aliutkus/torchsearchsorted -- requirements.txt:9; published semantics: numpy
 * searchsorted side='right', i.e. the first index i with cdf[i] > u).
 *
 * Declared order (what torch's CPU kernels do for fp32 rows -- SURVEY 0.7): the weight sum is accumulated
 * sequentially in double and rounded to float; the CDF is a sequential double running sum of the float pdf values,
 * every prefix rounded to float.  The HIP kernel (csrc/sample.hip) must reproduce cdf, inds and samples bit for bit.
 *
 * Only tests/ may load this file (built by __graft_entry__.build() into oracle/_build/libcdf_oracle.so).
 */
#include <stdint.h>

void oracle_sample_pdf(const float* bins, const float* weights, int64_t n, int nb, const float* u, int nf,
                       float* samples, int64_t* inds, float* cdf_out) {
    for (int64_t r = 0; r < n; ++r) {
        const float* b = bins + r * nb;
        const float* w = weights + r * (nb - 1);
        float* cdf = cdf_out + r * nb;
        double tot = 0.0;
        for (int i = 0; i < nb - 1; ++i) tot += (double)(float)(w[i] + 1e-5f);
        const float sum = (float)tot;
        double run = 0.0;
        cdf[0] = 0.0f;
        for (int i = 0; i < nb - 1; ++i) {
            const float wi = w[i] + 1e-5f;
            const float pdf = wi / sum;
            run += (double)pdf;
            cdf[i + 1] = (float)run;
        }
        for (int k = 0; k < nf; ++k) {
            const float uu = u[r * nf + k];
            int lo = 0, hi = nb;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (cdf[mid] <= uu)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            const int below = lo - 1 > 0 ? lo - 1 : 0;
            const int above = lo < nb - 1 ? lo : nb - 1;
            float denom = cdf[above] - cdf[below];
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (uu - cdf[below]) / denom;
            samples[r * nf + k] = b[below] + t * (b[above] - b[below]);
            inds[r * nf + k] = lo;
        }
    }
}

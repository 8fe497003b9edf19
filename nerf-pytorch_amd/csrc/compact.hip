// compact.hip -- the sample list of a COMPACTED backward (nerfhip_plan_set_bwd_compaction).
//
// The reference differentiates every sample point densely (autograd of nerf/models.py:233-256 under train_nerf.py:259).  But
// sigma_a = relu(raw[..., 3] + noise) (nerf/volume_rendering_utils.py:38): wherever that ReLU is off -- and behind the sample at which a
// ray's transmittance reaches 0 -- the sample's weight is exactly 0, so its d(loss)/d(raw) is exactly zero in all four channels, and
// with it every d(pre-activation) row of that sample in every layer: a zero term of every weight-gradient sum.  The compacted backward
// drops those terms instead of multiplying them: this file lists the samples whose cotangent row is not all zero (ascending sample
// index: a fixed order), the data-gradient kernels walk that list (mlp16.hip / mlp_f16w.hip: gather d(raw) and the ReLU masks by
// index, write the d(pre-activation) images compacted) and the weight-gradient kernels multiply the compacted images with the
// activation rows gathered through the same list (wgrad.hip / wgrad_f16.hip).  Same sums, zero terms dropped.
//
// Two launches, no atomics: k_compact_count (per 2048-sample block: the number of non-zero rows) and k_compact_scatter (block offset
// = sum of the counts in front of it; ranks inside the block by a prefix sum over its threads).  The list is padded with sample 0 up to
// the next multiple of 128 (the data-gradient kernels then write whole zero rows for the padding slots, and 0 * finite = 0).
#include "nh_mlp.h"

namespace {

constexpr int CB_THREADS = 256, CB_PER_THREAD = 8, CB_SAMPLES = CB_THREADS * CB_PER_THREAD;

NH_DEVICE int row_nonzero(const float* g_out, int64_t m) {
    const float4 v = *(const float4*)(g_out + (size_t)m * 4);
    return (v.x != 0.0f || v.y != 0.0f || v.z != 0.0f || v.w != 0.0f) ? 1 : 0;  // (a NaN row counts: it propagates as in the dense path)
}
NH_DEVICE int wave_sum_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += nh_shfl_xor_i(v, m);
    return v;
}

NH_KERNEL void k_compact_count(const float* __restrict__ g_out, int64_t M, int* __restrict__ counts) {
    NH_SHARED int wsum[CB_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * CB_SAMPLES + (int64_t)threadIdx.x * CB_PER_THREAD;
    int c = 0;
#pragma unroll
    for (int e = 0; e < CB_PER_THREAD; ++e)
        if (base + e < M) c += row_nonzero(g_out, base + e);
    c = wave_sum_i(c);
    if (nh_lane() == 0) wsum[nh_wave_in_block()] = c;
    nh_block_sync();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < CB_THREADS / 64; ++w) t += wsum[w];
        counts[blockIdx.x] = t;
    }
}

NH_KERNEL void k_compact_scatter(const float* __restrict__ g_out, int64_t M, const int* __restrict__ counts, int nblocks,
                                 int* __restrict__ idx, int* __restrict__ stats) {
    NH_SHARED int wsum[CB_THREADS / 64 + 1];
    const int lane = nh_lane(), wave = nh_wave_in_block();
    const int64_t base = (int64_t)blockIdx.x * CB_SAMPLES + (int64_t)threadIdx.x * CB_PER_THREAD;
    int flags = 0, c = 0;
#pragma unroll
    for (int e = 0; e < CB_PER_THREAD; ++e)
        if (base + e < M && row_nonzero(g_out, base + e)) flags |= 1 << e, ++c;
    // inclusive prefix sum of c over the wave
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = nh_shfl_i(incl, lane - d < 0 ? 0 : lane - d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    // the samples in front of this block: wave 0 sums the counts of blocks 0 .. blockIdx.x - 1
    if (wave == 0) {
        int s = 0;
        for (int b = lane; b < (int)blockIdx.x; b += 64) s += counts[b];
        s = wave_sum_i(s);
        if (lane == 0) wsum[CB_THREADS / 64] = s;
    }
    nh_block_sync();
    int off = wsum[CB_THREADS / 64];
    for (int w = 0; w < wave; ++w) off += wsum[w];
    off += incl - c;
#pragma unroll
    for (int e = 0; e < CB_PER_THREAD; ++e)
        if (flags & (1 << e)) idx[off++] = (int)(base + e);
    if ((int)blockIdx.x == nblocks - 1) {  // the last block knows the total: statistics and the padding slots
        int total = wsum[CB_THREADS / 64];
        for (int w = 0; w < CB_THREADS / 64; ++w) total += wsum[w];
        const int padded = (total + 127) & ~127;
        for (int q = total + (int)threadIdx.x; q < padded; q += CB_THREADS) idx[q] = 0;
        if (threadIdx.x == 0) {
            stats[NH_CSTAT_ACTIVE] = total;
            stats[NH_CSTAT_TOTAL] = (int)M;
        }
    }
}

}  // namespace

int64_t nh_compact_ints(int64_t M) {
    // statistics | per-block counts | the list: one slot per sample, whole 128-sample groups, + one 1-KiB copy piece of slack
    return NH_CSTAT_WORDS + ((nh_ceil_div(M, CB_SAMPLES) + 15) & ~(int64_t)15) + nh_ceil_div(M, 128) * 128 + 256;
}

NhCompact nh_compact_view(int* area, int64_t M) {
    NhCompact c;
    c.stats = area;
    c.counts = area + NH_CSTAT_WORDS;
    c.idx = c.counts + ((nh_ceil_div(M, CB_SAMPLES) + 15) & ~(int64_t)15);
    c.stash_in_list_order = false;
    return c;
}

int nh_compact_build(const float* g_out, int64_t M, const NhCompact& c, nerfhip_stream_t stream) {
    NH_REQUIRE(g_out && c.idx && M > 0 && M < ((int64_t)1 << 31), "compact: bad arguments");
    const int nblocks = (int)nh_ceil_div(M, CB_SAMPLES);
    NH_LAUNCH(k_compact_count, nblocks, CB_THREADS, 0, stream, g_out, M, c.counts);
    int rc = nh_launch_status("compact_count");
    if (rc) return rc;
    NH_LAUNCH(k_compact_scatter, nblocks, CB_THREADS, 0, stream, g_out, M, (const int*)c.counts, nblocks, c.idx, c.stats);
    return nh_launch_status("compact_scatter");
}

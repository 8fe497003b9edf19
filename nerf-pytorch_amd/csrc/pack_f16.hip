// pack_f16.hip -- the weight images of the fp16-piece plans (NERFHIP_PRECISION_F16X3*, include/nerfhip.h): gather the reference's
// parameters (nerf/models.py:205-229, state_dict order) into the MFMA operand order of mlp_f16w.hip, scale by 2^8, split every value
// into two IEEE fp16 pieces.  An image = 512 fp32 bias words (one per output row) + per (k-block kb, output tile t) a 1-KiB block of
// high pieces and a 1-KiB block of low pieces; lane l of a block holds the 8 elements W[16 t + (l & 15)][in(kb, l >> 4, e)] (nh_plan.h
// nhw_unit; the gather table is plan.cpp fill_spec_b's).  A weight's low piece is a normal fp16 number down to |w| = 2^-10.
// Limit (documented in include/nerfhip.h): |w| * 2^8 must stay below fp16's largest finite value -- |w| < 255.9; a larger weight is
// SATURATED to +-65504 / 2^8 here (not turned into Inf, which would make every output NaN): a net that large has left the range
// any NeRF trains in (torch's default init: |w| <= 1 / sqrt(fan_in)).
#include "nh_device.h"
#include "nh_mlp.h"

namespace {

constexpr float WS = NHB_F16_WSCALE;
constexpr float F16_MAX = 65504.0f;

struct PackArgs {
    int n_layers;
    int64_t first;  // first word of the fp16-piece images (what lies in front is the fp32 image)
    int64_t base[2 * NH_MAX_LAYERS + 10];  // word offset of every layer image, ascending
};

NH_KERNEL void k_pack_f16x3(const float* __restrict__ params, const int32_t* __restrict__ table, int64_t n, PackArgs la,
                            float* __restrict__ packed) {
    const int64_t i = la.first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int l = 0;
    while (l + 1 < la.n_layers && i >= la.base[l + 1]) ++l;
    const int64_t r = i - la.base[l];
    const int32_t s = table[i];
    const float v = (s >= 0 ? params[s] : 0.0f) * WS;  // (times 2^8, exact; every gemm takes it out again)
    if (r < 512) {  // bias word: joins accumulators of (scaled weights) x (scaled activations), fp32
        packed[i] = v;
        return;
    }
    const int64_t w = r - 512, blk = w >> 9;  // (kb * nt + t), element lane * 8 + e inside it
    const int q = (int)(w & 511);
    nh_f16* const img = (nh_f16*)(packed + la.base[l] + 512);
    const float vs = nh_med3(v, -F16_MAX, F16_MAX);  // (saturate instead of Inf: header)
    const nh_f16 hi = nh_to_f16(vs);
    img[(2 * blk) * 512 + q] = hi;
    img[(2 * blk + 1) * 512 + q] = nh_to_f16(vs - nh_from_f16(hi));
}

}  // namespace

// the fp16-piece layer images of a plan (behind its fp32 image, if it has one): gather, scale, split
int nh_pack_pieces_f16(nerfhip_plan* plan, const float* params, const int32_t* table, float* packed, nerfhip_stream_t stream) {
    const int64_t n = plan->packed_floats, n32 = plan->packed32_floats;
    PackArgs la;
    memset(&la, 0, sizeof(la));
    const NhPackedOffsets& o = plan->pob;
    int k = 0;
    la.base[k++] = o.f_layer1;
    for (int i = 0; i < plan->L - 1; ++i) la.base[k++] = o.f_xyz[i];
    la.base[k++] = o.f_head;
    if (plan->view) {
        la.base[k++] = o.f_dir;
        la.base[k++] = o.f_rgb;
    }
    if (nh_prec_level(plan->precision) >= 3) {  // (the order of plan.cpp for_each_spec_b)
        if (plan->view) {
            la.base[k++] = o.b_rgb;
            la.base[k++] = o.b_dir;
        }
        la.base[k++] = o.b_head;
        for (int i = 0; i < plan->L - 1; ++i) la.base[k++] = o.b_xyz[i];
    }
    la.n_layers = k;
    la.first = n32;
    NH_LAUNCH_NAMED("k_pack_f16x3", k_pack_f16x3, nh_ceil_div(n - n32, 256), 256, 0, stream, params, table, n, la, packed);
    return nh_launch_status("pack_weights_plan");
}

extern "C" int nerfhip_pack_weights_plan(nerfhip_plan_t plan, const float* params, const int32_t* table, float* packed,
                                         nerfhip_stream_t stream) {
    NH_REQUIRE(plan && params && table && packed, "pack_weights_plan: bad arguments");
    const int64_t n = plan->packed_floats;
    if (plan->precision == NERFHIP_PRECISION_FP32) return nerfhip_pack_weights(params, table, n, packed, stream);
    const int64_t n32 = plan->packed32_floats;  // (training-capable plans: the fp32 image in front -- a plain gather)
    if (n32 > 0) {
        const int rc = nerfhip_pack_weights(params, table, n32, packed, stream);
        if (rc) return rc;
    }
    return nh_pack_pieces_f16(plan, params, table, packed, stream);
}

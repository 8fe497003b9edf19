// mlp_f16.hip -- mlp_bf16.hip compiled for IEEE fp16 pieces.  The f16x3 plans (NERFHIP_PRECISION_F16X3*, include/nerfhip.h) run their
// forward / data-gradient chain on the two-waves-per-SIMD kernels of mlp_f16w.hip, so the product build takes the weight-image packer
// (k_pack_f16x3) from here and nothing else.  A/B builds (scripts/build_bf16_variant.sh w1 "-DNHB_W2_DEFAULT=0 -DNHB_F16_ONE_WAVE"
// "plan mlp_f16") compile the one-wave-per-SIMD kernels k_mlp_fwd_f16x3 / k_mlp_dgrad_f16x3 of round 4's first half as well.
#define NHB_F16 1
#ifndef NHB_F16_ONE_WAVE
#define NHB_PACK_ONLY 1
#endif
#include "mlp_bf16.hip"

// mlp_f16.hip -- mlp_bf16.hip compiled for IEEE fp16 pieces: the forward / data-gradient kernels and the weight-image packer of
// the f16x3 plans (NERFHIP_PRECISION_F16X3*, include/nerfhip.h): k_mlp_fwd_f16x3, k_mlp_dgrad_f16x3, k_pack_f16x3.
#define NHB_F16 1
#include "mlp_bf16.hip"

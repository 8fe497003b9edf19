// wgrad_f16.hip -- wgrad_bf16.hip compiled for IEEE fp16 pieces: the large weight-gradient blocks of NERFHIP_PRECISION_F16X3_TRAIN
// plans (k_wgrad_f16x3, k_wgrad_reduce_f16x3).
#define NHB_F16 1
#include "wgrad_bf16.hip"

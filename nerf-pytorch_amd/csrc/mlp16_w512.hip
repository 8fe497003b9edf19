// mlp16_w512.hip -- the 512-wide instantiations of the forward / data-gradient kernels of mlp16.hip (hidden_size in
// (256, 512]: nerf/models.py:186-196 takes any hidden_size): one wave per SIMD with the whole unified register file
// (254 VGPRs + 256 AGPRs, no scratch), 4-wave workgroups, one per CU.  A separate translation unit only because these
// three kernels take minutes to compile; the code is mlp16.hip's.
#define NH16_W512_TU
#include "mlp16.hip"

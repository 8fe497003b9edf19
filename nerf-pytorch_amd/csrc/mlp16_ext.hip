// mlp16_ext.hip -- the forward kernels of mlp16.hip with the extended encoding registers (num_encoding_fn_xyz up to 16,
// num_encoding_fn_dir up to 10: NH16_KRX_EXT / NH16_KRD_EXT of nh_plan.h), every width, compiled as a translation unit of
// their own so that the default kernels' build time does not grow.  The backward kernels do not depend on the slot count.
#define NH16_EXT_TU
#include "mlp16.hip"

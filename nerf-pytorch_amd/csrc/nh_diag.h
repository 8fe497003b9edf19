// nh_diag.h -- the fence around the cost-attribution switches of the kernel sources.
//
// Two kinds of preprocessor switches exist in csrc/.  A/B switches (NHW_DMA_BURST, NHW_STORES_FIRST, NH_WGS_NARROW, ...) select another
// schedule of the SAME arithmetic: every such build passes the parity suite.  Diagnostic switches (NHW_EXP_*: "what does the kernel cost
// without its conversions / its weight stream / its barriers ...") compute WRONG results on purpose.  The product build defines neither
// kind (Makefile HIPFLAGS; tests/test_host_abi.py asserts it), and a diagnostic switch without NH_DIAG does not compile: NH_DIAG is set
// only by `make variant` (Makefile), whose library reports nerfhip_version() + NH_DIAG_VERSION_FLAG and is refused by the Python
// package's get_lib() (nerf-pytorch_amd/_lib.py).
#pragma once

#define NH_DIAG_VERSION_FLAG 1000000

#if !defined(NH_DIAG) &&                                                                                                     \
    (defined(NHW_EXP_NO_EPI) || defined(NHW_EXP_NO_MAX) || defined(NHW_EXP_NO_STORE) || defined(NHW_EXP_NO_BARRIER) ||          \
     defined(NHW_EXP_NO_STREAM) || defined(NHW_EXP_SAME_SRC) || defined(NHW_EXP_HALF_LDS) || defined(NHW_EXP_NO_CONVERT) ||     \
     defined(NHW_EXP_ONE_MFMA) || defined(NHW_EXP_NO_WAIT))
#error "NHW_EXP_* switches build kernels that compute WRONG results: they need -DNH_DIAG (make variant NAME=... DEFS='-DNHW_EXP_...')"
#endif

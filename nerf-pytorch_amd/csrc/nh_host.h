// nh_host.h -- host-side plumbing shared by the C-ABI translation units.
#pragma once
#include "../../include/nerfhip.h"
#include <string.h>

#include "nh_device.h"

void nh_set_error(const char* fmt, ...);

#define NH_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            nh_set_error(__VA_ARGS__); \
            return NERFHIP_ERR_ARG;    \
        }                              \
    } while (0)

// Optional per-kernel timing (nerfhip_profile_enable / _report): while it is on, a launch goes through hipExtLaunchKernelGGL with
// a start and a stop event -- the timestamps of the kernel's own dispatch packet.  (Two hipEventRecord calls around the launch, as
// rounds 1-3 did it, put two more packets on the stream per launch: 0.3 ms of a 2-ms step.)  nh_prof_events hands out the pair of
// launch `name`, or NULLs while profiling is off.
void nh_prof_events(const char* name, void** start, void** stop);
// Device counters of the shader-clock probe for MLP kernel `kind` (NH_CLK_*), or NULL while profiling is off.
unsigned long long* nh_prof_clock_slot(int kind);

#ifdef NERFHIP_EMU
#define NH_LAUNCH(kern, grid, block, smem, stream, ...) \
    emu::launch(emu::Dim3((unsigned)(grid)), emu::Dim3((unsigned)(block)), (size_t)(smem), [&]() { kern(__VA_ARGS__); })
#define NH_LAUNCH_NAMED(name, kern, grid, block, smem, stream, ...) NH_LAUNCH(kern, grid, block, smem, stream, __VA_ARGS__)
static inline int nh_launch_status(const char*) { return NERFHIP_OK; }
#else
#include <hip/hip_ext.h>
#define NH_LAUNCH_NAMED(name, kern, grid, block, smem, stream, ...)                                                     \
    do {                                                                                                                \
        void *nh_e0_ = nullptr, *nh_e1_ = nullptr;                                                                      \
        nh_prof_events(name, &nh_e0_, &nh_e1_);                                                                         \
        if (nh_e0_)                                                                                                     \
            hipExtLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (unsigned)(smem), (hipStream_t)(stream), \
                                  (hipEvent_t)nh_e0_, (hipEvent_t)nh_e1_, 0u, __VA_ARGS__);                             \
        else                                                                                                            \
            hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (size_t)(smem), (hipStream_t)(stream), \
                               __VA_ARGS__);                                                                            \
    } while (0)
// (the profile name defaults to the kernel's spelling; the split-precision translation units, whose kernel names are assembled by
// a macro, spell it out)
#define NH_LAUNCH(kern, grid, block, smem, stream, ...) NH_LAUNCH_NAMED(#kern, kern, grid, block, smem, stream, __VA_ARGS__)
static inline int nh_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        nh_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return NERFHIP_ERR_LAUNCH;
    }
    return NERFHIP_OK;
}
#endif

static inline int64_t nh_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

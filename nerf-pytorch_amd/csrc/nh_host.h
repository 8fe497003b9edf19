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

// Optional per-kernel timing with HIP events recorded on the launch stream (nerfhip_profile_enable / _report).
void nh_prof_begin(const char* name, nerfhip_stream_t stream);
void nh_prof_end(nerfhip_stream_t stream);
// Device counters of the shader-clock probe for MLP kernel `kind` (NH_CLK_*), or NULL while profiling is off.
unsigned long long* nh_prof_clock_slot(int kind);

#ifdef NERFHIP_EMU
#define NH_LAUNCH(kern, grid, block, smem, stream, ...) \
    emu::launch(emu::Dim3((unsigned)(grid)), emu::Dim3((unsigned)(block)), (size_t)(smem), [&]() { kern(__VA_ARGS__); })
#define NH_LAUNCH_NAMED(name, kern, grid, block, smem, stream, ...) NH_LAUNCH(kern, grid, block, smem, stream, __VA_ARGS__)
static inline int nh_launch_status(const char*) { return NERFHIP_OK; }
#else
#define NH_LAUNCH(kern, grid, block, smem, stream, ...)                                                             \
    do {                                                                                                            \
        nh_prof_begin(#kern, stream);                                                                               \
        hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (size_t)(smem), (hipStream_t)(stream), \
                           __VA_ARGS__);                                                                            \
        nh_prof_end(stream);                                                                                        \
    } while (0)
// the same with the profile name spelled out (kernels whose C++ name is assembled by a macro: the split-precision translation units)
#define NH_LAUNCH_NAMED(name, kern, grid, block, smem, stream, ...)                                                 \
    do {                                                                                                            \
        nh_prof_begin(name, stream);                                                                                \
        hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (size_t)(smem), (hipStream_t)(stream), \
                           __VA_ARGS__);                                                                            \
        nh_prof_end(stream);                                                                                        \
    } while (0)
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

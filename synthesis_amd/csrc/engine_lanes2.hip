// synthesis_amd — third translation unit of the library: the two-trees-per-lane kernels (lane2_kernel.cuh), compiled beside
// engine.hip and engine_conv.hip (`make -j3`). engine.hip declares the same instantiations `extern template` and launches them
// through their host stubs.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/synthesis_amd.h"
#include "lane2_kernel.cuh"

namespace syn {
#define SYN_LANES2(MODE, COUNT)                                                               \
    template __global__ void selfplay_kernel_lanes2<MODE, COUNT, true, 8, 0>(EngineParams);   \
    template __global__ void selfplay_kernel_lanes2<MODE, COUNT, false, 8, 0>(EngineParams);  \
    template __global__ void selfplay_kernel_lanes2<MODE, COUNT, true, 8, 2>(EngineParams);   \
    template __global__ void selfplay_kernel_lanes2<MODE, COUNT, false, 8, 2>(EngineParams);  \
    template __global__ void selfplay_kernel_lanes2<MODE, COUNT, true, 12, 0>(EngineParams);  \
    template __global__ void selfplay_kernel_lanes2<MODE, COUNT, false, 12, 0>(EngineParams); \
    template __global__ void selfplay_kernel_lanes2<MODE, COUNT, true, 12, 2>(EngineParams);  \
    template __global__ void selfplay_kernel_lanes2<MODE, COUNT, false, 12, 2>(EngineParams);
template __global__ void selfplay_kernel_lanes2<MODE_SELFPLAY, false, true, 8, 0, 1>(EngineParams);
SYN_LANES2(MODE_SEARCH, false)
SYN_LANES2(MODE_SELFPLAY, false)
SYN_LANES2(MODE_SELFPLAY, true)
#undef SYN_LANES2
}  // namespace syn

// synthesis_amd — second translation unit of the library: the lane-per-tree kernels instantiated for Connect4ConvNet
// (POLICY == 2, convnet.cuh). They are 18 of the ~60 kernels of engine.hip and compile independently of everything else, so
// building them beside engine.hip (`make -j2`) takes the conv network's share out of the build's critical path. engine.hip
// declares the same instantiations `extern template` and launches them through their host stubs.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/synthesis_amd.h"
#include "lane_kernel.cuh"

namespace syn {
#define SYN_CONV_LANES(MODE, COUNT)                                                          \
    template __global__ void selfplay_kernel_lanes<MODE, COUNT, true, 4, false, 2>(EngineParams);   \
    template __global__ void selfplay_kernel_lanes<MODE, COUNT, false, 4, false, 2>(EngineParams);  \
    template __global__ void selfplay_kernel_lanes<MODE, COUNT, true, 8, false, 2>(EngineParams);   \
    template __global__ void selfplay_kernel_lanes<MODE, COUNT, false, 8, false, 2>(EngineParams);  \
    template __global__ void selfplay_kernel_lanes<MODE, COUNT, true, 16, false, 2>(EngineParams);  \
    template __global__ void selfplay_kernel_lanes<MODE, COUNT, false, 16, false, 2>(EngineParams);
SYN_CONV_LANES(MODE_SEARCH, false)
SYN_CONV_LANES(MODE_SELFPLAY, false)
SYN_CONV_LANES(MODE_SELFPLAY, true)
#undef SYN_CONV_LANES
}  // namespace syn

// synthesis_amd — translation unit of the library: the lane-per-tree kernels (lane_kernel.cuh) evaluating Connect4Net in the f16x2
// arithmetic (POLICY 3: f16x2_tile.cuh, two-term f16 split on v_mfma_f32_16x16x32_f16) for the compile-time-folded configuration
// families. engine.hip declares the same instantiations `extern template` (lane_instances.h); built beside it by `make -j`.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/synthesis_amd.h"
#include "lane_kernel.cuh"
#include "lane_instances.h"

namespace syn {
#define SYN_X(MODE, COUNT, FAST, NW, PROF, POLICY) template __global__ void selfplay_kernel_lanes<MODE, COUNT, FAST, NW, PROF, POLICY>(EngineParams);
SYN_LANES_F16_LIST(SYN_X)
#undef SYN_X
}  // namespace syn

// synthesis_amd — translation unit six of the library: the lane-per-tree kernels (lane_kernel.cuh) of the reference's own self-play configuration, Fpu::Func(|| Normal(mean, std)) folded at compile time (CfgView<FAST = 2>),
// Connect4Net / RolloutPolicy. engine.hip declares the same instantiations `extern template` (lane_instances.h) and launches them
// through their host stubs; built beside it by `make -j`.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/synthesis_amd.h"
#include "lane_kernel.cuh"
#include "lane_instances.h"

namespace syn {
#define SYN_X(MODE, COUNT, FAST, NW, PROF, POLICY) template __global__ void selfplay_kernel_lanes<MODE, COUNT, FAST, NW, PROF, POLICY>(EngineParams);
SYN_LANES_REF_LIST(SYN_X)
#undef SYN_X
}  // namespace syn

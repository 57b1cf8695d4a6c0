// synthesis_amd — translation unit of the library: the pool kernels (pool_kernel.cuh: a wave's lanes work on a pool of up to 128 trees,
// a lane binds the next READY tree as soon as its descent arrives) for Connect4Net on the f32 matrix cores (POLICY 0). engine.hip declares
// the same instantiations `extern template` (lane_instances.h); built beside it by `make -j`.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/synthesis_amd.h"
#include "pool_kernel.cuh"
#include "lane_instances.h"

namespace syn {
#define SYN_X(MODE, COUNT, FAST, NW, POLICY) template __global__ void selfplay_kernel_pool<MODE, COUNT, FAST, NW, POLICY>(EngineParams);
SYN_POOL_F32_LIST(SYN_X)
#undef SYN_X
}  // namespace syn

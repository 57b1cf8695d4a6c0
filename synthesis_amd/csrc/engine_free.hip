// synthesis_amd — translation unit of the library: the free-running kernels for at most 16 trees per CU (free_kernel.cuh: four waves of four
// trees, every wave evaluating its own leaves; f16x2 network arithmetic). engine.hip declares the same instantiations `extern template`; built beside it by `make -j`.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/synthesis_amd.h"
#include "free_kernel.cuh"

namespace syn {
#define SYN_FREE(MODE, COUNT)                                                               \
    template __global__ void selfplay_kernel_free<MODE, COUNT, true, false>(EngineParams);  \
    template __global__ void selfplay_kernel_free<MODE, COUNT, false, false>(EngineParams);
SYN_FREE(MODE_SEARCH, false)
SYN_FREE(MODE_SELFPLAY, false)
SYN_FREE(MODE_SELFPLAY, true)
#undef SYN_FREE
template __global__ void selfplay_kernel_free<MODE_SELFPLAY, false, true, true>(EngineParams);
}  // namespace syn

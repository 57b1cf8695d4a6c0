// synthesis_amd — translation unit of the library: the one-tree-per-wave mailbox kernels (mail_kernel.cuh: at most 16 trees per CU,
// f16x2 network arithmetic). engine.hip declares the same instantiations `extern template`; built beside it by `make -j`.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/synthesis_amd.h"
#include "mail_kernel.cuh"

namespace syn {
#define SYN_MAIL(MODE, COUNT)                                                               \
    template __global__ void selfplay_kernel_mail<MODE, COUNT, true, false>(EngineParams);  \
    template __global__ void selfplay_kernel_mail<MODE, COUNT, false, false>(EngineParams);
SYN_MAIL(MODE_SEARCH, false)
SYN_MAIL(MODE_SELFPLAY, false)
SYN_MAIL(MODE_SELFPLAY, true)
#undef SYN_MAIL
template __global__ void selfplay_kernel_mail<MODE_SELFPLAY, false, true, true>(EngineParams);
}  // namespace syn

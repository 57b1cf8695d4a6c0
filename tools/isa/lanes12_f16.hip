// Developer tool: ISA of the f16x2 lane-per-tree kernel at 12 waves (parity family), with line tables for tools/isa/attribute.py
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/synthesis_amd.h"
#include "../../synthesis_amd/csrc/lane_kernel.cuh"
namespace syn {
template __global__ void selfplay_kernel_lanes<MODE_SELFPLAY, false, true, 12, false, 3>(EngineParams);
}

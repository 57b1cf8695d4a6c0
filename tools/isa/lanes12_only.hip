// Developer tool: the 12-wave parity-family instantiation of the lane-per-tree kernel for ISA / register inspection
#include "../../synthesis_amd/csrc/lane_kernel.cuh"
template __global__ void syn::selfplay_kernel_lanes<syn::MODE_SELFPLAY, false, 1, 12, false, 0>(syn::EngineParams);

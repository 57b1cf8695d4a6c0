// Developer tool: compiles only the reference-configuration instantiations of the lane-per-tree kernel for ISA / register inspection
#include "../../synthesis_amd/csrc/lane_kernel.cuh"
template __global__ void syn::selfplay_kernel_lanes<syn::MODE_SELFPLAY, false, 2, 16, false, 0>(syn::EngineParams);
template __global__ void syn::selfplay_kernel_lanes<syn::MODE_SELFPLAY, false, 2, 8, false, 0>(syn::EngineParams);

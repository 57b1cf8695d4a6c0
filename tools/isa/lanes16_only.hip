// Developer tool: compiles only the headline instantiation of the lane-per-tree kernel (16 waves) for ISA / register inspection
#include "../../synthesis_amd/csrc/lane_kernel.cuh"
template __global__ void syn::selfplay_kernel_lanes<syn::MODE_SELFPLAY, false, true, 16, false, 0>(syn::EngineParams);

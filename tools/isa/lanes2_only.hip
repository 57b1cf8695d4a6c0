// Developer tool: compiles only the headline instantiation of the two-trees-per-lane kernel for ISA / register inspection
#include "../../synthesis_amd/csrc/lane2_kernel.cuh"
template __global__ void syn::selfplay_kernel_lanes2<syn::MODE_SELFPLAY, false, true, 8, 0>(syn::EngineParams);

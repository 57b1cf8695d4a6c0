#include "../../synthesis_amd/csrc/lane_kernel.cuh"
template __global__ void syn::selfplay_kernel_lanes<syn::MODE_SELFPLAY, false, 1, 16, false, 2>(syn::EngineParams);
template __global__ void syn::selfplay_kernel_lanes<syn::MODE_SELFPLAY, false, 2, 12, false, 0>(syn::EngineParams);
template __global__ void syn::selfplay_kernel_lanes<syn::MODE_SELFPLAY, false, 1, 8, false, 0>(syn::EngineParams);
template __global__ void syn::selfplay_kernel_lanes<syn::MODE_SELFPLAY, false, 0, 8, false, 0>(syn::EngineParams);
template __global__ void syn::selfplay_kernel_lanes<syn::MODE_SELFPLAY, false, 1, 4, false, 0>(syn::EngineParams);
template __global__ void syn::selfplay_kernel_lanes<syn::MODE_SELFPLAY, false, 1, 8, false, 2>(syn::EngineParams);

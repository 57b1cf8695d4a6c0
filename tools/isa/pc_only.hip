// Developer tool: compiles only the headline instantiation of the producer/consumer kernel so its ISA and register report
// can be inspected in seconds:  tools/isa/build.sh  ->  tools/isa/pc_only.s + pc_only.txt
#include "../../synthesis_amd/csrc/pc_kernel.cuh"
template __global__ void syn::selfplay_kernel_pc<syn::MODE_SELFPLAY, false, true, false>(syn::EngineParams);

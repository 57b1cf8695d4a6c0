// Developer tool: compiles only the stand-alone Connect4ConvNet evaluation kernel for ISA inspection
#include "../../synthesis_amd/csrc/convnet.cuh"
template __global__ void syn::policy_eval_conv_kernel<512>(const float*, const unsigned long long*, const unsigned long long*, int, float*, float*);

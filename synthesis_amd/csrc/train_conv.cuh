// synthesis_amd — Connect4ConvNet learner: parameter layout shared by the matrix-core learner (train_conv_mfma.cuh) and the
// device-side rebuild of the inference image (conv_image_kernel: publishing the trained network to the self-play engine without a
// host round trip). Round 2's VALU gradient kernel lived here; it was replaced by train_conv_mfma.cuh (4x faster, and the order
// of two of its chains — the head forward and the conv-gradient partials — is now the matrix cores').
#pragma once
#include "convnet.cuh"
#include "train_kernels.cuh"

namespace syn {

struct ConvTrainGeom {
    static constexpr int CHUNK = 32, FLAT = ConvGeom::FLAT, HW = ConvGeom::HW, C = ConvGeom::C;
    // canonical parameter offsets: conv.weight[16][2][3][3], conv.bias[16], head.weight[12][1008], head.bias[12]
    static constexpr int P_CW = 0, P_CB = ConvGeom::CONV_W, P_HW = P_CB + C, P_HB = P_HW + 12 * FLAT;
};

// canonical Connect4ConvNet parameters -> the fragment image of convnet.cuh (device side of build_conv_image: publishing the
// trained network to the self-play engine without a host round trip)
__global__ void conv_image_kernel(const float* __restrict__ blob, float* __restrict__ img) {
    using G = ConvGeom;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G::IMG_FLOATS) return;
    float v = 0.0f;
    if (i < G::CONVA_OFF) {
        const int p = i / 256, rem = i - p * 256, lane = rem >> 2, r = rem & 3;
        const int o = lane & 15, ch = 4 * (lane >> 4) + r;
        v = o < G::OUT ? blob[ConvTrainGeom::P_HW + o * G::FLAT + ch * G::HW + p] : 0.0f;
    } else if (i < G::CBIAS_OFF) {
        const int j = i - G::CONVA_OFF, s = j / 64, lane = j - s * 64;
        const int ch = lane & 15, t = 4 * s + (lane >> 4);
        v = t < 18 ? blob[ch * 18 + t] : 0.0f;
    } else if (i < G::HBIAS_OFF) {
        v = blob[ConvTrainGeom::P_CB + (i - G::CBIAS_OFF)];  // [q][r] = channel 4 q + r: the natural order
    } else {
        const int o = i - G::HBIAS_OFF;
        v = o < G::OUT ? blob[ConvTrainGeom::P_HB + o] : 0.0f;
    }
    img[i] = v;
}

}  // namespace syn

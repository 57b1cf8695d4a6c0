// synthesis_amd — the learner step for Connect4ConvNet (convnet.cuh; the conv policy/value network of north_star): forward,
// log_softmax + kl_div, backward of a minibatch of up to 32 positions in ONE workgroup; Adam is adam_kernel (train_kernels.cuh).
//
// The reference has neither this network nor a learner of its own for it (alpha_zero.rs:72-94 drives libtorch's autograd over
// whatever NNPolicy it is given): the published semantics restated in oracle/train.hpp apply unchanged, and every f32 chain here
// runs in the fixed order of oracle/train.hpp::ConvTrainer, so the kernel is bit-identical to it:
//   conv forward   per output (sample, channel, cell): bias, then the in-board taps in slimnn's order ci -> k1 -> k2, fma per tap
//   head forward   per (sample, output): bias, then the 1008 activations in NCHW flattening order, fma per term
//   dWh[o][i]      fma chain over the samples ascending;  dbh[o] plain sum over the samples
//   dY[b][i]       relu'(act) * (fma chain over the 12 outputs ascending)       (overwrites the activations in LDS)
//   dWc[c][tap]    eight partial fma chains over the sample groups [4 g, 4 g + 4) (samples ascending, cells row-major inside a
//                  sample), added in order ((p0 + p1) + p2) + ...; dbc[c] plain sums in the same order
// Everything runs on the VALU with the 32 x 1008 activations resident in LDS (129 KB): a (sample, cell)'s 18 tap inputs are
// extracted once for all 16 channels in the forward pass, the tap inputs of dWc are bits of convnet.cuh's pre-shifted boards, a
// thread owns one head input column for all 12 outputs in dWh. The matrix-core forms (the inference tile for the forward, a
// [16 x 2016] x [2016 x 18] GEMM for dWc) are the known next step. Measured in DESIGN.md §6.4.
#pragma once
#include "convnet.cuh"
#include "train_kernels.cuh"

namespace syn {

struct ConvTrainGeom {
    static constexpr int CHUNK = 32, FLAT = ConvGeom::FLAT, HW = ConvGeom::HW, C = ConvGeom::C;
    static constexpr int ASTR = FLAT + 1;   // LDS row stride of a sample's activations: odd, so that lanes = samples hit 32 banks
    static constexpr int NPART = 8;         // dWc / dbc: partial chains over the sample groups [4 g, 4 g + 4), added in order
    // LDS (floats): activations / dY, the 12 raw outputs, their gradients, per-sample KL terms, conv parameters, boards
    static constexpr int ACT_OFF = 0;
    static constexpr int OUT_OFF = ACT_OFF + CHUNK * ASTR;
    static constexpr int DZ_OFF = OUT_OFF + CHUNK * 12;
    static constexpr int KL_OFF = DZ_OFF + CHUNK * 12;
    static constexpr int CW_OFF = KL_OFF + CHUNK * 2;
    static constexpr int CB_OFF = CW_OFF + ConvGeom::CONV_W;
    static constexpr int BB_OFF = (CB_OFF + C + 1) & ~1;          // [CHUNK][2] u64
    static constexpr int PART_OFF = BB_OFF + CHUNK * 4;           // [NPART sample groups][288 + 16] partial conv gradients
    static constexpr int LDS_FLOATS = PART_OFF + NPART * (ConvGeom::CONV_W + C);
    // canonical parameter offsets
    static constexpr int P_CW = 0, P_CB = ConvGeom::CONV_W, P_HW = P_CB + C, P_HB = P_HW + 12 * FLAT;
};

// in-board source cell (row-major index) of tap (k1, k2) for output cell (r, col), or -1
SYN_DEV int conv_tap_src(int r, int col, int k1, int k2) {
    const int rr = r + k1 - 1, cc = col + k2 - 1;
    return (rr >= 0 && rr < 7 && cc >= 0 && cc < 9) ? rr * 9 + cc : -1;
}
// plane value (0 / 1) of row-major cell `cell` (bit row + 7 col of the board)
SYN_DEV float conv_plane_bit(uint64_t bb, int cell) {
    const int rr = cell / 9, cc = cell - 9 * rr;
    return (float)((uint32_t)(bb >> (rr + 7 * cc)) & 1u);
}

// grads[12412] receives d(loss)/d(param) of the minibatch (B <= 32); losses[0..1] = pi_loss, v_loss
__global__ __launch_bounds__(1024) void train_conv_grad_kernel(const float* __restrict__ w, const unsigned long long* __restrict__ my_bb,
                                                               const unsigned long long* __restrict__ op_bb,
                                                               const float* __restrict__ tpi, const float* __restrict__ tv, int B,
                                                               DevTrainHyper hp, float* __restrict__ grads, float* __restrict__ losses,
                                                               const int* __restrict__ idx = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using G = ConvTrainGeom;
    const int tid = threadIdx.x;
    float* act = lds + G::ACT_OFF;
    float* out = lds + G::OUT_OFF;
    float* dz = lds + G::DZ_OFF;
    uint64_t* bb = reinterpret_cast<uint64_t*>(lds + G::BB_OFF);
    const float* hw = w + G::P_HW;
    const float* hb = w + G::P_HB;
    const float bm = 1.0f / (float)B;

    // ---- stage the boards; the targets of the (sample, head) threads
    if (tid < 2 * B) {
        const int b = tid >> 1;
        const size_t si = idx ? (size_t)idx[b] : (size_t)b;
        bb[tid] = (tid & 1) ? op_bb[si] : my_bb[si];
    }
    float tgt[9];
#pragma unroll
    for (int j = 0; j < 9; j++) tgt[j] = 0.0f;
    if (tid < 2 * G::CHUNK && (tid >> 1) < B) {
        const int b = tid >> 1;
        const size_t si = idx ? (size_t)idx[b] : (size_t)b;
        if ((tid & 1) == 0) {
#pragma unroll
            for (int j = 0; j < 9; j++) tgt[j] = tpi[si * 9 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 3; j++) tgt[j] = tv[si * 3 + j];
        }
    }
    __syncthreads();

    // ---- conv forward + ReLU: one (sample, cell) per thread and step — its 18 tap inputs (0 / 1; a tap outside the board reads 0,
    //      and fma(w, 0, acc) = acc exactly, so no test is needed) are extracted once and serve all 16 channels
    for (int it = tid; it < B * G::HW; it += 1024) {
        const int b = it / G::HW, p = it - b * G::HW;
        const int r = p / 9, col = p - 9 * r;
        float xf[18];
#pragma unroll
        for (int ci = 0; ci < 2; ci++) {
            const uint64_t plane = bb[2 * b + ci];
#pragma unroll
            for (int k1 = 0; k1 < 3; k1++)
#pragma unroll
                for (int k2 = 0; k2 < 3; k2++) {
                    const int rr = r + k1 - 1, cc = col + k2 - 1;
                    const bool in = rr >= 0 && rr < 7 && cc >= 0 && cc < 9;
                    xf[(ci * 3 + k1) * 3 + k2] = in ? (float)((uint32_t)(plane >> (rr + 7 * cc)) & 1u) : 0.0f;
                }
        }
        float* ab = act + b * G::ASTR + p;
#pragma unroll 4
        for (int c = 0; c < G::C; c++) {
            // the weights are wave-uniform: read through the scalar cache, they reach the fma as scalar operands
            float acc = w[G::P_CB + c];
#pragma unroll
            for (int t = 0; t < 18; t++) acc = __builtin_fmaf(w[G::P_CW + c * 18 + t], xf[t], acc);
            ab[c * G::HW] = acc > 0.0f ? acc : 0.0f;
        }
    }
    __syncthreads();

    // ---- head forward: wave o (12 of the 16) runs output o for all samples, lane = sample: the weight row is wave-uniform (scalar
    //      loads, scalar fma operand), the activations come from LDS rows whose odd stride spreads the lanes over the banks
    {
        const int o = tid >> 6, b = tid & 63;
        if (o < 12 && b < B) {
            const float* a = act + b * G::ASTR;
            const float* wr = hw + (size_t)o * G::FLAT;
            float acc = hb[o];
#pragma unroll 16
            for (int i = 0; i < G::FLAT; i++) acc = __builtin_fmaf(a[i], wr[i], acc);
            out[b * 12 + o] = acc;
        }
    }
    __syncthreads();

    // ---- heads: log_softmax + kl_div and their gradient; one thread per (sample, head) — as train_grad_kernel
    if (tid < 2 * G::CHUNK) {
        const int b = tid >> 1, head = tid & 1;
        const int off = head == 0 ? 0 : 9, n = head == 0 ? 9 : 3;
        float kl = 0.0f;
        if (b < B) {
            const float* x = out + b * 12 + off;
            const float weight = head == 0 ? hp.policy_weight : hp.value_weight;
            float xv[9];
#pragma unroll
            for (int j = 0; j < 9; j++) xv[j] = j < n ? x[j] : 0.0f;
            float mx = xv[0];
#pragma unroll
            for (int j = 1; j < 9; j++) mx = (j < n && xv[j] > mx) ? xv[j] : mx;
            float se = 0.0f;
#pragma unroll
            for (int j = 0; j < 9; j++)
                if (j < n) se += det_expf(xv[j] - mx);
            const float lse = mx + det_logf(se);
            float tsum = 0.0f;
#pragma unroll
            for (int j = 0; j < 9; j++) {
                if (j < n) {
                    const float logp = xv[j] - lse;
                    if (tgt[j] > 0.0f) kl += tgt[j] * (det_logf(tgt[j]) - logp);
                    tsum += tgt[j];
                }
            }
            const float s = weight * bm;
#pragma unroll
            for (int j = 0; j < 9; j++)
                if (j < n) dz[b * 12 + off + j] = s * (det_expf(xv[j] - lse) * tsum - tgt[j]);
        }
        lds[G::KL_OFF + b * 2 + head] = kl;
    }
    __syncthreads();
    if (tid == 0) {
        float pi_acc = 0.0f, v_acc = 0.0f;
        for (int b = 0; b < B; b++) {
            pi_acc += lds[G::KL_OFF + b * 2 + 0];
            v_acc += lds[G::KL_OFF + b * 2 + 1];
        }
        losses[0] = bm * pi_acc;
        losses[1] = bm * v_acc;
    }

    // ---- head parameter gradients: a thread owns input column i for all 12 outputs (one activation read per sample feeds 12
    //      chains); dbh[o] plain sums
    if (tid < G::FLAT) {
        float acc[12];
#pragma unroll
        for (int o = 0; o < 12; o++) acc[o] = 0.0f;
        for (int b = 0; b < B; b++) {
            const float a = act[b * G::ASTR + tid];
#pragma unroll
            for (int o = 0; o < 12; o++) acc[o] = __builtin_fmaf(dz[b * 12 + o], a, acc[o]);
        }
#pragma unroll
        for (int o = 0; o < 12; o++) grads[G::P_HW + o * G::FLAT + tid] = acc[o];
    } else if (tid < G::FLAT + 12) {
        const int o = tid - G::FLAT;
        float a = 0.0f;
        for (int b = 0; b < B; b++) a += dz[b * 12 + o];
        grads[G::P_HB + o] = a;
    }
    __syncthreads();

    // ---- activation gradients through the ReLU, in place: dY[b][i]; a thread owns column i: its 12 head weights are loaded once
    if (tid < G::FLAT) {
        float wcol[12];
#pragma unroll
        for (int o = 0; o < 12; o++) wcol[o] = hw[(size_t)o * G::FLAT + tid];
        for (int b = 0; b < B; b++) {
            float a = 0.0f;
#pragma unroll
            for (int o = 0; o < 12; o++) a = __builtin_fmaf(dz[b * 12 + o], wcol[o], a);
            float* p = act + b * G::ASTR + tid;
            *p = *p > 0.0f ? a : 0.0f;
        }
    }
    __syncthreads();

    // ---- conv parameter gradients: a partial chain per (sample group g of 4, channel, tap) over the group's samples and their cells
    //      row-major (the tap's input for a cell is one bit of the pre-shifted board of convnet.cuh, conv_tap_board: a set bit adds
    //      dY — fma(dY, 1, a) —, a clear one adds +0); the eight partials meet in LDS and are added in order. The conv bias likewise.
    float* part = lds + G::PART_OFF;
    constexpr int NCH = ConvGeom::CONV_W + G::C;
    for (int it = tid; it < G::NPART * NCH; it += 1024) {
        const int gq = it / NCH, j = it - gq * NCH;
        float a = 0.0f;
        const int b1 = 4 * gq + 4 < B ? 4 * gq + 4 : B;
        if (j < ConvGeom::CONV_W) {
            const int c = j / 18, t = j - 18 * c;
            for (int b = 4 * gq; b < b1; b++) {
                const uint64_t S = conv_tap_board(bb[2 * b], bb[2 * b + 1], t);
                const float* dY = act + b * G::ASTR + c * G::HW;
                for (int r = 0; r < 7; r++) {
                    const uint64_t Sr = S >> r;
                    const uint32_t lo = (uint32_t)Sr, hi = (uint32_t)(Sr >> 32);
#pragma unroll
                    for (int col = 0; col < 9; col++) {
                        // sign-extended 1-bit field: 0 or ~0 — the mask of the term
                        const int m = 7 * col < 32 ? __builtin_amdgcn_sbfe((int)lo, 7 * col, 1) : __builtin_amdgcn_sbfe((int)hi, 7 * col - 32, 1);
                        a += bits_f32(f32_bits(dY[r * 9 + col]) & (uint32_t)m);
                    }
                }
            }
        } else {
            const int c = j - ConvGeom::CONV_W;
            for (int b = 4 * gq; b < b1; b++)
                for (int p = 0; p < G::HW; p++) a += act[b * G::ASTR + c * G::HW + p];
        }
        part[it] = a;
    }
    __syncthreads();
    if (tid < NCH) {
        float v = part[tid];
#pragma unroll
        for (int gq = 1; gq < G::NPART; gq++) v += part[gq * NCH + tid];
        grads[tid < ConvGeom::CONV_W ? G::P_CW + tid : G::P_CB + (tid - ConvGeom::CONV_W)] = v;
    }
}

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

// synthesis_amd — the learner step for Connect4ConvNet on the f32 matrix cores: forward, log_softmax + kl_div, backward of a
// minibatch of up to 32 positions by ONE 16-wave workgroup, every GEMM-shaped chain a k-ordered fma chain of
// v_mfma_f32_16x16x4_f32; and the persistent epoch kernel around it (all steps of an epoch in one launch, Adam inside).
//
// The reference has neither this network nor a learner of its own for it (alpha_zero.rs:72-94 drives libtorch's autograd over
// whatever NNPolicy it is given): the published semantics restated in oracle/train.hpp apply unchanged, and every f32 chain here
// runs in the fixed order of oracle/train.hpp::ConvTrainer, so the kernels are bit-identical to it. With
// A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], D[4 (lane >> 4) + r][lane & 15] (MI355X guide §3) and
// q = lane >> 4, j = lane & 15:
//   F   conv forward + head partials. Wave w owns the board cells p = w, w + 16, w + 32, w + 48 (< 63) for both 16-sample tiles:
//       conv    D[channel][sample] = bias + sum over the 18 taps — A = conv weights (five registers per lane), B = one BIT of the
//               sample's pre-shifted boards (convnet.cuh conv_tap_board), 5 MFMAs, taps in slimnn's order;
//       ReLU -> LDS act[sample][channel * 63 + p]  (what dWh, dY and dWc read), and straight from the registers
//       head    D[output][sample] += Wh[output][channel * 63 + p] * act — 4 MFMAs per cell, channels 0,4,8,12, 1,5,9,13, ...;
//               the wave's partial sums over its cells go to LDS, bias + P0 + P1 + ... + P15 in order makes the 12 outputs.
//   H   log_softmax / kl_div / dz per (sample, head) on the VALU (det_expf / det_logf), as train_kernels.cuh.
//   G1  dWh[o][i] = chain over the samples: A = dz^T (outputs x samples), B = act (samples x 16 input columns): 8 MFMAs per
//       16-column tile, 63 tiles dealt over the waves; dbh plain sums.
//   G2  dY[b][i] = relu'(act) * chain over the 12 outputs: A = Wh (16 input columns x outputs), B = dz^T (outputs x 16 samples):
//       3 MFMAs per (column tile, sample tile); overwrites act in LDS.
//   G3  dWc[c][tap] and dbc[c]: wave w owns the sample pair 2 w, 2 w + 1: A = dY (channels x 4 cells), B = the tap's input bit
//       of those cells (taps 0..15 in one column tile, taps 16, 17 and an all-ones "bias tap" in a second): 16 k-steps per sample
//       and tile (cell 63 is padding: A = 0); the sixteen partial results meet in LDS and are added in order.
// 72 + 32 + 24 + 64 MFMAs per wave and step. Measured in DESIGN.md §6.4.
#pragma once
#include "train_conv.cuh"
#include "train_epoch.cuh"

namespace syn {

struct ConvMfmaGeom {
    static constexpr int CHUNK = 32, FLAT = ConvGeom::FLAT, HW = ConvGeom::HW, C = ConvGeom::C;
    static constexpr int ASTR = FLAT + 1;                          // LDS row stride of a sample's activations (odd)
    static constexpr int ACT_OFF = 0;
    static constexpr int OUT_OFF = ACT_OFF + CHUNK * ASTR;         // [32][12] raw outputs
    static constexpr int DZ_OFF = OUT_OFF + CHUNK * 12;            // [32][12] d(loss)/d(output); rows >= B stay zero
    static constexpr int KL_OFF = DZ_OFF + CHUNK * 12;             // [32][2]
    static constexpr int BB_OFF = (KL_OFF + CHUNK * 2 + 1) & ~1;   // [32][2] u64 boards; rows >= B zero
    // one region, two uses: the head's partial outputs [wave 16][sample 32][12] (phase F -> H), then the conv gradients'
    // partial sums [wave 16][channel 16][20: taps 0..17, bias, pad] (phase G3 -> G4)
    static constexpr int PART_OFF = BB_OFF + CHUNK * 4;
    static constexpr int PART_FLOATS = 16 * CHUNK * 12;
    static constexpr int LDS_FLOATS = PART_OFF + PART_FLOATS;      // 39,392 floats = 157,568 B
    static constexpr int P_CW = 0, P_CB = ConvGeom::CONV_W, P_HW = P_CB + C, P_HB = P_HW + 12 * FLAT;
};
static_assert(16 * 16 * 20 <= ConvMfmaGeom::PART_FLOATS, "conv-gradient partials fit the head-partial region");
static_assert(ConvMfmaGeom::LDS_FLOATS * 4 <= 160 * 1024, "one workgroup: 160 KB of LDS");

// sum over b < B of x[b * stride], added in sample order (the oracle's order) — all 32 values are requested first so that their LDS
// latencies overlap instead of forming a chain of 32 dependent round trips (rows >= B are not added: x + 0 could flip a -0)
SYN_DEV float conv_ordered_sum32(const float* x, int stride, int B) {
    float v[ConvMfmaGeom::CHUNK];
#pragma unroll
    for (int b = 0; b < ConvMfmaGeom::CHUNK; b++) v[b] = x[b * stride];
    float a = 0.0f;
#pragma unroll
    for (int b = 0; b < ConvMfmaGeom::CHUNK; b++) a = b < B ? a + v[b] : a;
    return a;
}

// One minibatch (B <= 32 samples; sample b = my_bb[idx ? idx[b] : b]): grads[12412] <- d(loss)/d(param), losses[0..1] <- pi / v
// loss. Called by all 1024 threads of a workgroup; ends with every gradient written (no trailing barrier).
template <int NT>
SYN_DEV void conv_grad_step_mfma(const float* __restrict__ w, const unsigned long long* __restrict__ my_bb,
                                 const unsigned long long* __restrict__ op_bb, const float* __restrict__ tpi,
                                 const float* __restrict__ tv, int B, const DevTrainHyper& hp, float* __restrict__ grads,
                                 float* __restrict__ losses, const int* __restrict__ idx, float* lds, int tid, unsigned long long* prof = nullptr) {
    int pk = 0;   // SYN_TRAIN_PROFILE: cycle stamps of thread 0 after every barrier (stage, F, H, G1, G2, G3)
#define CONV_STAMP() do { if (prof && tid == 0) prof[pk++] = (unsigned long long)__builtin_readcyclecounter(); } while (0)
    CONV_STAMP();
    using G = ConvMfmaGeom;
    const int lane = tid & 63, rw = tid >> 6, j = lane & 15, q = lane >> 4;
    constexpr int NWV = NT / 64;   // real waves; the sixteen chains' owners ("virtual waves" wv) are dealt over them
    float* act = lds + G::ACT_OFF;
    float* dz = lds + G::DZ_OFF;
    float* part = lds + G::PART_OFF;
    uint64_t* bb = reinterpret_cast<uint64_t*>(lds + G::BB_OFF);
    const float bm = 1.0f / (float)B;

    // ---- stage the boards (samples >= B: empty boards, zero dz rows -> exact zeros in every gradient chain); targets
    if (tid < 2 * G::CHUNK) {
        const int b = tid >> 1;
        unsigned long long v = 0ull;
        if (b < B) {
            const size_t si = idx ? (size_t)idx[b] : (size_t)b;
            v = (tid & 1) ? op_bb[si] : my_bb[si];
        }
        bb[tid] = v;
    }
    // heads: 16 lanes per sample, one output entry per lane (0..8 policy, 9..11 outcome); its target and the target's logarithm
    static_assert(NT == 16 * G::CHUNK, "the heads phase maps thread -> (sample tid >> 4, entry tid & 15)");
    const int hb = tid >> 4, jx = tid & 15;
    float tgt = 0.0f;
    if (hb < B && jx < 12) {
        const size_t si = idx ? (size_t)idx[hb] : (size_t)hb;
        tgt = jx < 9 ? tpi[si * 9 + jx] : tv[si * 3 + (jx - 9)];
    }
    const float ltgt = tgt > 0.0f ? det_logf(tgt) : 0.0f;
    __syncthreads();
    CONV_STAMP();

    // ---- F: conv forward + ReLU + head partials
    for (int wv = rw; wv < 16; wv += NWV) {
            float ca[5];
    #pragma unroll
            for (int s = 0; s < 5; s++) ca[s] = 4 * s + q < 18 ? w[G::P_CW + j * 18 + 4 * s + q] : 0.0f;   // A[channel j][tap 4 s + q]
            const f32x4 cbv = *reinterpret_cast<const f32x4*>(w + G::P_CB + 4 * q);                         // D rows: channels 4 q + r
            // the head weights of this wave's (up to four) cells: A[output j][channel 4 q + r] per cell, loaded once for both sample tiles
            float hwv[4][4];
    #pragma unroll
            for (int c = 0; c < 4; c++)
    #pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int p = wv + 16 * c;
                    hwv[c][r] = (j < 12 && p < G::HW) ? w[G::P_HW + (size_t)j * G::FLAT + (4 * q + r) * G::HW + p] : 0.0f;
                }
    #pragma unroll 1
            for (int t = 0; t < 2; t++) {
                const int sample = 16 * t + j;
                const uint64_t my = bb[2 * sample], op = bb[2 * sample + 1];
                uint64_t S[5];
    #pragma unroll
                for (int s = 0; s < 5; s++) S[s] = conv_tap_board(my, op, 4 * s + q);
                f32x4 hacc = {0.0f, 0.0f, 0.0f, 0.0f};
    #pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int p = wv + 16 * c;
                    if (p < G::HW) {   // (wave-uniform: only wave 15 has three cells)
                        const int row = p / 9, col = p - 9 * row, pos = row + 7 * col;
                        f32x4 acc = cbv;
    #pragma unroll
                        for (int s = 0; s < 5; s++)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[s], (float)((uint32_t)(S[s] >> pos) & 1u), acc, 0, 0, 0);
    #pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float a = acc[r] > 0.0f ? acc[r] : 0.0f;
                            act[sample * G::ASTR + (4 * q + r) * G::HW + p] = a;
                            hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(hwv[c][r], a, hacc, 0, 0, 0);
                        }
                    }
                }
                if (q < 3) {
    #pragma unroll
                    for (int r = 0; r < 4; r++) part[(wv * G::CHUNK + sample) * 12 + 4 * q + r] = hacc[r];
                }
            }
    }
    __syncthreads();
    CONV_STAMP();

    // ---- H: the 12 outputs (bias + the sixteen partials in order), then log_softmax + kl_div and their gradient
    {
        // this lane's output: bias + the sixteen partials in order; then log_softmax + kl_div and their gradient with the row's
        // values exchanged as DPP row broadcasts (train_epoch.cuh ep_row_gather / ep_row_sum: sums sequential in entry order)
        const bool pol = jx < 9;
        const bool live = jx < 12 && hb < B;
        float xo = 0.0f;
        if (jx < 12) {
            xo = w[G::P_HB + jx];
#pragma unroll
            for (int g = 0; g < 16; g++) xo += part[(g * G::CHUNK + hb) * 12 + jx];
        }
        xo = live ? xo : 0.0f;
        float xs[12];
        ep_row_gather(xo, xs);
        float mxp = xs[0], mxv = xs[9];
#pragma unroll
        for (int t = 1; t < 9; t++) mxp = xs[t] > mxp ? xs[t] : mxp;
#pragma unroll
        for (int t = 10; t < 12; t++) mxv = xs[t] > mxv ? xs[t] : mxv;
        const float mx = pol ? mxp : mxv;
        const float e = live ? det_expf(xo - mx) : 0.0f;
        const float se = ep_row_sum(e, pol), tsum = ep_row_sum(tgt, pol);
        const float lse = mx + det_logf(live ? se : 1.0f);
        const float logp = xo - lse;
        const float term = (live && tgt > 0.0f) ? tgt * (ltgt - logp) : 0.0f;
        const float kl = ep_row_sum(term, pol);
        const float sc = (pol ? hp.policy_weight : hp.value_weight) * bm;
        if (jx < 12) dz[hb * 12 + jx] = live ? sc * (det_expf(xo - lse) * tsum - tgt) : 0.0f;   // rows >= B: zeros
        if (jx == 0 || jx == 9) lds[G::KL_OFF + hb * 2 + (pol ? 0 : 1)] = hb < B ? kl : 0.0f;
    }
    __syncthreads();
    CONV_STAMP();
    if (tid < 2) losses[tid] = bm * conv_ordered_sum32(lds + G::KL_OFF + tid, 2, B);

    // ---- G1: head parameter gradients
    {
        float dza[8];   // A[output j][sample 4 s + q]
#pragma unroll
        for (int s = 0; s < 8; s++) dza[s] = j < 12 ? dz[(4 * s + q) * 12 + j] : 0.0f;
#pragma unroll 1
        for (int ct = rw; ct < G::FLAT / 16; ct += NWV) {
            const int col = 16 * ct + j;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < 8; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dza[s], act[(4 * s + q) * G::ASTR + col], acc, 0, 0, 0);
            if (q < 3) {
#pragma unroll
                for (int r = 0; r < 4; r++) grads[G::P_HW + (size_t)(4 * q + r) * G::FLAT + col] = acc[r];
            }
        }
        if (tid >= NT - 12) {   // dbh: plain sums over the samples (threads of the last wave, which has the fewest tiles)
            const int o = tid - (NT - 12);
            grads[G::P_HB + o] = conv_ordered_sum32(dz + o, 12, B);
        }
    }
    __syncthreads();
    CONV_STAMP();

    // ---- G2: activation gradients through the ReLU, in place
    {
        float dzb[2][3];   // B[output 4 s + q][sample 16 bt + j]
#pragma unroll
        for (int bt = 0; bt < 2; bt++)
#pragma unroll
            for (int s = 0; s < 3; s++) dzb[bt][s] = dz[(16 * bt + j) * 12 + 4 * s + q];
        // column tiles dealt over the real waves, four at a time: all their weight loads first
#pragma unroll 1
        for (int base = rw; base < G::FLAT / 16; base += 4 * NWV) {
            float wa[4][3];   // A[input column 16 ct + j][output 4 s + q]
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    const int ct = base + NWV * c;
                    wa[c][s] = ct < G::FLAT / 16 ? w[G::P_HW + (size_t)(4 * s + q) * G::FLAT + 16 * ct + j] : 0.0f;
                }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int ct = base + NWV * c;
                if (ct < G::FLAT / 16) {
#pragma unroll
                    for (int bt = 0; bt < 2; bt++) {
                        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int s = 0; s < 3; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][s], dzb[bt][s], acc, 0, 0, 0);
                        float* pa = act + (16 * bt + j) * G::ASTR + 16 * ct + 4 * q;   // D rows: input columns 16 ct + 4 q + r
#pragma unroll
                        for (int r = 0; r < 4; r++) pa[r] = pa[r] > 0.0f ? acc[r] : 0.0f;
                    }
                }
            }
        }
    }
    __syncthreads();
    CONV_STAMP();

    // ---- G3: conv parameter gradients, one partial per wave (= sample pair)
    for (int wv = rw; wv < 16; wv += NWV) {
            const FeatureTable FT = make_feature_table(q);   // bit position (row + 7 col) of cell 4 m + q; 63 (an always-clear bit) for cell 63
            f32x4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0;
    #pragma unroll 1
            for (int k = 0; k < 2; k++) {
                const int b = 2 * wv + k;
                const uint64_t my = bb[2 * b], op = bb[2 * b + 1];
                const uint64_t S0 = conv_tap_board(my, op, j);                                               // taps 0..15
                const uint64_t S1 = j < 2 ? conv_tap_board(my, op, 16 + j) : (j == 2 ? c4::FULL : 0ull);     // taps 16, 17, the bias "tap"
                const float* ya = act + b * G::ASTR + j * G::HW + q;                                         // A[channel j][cell 4 s + q]
    #pragma unroll
                for (int s = 0; s < 16; s++) {
                    const uint32_t pos = (FT.t[s >> 2] >> (8 * (s & 3))) & 0xFFu;
                    const float y = 4 * s + q < G::HW ? ya[4 * s] : 0.0f;
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, (float)((uint32_t)(S0 >> pos) & 1u), a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, (float)((uint32_t)(S1 >> pos) & 1u), a1, 0, 0, 0);
                }
            }
            // D[channel 4 q + r][tap 16 n + j]
    #pragma unroll
            for (int r = 0; r < 4; r++) {
                part[(wv * 16 + 4 * q + r) * 20 + j] = a0[r];
                if (j < 3) part[(wv * 16 + 4 * q + r) * 20 + 16 + j] = a1[r];
            }
    }
    __syncthreads();
    CONV_STAMP();
    // ---- G4: the sixteen partials, added in order
    if (tid < ConvGeom::CONV_W + G::C) {
        const int c = tid < ConvGeom::CONV_W ? tid / 18 : tid - ConvGeom::CONV_W;
        const int t = tid < ConvGeom::CONV_W ? tid - 18 * c : 18;
        float v = part[c * 20 + t];
#pragma unroll
        for (int g = 1; g < 16; g++) v += part[(g * 16 + c) * 20 + t];
        grads[tid < ConvGeom::CONV_W ? G::P_CW + tid : G::P_CB + c] = v;
    }
    CONV_STAMP();
#undef CONV_STAMP
}

// ---------------------------------------------------------------------------------------------- bf16 variant
// BASELINE configs[4] words the on-node training step "bf16 conv": the same step with every matrix operand rounded to bf16
// (round-to-nearest-even, v_cvt_pk_bf16_f32) and multiplied on the bf16 matrix cores (v_mfma_f32_16x16x16_bf16: A[i][4 q + e],
// B[4 q + e][j], e = 0..3, f32 accumulation), master weights, Adam moments, the softmax / KL head and every accumulator in f32.
// K = 16 per instruction: the conv is 2 MFMAs per cell (18 taps, padded to 32), the head ONE (its k = the 16 channels: a lane's
// four ReLU outputs are exactly its four k elements), dWh 2 per column tile, dY 1, dWc 4 per sample and tap tile. NOT bit-exact
// with anything and never used for inference (bf16 cannot hold the 1e-5 bar): tests hold its gradients and an 8-step run to the f32
// learner within bf16's error. Tap inputs (0 / 1) are exact in bf16; what is rounded: weights, activations, dz, dY.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
SYN_DEV bf16x4 pack_bf16(float a, float b, float c, float d) { return bf16x4{(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d}; }

template <int NT>
SYN_DEV void conv_grad_step_bf16(const float* __restrict__ w, const unsigned long long* __restrict__ my_bb,
                                 const unsigned long long* __restrict__ op_bb, const float* __restrict__ tpi,
                                 const float* __restrict__ tv, int B, const DevTrainHyper& hp, float* __restrict__ grads,
                                 float* __restrict__ losses, const int* __restrict__ idx, float* lds, int tid) {
    using G = ConvMfmaGeom;
    const int lane = tid & 63, rw = tid >> 6, j = lane & 15, q = lane >> 4;
    constexpr int NWV = NT / 64;   // real waves; the sixteen chains' owners ("virtual waves" wv) are dealt over them
    float* act = lds + G::ACT_OFF;
    float* dz = lds + G::DZ_OFF;
    float* part = lds + G::PART_OFF;
    uint64_t* bb = reinterpret_cast<uint64_t*>(lds + G::BB_OFF);
    const float bm = 1.0f / (float)B;
    if (tid < 2 * G::CHUNK) {
        const int b = tid >> 1;
        unsigned long long v = 0ull;
        if (b < B) {
            const size_t si = idx ? (size_t)idx[b] : (size_t)b;
            v = (tid & 1) ? op_bb[si] : my_bb[si];
        }
        bb[tid] = v;
    }
    // heads: 16 lanes per sample, one output entry per lane (0..8 policy, 9..11 outcome); its target and the target's logarithm
    static_assert(NT == 16 * G::CHUNK, "the heads phase maps thread -> (sample tid >> 4, entry tid & 15)");
    const int hb = tid >> 4, jx = tid & 15;
    float tgt = 0.0f;
    if (hb < B && jx < 12) {
        const size_t si = idx ? (size_t)idx[hb] : (size_t)hb;
        tgt = jx < 9 ? tpi[si * 9 + jx] : tv[si * 3 + (jx - 9)];
    }
    const float ltgt = tgt > 0.0f ? det_logf(tgt) : 0.0f;
    __syncthreads();

    // ---- F: conv (k = tap 16 h + 4 q + e) + ReLU + head partials (k = channel 4 q + e)
    for (int wv = rw; wv < 16; wv += NWV) {
            bf16x4 ca[2];
    #pragma unroll
            for (int h = 0; h < 2; h++) {
                float t4[4];
    #pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int tap = 16 * h + 4 * q + e;
                    t4[e] = tap < 18 ? w[G::P_CW + j * 18 + tap] : 0.0f;
                }
                ca[h] = pack_bf16(t4[0], t4[1], t4[2], t4[3]);
            }
            const f32x4 cbv = *reinterpret_cast<const f32x4*>(w + G::P_CB + 4 * q);
            bf16x4 hwb[4];   // the head weights of this wave's cells, rounded once for both sample tiles
    #pragma unroll
            for (int c = 0; c < 4; c++) {
                const int p = wv + 16 * c;
                float h4[4];
    #pragma unroll
                for (int r = 0; r < 4; r++) h4[r] = (j < 12 && p < G::HW) ? w[G::P_HW + (size_t)j * G::FLAT + (4 * q + r) * G::HW + p] : 0.0f;
                hwb[c] = pack_bf16(h4[0], h4[1], h4[2], h4[3]);
            }
    #pragma unroll 1
            for (int t = 0; t < 2; t++) {
                const int sample = 16 * t + j;
                const uint64_t my = bb[2 * sample], op = bb[2 * sample + 1];
                uint64_t S[2][4];
    #pragma unroll
                for (int h = 0; h < 2; h++)
    #pragma unroll
                    for (int e = 0; e < 4; e++) S[h][e] = conv_tap_board(my, op, 16 * h + 4 * q + e);   // (taps >= 18: empty boards)
                f32x4 hacc = {0.0f, 0.0f, 0.0f, 0.0f};
    #pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int p = wv + 16 * c;
                    if (p < G::HW) {
                        const int row = p / 9, col = p - 9 * row, pos = row + 7 * col;
                        f32x4 acc = cbv;
    #pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const bf16x4 xb = pack_bf16((float)((uint32_t)(S[h][0] >> pos) & 1u), (float)((uint32_t)(S[h][1] >> pos) & 1u),
                                                        (float)((uint32_t)(S[h][2] >> pos) & 1u), (float)((uint32_t)(S[h][3] >> pos) & 1u));
                            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ca[h], xb, acc, 0, 0, 0);
                        }
                        float a4[4];
    #pragma unroll
                        for (int r = 0; r < 4; r++) {
                            a4[r] = acc[r] > 0.0f ? acc[r] : 0.0f;
                            act[sample * G::ASTR + (4 * q + r) * G::HW + p] = a4[r];
                        }
                        hacc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hwb[c], pack_bf16(a4[0], a4[1], a4[2], a4[3]), hacc, 0, 0, 0);
                    }
                }
                if (q < 3) {
    #pragma unroll
                    for (int r = 0; r < 4; r++) part[(wv * G::CHUNK + sample) * 12 + 4 * q + r] = hacc[r];
                }
            }
    }
    __syncthreads();
    // ---- H (f32, as conv_grad_step_mfma)
    {
        // this lane's output: bias + the sixteen partials in order; then log_softmax + kl_div and their gradient with the row's
        // values exchanged as DPP row broadcasts (train_epoch.cuh ep_row_gather / ep_row_sum: sums sequential in entry order)
        const bool pol = jx < 9;
        const bool live = jx < 12 && hb < B;
        float xo = 0.0f;
        if (jx < 12) {
            xo = w[G::P_HB + jx];
#pragma unroll
            for (int g = 0; g < 16; g++) xo += part[(g * G::CHUNK + hb) * 12 + jx];
        }
        xo = live ? xo : 0.0f;
        float xs[12];
        ep_row_gather(xo, xs);
        float mxp = xs[0], mxv = xs[9];
#pragma unroll
        for (int t = 1; t < 9; t++) mxp = xs[t] > mxp ? xs[t] : mxp;
#pragma unroll
        for (int t = 10; t < 12; t++) mxv = xs[t] > mxv ? xs[t] : mxv;
        const float mx = pol ? mxp : mxv;
        const float e = live ? det_expf(xo - mx) : 0.0f;
        const float se = ep_row_sum(e, pol), tsum = ep_row_sum(tgt, pol);
        const float lse = mx + det_logf(live ? se : 1.0f);
        const float logp = xo - lse;
        const float term = (live && tgt > 0.0f) ? tgt * (ltgt - logp) : 0.0f;
        const float kl = ep_row_sum(term, pol);
        const float sc = (pol ? hp.policy_weight : hp.value_weight) * bm;
        if (jx < 12) dz[hb * 12 + jx] = live ? sc * (det_expf(xo - lse) * tsum - tgt) : 0.0f;   // rows >= B: zeros
        if (jx == 0 || jx == 9) lds[G::KL_OFF + hb * 2 + (pol ? 0 : 1)] = hb < B ? kl : 0.0f;
    }
    __syncthreads();
    if (tid < 2) losses[tid] = bm * conv_ordered_sum32(lds + G::KL_OFF + tid, 2, B);
    // ---- G1: dWh (k = sample 16 h + 4 q + e)
    {
        bf16x4 dza[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float t4[4];
#pragma unroll
            for (int e = 0; e < 4; e++) t4[e] = j < 12 ? dz[(16 * h + 4 * q + e) * 12 + j] : 0.0f;
            dza[h] = pack_bf16(t4[0], t4[1], t4[2], t4[3]);
        }
#pragma unroll 1
        for (int ct = rw; ct < G::FLAT / 16; ct += NWV) {
            const int col = 16 * ct + j;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float* pa = act + (16 * h + 4 * q) * G::ASTR + col;
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dza[h], pack_bf16(pa[0], pa[G::ASTR], pa[2 * G::ASTR], pa[3 * G::ASTR]), acc, 0, 0, 0);
            }
            if (q < 3) {
#pragma unroll
                for (int r = 0; r < 4; r++) grads[G::P_HW + (size_t)(4 * q + r) * G::FLAT + col] = acc[r];
            }
        }
        if (tid >= NT - 12) {
            const int o = tid - (NT - 12);
            grads[G::P_HB + o] = conv_ordered_sum32(dz + o, 12, B);
        }
    }
    __syncthreads();
    // ---- G2: dY (k = output 4 q + e; outputs 12..15: zeros)
    {
        bf16x4 dzb[2];
#pragma unroll
        for (int bt = 0; bt < 2; bt++) {
            float t4[4];
#pragma unroll
            for (int e = 0; e < 4; e++) t4[e] = q < 3 ? dz[(16 * bt + j) * 12 + 4 * q + e] : 0.0f;
            dzb[bt] = pack_bf16(t4[0], t4[1], t4[2], t4[3]);
        }
#pragma unroll 1
        for (int ct = rw; ct < G::FLAT / 16; ct += NWV) {
            float t4[4];
#pragma unroll
            for (int e = 0; e < 4; e++) t4[e] = q < 3 ? w[G::P_HW + (size_t)(4 * q + e) * G::FLAT + 16 * ct + j] : 0.0f;
            const bf16x4 wa = pack_bf16(t4[0], t4[1], t4[2], t4[3]);
#pragma unroll
            for (int bt = 0; bt < 2; bt++) {
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa, dzb[bt], acc, 0, 0, 0);
                float* pa = act + (16 * bt + j) * G::ASTR + 16 * ct + 4 * q;
#pragma unroll
                for (int r = 0; r < 4; r++) pa[r] = pa[r] > 0.0f ? acc[r] : 0.0f;
            }
        }
    }
    __syncthreads();
    // ---- G3: dWc partials (k = cell 16 h + 4 q + e, h = 0..3; cell 63: padding)
    for (int wv = rw; wv < 16; wv += NWV) {
            f32x4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0;
    #pragma unroll 1
            for (int k = 0; k < 2; k++) {
                const int b = 2 * wv + k;
                const uint64_t my = bb[2 * b], op = bb[2 * b + 1];
                const uint64_t S0 = conv_tap_board(my, op, j);
                const uint64_t S1 = j < 2 ? conv_tap_board(my, op, 16 + j) : (j == 2 ? c4::FULL : 0ull);
                const float* ya = act + b * G::ASTR + j * G::HW;
    #pragma unroll
                for (int h = 0; h < 4; h++) {
                    float y4[4], x0[4], x1[4];
    #pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int cell = 16 * h + 4 * q + e;                       // per-lane (q), compile-time h, e
                        const int cc = cell < G::HW ? cell : 0;
                        const int row = cc / 9, col = cc - 9 * row, pos = row + 7 * col;
                        y4[e] = cell < G::HW ? ya[cc] : 0.0f;
                        x0[e] = cell < G::HW ? (float)((uint32_t)(S0 >> pos) & 1u) : 0.0f;
                        x1[e] = cell < G::HW ? (float)((uint32_t)(S1 >> pos) & 1u) : 0.0f;
                    }
                    const bf16x4 yb = pack_bf16(y4[0], y4[1], y4[2], y4[3]);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(yb, pack_bf16(x0[0], x0[1], x0[2], x0[3]), a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(yb, pack_bf16(x1[0], x1[1], x1[2], x1[3]), a1, 0, 0, 0);
                }
            }
    #pragma unroll
            for (int r = 0; r < 4; r++) {
                part[(wv * 16 + 4 * q + r) * 20 + j] = a0[r];
                if (j < 3) part[(wv * 16 + 4 * q + r) * 20 + 16 + j] = a1[r];
            }
    }
    __syncthreads();
    if (tid < ConvGeom::CONV_W + G::C) {
        const int c = tid < ConvGeom::CONV_W ? tid / 18 : tid - ConvGeom::CONV_W;
        const int t = tid < ConvGeom::CONV_W ? tid - 18 * c : 18;
        float v = part[c * 20 + t];
#pragma unroll
        for (int g = 1; g < 16; g++) v += part[(g * 16 + c) * 20 + t];
        grads[tid < ConvGeom::CONV_W ? G::P_CW + tid : G::P_CB + c] = v;
    }
}

// The workgroup: 8 waves (2 per SIMD, 256 VGPRs each: the step's prefetched operands do not fit 128), each owning two of the sixteen
// chains' "virtual waves".
constexpr int CONV_TRAIN_THREADS = 512;

// One launch = one minibatch (syn_train_step, the data-parallel gradient half): <<<1, 512>>>. BF16: the bf16 matrix-core variant.
template <bool BF16>
__global__ __launch_bounds__(CONV_TRAIN_THREADS) void train_conv_grad_kernel_mfma(const float* __restrict__ w, const unsigned long long* __restrict__ my_bb,
                                                                                  const unsigned long long* __restrict__ op_bb,
                                                                                  const float* __restrict__ tpi, const float* __restrict__ tv, int B,
                                                                                  DevTrainHyper hp, float* __restrict__ grads,
                                                                                  float* __restrict__ losses, const int* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (BF16) conv_grad_step_bf16<CONV_TRAIN_THREADS>(w, my_bb, op_bb, tpi, tv, B, hp, grads, losses, idx, lds, threadIdx.x);
    else conv_grad_step_mfma<CONV_TRAIN_THREADS>(w, my_bb, op_bb, tpi, tv, B, hp, grads, losses, idx, lds, threadIdx.x);
}

// One launch = every optimiser step of an epoch (syn_train_epoch): the step-ordered batches come from train_gather_kernel, the
// per-step Adam scalars (bias corrections, in double on the host like libtorch) from the caller. One workgroup: the steps are a
// dependent chain and a step is a few microseconds of one CU's matrix pipes, so nothing is gained by spreading it.
struct ConvEpochParams {
    float *w, *m, *v;
    const unsigned long long *my_bb, *op_bb;   // [n_steps * batch], step-ordered
    const float *tpi, *tv;
    const float *step_size, *inv_sqrt_bc2;     // [n_steps]
    float* losses;                             // [n_steps][2]
    float* grads;                              // the last step's gradients stay here (syn_trainer_get_state)
    int n_steps, batch;
    DevTrainHyper hp;
    unsigned long long* prof;                  // SYN_TRAIN_PROFILE: stamps of step 2 (f32 variant), else null
};
template <bool BF16>
__global__ __launch_bounds__(CONV_TRAIN_THREADS) void train_conv_epoch_kernel(ConvEpochParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = CONV_TRAIN_THREADS;
    const int tid = threadIdx.x;
    const int B = P.batch;
    for (int s = 0; s < P.n_steps; s++) {
        const size_t o = (size_t)s * B;
        if (BF16) conv_grad_step_bf16<NT>(P.w, P.my_bb + o, P.op_bb + o, P.tpi + o * 9, P.tv + o * 3, B, P.hp, P.grads, P.losses + 2 * s, nullptr, lds, tid);
        else conv_grad_step_mfma<NT>(P.w, P.my_bb + o, P.op_bb + o, P.tpi + o * 9, P.tv + o * 3, B, P.hp, P.grads, P.losses + 2 * s, nullptr, lds, tid,
                                     (P.prof && s == 2) ? P.prof : nullptr);
        // the gradients were written by other threads of THIS workgroup (one CU, one vector L1): workgroup scope is all the
        // visibility the step needs — an agent-scope release / acquire here writes back and invalidates the XCD's L2 four times a
        // step (measured: 41 us per step). Then Adam (adam_kernel's expression) over the 12,412 parameters.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const float step_size = P.step_size[s], inv_sqrt_bc2 = P.inv_sqrt_bc2[s];
        // 25 parameters per thread in two batches: all loads of a batch first, so that their latencies overlap
        constexpr int PER = 13;
#pragma unroll 1
        for (int b0 = 0; b0 < ConvGeom::NUM_PARAMS; b0 += PER * NT) {
            float g0[PER], wi[PER], mo[PER], vo[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int i = b0 + tid + NT * k;
                const bool ok = i < ConvGeom::NUM_PARAMS;
                g0[k] = ok ? P.grads[i] : 0.0f;
                wi[k] = ok ? P.w[i] : 0.0f;
                mo[k] = ok ? P.m[i] : 0.0f;
                vo[k] = ok ? P.v[i] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int i = b0 + tid + NT * k;
                if (i < ConvGeom::NUM_PARAMS) {
                    const float g = P.hp.weight_decay != 0.0f ? __builtin_fmaf(P.hp.weight_decay, wi[k], g0[k]) : g0[k];
                    const float mi = __builtin_fmaf(1.0f - P.hp.beta1, g, P.hp.beta1 * mo[k]);
                    const float vi = __builtin_fmaf((1.0f - P.hp.beta2) * g, g, P.hp.beta2 * vo[k]);
                    const float denom = sqrtf(vi) * inv_sqrt_bc2 + P.hp.eps;
                    P.m[i] = mi;
                    P.v[i] = vi;
                    P.w[i] = wi[k] - step_size * (mi / denom);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (P.prof && s == 2 && tid == 0) P.prof[15] = (unsigned long long)__builtin_readcyclecounter();   // end of the step incl. Adam
    }
}

// ---------------------------------------------------------------------------------------------- the epoch on four workgroups
// One CU cannot take a step below ~15 us (384 f32 MFMAs per wave and step, two waves per SIMD, + Adam on the same FP32 datapath);
// the persistent one-workgroup kernel above runs at 29 us. This kernel spreads a step over CONV_MW_WGS workgroups of one XCD, the
// way train_epoch.cuh does for Connect4Net, WITHOUT changing a single chain: every f32 result is produced by the same instruction
// sequence on the same operands as in conv_grad_step_mfma, only by another workgroup. Workgroup g owns
//   F, G1, G2   the board cells p = o, o + 16, o + 32, o + 48 of the chain owners o = 4 g .. 4 g + 3 (G1 / G2 take their 16-column
//               tiles as "the 16 channels of one cell": a column's chain never leaves its column, so any tiling gives the same bits),
//               and with them the head weights of those cells: their gradients, their Adam update and their only reader are here;
//   G3          the sample pairs 4 g .. 4 g + 3 (the two tap tiles of a pair on two waves);
// H (the 12 outputs, losses, dz, dbh) is computed by every workgroup, and so is the Adam update of the 316 parameters every workgroup
// reads (conv weights and biases, head biases): each keeps its own copy of them and of their moments in LDS for the whole epoch
// (identical bits everywhere; workgroup 0 writes them back at the end). What crosses workgroups goes through a 170 KB exchange
// buffer in L2 — the head partials (F -> H), dY (G2 -> G3), the conv-gradient partials (G3 -> G4 + Adam) — behind three barriers per
// step, the barrier of train_epoch.cuh (one XCD: stores acknowledged by L2 + `buffer_inv sc0`; otherwise device-scope release /
// acquire).
constexpr int CONV_MW_WGS = 4, CONV_MW_XCDS = 8;
struct ConvMwGeom {
    static constexpr int XPART = 0;                                  // [16 owners][32 samples][12]
    static constexpr int XDY = XPART + 16 * 32 * 12;                 // [32 samples][4 workgroups][16 channels][16 cell indices]
    static constexpr int XCONV = XDY + 32 * 1024;                    // [16 pairs][16 channels][20]
    static constexpr int FLOATS = XCONV + 16 * 16 * 20;
    // the shared parameters' private copies, in the (otherwise unused) head-partial region of LDS: index si = parameter index for the
    // conv weights / biases (0..303), 304 + o for head bias o; [w][m][v] of SMALL_STRIDE floats each
    static constexpr int SMALL = ConvGeom::CONV_W + ConvGeom::C + 12, SMALL_STRIDE = 320;
    // this workgroup's head weights and their gradients, entry e = output * 256 + channel * 16 + cell index (cell index ci -> owner
    // 4 g + (ci & 3), cell p = owner + 16 (ci >> 2)): F / G2 read their operands here, G1 leaves the gradients here, Adam keeps the
    // weights and moments of its six entries per thread in registers for the whole epoch
    // (LDS strides OWN_SO per output and OWN_SC per channel chosen so that F's, G1's and G2's fragment accesses are at most 3-way
    // bank conflicts; the natural 256 / 16 make every one of them a 16- to 64-way conflict)
    static constexpr int OWN = 12 * 256, OWN_SO = 279, OWN_SC = 17;
    static constexpr int LDS_FLOATS = ConvMfmaGeom::PART_OFF + 3 * SMALL_STRIDE + 2 * 12 * OWN_SO;
};
static_assert(ConvMwGeom::LDS_FLOATS * 4 + 64 <= 160 * 1024, "one workgroup: 160 KB of LDS");
struct ConvMwParams {
    ConvEpochParams e;
    float* xbuf;          // ConvMwGeom::FLOATS
    unsigned* sync;       // [0] arrivals, [1] abort flag, [2] start-up arrivals, [3] one-XCD mode (out), [8..] XCC ids
    int force_device_scope;
};

template <bool BF16>
__global__ __launch_bounds__(CONV_TRAIN_THREADS) void train_conv_epoch_kernel_mw(ConvMwParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using G = ConvMfmaGeom;
    using X = ConvMwGeom;
    constexpr int NT = CONV_TRAIN_THREADS, NWG = CONV_MW_WGS;
    if (blockIdx.x % CONV_MW_XCDS != 0) return;
    const int g = blockIdx.x / CONV_MW_XCDS;
    const int tid0 = threadIdx.x;
    const int B = P.e.batch;
    const float bm = 1.0f / (float)B;
    __shared__ unsigned mw_abort, mw_fast;
    float* act = lds + G::ACT_OFF;
    float* dz = lds + G::DZ_OFF;
    uint64_t* bb = reinterpret_cast<uint64_t*>(lds + G::BB_OFF);
    float* sw = lds + G::PART_OFF;               // shared parameters: weights, moments
    float* sm = sw + X::SMALL_STRIDE;
    float* sv = sm + X::SMALL_STRIDE;
    float* whead = sv + X::SMALL_STRIDE;         // this workgroup's head weights [12][16][16]
    float* ghead = whead + 12 * X::OWN_SO;       // ... and their gradients
    float* xpart = P.xbuf + X::XPART;
    float* xdy = P.xbuf + X::XDY;
    float* xconv = P.xbuf + X::XCONV;

    // ---- where did the workgroups land? (train_epoch.cuh)
    if (tid0 == 0) {
        mw_abort = 0u;
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) + 1u;  // HW_REG_XCC_ID[3:0]
        __hip_atomic_store(P.sync + 8 + g, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(P.sync + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(P.sync + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NWG) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) {
                __hip_atomic_store(P.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mw_abort = 1u;
                break;
            }
        }
        bool same = true;
        for (int i = 0; i < NWG; i++) same = same && __hip_atomic_load(P.sync + 8 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xcc;
        mw_fast = (same && !P.force_device_scope) ? 1u : 0u;
        if (g == 0) P.sync[3] = mw_fast;
    }
    // the shared parameters and their moments: this workgroup's copies
    if (tid0 < X::SMALL) {
        const int i = tid0 < ConvGeom::CONV_W + G::C ? tid0 : G::P_HB + (tid0 - (ConvGeom::CONV_W + G::C));
        sw[tid0] = P.e.w[i];
        sm[tid0] = P.e.m[i];
        sv[tid0] = P.e.v[i];
    }
    // this thread's six head-weight entries (e = tid + NT k) with their moments: registers for the whole epoch, weights also in LDS
    constexpr int OWN_PER = X::OWN / NT;
    int own_pi[OWN_PER];
    float own_w[OWN_PER], own_m[OWN_PER], own_v[OWN_PER];
#pragma unroll
    for (int k = 0; k < OWN_PER; k++) {
        const int e = tid0 + NT * k, ci = e & 15;
        const int p = 4 * g + (ci & 3) + 16 * (ci >> 2);
        own_pi[k] = p < G::HW ? G::P_HW + (e >> 8) * G::FLAT + ((e & 255) >> 4) * G::HW + p : -1;
        own_w[k] = own_pi[k] >= 0 ? P.e.w[own_pi[k]] : 0.0f;
        own_m[k] = own_pi[k] >= 0 ? P.e.m[own_pi[k]] : 0.0f;
        own_v[k] = own_pi[k] >= 0 ? P.e.v[own_pi[k]] : 0.0f;
        whead[(e >> 8) * X::OWN_SO + ((e & 255) >> 4) * X::OWN_SC + ci] = own_w[k];
    }
    __syncthreads();
    if (mw_abort) return;
    const bool one_xcd = mw_fast != 0u;
    unsigned barriers = 0;
    // split-phase barrier: xarrive() releases what this workgroup stored and arrives; xwait() waits for all NWG arrivals and acquires
    // (false when a workgroup never arrived). Work that needs nothing from the other workgroups goes in between.
    auto xarrive = [&]() {
        if (one_xcd) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        barriers++;
        if (tid0 == 0) __hip_atomic_fetch_add(P.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto xwait = [&]() -> bool {
        if (tid0 == 0) {
            const unsigned want = (unsigned)NWG * barriers;
            unsigned spins = 0;
            for (;;) {
                const unsigned have = one_xcd ? __hip_atomic_load(P.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                              : __hip_atomic_load(P.sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (have >= want) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24) || __hip_atomic_load(P.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(P.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mw_abort = 1u;
                    break;
                }
            }
        }
        __syncthreads();
        if (mw_abort) return false;
        if (one_xcd) asm volatile("buffer_inv sc0" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return true;
    };
    auto xbarrier = [&]() -> bool {
        xarrive();
        return xwait();
    };

    // the batch of step 0 (later batches are requested one step ahead, before the step's last barrier)
    unsigned long long board_next = 0ull;   // thread tid < 64: board word tid (sample tid >> 1, mine / theirs)
    float tgt_next = 0.0f;                  // this thread's head entry (sample tid >> 4, entry tid & 15)
    auto request_batch = [&](int s, int tid) {
        const size_t so = (size_t)s * B;
        board_next = 0ull;
        if (tid < 2 * G::CHUNK && (tid >> 1) < B) board_next = (tid & 1) ? P.e.op_bb[so + (tid >> 1)] : P.e.my_bb[so + (tid >> 1)];
        const int hb = tid >> 4, jx = tid & 15;
        tgt_next = 0.0f;
        if (hb < B && jx < 12) tgt_next = jx < 9 ? P.e.tpi[(so + hb) * 9 + jx] : P.e.tv[(so + hb) * 3 + (jx - 9)];
    };
    request_batch(0, tid0);

    int pk = 0;
#define MW_STAMP() do { if (P.e.prof && tid0 == 0 && s == 2) P.e.prof[g * 16 + pk++] = (unsigned long long)__builtin_readcyclecounter(); } while (0)
    for (int s = 0; s < P.e.n_steps; s++) {
        // per-iteration opaque copy of the thread index: without it the compiler hoists every step-invariant address and operand
        // out of the step loop and keeps hundreds of values in scratch for the whole epoch (train_epoch.cuh, same remedy)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, rw = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, q = lane >> 4;
        MW_STAMP();
        // ---- stage: boards of all samples, this thread's head entry and its target
        if (tid < 2 * G::CHUNK) bb[tid] = board_next;
        const int hb = tid >> 4, jx = tid & 15;
        const float tgt = tgt_next;
        const float ltgt = tgt > 0.0f ? det_logf(tgt) : 0.0f;
        // this workgroup's two cells per wave (G1 / G2): cell index ci = wave + 8 k -> owner 4 g + (ci & 3), p = owner + 16 (ci >> 2);
        // G2's head-weight operands are requested here, a phase and a barrier ahead of their use
        float wa[2][3];   // f32: A[channel j of cell p][output 4 s + q]
        bf16x4 wab[2];    // bf16: A[channel j of cell p][output 4 q + e] (outputs 12..15: zeros)
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int ci = rw + 8 * kk;
            const int p = 4 * g + (ci & 3) + 16 * (ci >> 2);
            if (BF16) {
                float t4[4];
#pragma unroll
                for (int e = 0; e < 4; e++) t4[e] = (q < 3 && p < G::HW) ? whead[(4 * q + e) * X::OWN_SO + j * X::OWN_SC + ci] : 0.0f;
                wab[kk] = pack_bf16(t4[0], t4[1], t4[2], t4[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 3; k++) wa[kk][k] = p < G::HW ? whead[(4 * k + q) * X::OWN_SO + j * X::OWN_SC + ci] : 0.0f;
            }
        }
        __syncthreads();
        MW_STAMP();

        // ---- F: chain owner wv = 4 g + (wave >> 1), sample tile t = wave & 1 (conv_grad_step_mfma's / _bf16's F for that (owner, tile))
        if (!BF16) {
            const int wv = 4 * g + (rw >> 1), t = rw & 1;
            float ca[5];
#pragma unroll
            for (int k = 0; k < 5; k++) ca[k] = 4 * k + q < 18 ? sw[G::P_CW + j * 18 + 4 * k + q] : 0.0f;
            const f32x4 cbv = *reinterpret_cast<const f32x4*>(sw + G::P_CB + 4 * q);
            float hwv[4][4];
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int p = wv + 16 * c;
                    hwv[c][r] = (j < 12 && p < G::HW) ? whead[j * X::OWN_SO + (4 * q + r) * X::OWN_SC + (rw >> 1) + 4 * c] : 0.0f;
                }
            const int sample = 16 * t + j;
            const uint64_t my = bb[2 * sample], op = bb[2 * sample + 1];
            uint64_t S[5];
#pragma unroll
            for (int k = 0; k < 5; k++) S[k] = conv_tap_board(my, op, 4 * k + q);
            f32x4 hacc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int p = wv + 16 * c;
                if (p < G::HW) {
                    const int row = p / 9, col = p - 9 * row, pos = row + 7 * col;
                    f32x4 acc = cbv;
#pragma unroll
                    for (int k = 0; k < 5; k++)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[k], (float)((uint32_t)(S[k] >> pos) & 1u), acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float a = acc[r] > 0.0f ? acc[r] : 0.0f;
                        act[sample * G::ASTR + (4 * q + r) * G::HW + p] = a;
                        hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(hwv[c][r], a, hacc, 0, 0, 0);
                    }
                }
            }
            if (q < 3) {
#pragma unroll
                for (int r = 0; r < 4; r++) xpart[(wv * G::CHUNK + sample) * 12 + 4 * q + r] = hacc[r];
            }
        } else {
            const int wv = 4 * g + (rw >> 1), t = rw & 1;
            bf16x4 ca[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float t4[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int tap = 16 * h + 4 * q + e;
                    t4[e] = tap < 18 ? sw[G::P_CW + j * 18 + tap] : 0.0f;
                }
                ca[h] = pack_bf16(t4[0], t4[1], t4[2], t4[3]);
            }
            const f32x4 cbv = *reinterpret_cast<const f32x4*>(sw + G::P_CB + 4 * q);
            bf16x4 hwb[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int p = wv + 16 * c;
                float h4[4];
#pragma unroll
                for (int r = 0; r < 4; r++) h4[r] = (j < 12 && p < G::HW) ? whead[j * X::OWN_SO + (4 * q + r) * X::OWN_SC + (rw >> 1) + 4 * c] : 0.0f;
                hwb[c] = pack_bf16(h4[0], h4[1], h4[2], h4[3]);
            }
            const int sample = 16 * t + j;
            const uint64_t my = bb[2 * sample], op = bb[2 * sample + 1];
            uint64_t S[2][4];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int e = 0; e < 4; e++) S[h][e] = conv_tap_board(my, op, 16 * h + 4 * q + e);   // (taps >= 18: empty boards)
            f32x4 hacc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int p = wv + 16 * c;
                if (p < G::HW) {
                    const int row = p / 9, col = p - 9 * row, pos = row + 7 * col;
                    f32x4 acc = cbv;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const bf16x4 xb = pack_bf16((float)((uint32_t)(S[h][0] >> pos) & 1u), (float)((uint32_t)(S[h][1] >> pos) & 1u),
                                                    (float)((uint32_t)(S[h][2] >> pos) & 1u), (float)((uint32_t)(S[h][3] >> pos) & 1u));
                        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ca[h], xb, acc, 0, 0, 0);
                    }
                    float a4[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        a4[r] = acc[r] > 0.0f ? acc[r] : 0.0f;
                        act[sample * G::ASTR + (4 * q + r) * G::HW + p] = a4[r];
                    }
                    hacc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hwb[c], pack_bf16(a4[0], a4[1], a4[2], a4[3]), hacc, 0, 0, 0);
                }
            }
            if (q < 3) {
#pragma unroll
                for (int r = 0; r < 4; r++) xpart[(wv * G::CHUNK + sample) * 12 + 4 * q + r] = hacc[r];
            }
        }
        MW_STAMP();
        if (!xbarrier()) return;
        MW_STAMP();

        // ---- H (every workgroup): bias + the sixteen partials in order, log_softmax + kl_div, dz
        {
            const bool pol = jx < 9;
            const bool live = jx < 12 && hb < B;
            float xo = 0.0f;
            if (jx < 12) {
                float pv[16];
#pragma unroll
                for (int o = 0; o < 16; o++) pv[o] = xpart[(o * G::CHUNK + hb) * 12 + jx];
                xo = sw[ConvGeom::CONV_W + G::C + jx];
#pragma unroll
                for (int o = 0; o < 16; o++) xo += pv[o];
            }
            // the next step's batch: a cold read from memory, and loads return in order — issued here, behind the partials' loads and
            // in front of the longest stretch without global loads (the rest of H, G1, G2), it costs nothing; issued in front of a
            // barrier's arrival it added its latency to the barrier (measured: 10k instead of 3.4k cycles)
            if (s + 1 < P.e.n_steps) request_batch(s + 1, tid);
            xo = live ? xo : 0.0f;
            float xs[12];
            ep_row_gather(xo, xs);
            float mxp = xs[0], mxv = xs[9];
#pragma unroll
            for (int t = 1; t < 9; t++) mxp = xs[t] > mxp ? xs[t] : mxp;
#pragma unroll
            for (int t = 10; t < 12; t++) mxv = xs[t] > mxv ? xs[t] : mxv;
            const float mx = pol ? mxp : mxv;
            const float e = live ? det_expf(xo - mx) : 0.0f;
            const float se = ep_row_sum(e, pol), tsum = ep_row_sum(tgt, pol);
            const float lse = mx + det_logf(live ? se : 1.0f);
            const float logp = xo - lse;
            const float term = (live && tgt > 0.0f) ? tgt * (ltgt - logp) : 0.0f;
            const float kl = ep_row_sum(term, pol);
            const float sc = (pol ? P.e.hp.policy_weight : P.e.hp.value_weight) * bm;
            if (jx < 12) dz[hb * 12 + jx] = live ? sc * (det_expf(xo - lse) * tsum - tgt) : 0.0f;
            if (jx == 0 || jx == 9) lds[G::KL_OFF + hb * 2 + (pol ? 0 : 1)] = hb < B ? kl : 0.0f;
        }
        __syncthreads();
        MW_STAMP();
        // (losses: by the wave whose second cell slot is empty — cell 63 does not exist)
        if (g == NWG - 1 && tid >= NT - 64 && tid < NT - 62) P.e.losses[2 * s + (tid - (NT - 64))] = bm * conv_ordered_sum32(lds + G::KL_OFF + (tid - (NT - 64)), 2, B);

        // ---- G1 + G2 on this workgroup's cells
        float dbh = 0.0f;   // threads NT - 12 ..: dbh[o], plain sum over the samples (every workgroup: it feeds its own copy's Adam)
        {
            float dza[8];      // f32: A[output j][sample 4 s + q]
            float dzb[2][3];   // f32: B[output 4 s + q][sample 16 bt + j]
            bf16x4 dzab[2];    // bf16: A[output j][sample 16 h + 4 q + e]
            bf16x4 dzbb[2];    // bf16: B[output 4 q + e][sample 16 bt + j] (outputs 12..15: zeros)
            if (BF16) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    float t4[4], u4[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        t4[e] = j < 12 ? dz[(16 * h + 4 * q + e) * 12 + j] : 0.0f;
                        u4[e] = q < 3 ? dz[(16 * h + j) * 12 + 4 * q + e] : 0.0f;
                    }
                    dzab[h] = pack_bf16(t4[0], t4[1], t4[2], t4[3]);
                    dzbb[h] = pack_bf16(u4[0], u4[1], u4[2], u4[3]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) dza[k] = j < 12 ? dz[(4 * k + q) * 12 + j] : 0.0f;
#pragma unroll
                for (int bt = 0; bt < 2; bt++)
#pragma unroll
                    for (int k = 0; k < 3; k++) dzb[bt][k] = dz[(16 * bt + j) * 12 + 4 * k + q];
            }
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                const int ci = rw + 8 * kk;
                const int p = 4 * g + (ci & 3) + 16 * (ci >> 2);
                if (p < G::HW) {   // (wave-uniform)
                    // G1: dWh[output][channel j of cell p] = chain over the samples
                    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (BF16) {
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const float* pa = act + (16 * h + 4 * q) * G::ASTR + j * G::HW + p;
                            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dzab[h], pack_bf16(pa[0], pa[G::ASTR], pa[2 * G::ASTR], pa[3 * G::ASTR]), acc, 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dza[k], act[(4 * k + q) * G::ASTR + j * G::HW + p], acc, 0, 0, 0);
                    }
                    if (q < 3) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            ghead[(4 * q + r) * X::OWN_SO + j * X::OWN_SC + ci] = acc[r];
                            // (the caller reads the last step's gradients: syn_trainer_get_state)
                            if (s + 1 == P.e.n_steps) P.e.grads[G::P_HW + (size_t)(4 * q + r) * G::FLAT + j * G::HW + p] = acc[r];
                        }
                    }
                }
            }
            if (tid >= NT - 12) dbh = conv_ordered_sum32(dz + (tid - (NT - 12)), 12, B);
            MW_STAMP();   // (G2 overwrites this wave's own cells' columns of act, which only this wave's G1 read)
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                const int ci = rw + 8 * kk;
                const int p = 4 * g + (ci & 3) + 16 * (ci >> 2);
                if (p < G::HW) {
#pragma unroll
                    for (int bt = 0; bt < 2; bt++) {
                        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (BF16) {
                            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wab[kk], dzbb[bt], acc, 0, 0, 0);
                        } else {
#pragma unroll
                            for (int k = 0; k < 3; k++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[kk][k], dzb[bt][k], acc, 0, 0, 0);
                        }
                        // D rows: channels 4 q + r of cell p; column: sample 16 bt + j
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            float* pa = act + (16 * bt + j) * G::ASTR + (4 * q + r) * G::HW + p;
                            *pa = *pa > 0.0f ? acc[r] : 0.0f;
                        }
                    }
                }
            }
            // dY of this workgroup's cells -> exchange buffer, [sample][workgroup][channel][cell index]: a thread moves the four
            // cells 4 g + 16 c .. + 3 of one (sample, channel) as one 16-byte store, a wave 1 KB of consecutive addresses (the
            // per-element stores of the fragments' owners are 64 different lines per instruction: measured 2x the whole phase)
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int it = tid + NT * k, b = it >> 6, ch = (it >> 2) & 15, c = it & 3;
                const int p0 = 4 * g + 16 * c;
                const float* pa = act + b * G::ASTR + ch * G::HW + p0;
                f32x4 v4;
                v4[0] = pa[0]; v4[1] = pa[1]; v4[2] = pa[2];
                v4[3] = p0 + 3 < G::HW ? pa[3] : 0.0f;
                *reinterpret_cast<f32x4*>(xdy + b * 1024 + g * 256 + ch * 16 + 4 * c) = v4;
            }
        }
        MW_STAMP();
        xarrive();   // (dY published; the barrier's latency is covered by the Adam update of this workgroup's head weights)
        const float step_size = P.e.step_size[s], inv_sqrt_bc2 = P.e.inv_sqrt_bc2[s];
        // ---- Adam (adam_kernel's expression) of this workgroup's head weights: gradients from G1 (LDS), weights and moments in registers
        {
#pragma unroll
            for (int k = 0; k < OWN_PER; k++) {
                if (own_pi[k] >= 0) {
                    const int e = tid + NT * k, la = (e >> 8) * X::OWN_SO + ((e & 255) >> 4) * X::OWN_SC + (e & 15);
                    const float g0 = ghead[la];
                    const float gr = P.e.hp.weight_decay != 0.0f ? __builtin_fmaf(P.e.hp.weight_decay, own_w[k], g0) : g0;
                    const float mi = __builtin_fmaf(1.0f - P.e.hp.beta1, gr, P.e.hp.beta1 * own_m[k]);
                    const float vi = __builtin_fmaf((1.0f - P.e.hp.beta2) * gr, gr, P.e.hp.beta2 * own_v[k]);
                    const float denom = sqrtf(vi) * inv_sqrt_bc2 + P.e.hp.eps;
                    own_m[k] = mi;
                    own_v[k] = vi;
                    own_w[k] = own_w[k] - step_size * (mi / denom);
                    whead[la] = own_w[k];   // (G2 took its operands before the arrival above; F of the next step reads these)
                }
            }
        }
        MW_STAMP();
        if (!xwait()) return;
        MW_STAMP();

        // ---- dY of this workgroup's eight samples (pairs 4 g .. 4 g + 3), all cells, from the exchange buffer into their rows of act:
        //      16-byte loads, 1 KB of consecutive addresses per wave instruction, all in flight together (G3 reading its operands
        //      straight from the buffer — 4 bytes per lane, 16 of every 64 — took 4k cycles longer and another 5k in the barrier behind it)
        {
            f32x4 v4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int it = tid + NT * k;
                v4[k] = *reinterpret_cast<const f32x4*>(xdy + (8 * g + (it >> 8)) * 1024 + (it & 255) * 4);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int it = tid + NT * k, rest = it & 255;
                const int p0 = 4 * (rest >> 6) + 16 * (rest & 3);   // block (= writer) rest >> 6, channel (rest >> 2) & 15, cell group rest & 3
                float* pa = act + (8 * g + (it >> 8)) * G::ASTR + ((rest >> 2) & 15) * G::HW + p0;
                pa[0] = v4[k][0]; pa[1] = v4[k][1]; pa[2] = v4[k][2];
                if (p0 + 3 < G::HW) pa[3] = v4[k][3];
            }
        }
        __syncthreads();
        MW_STAMP();
        // ---- G3: sample pair wv = 4 g + (wave >> 1); wave & 1 = tap tile (0: taps 0..15, 1: taps 16, 17 and the bias "tap")
        {
            const int wv = 4 * g + (rw >> 1), half = rw & 1;
            f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
            if (!BF16) {
                const FeatureTable FT = make_feature_table(q);
                float y[2][16];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const float* ya = act + (2 * wv + k) * G::ASTR + j * G::HW + q;   // A[channel j][cell 4 i + q]
#pragma unroll
                    for (int i = 0; i < 16; i++) y[k][i] = 4 * i + q < G::HW ? ya[4 * i] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int b = 2 * wv + k;
                    const uint64_t my = bb[2 * b], op = bb[2 * b + 1];
                    const uint64_t Sx = half == 0 ? conv_tap_board(my, op, j)
                                                  : (j < 2 ? conv_tap_board(my, op, 16 + j) : (j == 2 ? c4::FULL : 0ull));
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint32_t pos = (FT.t[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                        a = __builtin_amdgcn_mfma_f32_16x16x4f32(y[k][i], (float)((uint32_t)(Sx >> pos) & 1u), a, 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int b = 2 * wv + k;
                    const uint64_t my = bb[2 * b], op = bb[2 * b + 1];
                    const uint64_t Sx = half == 0 ? conv_tap_board(my, op, j)
                                                  : (j < 2 ? conv_tap_board(my, op, 16 + j) : (j == 2 ? c4::FULL : 0ull));
                    const float* ya = act + b * G::ASTR + j * G::HW;
#pragma unroll
                    for (int h = 0; h < 4; h++) {   // k = cell 16 h + 4 q + e; cell 63: padding
                        float y4[4], x4[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int cell = 16 * h + 4 * q + e;
                            const int cc = cell < G::HW ? cell : 0;
                            const int row = cc / 9, col = cc - 9 * row, pos = row + 7 * col;
                            y4[e] = cell < G::HW ? ya[cc] : 0.0f;
                            x4[e] = cell < G::HW ? (float)((uint32_t)(Sx >> pos) & 1u) : 0.0f;
                        }
                        a = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pack_bf16(y4[0], y4[1], y4[2], y4[3]), pack_bf16(x4[0], x4[1], x4[2], x4[3]), a, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (half == 0) xconv[(wv * 16 + 4 * q + r) * 20 + j] = a[r];
                else if (j < 3) xconv[(wv * 16 + 4 * q + r) * 20 + 16 + j] = a[r];
            }
        }
        MW_STAMP();
        if (!xbarrier()) return;
        MW_STAMP();
        // ---- G4 + Adam of the shared parameters, on this workgroup's own copy: conv weights / biases (gradient = the sixteen partials
        //      added in order), head biases (dbh from above)
        {
            const bool is_conv = tid < ConvGeom::CONV_W + G::C, is_hb = tid >= NT - 12;
            if (is_conv || is_hb) {
                float gsum;
                int si, gi;
                if (is_conv) {
                    const int c = tid < ConvGeom::CONV_W ? tid / 18 : tid - ConvGeom::CONV_W;
                    const int t = tid < ConvGeom::CONV_W ? tid - 18 * c : 18;
                    float pv[16];
#pragma unroll
                    for (int o = 0; o < 16; o++) pv[o] = xconv[(o * 16 + c) * 20 + t];
                    gsum = pv[0];
#pragma unroll
                    for (int o = 1; o < 16; o++) gsum += pv[o];
                    si = tid;
                    gi = tid;
                } else {
                    gsum = dbh;
                    si = ConvGeom::CONV_W + G::C + (tid - (NT - 12));
                    gi = G::P_HB + (tid - (NT - 12));
                }
                if (g == 0) P.e.grads[gi] = gsum;
                const float wo = sw[si];
                const float gr = P.e.hp.weight_decay != 0.0f ? __builtin_fmaf(P.e.hp.weight_decay, wo, gsum) : gsum;
                const float mi = __builtin_fmaf(1.0f - P.e.hp.beta1, gr, P.e.hp.beta1 * sm[si]);
                const float vi = __builtin_fmaf((1.0f - P.e.hp.beta2) * gr, gr, P.e.hp.beta2 * sv[si]);
                const float denom = sqrtf(vi) * inv_sqrt_bc2 + P.e.hp.eps;
                sm[si] = mi;
                sv[si] = vi;
                sw[si] = wo - step_size * (mi / denom);
            }
        }
        __syncthreads();   // the shared copies (and, since the last barrier, the head weights) in LDS are the next step's
        MW_STAMP();
    }
#undef MW_STAMP
#pragma unroll
    for (int k = 0; k < OWN_PER; k++) {
        if (own_pi[k] >= 0) {
            P.e.w[own_pi[k]] = own_w[k];
            P.e.m[own_pi[k]] = own_m[k];
            P.e.v[own_pi[k]] = own_v[k];
        }
    }
    if (g == 0 && tid0 < X::SMALL) {
        const int i = tid0 < ConvGeom::CONV_W + G::C ? tid0 : G::P_HB + (tid0 - (ConvGeom::CONV_W + G::C));
        P.e.w[i] = sw[tid0];
        P.e.m[i] = sm[tid0];
        P.e.v[i] = sv[tid0];
    }
}

}  // namespace syn

// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of the reference's layer semantics and of the study-connect4 policy/value network:
//   slimnn/src/linear.rs:17-25        Linear::forward  (out = bias; for i { for o { out[o] += x[i]*W[o][i] } })
//   slimnn/src/conv.rs:45-85          Conv2d::forward  (NCHW cross-correlation, row/col zero padding, stride)
//   slimnn/src/activations.rs:31-63   ReLU, Tanh, Softmax (Softmax is NOT max-subtracted there)
//   study-connect4/src/policies.rs:14-59  Connect4Net: 63->128->96->64->48->12, ReLU between, logits = out[0..9],
//                                         value = softmax(out[9..12]) (libtorch softmax: max-subtracted)
// The reference's Connect4Net runs on libtorch (tch 0.4.1, not in /root/reference): y = x W^T + b with an
// unspecified summation order. Two accumulation modes are provided:
//   ACC_SLIMNN : separate multiply and add, input index ascending — exactly slimnn's loop (canonical restatement).
//   ACC_FMA    : same ascending order with a fused multiply-add per term — the order/rounding the HIP engine's
//                f32 MFMA path produces bit for bit. Used so that MCTS visit counts can be compared exactly.
//   ACC_F16X2  : (Connect4Net only) the engine's SYN_NET_ARITH_F16X2 arithmetic bit for bit — every operand a pair of f16 numbers,
//                products on v_mfma_f32_16x16x32_f16, whose accumulation nn_f16x2.hpp restates. A third definition of the same
//                function (not the reference's arithmetic); as close to an f64 evaluation as the f32 modes are.
// The modes agree to ~1e-6; north_star's 1e-5 tolerance is asserted between the engine and ACC_SLIMNN.
// Parity: Linear/Conv2d/ReLU pinned by slimnn's KATs (linear.rs:105-112, conv.rs:92-602, activations.rs:70-75);
// Connect4Net has no test in the reference -> PARITY UNPINNED there (torch-generated goldens in tests/golden/).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

#include <memory>

#include "connect4.hpp"
#include "det_math.hpp"
#include "nn_f16x2.hpp"

namespace oracle {

enum AccMode : int { ACC_SLIMNN = 0, ACC_FMA = 1, ACC_F16X2 = 2 };

// W is [O][I] row-major (slimnn `weight: [[f32; I]; O]`, same as torch nn.Linear.weight).
inline void linear_forward(int I, int O, const float* W, const float* b, const float* x, float* out, int mode) {
    for (int o = 0; o < O; o++) out[o] = b[o];
    if (mode == ACC_FMA) {
        for (int i = 0; i < I; i++)
            for (int o = 0; o < O; o++) out[o] = std::fmaf(x[i], W[(size_t)o * I + i], out[o]);
    } else {
        for (int i = 0; i < I; i++)
            for (int o = 0; o < O; o++) out[o] += x[i] * W[(size_t)o * I + i];
    }
}

inline void relu_inplace(float* x, int n) {
    for (int i = 0; i < n; i++) x[i] = x[i] > 0.0f ? x[i] : 0.0f;  // x.max(0.0): NaN -> 0.0 as Rust's f32::max
}
inline void tanh_inplace(float* x, int n) {
    for (int i = 0; i < n; i++) x[i] = det_tanhf(x[i]);  // activations.rs:39-44 (x.tanh(): libm, unpinned -> det_tanhf)
}
// activations.rs:46-63 (exp via det_expf, see det_math.hpp)
inline void softmax_slimnn(const float* x, float* y, int n) {
    float total = 0.0f;
    for (int i = 0; i < n; i++) {
        y[i] = det_expf(x[i]);
        total += y[i];
    }
    for (int i = 0; i < n; i++) y[i] /= total;
}
// libtorch softmax semantics used by Connect4Net::eval (policies.rs:54-57): subtract the max first.
inline void softmax_stable(const float* x, float* y, int n) {
    float m = x[0];
    for (int i = 1; i < n; i++) m = x[i] > m ? x[i] : m;
    float total = 0.0f;
    for (int i = 0; i < n; i++) {
        y[i] = det_expf(x[i] - m);
        total += y[i];
    }
    for (int i = 0; i < n; i++) y[i] /= total;
}

// conv.rs:45-85. x is [CIN][H_IN][W_IN], W is [COUT][CIN][K][K], y is [COUT][H_OUT][W_OUT].
// Returns false if the caller's output dims violate the reference's asserts (conv.rs:50-51).
inline bool conv2d_forward(int CIN, int COUT, int K, int ROW_PAD, int COL_PAD, int STRIDE, int H_IN, int W_IN,
                           int H_OUT, int W_OUT, const float* W, const float* b, const float* x, float* y, int mode) {
    if (W_OUT != ((W_IN + 2 * COL_PAD - K) / STRIDE) + 1) return false;
    if (H_OUT != ((H_IN + 2 * ROW_PAD - K) / STRIDE) + 1) return false;
    for (int co = 0; co < COUT; co++)
        for (int r = 0; r < H_OUT; r++)
            for (int c = 0; c < W_OUT; c++) y[((size_t)co * H_OUT + r) * W_OUT + c] = b[co];
    for (int co = 0; co < COUT; co++)
        for (int ci = 0; ci < CIN; ci++)
            for (int r = 0; r < H_OUT; r++)
                for (int c = 0; c < W_OUT; c++)
                    for (int k1 = 0; k1 < K; k1++) {
                        int in_row = r * STRIDE + k1;
                        if (ROW_PAD <= in_row && in_row < H_IN + ROW_PAD) {
                            for (int k2 = 0; k2 < K; k2++) {
                                int in_col = c * STRIDE + k2;
                                if (COL_PAD <= in_col && in_col < W_IN + COL_PAD) {
                                    float w = W[(((size_t)co * CIN + ci) * K + k1) * K + k2];
                                    float v = x[((size_t)ci * H_IN + (in_row - ROW_PAD)) * W_IN + (in_col - COL_PAD)];
                                    float& acc = y[((size_t)co * H_OUT + r) * W_OUT + c];
                                    if (mode == ACC_FMA) acc = std::fmaf(w, v, acc);
                                    else acc += w * v;
                                }
                            }
                        }
                    }
    return true;
}

// Connect4Net parameter blob: l_1.weight[128x63], l_1.bias[128], l_2.weight[96x128], l_2.bias[96],
// l_3.weight[64x96], l_3.bias[64], l_4.weight[48x64], l_4.bias[48], l_5.weight[12x48], l_5.bias[12]
// (VarStore names policies.rs:20-24) = 30,492 f32.
struct Connect4Net {
    static constexpr int NL = 5;
    static constexpr int DIMS[NL + 1] = {63, 128, 96, 64, 48, 12};
    static constexpr size_t NUM_PARAMS = 63 * 128 + 128 + 128 * 96 + 96 + 96 * 64 + 64 + 64 * 48 + 48 + 48 * 12 + 12;
    const float* blob = nullptr;
    int mode = ACC_SLIMNN;

    const float* weight(int l) const {
        size_t off = 0;
        for (int i = 0; i < l; i++) off += (size_t)DIMS[i] * DIMS[i + 1] + DIMS[i + 1];
        return blob + off;
    }
    const float* bias(int l) const { return weight(l) + (size_t)DIMS[l] * DIMS[l + 1]; }

    // ACC_F16X2: the plan and the split operands, built from the blob on first use (one Connect4Net object per thread)
    mutable std::shared_ptr<F16x2Net> f16;
    mutable const float* f16_blob = nullptr;
    const F16x2Net& f16_net() const {
        if (!f16 || f16_blob != blob) { f16 = std::make_shared<F16x2Net>(blob); f16_blob = blob; }
        return *f16;
    }

    // policies.rs:28-44 on a single state; out12 = the 12 raw outputs
    void forward(const float* x63, float* out12) const {
        if (mode == ACC_F16X2) {   // the f16x2 arithmetic starts from the bitboards: +1 = mine, -1 = theirs (connect4.rs:235-258)
            uint64_t my = 0, op = 0;
            for (int f = 0; f < 63; f++) {
                const uint64_t bit = 1ull << ((f / 9) + 7 * (f % 9));
                if (x63[f] == 1.0f) my |= bit;
                if (x63[f] == -1.0f) op |= bit;
            }
            f16_net().forward(my, op, out12);
            return;
        }
        float a[128], c[128];
        const float* in = x63;
        float* bufs[2] = {a, c};
        for (int l = 0; l < NL; l++) {
            float* out = (l == NL - 1) ? out12 : bufs[l & 1];
            linear_forward(DIMS[l], DIMS[l + 1], weight(l), bias(l), in, out, mode);
            if (l != NL - 1) relu_inplace(out, DIMS[l + 1]);
            in = out;
        }
    }

    // Policy::eval (policies.rs:47-59): raw policy logits, softmaxed outcome distribution [lose, draw, win].
    void eval(const Connect4& game, float logits[9], float value[3]) const {
        float x[63], out[12];
        if (mode == ACC_F16X2) f16_net().forward(game.my_bb, game.op_bb, out);
        else {
            game.features(x);
            forward(x, out);
        }
        for (int i = 0; i < 9; i++) logits[i] = out[i];
        softmax_stable(out + 9, value, 3);
    }
};

// ---- Connect4ConvNet: the conv policy/value network BASELINE.json's north_star words ("slimnn Conv2d over the 2x9x7 bitplane
// state + Linear policy/value heads"). The reference defines the LAYERS (slimnn Conv2d, conv.rs:45-85; Linear, linear.rs:17-25)
// but no such network (nothing in the reference instantiates Conv2d in a policy), so the architecture is this
// build's instantiation, fixed here:
//     x[2][7][9]   plane 0 = stones of the side to move, plane 1 = the opponent's (1.0 / 0.0), row 0 = bottom
//     Conv2d<IN 2, OUT 16, K 3, row_pad 1, col_pad 1, stride 1> -> [16][7][9], ReLU
//     Linear<1008, 12> on the NCHW flattening (i = channel*63 + row*9 + col): logits = out[0..9] (raw), value = softmax(out[9..12])
// Parameter blob: conv.weight[16][2][3][3], conv.bias[16], head.weight[12][1008], head.bias[12] = 12,412 f32.
// Accumulation modes:
//   ACC_SLIMNN : the layers exactly as slimnn loops them (conv taps ci -> k1 -> k2 with padded taps skipped, head inputs in
//                flattening order, separate multiply and add) — the canonical restatement, 1e-5 tolerance against the engine;
//   ACC_FMA    : the order the HIP engine's matrix-core chains produce bit for bit: conv taps in the same order (a padded tap is
//                fma(w, 0, acc): the same value), head inputs position-major (p = row*9 + col ascending; inside a position the
//                channels in the order 0,4,8,12, 1,5,9,13, 2,6,10,14, 3,7,11,15 — the D-register order of the conv tile),
//                one fused multiply-add per term. With 0/1 inputs the conv layer is bit-identical in both modes.
// PARITY: the layers are pinned by slimnn's KATs; the network has no counterpart in the reference -> unpinned by construction.
struct Connect4ConvNet {
    static constexpr int C = 16, H = 7, W = 9, HW = 63, FLAT = C * HW, OUT = 12;
    static constexpr size_t CONV_W = (size_t)C * 2 * 3 * 3, NUM_PARAMS = CONV_W + C + (size_t)OUT * FLAT + OUT;
    const float* blob = nullptr;
    int mode = ACC_SLIMNN;

    const float* conv_w() const { return blob; }
    const float* conv_b() const { return blob + CONV_W; }
    const float* head_w() const { return blob + CONV_W + C; }
    const float* head_b() const { return blob + CONV_W + C + (size_t)OUT * FLAT; }

    static void planes(const Connect4& game, float* x /*[2][7][9]*/) {
        for (int pl = 0; pl < 2; pl++) {
            const uint64_t bb = pl == 0 ? game.my_bb : game.op_bb;
            for (int r = 0; r < H; r++)
                for (int c = 0; c < W; c++) x[(pl * H + r) * W + c] = ((bb >> (r + 7 * c)) & 1ull) ? 1.0f : 0.0f;
        }
    }
    void forward(const float* x, float* out12) const {
        float y[FLAT];
        conv2d_forward(2, C, 3, 1, 1, 1, H, W, H, W, conv_w(), conv_b(), x, y, mode);
        relu_inplace(y, FLAT);
        if (mode == ACC_FMA) {
            for (int o = 0; o < OUT; o++) out12[o] = head_b()[o];
            for (int p = 0; p < HW; p++)
                for (int r = 0; r < 4; r++)
                    for (int q = 0; q < 4; q++) {
                        const int i = (4 * q + r) * HW + p;
                        for (int o = 0; o < OUT; o++) out12[o] = std::fmaf(y[i], head_w()[(size_t)o * FLAT + i], out12[o]);
                    }
        } else {
            linear_forward(FLAT, OUT, head_w(), head_b(), y, out12, ACC_SLIMNN);
        }
    }
    void eval(const Connect4& game, float logits[9], float value[3]) const {
        float x[2 * HW], out[OUT];
        planes(game, x);
        forward(x, out);
        for (int i = 0; i < 9; i++) logits[i] = out[i];
        softmax_stable(out + 9, value, 3);
    }
};

}  // namespace oracle

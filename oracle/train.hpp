// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of the step that follows the self-play path in the reference's learner:
//   synthesis/src/data.rs:196-235        ReplayBuffer::deduplicate (average the targets of identical states)
//   synthesis/src/alpha_zero.rs:72-94    one optimiser step: forward, log_softmax, kl_div(Sum) * (1/batch) for the
//                                        policy and the outcome head, loss = policy_weight*pi + value_weight*v, Adam
//   synthesis/src/alpha_zero.rs:33-36    Adam::default() + set_weight_decay
// The arithmetic lives in libtorch (tch 0.4.1, not in /root/reference): Linear / relu / log_softmax / kl_div / Adam.
// What is restated is their published semantics:
//   kl_div(input = log p, target t, Sum, log_target = false) = sum t * (log t - log p), with 0 * log 0 = 0
//   d/dlogits = scale * (softmax * sum(t) - t)                       (kl_div backward, then log_softmax backward)
//   Adam (torch::optim::Adam, amsgrad off): g += wd * p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g g;
//                                           p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// No reference test exercises this step -> PARITY UNPINNED against the reference; an independent second opinion
// (this container's torch, float64) is committed as tests/golden/train_torch_goldens.json and checked to 1e-5.
// Every f32 operation below is written in a fixed order (fma chains, ascending indices) so the HIP implementation
// can match bit for bit.
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <utility>
#include <vector>

#include "det_math.hpp"
#include "nn.hpp"

namespace oracle {

struct TrainHyper {
    float weight_decay = 1e-6f;   // study-connect4/src/main.rs:20
    float policy_weight = 1.0f;   // main.rs:23
    float value_weight = 1.0f;    // main.rs:24
    float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;  // Adam::default()
};

struct Trainer {
    static constexpr int NL = 5;
    std::vector<float> w, m, v, grad;
    int64_t step = 0;
    TrainHyper hp;

    explicit Trainer(const float* blob, TrainHyper h = TrainHyper())
        : w(blob, blob + Connect4Net::NUM_PARAMS), m(Connect4Net::NUM_PARAMS, 0.0f), v(Connect4Net::NUM_PARAMS, 0.0f),
          grad(Connect4Net::NUM_PARAMS, 0.0f), hp(h) {}

    static size_t w_off(int l) {
        size_t off = 0;
        for (int i = 0; i < l; i++) off += (size_t)Connect4Net::DIMS[i] * Connect4Net::DIMS[i + 1] + Connect4Net::DIMS[i + 1];
        return off;
    }
    static size_t b_off(int l) { return w_off(l) + (size_t)Connect4Net::DIMS[l] * Connect4Net::DIMS[l + 1]; }

    // Gradients of loss = pw * (1/B) * KL(pi) + vw * (1/B) * KL(v) for one minibatch; losses[0..1] = pi_loss, v_loss.
    // X[B][63] features, tpi[B][9], tv[B][3].
    void gradients(const float* X, const float* tpi, const float* tv, int B, float losses[2]) {
        const int* D = Connect4Net::DIMS;
        std::vector<std::vector<float>> A(NL + 1), dZ(NL + 1);
        A[0].assign(X, X + (size_t)B * 63);
        for (int l = 0; l < NL; l++) {
            int K = D[l], O = D[l + 1];
            A[l + 1].resize((size_t)B * O);
            const float* W = w.data() + w_off(l);
            const float* bias = w.data() + b_off(l);
            for (int b = 0; b < B; b++)
                for (int o = 0; o < O; o++) {
                    float acc = bias[o];
                    for (int k = 0; k < K; k++) acc = std::fmaf(A[l][(size_t)b * K + k], W[(size_t)o * K + k], acc);
                    if (l < NL - 1) acc = acc > 0.0f ? acc : 0.0f;
                    A[l + 1][(size_t)b * O + o] = acc;
                }
        }
        // heads: out[b][0..9) policy logits, out[b][9..12) outcome logits
        const float bm = 1.0f / (float)B;  // batch_mean (alpha_zero.rs:44)
        dZ[NL].assign((size_t)B * 12, 0.0f);
        float pi_loss = 0.0f, v_loss = 0.0f;
        for (int b = 0; b < B; b++) {
            for (int head = 0; head < 2; head++) {
                const int off = head == 0 ? 0 : 9, n = head == 0 ? 9 : 3;
                const float* x = &A[NL][(size_t)b * 12 + off];
                const float* t = head == 0 ? tpi + (size_t)b * 9 : tv + (size_t)b * 3;
                const float weight = head == 0 ? hp.policy_weight : hp.value_weight;
                float mx = x[0];
                for (int j = 1; j < n; j++) mx = x[j] > mx ? x[j] : mx;
                float se = 0.0f;
                for (int j = 0; j < n; j++) se += det_expf(x[j] - mx);
                const float lse = mx + det_logf(se);
                float kl = 0.0f, tsum = 0.0f;
                for (int j = 0; j < n; j++) {
                    float logp = x[j] - lse;
                    if (t[j] > 0.0f) kl += t[j] * (det_logf(t[j]) - logp);
                    tsum += t[j];
                }
                (head == 0 ? pi_loss : v_loss) += kl;
                const float s = weight * bm;
                for (int j = 0; j < n; j++) {
                    float p = det_expf(x[j] - lse);
                    dZ[NL][(size_t)b * 12 + off + j] = s * (p * tsum - t[j]);
                }
            }
        }
        losses[0] = bm * pi_loss;
        losses[1] = bm * v_loss;
        // backward
        for (int l = NL - 1; l >= 0; l--) {
            int K = D[l], O = D[l + 1];
            const float* W = w.data() + w_off(l);
            float* gW = grad.data() + w_off(l);
            float* gb = grad.data() + b_off(l);
            for (int o = 0; o < O; o++) {
                float acc = 0.0f;
                for (int b = 0; b < B; b++) acc += dZ[l + 1][(size_t)b * O + o];
                gb[o] = acc;
                for (int k = 0; k < K; k++) {
                    float a = 0.0f;
                    for (int b = 0; b < B; b++) a = std::fmaf(dZ[l + 1][(size_t)b * O + o], A[l][(size_t)b * K + k], a);
                    gW[(size_t)o * K + k] = a;
                }
            }
            if (l > 0) {
                dZ[l].resize((size_t)B * K);
                for (int b = 0; b < B; b++)
                    for (int k = 0; k < K; k++) {
                        float a = 0.0f;
                        for (int o = 0; o < O; o++) a = std::fmaf(dZ[l + 1][(size_t)b * O + o], W[(size_t)o * K + k], a);
                        dZ[l][(size_t)b * K + k] = A[l][(size_t)b * K + k] > 0.0f ? a : 0.0f;  // relu'
                    }
            }
        }
    }

    // torch::optim::Adam step on `grad` (already summed/averaged over ranks if data-parallel)
    void adam(float lr) {
        step += 1;
        const double bc1 = 1.0 - std::pow((double)hp.beta1, (double)step);
        const double bc2 = 1.0 - std::pow((double)hp.beta2, (double)step);
        const float step_size = (float)((double)lr / bc1);
        const float inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
        for (size_t i = 0; i < w.size(); i++) {
            float g = hp.weight_decay != 0.0f ? std::fmaf(hp.weight_decay, w[i], grad[i]) : grad[i];
            m[i] = std::fmaf(1.0f - hp.beta1, g, hp.beta1 * m[i]);
            v[i] = std::fmaf((1.0f - hp.beta2) * g, g, hp.beta2 * v[i]);
            float denom = std::sqrt(v[i]) * inv_sqrt_bc2 + hp.eps;
            w[i] = w[i] - step_size * (m[i] / denom);
        }
    }

    void train_step(const float* X, const float* tpi, const float* tv, int B, float lr, float losses[2]) {
        gradients(X, tpi, tv, B, losses);
        adam(lr);
    }
};

// ---- the same step for Connect4ConvNet (oracle/nn.hpp: Conv2d<2,16,3,pad 1> + ReLU + Linear<1008,12>; the network of north_star).
// The reference has neither this network nor a learner for it (its learner is libtorch's autograd over whatever NNPolicy is
// given): the published semantics restated above apply unchanged; every f32 chain below runs in one fixed order that the HIP
// kernel (synthesis_amd/csrc/train_conv_mfma.cuh: every chain a k-ordered fma chain of the f32 matrix cores) reproduces bit for bit:
//   forward   conv taps in slimnn's order ci -> k1 -> k2 (in-board taps only), fma per term; head = sixteen partial chains over
//             the cells p = g (mod 16), channels 0,4,8,12, 1,5,9,13, ... inside a cell, added to the bias in order
//   dWh[o][i] over the samples ascending; dAct[b][i] over the outputs ascending; dWc[c][tap] over the samples ascending and, inside
//   a sample, the cells ascending (row-major, in-board taps only), as sixteen partial chains over the sample pairs [2g, 2g + 2)
//   that are then added in order; bias gradients as plain sums in the same orders.
// Checked against this container's torch in float64 (tests/golden/conv_train_torch_goldens.npz).
struct ConvTrainer {
    using Net = Connect4ConvNet;
    std::vector<float> w, m, v, grad;
    int64_t step = 0;
    TrainHyper hp;

    explicit ConvTrainer(const float* blob, TrainHyper h = TrainHyper())
        : w(blob, blob + Net::NUM_PARAMS), m(Net::NUM_PARAMS, 0.0f), v(Net::NUM_PARAMS, 0.0f), grad(Net::NUM_PARAMS, 0.0f), hp(h) {}

    // in-board source cell of tap (k1, k2) for output cell (r, c), or -1
    static int tap_src(int r, int c, int k1, int k2) {
        const int rr = r + k1 - 1, cc = c + k2 - 1;
        return (rr >= 0 && rr < Net::H && cc >= 0 && cc < Net::W) ? rr * Net::W + cc : -1;
    }

    // my_bb / op_bb [B], tpi[B][9], tv[B][3]
    void gradients(const uint64_t* my_bb, const uint64_t* op_bb, const float* tpi, const float* tv, int B, float losses[2]) {
        const float* cw = w.data();
        const float* cb = cw + Net::CONV_W;
        const float* hw = cb + Net::C;
        const float* hb = hw + (size_t)Net::OUT * Net::FLAT;
        std::vector<float> X((size_t)B * 2 * Net::HW), A((size_t)B * Net::FLAT), out((size_t)B * 12), dz((size_t)B * 12, 0.0f);
        for (int b = 0; b < B; b++) {
            Connect4 g = Connect4::from_bitboards(my_bb[b], op_bb[b]);
            Net::planes(g, &X[(size_t)b * 2 * Net::HW]);
            const float* x = &X[(size_t)b * 2 * Net::HW];
            for (int c = 0; c < Net::C; c++)
                for (int r = 0; r < Net::H; r++)
                    for (int col = 0; col < Net::W; col++) {
                        float acc = cb[c];
                        for (int ci = 0; ci < 2; ci++)
                            for (int k1 = 0; k1 < 3; k1++)
                                for (int k2 = 0; k2 < 3; k2++) {
                                    const int src = tap_src(r, col, k1, k2);
                                    if (src >= 0) acc = std::fmaf(cw[((c * 2 + ci) * 3 + k1) * 3 + k2], x[ci * Net::HW + src], acc);
                                }
                        A[(size_t)b * Net::FLAT + c * Net::HW + r * Net::W + col] = acc > 0.0f ? acc : 0.0f;
                    }
            // head: sixteen partial fma chains — chain g over the cells p = g, g + 16, g + 32, g + 48 (< 63), inside a cell the
            // channels 0,4,8,12, 1,5,9,13, ... (the k order of the matrix-core tile, convnet.cuh) — then bias + P0 + P1 + ... in order:
            // the order in which the sixteen waves of the HIP learner (train_conv_mfma.cuh) produce them
            for (int o = 0; o < 12; o++) {
                float acc = hb[o];
                for (int g = 0; g < 16; g++) {
                    float part = 0.0f;
                    for (int p = g; p < Net::HW; p += 16)
                        for (int r = 0; r < 4; r++)
                            for (int q = 0; q < 4; q++) {
                                const size_t i = (size_t)(4 * q + r) * Net::HW + p;
                                part = std::fmaf(A[(size_t)b * Net::FLAT + i], hw[(size_t)o * Net::FLAT + i], part);
                            }
                    acc += part;
                }
                out[(size_t)b * 12 + o] = acc;
            }
        }
        // heads (identical to Trainer::gradients)
        const float bm = 1.0f / (float)B;
        float pi_loss = 0.0f, v_loss = 0.0f;
        for (int b = 0; b < B; b++)
            for (int head = 0; head < 2; head++) {
                const int off = head == 0 ? 0 : 9, n = head == 0 ? 9 : 3;
                const float* x = &out[(size_t)b * 12 + off];
                const float* t = head == 0 ? tpi + (size_t)b * 9 : tv + (size_t)b * 3;
                const float weight = head == 0 ? hp.policy_weight : hp.value_weight;
                float mx = x[0];
                for (int j = 1; j < n; j++) mx = x[j] > mx ? x[j] : mx;
                float se = 0.0f;
                for (int j = 0; j < n; j++) se += det_expf(x[j] - mx);
                const float lse = mx + det_logf(se);
                float kl = 0.0f, tsum = 0.0f;
                for (int j = 0; j < n; j++) {
                    float logp = x[j] - lse;
                    if (t[j] > 0.0f) kl += t[j] * (det_logf(t[j]) - logp);
                    tsum += t[j];
                }
                (head == 0 ? pi_loss : v_loss) += kl;
                const float s = weight * bm;
                for (int j = 0; j < n; j++) dz[(size_t)b * 12 + off + j] = s * (det_expf(x[j] - lse) * tsum - t[j]);
            }
        losses[0] = bm * pi_loss;
        losses[1] = bm * v_loss;
        // backward: heads
        float* gcw = grad.data();
        float* gcb = gcw + Net::CONV_W;
        float* ghw = gcb + Net::C;
        float* ghb = ghw + (size_t)Net::OUT * Net::FLAT;
        for (int o = 0; o < 12; o++) {
            float acc = 0.0f;
            for (int b = 0; b < B; b++) acc += dz[(size_t)b * 12 + o];
            ghb[o] = acc;
            for (int i = 0; i < Net::FLAT; i++) {
                float a = 0.0f;
                for (int b = 0; b < B; b++) a = std::fmaf(dz[(size_t)b * 12 + o], A[(size_t)b * Net::FLAT + i], a);
                ghw[(size_t)o * Net::FLAT + i] = a;
            }
        }
        // activation gradients through the ReLU: dY[b][i]
        std::vector<float> dY((size_t)B * Net::FLAT);
        for (int b = 0; b < B; b++)
            for (int i = 0; i < Net::FLAT; i++) {
                float a = 0.0f;
                for (int o = 0; o < 12; o++) a = std::fmaf(dz[(size_t)b * 12 + o], hw[(size_t)o * Net::FLAT + i], a);
                dY[(size_t)b * Net::FLAT + i] = A[(size_t)b * Net::FLAT + i] > 0.0f ? a : 0.0f;
            }
        // conv parameters: sixteen partial chains over the sample pairs [2 g, 2 g + 2) (samples ascending, cells row-major inside a
        // sample), then added in order — the order in which the HIP kernel's sixteen waves produce them (groups without a sample
        // contribute +0)
        for (int c = 0; c < Net::C; c++) {
            float sq[16];
            for (int g = 0; g < 16; g++) sq[g] = 0.0f;
            for (int b = 0; b < B; b++)
                for (int p = 0; p < Net::HW; p++) sq[b >> 1] += dY[(size_t)b * Net::FLAT + c * Net::HW + p];
            float sb = sq[0];
            for (int g = 1; g < 16; g++) sb += sq[g];
            gcb[c] = sb;
            for (int ci = 0; ci < 2; ci++)
                for (int k1 = 0; k1 < 3; k1++)
                    for (int k2 = 0; k2 < 3; k2++) {
                        float a[16];
                        for (int g = 0; g < 16; g++) a[g] = 0.0f;
                        for (int b = 0; b < B; b++)
                            for (int r = 0; r < Net::H; r++)
                                for (int col = 0; col < Net::W; col++) {
                                    const int src = tap_src(r, col, k1, k2);
                                    if (src >= 0)
                                        a[b >> 1] = std::fmaf(dY[(size_t)b * Net::FLAT + c * Net::HW + r * Net::W + col],
                                                              X[(size_t)b * 2 * Net::HW + ci * Net::HW + src], a[b >> 1]);
                                }
                        float sa = a[0];
                        for (int g = 1; g < 16; g++) sa += a[g];
                        gcw[((c * 2 + ci) * 3 + k1) * 3 + k2] = sa;
                    }
        }
    }

    void adam(float lr) {  // as Trainer::adam
        step += 1;
        const double bc1 = 1.0 - std::pow((double)hp.beta1, (double)step);
        const double bc2 = 1.0 - std::pow((double)hp.beta2, (double)step);
        const float step_size = (float)((double)lr / bc1);
        const float inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
        for (size_t i = 0; i < w.size(); i++) {
            float g = hp.weight_decay != 0.0f ? std::fmaf(hp.weight_decay, w[i], grad[i]) : grad[i];
            m[i] = std::fmaf(1.0f - hp.beta1, g, hp.beta1 * m[i]);
            v[i] = std::fmaf((1.0f - hp.beta2) * g, g, hp.beta2 * v[i]);
            float denom = std::sqrt(v[i]) * inv_sqrt_bc2 + hp.eps;
            w[i] = w[i] - step_size * (m[i] / denom);
        }
    }
};

// data.rs:196-235: average the targets of identical states. Output order of the reference is HashMap iteration order
// (unspecified); here: ascending (my_bb, op_bb). Sums run in buffer order, then one division by the count.
struct DedupEntry {
    uint64_t my_bb, op_bb;
    float pi[9], v[3];
    uint32_t num;
};
inline std::vector<DedupEntry> deduplicate(const uint64_t* my_bb, const uint64_t* op_bb, const float* pis,
                                           const float* vs, size_t n) {
    std::map<std::pair<uint64_t, uint64_t>, DedupEntry> stats;
    for (size_t i = 0; i < n; i++) {
        auto key = std::make_pair(my_bb[i], op_bb[i]);
        auto it = stats.find(key);
        if (it == stats.end()) {
            DedupEntry e{};
            e.my_bb = my_bb[i];
            e.op_bb = op_bb[i];
            it = stats.emplace(key, e).first;
        }
        for (int j = 0; j < 9; j++) it->second.pi[j] += pis[i * 9 + j];
        for (int j = 0; j < 3; j++) it->second.v[j] += vs[i * 3 + j];
        it->second.num += 1;
    }
    std::vector<DedupEntry> out;
    out.reserve(stats.size());
    for (auto& kv : stats) {
        DedupEntry e = kv.second;
        for (int j = 0; j < 9; j++) e.pi[j] = e.pi[j] / (float)e.num;
        for (int j = 0; j < 3; j++) e.v[j] = e.v[j] / (float)e.num;
        out.push_back(e);
    }
    return out;
}

}  // namespace oracle

// synthesis_amd — the learner step on the f32 matrix cores (replaces the VALU tiles of train_kernels.cuh).
//
// Same arithmetic as train_grad_kernel / oracle/train.hpp — alpha_zero.rs:72-94: forward, log_softmax + kl_div, backward — and the
// same fixed-order fused-multiply-add chains, but every chain runs as v_mfma_f32_16x16x4_f32 (a k-ordered chain of fmas with one
// rounding per term, the property the inference kernels already rely on):
//   forward   Z[b][o]  = fma(x[K-1], w[K-1], ... fma(x[0], w[0], bias))     rows = 16 outputs, cols = 16 samples, k = inputs
//   dA        dA[b][k] = chain over o ascending of fma(dZ[b][o], W[o][k], .) rows = 16 inputs,  cols = 16 samples, k = outputs
//   dW        dW[o][k] = chain over b ascending of fma(dZ[b][o], A[b][k], .) rows = 16 outputs, cols = 16 inputs,  k = samples
//             (continued across 32-sample chunks through the gradient buffer, as before)
// One workgroup of 16 waves; a layer's (row block x column block) tiles are dealt to the waves, a workgroup barrier separates the
// layers. Activations and activation gradients live in LDS ([sample][unit] rows, as in train_kernels.cuh); the weights are NOT
// staged: the A operands come straight from two fragment-order images in global memory (L2-resident, one coalesced 16-byte
// load per lane = 4 k-steps) that adam_image_kernel rewrites together with the canonical weights:
//   wimg   the inference image (mlp.cuh MlpGeom: [layer][s4][ob][lane][r] with the unit permutation) — forward; publishing the
//          trained network to the self-play engine is a device-to-device copy of this image
//   timg   the transposed image [layer 1..4][o-group s4][k-block kb][lane = 16 q + i][r] = W[16 s4 + 4 r + q][16 kb + i] — dA
// Padded terms multiply exact zeros (feature 63, output rows 12..15 of the last layer, samples past a partial chunk): fma(0, 0, acc)
// = acc, so they leave every chain's value unchanged.
#pragma once
#include "mlp.cuh"
#include "train_kernels.cuh"

namespace syn {

struct TrainImg {
    // transposed image: float offsets per layer l = 1..4 (index l), [s4 (O/16 groups)][kb (K/16 blocks)][64 lanes][4]
    static constexpr int T_S4[5] = {0, 6, 4, 3, 1};
    static constexpr int T_KB[5] = {0, 8, 6, 4, 3};
    static constexpr int T_OFF[6] = {0, 0, 12288, 18432, 21504, 22272};
    static constexpr int T_FLOATS = 22272;
};

// canonical parameter index -> its slots in the two images (-1 = none). Used by adam_image_kernel and by the host init.
__host__ __device__ inline void train_image_slots(int p, int& fwd, int& tr) {
    using G = TrainGeom;
    fwd = -1;
    tr = -1;
    int l = 0;
    while (l < 4 && p >= G::w_off(l + 1)) l++;
    const int K = G::D[l], O = G::D[l + 1];
    const int rel = p - G::w_off(l);
    const int NOB = MlpGeom::NOB[l];
    if (rel < K * O) {
        const int o = rel / K, k = rel - o * K;
        const int ob = o >> 4, c = o & 15;
        const int i = l == 4 ? c : (c >> 2) + 4 * (c & 3);  // A-row of unit o inside its block (inverse of mlp_unit_of_row)
        const int s4 = k >> 4, r = (k & 15) >> 2, q = k & 3;
        fwd = MlpGeom::W_OFF[l] + ((s4 * NOB + ob) * 64 + q * 16 + i) * 4 + r;
        if (l >= 1) {
            const int ts4 = o >> 4, tr_ = (o & 15) >> 2, tq = o & 3, kb = k >> 4, ti = k & 15;
            tr = TrainImg::T_OFF[l] + ((ts4 * TrainImg::T_KB[l] + kb) * 64 + tq * 16 + ti) * 4 + tr_;
        }
    } else {
        const int o = rel - K * O;
        const int ob = o >> 4, c = o & 15;
        const int i = l == 4 ? c : (c >> 2) + 4 * (c & 3);
        fwd = MlpGeom::W_FLOATS + MlpGeom::B_OFF[l] + (ob * 4 + (i >> 2)) * 4 + (i & 3);
    }
}

// torch::optim::Adam (as adam_kernel) + the two fragment images kept in step with the canonical weights
__global__ void adam_image_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ grads,
                                  int n, DevTrainHyper hp, float step_size, float inv_sqrt_bc2, float grad_scale,
                                  float* __restrict__ wimg, float* __restrict__ timg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g0 = grad_scale == 1.0f ? grads[i] : grads[i] * grad_scale;
    const float g = hp.weight_decay != 0.0f ? __builtin_fmaf(hp.weight_decay, w[i], g0) : g0;
    const float mi = __builtin_fmaf(1.0f - hp.beta1, g, hp.beta1 * m[i]);
    const float vi = __builtin_fmaf((1.0f - hp.beta2) * g, g, hp.beta2 * v[i]);
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + hp.eps;
    m[i] = mi;
    v[i] = vi;
    const float wn = w[i] - step_size * (mi / denom);
    w[i] = wn;
    int fwd, tr;
    train_image_slots(i, fwd, tr);
    wimg[fwd] = wn;
    if (tr >= 0) timg[tr] = wn;
}

// ---- forward: one (ob, cb) tile of layer L per call; lane (j, q): sample 16 cb + j
template <int L>
SYN_DEV void tm_forward_tile(const float* __restrict__ wimg, float* __restrict__ lds, int ob, int cb, int lane) {
    using G = TrainGeom;
    constexpr int S4 = MlpGeom::S4[L], NOB = MlpGeom::NOB[L], SA = G::stride(L), SO = G::stride(L + 1), O = G::D[L + 1];
    const int j = lane & 15, q = lane >> 4;
    const float* wl = wimg + MlpGeom::W_OFF[L] + lane * 4;
    f32x4 a[S4];
#pragma unroll
    for (int s4 = 0; s4 < S4; s4++) a[s4] = *reinterpret_cast<const f32x4*>(wl + (s4 * NOB + ob) * 256);
    f32x4 acc = *reinterpret_cast<const f32x4*>(wimg + MlpGeom::W_FLOATS + MlpGeom::B_OFF[L] + (ob * 4 + q) * 4);
    const float* A = lds + G::a_off(L) + (16 * cb + j) * SA + q;
#pragma unroll
    for (int s4 = 0; s4 < S4; s4++) {
        float b[4];
#pragma unroll
        for (int r = 0; r < 4; r++) b[r] = A[16 * s4 + 4 * r];  // x[k = 16 s4 + 4 r + q]
#pragma unroll
        for (int r = 0; r < 4; r++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4][r], b[r], acc, 0, 0, 0);
    }
    // D register r of lane (j, q) = unit 16 ob + 4 r + q (last layer: 16 ob + 4 q + r) of sample j
    float* out = lds + G::a_off(L + 1) + (16 * cb + j) * SO;
    if (L < G::NL - 1) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int u = 16 * ob + 4 * r + q;
            if (u < O) out[u] = acc[r] > 0.0f ? acc[r] : 0.0f;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int u = 16 * ob + 4 * q + r;
            if (u < O) out[u] = acc[r];
        }
    }
}

// ---- dA of layer L's input (L = 4..1): one (kb, cb) tile; rows = inputs 16 kb + i, cols = samples, chain over outputs
template <int L>
SYN_DEV void tm_backward_tile(const float* __restrict__ timg, float* __restrict__ lds, int kb, int cb, int lane) {
    using G = TrainGeom;
    constexpr int S4 = TrainImg::T_S4[L], NKB = TrainImg::T_KB[L], SA = G::stride(L), SZ = G::stride(L + 1), K = G::D[L];
    const int j = lane & 15, q = lane >> 4;
    const float* tl = timg + TrainImg::T_OFF[L] + lane * 4;
    f32x4 a[S4];
#pragma unroll
    for (int s4 = 0; s4 < S4; s4++) a[s4] = *reinterpret_cast<const f32x4*>(tl + (s4 * NKB + kb) * 256);
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* dZ = lds + G::d_off(L + 1) + (16 * cb + j) * SZ + q;  // columns past O are zero (see the heads / dA stores)
#pragma unroll
    for (int s4 = 0; s4 < S4; s4++) {
        float b[4];
#pragma unroll
        for (int r = 0; r < 4; r++) b[r] = dZ[16 * s4 + 4 * r];  // dZ[b][o = 16 s4 + 4 r + q]
#pragma unroll
        for (int r = 0; r < 4; r++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4][r], b[r], acc, 0, 0, 0);
    }
    // D register r of lane (j, q) = input 16 kb + 4 q + r of sample j: relu' mask, one 16-byte store
    const int k0 = 16 * kb + 4 * q;
    if (k0 < ((K + 3) & ~3)) {
        const f32x4 act = *reinterpret_cast<const f32x4*>(lds + G::a_off(L) + (16 * cb + j) * SA + k0);
        f32x4 rr;
#pragma unroll
        for (int r = 0; r < 4; r++) rr[r] = act[r] > 0.0f ? acc[r] : 0.0f;
        *reinterpret_cast<f32x4*>(lds + G::d_off(L) + (16 * cb + j) * SA + k0) = rr;
    }
}

// ---- dW of layer L: one (ob, kb) tile; rows = outputs, cols = inputs, chain over the chunk's 32 samples (continued from `grads`)
template <int L>
SYN_DEV void tm_param_tile(const float* __restrict__ lds, float* __restrict__ grads, int ob, int kb, bool first_chunk, int lane) {
    using G = TrainGeom;
    constexpr int K = G::D[L], O = G::D[L + 1], SA = G::stride(L), SZ = G::stride(L + 1);
    const int j = lane & 15, q = lane >> 4;
    float* gW = grads + G::w_off(L);
    f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int o = 16 * ob + 4 * q + r, k = 16 * kb + j;
        acc[r] = (first_chunk || o >= O || k >= K) ? 0.0f : gW[o * K + k];
    }
    const float* dZ = lds + G::d_off(L + 1) + q * SZ + 16 * ob + j;   // A operand: dZ[b = 4 s + q][o = 16 ob + i], i = lane & 15
    const float* A = lds + G::a_off(L) + q * SA + 16 * kb + j;        // B operand: A[b = 4 s + q][k = 16 kb + j]
    // (every index stays inside its padded LDS row: widths are multiples of 16 except 63 -> 64 and 12 -> 16, both zero-filled)
    float av[G::CHUNK / 4], bv[G::CHUNK / 4];
#pragma unroll
    for (int s = 0; s < G::CHUNK / 4; s++) {
        av[s] = dZ[4 * s * SZ];
        bv[s] = A[4 * s * SA];
    }
#pragma unroll
    for (int s = 0; s < G::CHUNK / 4; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int o = 16 * ob + 4 * q + r, k = 16 * kb + j;
        if (o < O && k < K) gW[o * K + k] = acc[r];
    }
}

template <int L>
SYN_DEV void tm_param_layer(const float* __restrict__ lds, float* __restrict__ grads, bool first_chunk, int wave, int lane, int& slot) {
    using G = TrainGeom;
    constexpr int NOB = (G::D[L + 1] + 15) / 16, NKB = (G::D[L] + 15) / 16;
    // tiles are dealt round-robin over the 16 waves across ALL layers (slot = running tile index)
#pragma unroll 1
    for (int t = 0; t < NOB * NKB; t++, slot++)
        if ((slot & 15) == wave) tm_param_tile<L>(lds, grads, t / NKB, t % NKB, first_chunk, lane);
}

// bias gradients: db[o] = sum over the chunk's samples in ascending order (plain adds, continued across chunks)
template <int L>
SYN_DEV void tm_bias_grads(const float* __restrict__ lds, float* __restrict__ grads, int nb, bool first_chunk, int t) {
    using G = TrainGeom;
    constexpr int O = G::D[L + 1], SZ = G::stride(L + 1);
    if (t >= 0 && t < O) {
        float* gb = grads + G::b_off(L);
        float acc = first_chunk ? 0.0f : gb[t];
        const float* dZ = lds + G::d_off(L + 1) + t;
        for (int b = 0; b < nb; b++) acc += dZ[b * SZ];
        gb[t] = acc;
    }
}

// Gradients of one minibatch by the whole 1024-thread workgroup (the body of train_grad_kernel_mfma).
SYN_DEV void tm_gradients(const float* wimg, const float* timg, const unsigned long long* __restrict__ my_bb,
                          const unsigned long long* __restrict__ op_bb, const float* __restrict__ tpi,
                          const float* __restrict__ tv, int B, DevTrainHyper hp, float* grads, float* losses,
                          const int* __restrict__ idx, unsigned long long* __restrict__ prof) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using G = TrainGeom;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int pslot = 0;
#define SYN_TSTAMP() do { if (prof && tid == 0) prof[pslot++] = (unsigned long long)__builtin_readcyclecounter(); } while (0)
    SYN_TSTAMP();
    const float bm = 1.0f / (float)B;
    float pi_acc = 0.0f, v_acc = 0.0f;  // thread 0 only

    for (int c0 = 0; c0 < B; c0 += G::CHUNK) {
        const int nb = B - c0 < G::CHUNK ? B - c0 : G::CHUNK;
        float tgt[9];
#pragma unroll
        for (int j = 0; j < 9; j++) tgt[j] = 0.0f;
        if (tid < 2 * G::CHUNK && (tid >> 1) < nb) {
            const int b = tid >> 1;
            const size_t si = idx ? (size_t)idx[c0 + b] : (size_t)(c0 + b);
            if ((tid & 1) == 0) {
#pragma unroll
                for (int j = 0; j < 9; j++) tgt[j] = tpi[si * 9 + j];
            } else {
#pragma unroll
                for (int j = 0; j < 3; j++) tgt[j] = tv[si * 3 + j];
            }
        }
        // features -> A[0] (columns 63.. of the padded rows are zero; rows of samples past nb are zero)
        {
            uint64_t bmy[2], bop[2];
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int b = (tid + it * 1024) >> 6;
                const size_t s = b < nb ? (idx ? (size_t)idx[c0 + b] : (size_t)(c0 + b)) : 0;
                bmy[it] = b < nb ? my_bb[s] : 0ull;
                bop[it] = b < nb ? op_bb[s] : 0ull;
            }
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int i = tid + it * 1024, b = i >> 6, f = i & 63;
                float x = 0.0f;
                if (b < nb && f < 63) x = c4::feature(bmy[it], bop[it], c4::next_free_cells(bmy[it] | bop[it]), f);
                lds[G::a_off(0) + b * G::stride(0) + f] = x;
            }
            // the last layer's outputs / gradients have 12 real columns in rows of 16: the 4 padding columns feed the matrix
            // cores as exact zeros
            if (tid < G::CHUNK * 4) lds[G::d_off(5) + (tid >> 2) * G::stride(5) + 12 + (tid & 3)] = 0.0f;
        }
        __syncthreads();
        SYN_TSTAMP();  // features
        // ---- forward: layer l has NOB x 2 tiles
#define TM_FWD(L)                                                                                           \
    for (int t = wave; t < MlpGeom::NOB[L] * 2; t += 16) tm_forward_tile<L>(wimg, lds, t >> 1, t & 1, lane); \
    __syncthreads();
        TM_FWD(0) TM_FWD(1) TM_FWD(2) TM_FWD(3) TM_FWD(4)
#undef TM_FWD
        SYN_TSTAMP();  // forward
        // ---- heads: log_softmax + kl_div and their gradient; one thread per (sample, head) — as train_grad_kernel
        if (tid < 2 * G::CHUNK) {
            const int b = tid >> 1, head = tid & 1;
            const int off = head == 0 ? 0 : 9, n = head == 0 ? 9 : 3;
            float* dz = lds + G::d_off(5) + b * G::stride(5) + off;
            float kl = 0.0f;
            if (b < nb) {
                const float* x = lds + G::a_off(5) + b * G::stride(5) + off;
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
                    if (j < n) dz[j] = s * (det_expf(xv[j] - lse) * tsum - tgt[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 9; j++)
                    if (j < n) dz[j] = 0.0f;
            }
            lds[G::KL_OFF + b * 2 + head] = kl;
        }
        __syncthreads();
        if (tid == 0) {
            for (int b = 0; b < nb; b++) {
                pi_acc += lds[G::KL_OFF + b * 2 + 0];
                v_acc += lds[G::KL_OFF + b * 2 + 1];
            }
        }
        SYN_TSTAMP();  // heads
        // ---- backward: activation gradients for the inputs of layers 4..1
#define TM_BWD(L)                                                                                               \
    for (int t = wave; t < TrainImg::T_KB[L] * 2; t += 16) tm_backward_tile<L>(timg, lds, t >> 1, t & 1, lane); \
    __syncthreads();
        TM_BWD(4) TM_BWD(3) TM_BWD(2) TM_BWD(1)
#undef TM_BWD
        SYN_TSTAMP();  // backward
        // ---- parameter gradients: 119 16x16 tiles over the 16 waves (they only read LDS), bias sums on spare threads
        {
            int slot = 0;
            tm_param_layer<1>(lds, grads, c0 == 0, wave, lane, slot);
            tm_param_layer<0>(lds, grads, c0 == 0, wave, lane, slot);
            tm_param_layer<2>(lds, grads, c0 == 0, wave, lane, slot);
            tm_param_layer<3>(lds, grads, c0 == 0, wave, lane, slot);
            tm_param_layer<4>(lds, grads, c0 == 0, wave, lane, slot);
            tm_bias_grads<0>(lds, grads, nb, c0 == 0, tid);
            tm_bias_grads<1>(lds, grads, nb, c0 == 0, tid - 128);
            tm_bias_grads<2>(lds, grads, nb, c0 == 0, tid - 256);
            tm_bias_grads<3>(lds, grads, nb, c0 == 0, tid - 384);
            tm_bias_grads<4>(lds, grads, nb, c0 == 0, tid - 512);
        }
        __syncthreads();
        SYN_TSTAMP();  // parameter gradients
    }
    if (tid == 0) {
        losses[0] = bm * pi_acc;
        losses[1] = bm * v_acc;
    }
#undef SYN_TSTAMP
}

// grads[NUM_PARAMS] (device) receives d(loss)/d(param) of the minibatch; losses[0..1] = pi_loss, v_loss.
__global__ __launch_bounds__(1024) void train_grad_kernel_mfma(const float* __restrict__ w, const float* __restrict__ wimg,
                                                               const float* __restrict__ timg,
                                                               const unsigned long long* __restrict__ my_bb,
                                                               const unsigned long long* __restrict__ op_bb,
                                                               const float* __restrict__ tpi, const float* __restrict__ tv, int B,
                                                               DevTrainHyper hp, float* __restrict__ grads, float* __restrict__ losses,
                                                               const int* __restrict__ idx = nullptr,
                                                               unsigned long long* __restrict__ prof = nullptr) {
    (void)w;
    tm_gradients(wimg, timg, my_bb, op_bb, tpi, tv, B, hp, grads, losses, idx, prof);
}

}  // namespace syn

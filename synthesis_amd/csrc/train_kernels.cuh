// synthesis_amd — the learner step that follows the self-play path.
//
// Replaces   synthesis/src/alpha_zero.rs:72-94   forward, log_softmax, kl_div(Sum) * (1/batch), loss, backward, Adam step
//            synthesis/src/alpha_zero.rs:33-36   Adam::default() + weight decay (libtorch semantics, see oracle/train.hpp)
//            synthesis/src/data.rs:196-235       ReplayBuffer::deduplicate
//
// The reference trains with batch_size 32 on a 30,492-parameter MLP: one optimiser step is ~3 MFLOP — far too small for
// anything but latency to matter. train_grad_kernel therefore runs the whole forward + backward of a minibatch in ONE
// workgroup with every activation and activation-gradient resident in LDS (104 KB) plus the current layer's weights (49 KB,
// staged per layer), register-tiled threads (4 outputs x 2 samples forward/backward, 4 x 4 parameters for dW) and
// fixed-order fma chains (bit-identical to oracle/train.hpp); gradients go to a caller-provided device buffer so that a
// data-parallel run can all-reduce them (RCCL, 122 KB) before adam_kernel applies the update.
#pragma once
#include "device_common.cuh"

namespace syn {

struct TrainGeom {
    static constexpr int NL = 5;
    static constexpr int D[NL + 1] = {63, 128, 96, 64, 48, 12};
    static constexpr int NUM_PARAMS = 30492;
    static constexpr int CHUNK = 32;  // samples resident in LDS at a time (the reference's batch_size)
    // activation rows: width rounded up to a multiple of 4 (16-byte vector reads along k) plus 4 floats of skew, which
    // spreads the rows of the 16 sample tiles of a wave over all LDS banks
    __host__ __device__ static constexpr int kp(int l) { return (D[l] + 3) & ~3; }     // padded width of layer l's input
    __host__ __device__ static constexpr int stride(int l) { return kp(l) + 4; }
    __host__ __device__ static constexpr int w_off(int l) {
        int off = 0;
        for (int i = 0; i < l; i++) off += D[i] * D[i + 1] + D[i + 1];
        return off;
    }
    __host__ __device__ static constexpr int b_off(int l) { return w_off(l) + D[l] * D[l + 1]; }
    __host__ __device__ static constexpr int a_off(int l) {  // activations A[l] in LDS (floats)
        int off = 0;
        for (int i = 0; i < l; i++) off += CHUNK * stride(i);
        return off;
    }
    static constexpr int A_FLOATS = CHUNK * (68 + 132 + 100 + 68 + 52 + 16);
    __host__ __device__ static constexpr int d_off(int l) {  // activation gradients dZ[l], l = 1..5
        int off = A_FLOATS;
        for (int i = 1; i < l; i++) off += CHUNK * stride(i);
        return off;
    }
    static constexpr int KL_OFF = A_FLOATS + CHUNK * (132 + 100 + 68 + 52 + 16);  // per-sample KL terms [CHUNK][2]
    static constexpr int WL_OFF = (KL_OFF + 2 * CHUNK + 3) & ~3;   // the current layer's weights [O][kp], staged per layer
    static constexpr int WL_FLOATS = 128 * 96;                    // largest layer
    static constexpr int LDS_FLOATS = WL_OFF + WL_FLOATS;         // 38,464 floats = 153,856 B
};

struct DevTrainHyper {
    float weight_decay, policy_weight, value_weight, beta1, beta2, eps;
};

typedef float tf4 __attribute__((ext_vector_type(4)));

// Where layer L's weights sit inside the staging buffer: layers 0 and 1 use it alone, layers 2, 3 and 4 (24 + 12 + 2 KB)
// share it, so they are staged once for the forward pass and are still there for the backward pass.
__host__ __device__ constexpr int train_wl_off(int L) {
    return TrainGeom::WL_OFF + (L == 3 ? 96 * 64 : (L == 4 ? 96 * 64 + 64 * 48 : 0));
}

// Copies layer L's weights [O][K] (global, row-major) into LDS rows of kp floats (zero padded).
template <int L>
SYN_DEV void train_stage_weights(const float* __restrict__ w, float* __restrict__ lds, int tid) {
    using G = TrainGeom;
    constexpr int K = G::D[L], O = G::D[L + 1], KP = G::kp(L);
    const float* src = w + G::w_off(L);
    float* dst = lds + train_wl_off(L);
    // all loads are issued before the first store (a load -> store loop would expose one global round trip per iteration)
    if (K == KP) {
        constexpr int N = K * O / 4, IT = (N + 1023) / 1024;
        tf4 r[IT];
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int i = tid + it * 1024;
            r[it] = i < N ? reinterpret_cast<const tf4*>(src)[i] : tf4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int i = tid + it * 1024;
            if (i < N) reinterpret_cast<tf4*>(dst)[i] = r[it];
        }
    } else {
        constexpr int N = KP * O, IT = (N + 1023) / 1024;
        float r[IT];
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int i = tid + it * 1024, o = i / KP, kk = i - o * KP;
            r[it] = (i < N && kk < K) ? src[o * K + kk] : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int i = tid + it * 1024;
            if (i < N) dst[i] = r[it];
        }
    }
}

// Forward layer L for the 32 samples of the chunk: thread tile = 4 outputs x 2 samples, each output an independent
// chain  fma(x[K-1], w[K-1], ... fma(x[0], w[0], bias))  in ascending k (oracle/train.hpp).
template <int L>
SYN_DEV void train_forward_layer(const float* __restrict__ w, float* __restrict__ lds, int tid) {
    using G = TrainGeom;
    constexpr int K = G::D[L], O = G::D[L + 1], KP = G::kp(L), SA = G::stride(L), SO = G::stride(L + 1);
    constexpr int K4 = K & ~3;
    if (tid < O * 4) {
        const int o0 = (tid >> 4) * 4, b0 = (tid & 15) * 2;
        const float* W = lds + train_wl_off(L) + o0 * KP;
        const float* A = lds + G::a_off(L) + b0 * SA;
        const float* bias = w + G::b_off(L) + o0;
        float acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i][0] = acc[i][1] = bias[i];
#pragma unroll 2
        for (int k = 0; k < K4; k += 4) {
            tf4 wv[4], av[2];
#pragma unroll
            for (int i = 0; i < 4; i++) wv[i] = *reinterpret_cast<const tf4*>(W + i * KP + k);
#pragma unroll
            for (int j = 0; j < 2; j++) av[j] = *reinterpret_cast<const tf4*>(A + j * SA + k);
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[i][j] = __builtin_fmaf(av[j][kk], wv[i][kk], acc[i][j]);
        }
#pragma unroll
        for (int k = K4; k < K; k++)  // tail of the 63-wide first layer (no padded term: -0 + 0 would not be an identity)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_fmaf(A[j * SA + k], W[i * KP + k], acc[i][j]);
        float* Aout = lds + G::a_off(L + 1) + b0 * SO + o0;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            tf4 r;
#pragma unroll
            for (int i = 0; i < 4; i++) r[i] = (L < G::NL - 1) ? (acc[i][j] > 0.0f ? acc[i][j] : 0.0f) : acc[i][j];
            *reinterpret_cast<tf4*>(Aout + j * SO) = r;
        }
    }
}

// dA of layer L's INPUT (L = 4..1): dA[b][k] = relu'(A[b][k]) * chain over o ascending of fma(dZ[b][o], W[o][k], .)
template <int L>
SYN_DEV void train_backward_act(float* __restrict__ lds, int tid) {
    using G = TrainGeom;
    constexpr int K = G::D[L], O = G::D[L + 1], KP = G::kp(L), SA = G::stride(L), SZ = G::stride(L + 1);
    if (tid < K * 4) {  // K / 4 column tiles x 16 sample tiles
        const int k0 = (tid >> 4) * 4, b0 = (tid & 15) * 2;
        const float* W = lds + train_wl_off(L) + k0;
        const float* dZ = lds + G::d_off(L + 1) + b0 * SZ;
        tf4 a[2];
        a[0] = a[1] = tf4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (int o = 0; o < O; o += 4) {
            tf4 dz[2];
#pragma unroll
            for (int j = 0; j < 2; j++) dz[j] = *reinterpret_cast<const tf4*>(dZ + j * SZ + o);
#pragma unroll
            for (int oo = 0; oo < 4; oo++) {
                const tf4 wv = *reinterpret_cast<const tf4*>(W + (o + oo) * KP);
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) a[j][kk] = __builtin_fmaf(dz[j][oo], wv[kk], a[j][kk]);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const tf4 act = *reinterpret_cast<const tf4*>(lds + G::a_off(L) + (b0 + j) * SA + k0);
            tf4 r;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) r[kk] = act[kk] > 0.0f ? a[j][kk] : 0.0f;
            *reinterpret_cast<tf4*>(lds + G::d_off(L) + (b0 + j) * SA + k0) = r;
        }
    }
}

// dW / db of layer L: thread tile = 4 outputs x 4 inputs, chains over the samples in ascending order, continued across
// chunks through the gradient buffer.
// T0 = first thread of the range that works on this layer (several layers run side by side on disjoint thread ranges).
template <int L, int T0>
SYN_DEV void train_param_grads(const float* __restrict__ lds, float* __restrict__ grads, int nb, bool first_chunk, int tid_) {
    using G = TrainGeom;
    constexpr int K = G::D[L], O = G::D[L + 1], KP = G::kp(L), SA = G::stride(L), SZ = G::stride(L + 1);
    constexpr int KT = KP / 4;
    static_assert(T0 + (O / 4) * KT <= 1024, "thread range");
    const int tid = tid_ - T0;
    if (tid >= 0 && tid < (O / 4) * KT) {
        const int o0 = (tid / KT) * 4, k0 = (tid % KT) * 4;
        const float* dZ = lds + G::d_off(L + 1) + o0;
        const float* A = lds + G::a_off(L) + k0;
        float* gW = grads + G::w_off(L);
        float* gb = grads + G::b_off(L);
        float g[4][4], db[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            db[i] = (first_chunk || k0 != 0) ? 0.0f : gb[o0 + i];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) g[i][kk] = (first_chunk || k0 + kk >= K) ? 0.0f : gW[(o0 + i) * K + k0 + kk];
        }
#pragma unroll 4
        for (int b = 0; b < nb; b++) {
            const tf4 dz = *reinterpret_cast<const tf4*>(dZ + b * SZ);
            const tf4 av = *reinterpret_cast<const tf4*>(A + b * SA);
#pragma unroll
            for (int i = 0; i < 4; i++) {
#pragma unroll
                for (int kk = 0; kk < 4; kk++) g[i][kk] = __builtin_fmaf(dz[i], av[kk], g[i][kk]);
                db[i] += dz[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
                if (k0 + kk < K) gW[(o0 + i) * K + k0 + kk] = g[i][kk];
            if (k0 == 0) gb[o0 + i] = db[i];
        }
    }
}

// grads[NUM_PARAMS] (device) receives d(loss)/d(param) of the minibatch; losses[0..1] = pi_loss, v_loss.
// Positions are given as bitboards; features are generated in the kernel (connect4.rs:235-258).
__global__ __launch_bounds__(1024) void train_grad_kernel(const float* __restrict__ w,
                                                          const unsigned long long* __restrict__ my_bb,
                                                          const unsigned long long* __restrict__ op_bb,
                                                          const float* __restrict__ tpi, const float* __restrict__ tv,
                                                          int B, DevTrainHyper hp, float* __restrict__ grads,
                                                          float* __restrict__ losses,
                                                          const int* __restrict__ idx = nullptr,
                                                          unsigned long long* __restrict__ prof = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using G = TrainGeom;
    const int tid = threadIdx.x;
    int pslot = 0;
#define SYN_TSTAMP() do { if (prof && tid == 0) prof[pslot++] = (unsigned long long)__builtin_readcyclecounter(); } while (0)
    SYN_TSTAMP();
    const float bm = 1.0f / (float)B;
    float pi_acc = 0.0f, v_acc = 0.0f;  // thread 0 only

    for (int c0 = 0; c0 < B; c0 += G::CHUNK) {
        const int nb = B - c0 < G::CHUNK ? B - c0 : G::CHUNK;
        // ---- layer 0's weights, the features and the targets are all global-latency bound: issue them together
        train_stage_weights<0>(w, lds, tid);
        float tgt[9];  // this (sample, head) thread's target row, consumed after the forward pass
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
        // ---- features -> A[0] (columns 63.. of the padded rows are zero); 2 elements per thread, boards loaded first
        {
            uint64_t bmy[2], bop[2];
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int b = (tid + it * 1024) >> 6;
                const size_t s = b < nb ? (idx ? (size_t)idx[c0 + b] : (size_t)(c0 + b)) : 0;  // BatchRandSampler's index_select
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
        }
        __syncthreads();
        SYN_TSTAMP();  // features
        // ---- forward: stage the weights (one coalesced pass), then 4x2 register tiles
        train_forward_layer<0>(w, lds, tid); __syncthreads();
        train_stage_weights<1>(w, lds, tid); __syncthreads(); train_forward_layer<1>(w, lds, tid); __syncthreads();
        train_stage_weights<2>(w, lds, tid); train_stage_weights<3>(w, lds, tid); train_stage_weights<4>(w, lds, tid);
        __syncthreads();
        train_forward_layer<2>(w, lds, tid); __syncthreads();
        train_forward_layer<3>(w, lds, tid); __syncthreads();
        train_forward_layer<4>(w, lds, tid); __syncthreads();
        SYN_TSTAMP();  // forward
        // ---- heads: log_softmax + kl_div and their gradient; one thread per (sample, head)
        if (tid < 2 * G::CHUNK) {
            const int b = tid >> 1, head = tid & 1;
            const int off = head == 0 ? 0 : 9, n = head == 0 ? 9 : 3;
            float* dz = lds + G::d_off(5) + b * G::stride(5) + off;
            float kl = 0.0f;
            if (b < nb) {
                const float* x = lds + G::a_off(5) + b * G::stride(5) + off;
                const float weight = head == 0 ? hp.policy_weight : hp.value_weight;
                // fixed trip count 9 with predicates (the value head has 3 entries): tgt[] stays in registers
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
            lds[G::KL_OFF + b * 2 + head] = kl;  // summed in sample order by thread 0 below
        }
        __syncthreads();
        if (tid == 0) {
            for (int b = 0; b < nb; b++) {
                pi_acc += lds[G::KL_OFF + b * 2 + 0];
                v_acc += lds[G::KL_OFF + b * 2 + 1];
            }
        }
        SYN_TSTAMP();  // heads
        // ---- backward: activation gradients for layers 4..1 (layers 4, 3, 2 are still staged)
        train_backward_act<4>(lds, tid); __syncthreads();
        train_backward_act<3>(lds, tid); __syncthreads();
        train_backward_act<2>(lds, tid); __syncthreads();
        train_stage_weights<1>(w, lds, tid); __syncthreads(); train_backward_act<1>(lds, tid); __syncthreads();
        SYN_TSTAMP();  // backward
        // ---- parameter gradients (no barrier between the layers: they only read LDS)
        // pass 1: layers 1 (768 threads), 3 (192) and 4 (36) side by side; pass 2: layers 0 (512) and 2 (384)
        train_param_grads<1, 0>(lds, grads, nb, c0 == 0, tid);
        train_param_grads<3, 768>(lds, grads, nb, c0 == 0, tid);
        train_param_grads<4, 960>(lds, grads, nb, c0 == 0, tid);
        train_param_grads<0, 0>(lds, grads, nb, c0 == 0, tid);
        train_param_grads<2, 512>(lds, grads, nb, c0 == 0, tid);
        __syncthreads();
        SYN_TSTAMP();  // parameter gradients
    }
    if (tid == 0) {
        losses[0] = bm * pi_acc;
        losses[1] = bm * v_acc;
    }
#undef SYN_TSTAMP
}

// torch::optim::Adam (amsgrad off) on device gradients; scalars prepared on the host in double like libtorch does.
__global__ void adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ grads, int n, DevTrainHyper hp, float step_size,
                            float inv_sqrt_bc2, float grad_scale) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g0 = grad_scale == 1.0f ? grads[i] : grads[i] * grad_scale;  // data-parallel mean = all-reduce sum * 1/ranks
    float g = hp.weight_decay != 0.0f ? __builtin_fmaf(hp.weight_decay, w[i], g0) : g0;
    float mi = __builtin_fmaf(1.0f - hp.beta1, g, hp.beta1 * m[i]);
    float vi = __builtin_fmaf((1.0f - hp.beta2) * g, g, hp.beta2 * v[i]);
    float denom = sqrtf(vi) * inv_sqrt_bc2 + hp.eps;
    m[i] = mi;
    v[i] = vi;
    w[i] = w[i] - step_size * (mi / denom);
}

// ---------------------------------------------------------------------------------------------- deduplicate
// After a stable sort of the buffer indices by (my_bb, op_bb): head[i] = 1 where a new state starts.
__global__ void dedup_heads_kernel(const unsigned long long* __restrict__ my_sorted,
                                   const unsigned long long* __restrict__ op_sorted, int n, unsigned* __restrict__ head) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || my_sorted[i] != my_sorted[i - 1] || op_sorted[i] != op_sorted[i - 1]) ? 1u : 0u;
}
// seg_start[s] = first sorted position of unique state s (seg_id = inclusive scan of head, minus 1)
__global__ void dedup_starts_kernel(const unsigned* __restrict__ head, const unsigned* __restrict__ seg_incl, int n,
                                    unsigned* __restrict__ seg_start) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (head[i]) seg_start[seg_incl[i] - 1] = (unsigned)i;
}
// One 16-lane row per unique state, lane j < 12 owns one target component and sums it over the duplicates in buffer
// order (the stable sort keeps buffer order inside a segment), then divides by the count (data.rs:206-226).
__global__ void dedup_reduce_kernel(const unsigned* __restrict__ order, const unsigned* __restrict__ seg_start, int m,
                                    int n, const unsigned long long* __restrict__ my_bb,
                                    const unsigned long long* __restrict__ op_bb, const float* __restrict__ pis,
                                    const float* __restrict__ vs, unsigned long long* __restrict__ out_my,
                                    unsigned long long* __restrict__ out_op, float* __restrict__ out_pi,
                                    float* __restrict__ out_v, unsigned* __restrict__ out_num) {
    int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    int j = threadIdx.x & 15;
    if (row >= m) return;
    unsigned s0 = seg_start[row], s1 = row + 1 < m ? seg_start[row + 1] : (unsigned)n;
    float acc = 0.0f;
    if (j < 12) {
        for (unsigned p = s0; p < s1; p++) {
            unsigned i = order[p];
            acc += j < 9 ? pis[(size_t)i * 9 + j] : vs[(size_t)i * 3 + (j - 9)];
        }
        float avg = acc / (float)(s1 - s0);
        if (j < 9) out_pi[(size_t)row * 9 + j] = avg;
        else out_v[(size_t)row * 3 + (j - 9)] = avg;
    }
    if (j == 12) {
        unsigned i = order[s0];
        out_my[row] = my_bb[i];
        out_op[row] = op_bb[i];
        out_num[row] = s1 - s0;
    }
}
// BatchRandSampler's index_select for a whole epoch at once: sample i of the step-ordered arrays = buffer entry perm[i]
__global__ void train_gather_kernel(const int* __restrict__ perm, int n, const unsigned long long* __restrict__ my,
                                    const unsigned long long* __restrict__ op, const float* __restrict__ tpi,
                                    const float* __restrict__ tv, unsigned long long* __restrict__ g_my,
                                    unsigned long long* __restrict__ g_op, float* __restrict__ g_tpi,
                                    float* __restrict__ g_tv) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, j = threadIdx.x & 15;
    if (i >= n) return;
    const size_t s = (size_t)perm[i];
    if (j < 9) g_tpi[(size_t)i * 9 + j] = tpi[s * 9 + j];
    else if (j < 12) g_tv[(size_t)i * 3 + (j - 9)] = tv[s * 3 + (j - 9)];
    else if (j == 12) g_my[i] = my[s];
    else if (j == 13) g_op[i] = op[s];
}
__global__ void iota_kernel(unsigned* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (unsigned)i;
}
__global__ void gather_u64_kernel(const unsigned long long* __restrict__ src, const unsigned* __restrict__ idx, int n,
                                  unsigned long long* __restrict__ dst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

}  // namespace syn

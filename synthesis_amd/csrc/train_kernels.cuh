// synthesis_amd — the learner step that follows the self-play path.
//
// Replaces   synthesis/src/alpha_zero.rs:72-94   forward, log_softmax, kl_div(Sum) * (1/batch), loss, backward, Adam step
//            synthesis/src/alpha_zero.rs:33-36   Adam::default() + weight decay (libtorch semantics, see oracle/train.hpp)
//            synthesis/src/data.rs:196-235       ReplayBuffer::deduplicate
//
// The reference trains with batch_size 32 on a 30,492-parameter MLP: one optimiser step is ~3 MFLOP — far too small for
// anything but latency to matter. train_grad_kernel therefore runs the whole forward + backward of a minibatch in ONE
// workgroup with every activation and activation-gradient resident in LDS (99 KB) plus the current layer's weights (49 KB, staged per layer), one thread per output element and
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
    __host__ __device__ static constexpr int stride(int l) { return D[l] | 1; }  // odd row stride: conflict-free columns
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
    static constexpr int A_FLOATS = CHUNK * (65 + 129 + 97 + 65 + 49 + 13);
    __host__ __device__ static constexpr int d_off(int l) {  // activation gradients dZ[l], l = 1..5
        int off = A_FLOATS;
        for (int i = 1; i < l; i++) off += CHUNK * stride(i);
        return off;
    }
    static constexpr int KL_OFF = A_FLOATS + CHUNK * (129 + 97 + 65 + 49 + 13);  // per-sample KL terms [CHUNK][2]
    static constexpr int WL_OFF = (KL_OFF + 2 * CHUNK + 3) & ~3;   // the current layer's weights, staged per layer (16-B aligned)
    static constexpr int WL_FLOATS = 128 * 96;                    // largest layer
    static constexpr int LDS_FLOATS = WL_OFF + WL_FLOATS;
};

struct DevTrainHyper {
    float weight_decay, policy_weight, value_weight, beta1, beta2, eps;
};

// grads[NUM_PARAMS] (device) receives d(loss)/d(param) of the minibatch; losses[0..1] = pi_loss, v_loss.
// Positions are given as bitboards; features are generated in the kernel (connect4.rs:235-258).
__global__ __launch_bounds__(1024) void train_grad_kernel(const float* __restrict__ w,
                                                          const unsigned long long* __restrict__ my_bb,
                                                          const unsigned long long* __restrict__ op_bb,
                                                          const float* __restrict__ tpi, const float* __restrict__ tv,
                                                          int B, DevTrainHyper hp, float* __restrict__ grads,
                                                          float* __restrict__ losses,
                                                          const int* __restrict__ idx = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using G = TrainGeom;
    const int tid = threadIdx.x;
    const float bm = 1.0f / (float)B;
    float pi_acc = 0.0f, v_acc = 0.0f;  // thread 0 only

    for (int c0 = 0; c0 < B; c0 += G::CHUNK) {
        const int nb = B - c0 < G::CHUNK ? B - c0 : G::CHUNK;
        // ---- features -> A[0]
        for (int i = tid; i < G::CHUNK * 63; i += 1024) {
            int b = i / 63, f = i - b * 63;
            float x = 0.0f;
            if (b < nb) {
                const size_t s = idx ? (size_t)idx[c0 + b] : (size_t)(c0 + b);  // BatchRandSampler's index_select
                uint64_t my = my_bb[s], op = op_bb[s];
                x = c4::feature(my, op, c4::next_free_cells(my | op), f);
            }
            lds[G::a_off(0) + b * G::stride(0) + f] = x;
        }
        __syncthreads();
        // ---- forward
#pragma unroll
        for (int l = 0; l < G::NL; l++) {
            const int K = G::D[l], O = G::D[l + 1];
            // stage the layer's weights in LDS: one coalesced pass instead of K dependent global loads per output
            {
                const float4* src = reinterpret_cast<const float4*>(w + G::w_off(l));
                float4* dst = reinterpret_cast<float4*>(lds + G::WL_OFF);
                for (int i = tid; i < K * O / 4; i += 1024) dst[i] = src[i];
            }
            __syncthreads();
            const float* W = lds + G::WL_OFF;
            const float* bias = w + G::b_off(l);
            const float* Ain = lds + G::a_off(l);
            float* Aout = lds + G::a_off(l + 1);
            for (int i = tid; i < O * G::CHUNK; i += 1024) {
                int o = i >> 5, b = i & 31;
                float acc = bias[o];
                const float* wr = W + o * K;
                const float* ar = Ain + b * G::stride(l);
                for (int k = 0; k < K; k++) acc = __builtin_fmaf(ar[k], wr[k], acc);
                if (l < G::NL - 1) acc = acc > 0.0f ? acc : 0.0f;
                Aout[b * G::stride(l + 1) + o] = acc;
            }
            __syncthreads();
        }
        // ---- heads: log_softmax + kl_div and their gradient; one thread per (sample, head)
        if (tid < 2 * G::CHUNK) {
            const int b = tid >> 1, head = tid & 1;
            const int off = head == 0 ? 0 : 9, n = head == 0 ? 9 : 3;
            float* dz = lds + G::d_off(5) + b * G::stride(5) + off;
            float kl = 0.0f;
            if (b < nb) {
                const float* x = lds + G::a_off(5) + b * G::stride(5) + off;
                const size_t si = idx ? (size_t)idx[c0 + b] : (size_t)(c0 + b);
                const float* t = head == 0 ? tpi + si * 9 : tv + si * 3;
                const float weight = head == 0 ? hp.policy_weight : hp.value_weight;
                float mx = x[0];
                for (int j = 1; j < n; j++) mx = x[j] > mx ? x[j] : mx;
                float se = 0.0f;
                for (int j = 0; j < n; j++) se += det_expf(x[j] - mx);
                const float lse = mx + det_logf(se);
                float tsum = 0.0f;
                for (int j = 0; j < n; j++) {
                    float logp = x[j] - lse;
                    if (t[j] > 0.0f) kl += t[j] * (det_logf(t[j]) - logp);
                    tsum += t[j];
                }
                const float s = weight * bm;
                for (int j = 0; j < n; j++) dz[j] = s * (det_expf(x[j] - lse) * tsum - t[j]);
            } else {
                for (int j = 0; j < n; j++) dz[j] = 0.0f;
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
        // ---- backward: activation gradients for layers 4..1 (needs the OLD weights of every layer)
#pragma unroll
        for (int l = G::NL - 1; l >= 1; l--) {
            const int K = G::D[l], O = G::D[l + 1];
            const float* W = lds + G::WL_OFF;
            const float* dZ = lds + G::d_off(l + 1);
            const float* A = lds + G::a_off(l);
            float* dA = lds + G::d_off(l);
            __syncthreads();
            {
                const float4* src = reinterpret_cast<const float4*>(w + G::w_off(l));
                float4* dst = reinterpret_cast<float4*>(lds + G::WL_OFF);
                for (int i = tid; i < K * O / 4; i += 1024) dst[i] = src[i];
            }
            __syncthreads();
            for (int i = tid; i < K * G::CHUNK; i += 1024) {
                int k = i >> 5, b = i & 31;
                float a = 0.0f;
                const float* dz = dZ + b * G::stride(l + 1);
                for (int o = 0; o < O; o++) a = __builtin_fmaf(dz[o], W[o * K + k], a);
                dA[b * G::stride(l) + k] = A[b * G::stride(l) + k] > 0.0f ? a : 0.0f;
            }
        }
        __syncthreads();
        // ---- parameter gradients: one thread per parameter, chain over the samples continues across chunks
#pragma unroll
        for (int l = 0; l < G::NL; l++) {
            const int K = G::D[l], O = G::D[l + 1];
            const float* dZ = lds + G::d_off(l + 1);
            const float* A = lds + G::a_off(l);
            float* gW = grads + G::w_off(l);
            float* gb = grads + G::b_off(l);
            for (int i = tid; i < O * K; i += 1024) {
                int o = i / K, k = i - o * K;
                float a = c0 == 0 ? 0.0f : gW[i];
                for (int b = 0; b < nb; b++)
                    a = __builtin_fmaf(dZ[b * G::stride(l + 1) + o], A[b * G::stride(l) + k], a);
                gW[i] = a;
            }
            for (int o = tid; o < O; o += 1024) {
                float a = c0 == 0 ? 0.0f : gb[o];
                for (int b = 0; b < nb; b++) a += dZ[b * G::stride(l + 1) + o];
                gb[o] = a;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        losses[0] = bm * pi_acc;
        losses[1] = bm * v_acc;
    }
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

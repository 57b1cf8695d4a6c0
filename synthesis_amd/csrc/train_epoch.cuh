// synthesis_amd — a whole epoch of optimiser steps in ONE launch (alpha_zero.rs:72-94 is the loop it replaces).
//
// train_mfma.cuh made the arithmetic of a step cheap; what was left of its 45 us were two launches per step, cold instruction
// and data fetches of a one-workgroup kernel, and weights that another kernel had just rewritten on other XCDs. Here the steps
// of an epoch (weights of step s feed step s + 1) run inside one persistent kernel of EP_WGS = 16 workgroups of 8 waves:
//   * forward, heads and activation gradients are computed REDUNDANTLY by every workgroup (they are the dependent chain: 1,648
//     matrix instructions = 5.5 us of one CU's four matrix pipes, nothing to share) with the fma chains of train_mfma.cuh. A
//     wave owns at most one unit (16 outputs x both 16-sample column blocks, two interleaved chains) per layer, so every weight
//     fragment is fetched once per workgroup and step; 2 waves per SIMD leave 256 VGPRs each, room for the fragments of several
//     layers in flight. LDS-only barriers between the layers: nothing waits for vmcnt.
//   * the parameter gradients and Adam are PARTITIONED: the 119 16x16 weight tiles and 7 bias groups are 126 jobs for the 128
//     waves of the launch. A wave keeps ITS parameters and their Adam moments in registers for the whole epoch. The dW tile is
//     computed in the orientation (rows = inputs in fragment order, columns = outputs in image order) whose D registers are
//     exactly the forward A-operand fragment of that tile, so dW -> Adam -> new fragment happens in registers and the new
//     weights go out as one coalesced 16-byte store per lane (plus the transposed copy the activation gradients read);
//   * one step barrier: every workgroup publishes its tiles into the OTHER of two image buffers, arrives on a counter, and reads
//     the complete new network after the last arrival. Two buffers make one barrier per step enough: nobody writes buffer
//     (s & 1) again before everyone has passed the barrier after reading it.
//   * the 16 workers are launched as every 8th workgroup of a 128-workgroup grid: workgroups are dealt round-robin to the 8
//     XCDs, so the workers share ONE XCD and its L2 is their coherence point — the barrier is "stores acknowledged by L2,
//     arrive, drop the vector L1" (the workgroup-scope release/acquire of threadgroup-split mode). The placement is CHECKED at
//     run time (HW_REG_XCC_ID of every worker, exchanged once with device-scope atomics); if the workers are not on one XCD the
//     barrier is a device-scope release/acquire (L2 write-back + invalidate: correct anywhere, 3x slower per step).
// The values are those of train_grad_kernel_mfma + adam_image_kernel (same chains, same Adam expression): bit-identical to
// oracle/train.hpp — tests/test_gpu_training.py::test_persistent_epoch_kernel_chains_epochs, ::test_epoch_kernel_modes_agree.
// Measured: 13.4 us per step = 74k steps/s at the reference's batch of 32 (two launches per step: 45 us).
// Batches of more than TrainGeom::CHUNK samples keep the queued two-kernel path (engine.hip).
#pragma once
#include "train_mfma.cuh"

namespace syn {

constexpr int EP_WGS = 16;    // working workgroups of the persistent launch (126 parameter jobs over 16 x 8 waves)
constexpr int EP_THREADS = 512;  // 8 waves = 2 per SIMD: 256 VGPRs each, room for every fragment a wave consumes in a step
constexpr int EP_WAVES = EP_THREADS / 64;
constexpr int EP_XCDS = 8;    // workgroups are dealt round-robin to the 8 XCDs: the launch has EP_WGS * EP_XCDS workgroups and only
                              // every 8th works, so that all workers share ONE XCD's L2 (checked at run time, see the kernel)
constexpr int EP_JOBS_W = 119;  // weight-tile jobs (layer order), followed by 7 bias jobs

struct EpochParams {
    float* w;                    // canonical parameters and Adam moments: read at the start, written back at the end
    float* m;
    float* v;
    float* img[2];               // fragment images, step s reads [s & 1] and writes [(s + 1) & 1]
    float* timg[2];
    const unsigned long long* my_bb;  // step-ordered batches (train_gather_kernel)
    const unsigned long long* op_bb;
    const float* tpi;
    const float* tv;
    const float* step_size;      // per step, prepared on the host in double like libtorch
    const float* inv_sqrt_bc2;
    float* losses;               // [n_steps][2]
    float* grads;                // canonical gradient of the LAST step (syn_trainer_get_state)
    unsigned* sync;              // [0] step arrivals, [1] status: 1 = a workgroup gave up waiting, [2] start arrivals,
                                 // [3] mode the launch ran in (1 = one XCD, L2-coherent), [8 .. 8 + EP_WGS) XCC id + 1 of each workgroup
    int force_device_scope;      // debug: keep the device-scope release/acquire even when all workgroups share an XCD
    unsigned long long* prof;    // optional cycle stamps of workgroup 0, step 1
    int n_steps, batch;
    DevTrainHyper hp;
};

// torch::optim::Adam for one parameter — the expression of adam_kernel / oracle/train.hpp (compiled with -ffp-contract=off)
SYN_DEV void adam_update(float& w, float& m, float& v, float g0, const DevTrainHyper& hp, float step_size, float inv_sqrt_bc2) {
    const float g = hp.weight_decay != 0.0f ? __builtin_fmaf(hp.weight_decay, w, g0) : g0;
    m = __builtin_fmaf(1.0f - hp.beta1, g, hp.beta1 * m);
    v = __builtin_fmaf((1.0f - hp.beta2) * g, g, hp.beta2 * v);
    const float denom = sqrtf(v) * inv_sqrt_bc2 + hp.eps;
    w = w - step_size * (m / denom);
}

// ---- forward: wave-unit = output block ob of layer L for BOTH 16-sample column blocks (every fragment is fetched once per
//      workgroup: a CU's load path, ~25 B/clk, is what bounds a step once the matrix work is spread over the pipes), fragments
//      requested ahead of the computation. The waves of consecutive layers differ (EP_FWD_BASE): the idle ones are already waiting
//      for the next layer's fragments, and a layer's units fall on different SIMDs.
constexpr int EP_FWD_BASE[5] = {0, 0, 6, 2, 5};   // layer L's slot u runs on wave (EP_FWD_BASE[L] + u) & 7
constexpr int EP_BWD_BASE[5] = {0, 0, 5, 1, 6};   // activation gradients: layer L's slot u on wave (EP_BWD_BASE[L] + u) & 7
// Slots of a layer: the first NF are whole units (both column blocks), then NH units are split into their two column blocks
// (two slots each). Six-unit layers would otherwise put two whole units on two of the four SIMDs (4,096 cycles of matrix pipe
// instead of 3,072); the one-unit last layer runs as two halves on two SIMDs.
constexpr int EP_FWD_NF[5] = {8, 4, 4, 3, 0}, EP_FWD_NH[5] = {0, 2, 0, 0, 1};
constexpr int EP_BWD_NF[5] = {0, 8, 4, 4, 3}, EP_BWD_NH[5] = {0, 0, 2, 0, 0};
struct EpSlot {
    int unit;   // output block (forward) / input block (activation gradients), -1 = this wave has no slot in the layer
    int cb0, ncb;
};
SYN_DEV EpSlot ep_slot(int wave, int base, int nf, int nh) {
    const int u = (wave - base) & (EP_WAVES - 1);
    if (u < nf) return EpSlot{u, 0, 2};
    if (u < nf + 2 * nh) return EpSlot{nf + ((u - nf) >> 1), (u - nf) & 1, 1};
    return EpSlot{-1, 0, 0};
}

template <int L>
struct EpFwd {
    f32x4 a[MlpGeom::S4[L]];
    f32x4 bias;
};
template <int L>
SYN_DEV void ep_fwd_load(const float* img, int wave, int lane, EpFwd<L>& F) {
    constexpr int S4 = MlpGeom::S4[L], NOB = MlpGeom::NOB[L];
    static_assert(EP_FWD_NF[L] + EP_FWD_NH[L] == NOB, "slots cover the layer");
    const EpSlot sl = ep_slot(wave, EP_FWD_BASE[L], EP_FWD_NF[L], EP_FWD_NH[L]);
    if (sl.unit >= 0) {
        const int ob = sl.unit, q = lane >> 4;
        const float* wl = img + MlpGeom::W_OFF[L] + lane * 4;
#pragma unroll
        for (int s4 = 0; s4 < S4; s4++) F.a[s4] = *reinterpret_cast<const f32x4*>(wl + (s4 * NOB + ob) * 256);
        F.bias = *reinterpret_cast<const f32x4*>(img + MlpGeom::W_FLOATS + MlpGeom::B_OFF[L] + (ob * 4 + q) * 4);
    }
}
template <int L>
SYN_DEV void ep_fwd_store(float* __restrict__ out, int ob, int q, const f32x4& acc) {
    using G = TrainGeom;
    constexpr int O = G::D[L + 1];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int u = L < G::NL - 1 ? 16 * ob + 4 * r + q : 16 * ob + 4 * q + r;
        if (u < O) out[u] = L < G::NL - 1 ? (acc[r] > 0.0f ? acc[r] : 0.0f) : acc[r];
    }
}
template <int L>
SYN_DEV void ep_fwd_compute(const EpFwd<L>& F, float* __restrict__ lds, int wave, int lane) {
    using G = TrainGeom;
    constexpr int S4 = MlpGeom::S4[L], SA = G::stride(L), SO = G::stride(L + 1);
    const EpSlot sl = ep_slot(wave, EP_FWD_BASE[L], EP_FWD_NF[L], EP_FWD_NH[L]);
    if (sl.unit < 0) return;
    const int ob = sl.unit, j = lane & 15, q = lane >> 4;
    const float* A = lds + G::a_off(L) + (16 * sl.cb0 + j) * SA + q;
    float* out = lds + G::a_off(L + 1) + (16 * sl.cb0 + j) * SO;
    if (sl.ncb == 2) {
        f32x4 acc0 = F.bias, acc1 = F.bias;  // samples j and 16 + j: two independent chains keep the matrix pipe busy
#pragma unroll
        for (int s4 = 0; s4 < S4; s4++) {
            float b0[4], b1[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                b0[r] = A[16 * s4 + 4 * r];
                b1[r] = A[16 * SA + 16 * s4 + 4 * r];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[s4][r], b0[r], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[s4][r], b1[r], acc1, 0, 0, 0);
            }
        }
        ep_fwd_store<L>(out, ob, q, acc0);
        ep_fwd_store<L>(out + 16 * SO, ob, q, acc1);
    } else {
        f32x4 acc = F.bias;
#pragma unroll
        for (int s4 = 0; s4 < S4; s4++) {
            float b0[4];
#pragma unroll
            for (int r = 0; r < 4; r++) b0[r] = A[16 * s4 + 4 * r];
#pragma unroll
            for (int r = 0; r < 4; r++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[s4][r], b0[r], acc, 0, 0, 0);
        }
        ep_fwd_store<L>(out, ob, q, acc);
    }
}

// ---- activation gradients: unit = input block kb of layer L
template <int L>
struct EpBwd {
    f32x4 a[TrainImg::T_S4[L]];
};
template <int L>
SYN_DEV void ep_bwd_load(const float* timg, int wave, int lane, EpBwd<L>& F) {
    constexpr int S4 = TrainImg::T_S4[L], NKB = TrainImg::T_KB[L];
    static_assert(EP_BWD_NF[L] + EP_BWD_NH[L] == NKB, "slots cover the layer");
    const EpSlot sl = ep_slot(wave, EP_BWD_BASE[L], EP_BWD_NF[L], EP_BWD_NH[L]);
    if (sl.unit >= 0) {
        const float* tl = timg + TrainImg::T_OFF[L] + lane * 4;
#pragma unroll
        for (int s4 = 0; s4 < S4; s4++) F.a[s4] = *reinterpret_cast<const f32x4*>(tl + (s4 * NKB + sl.unit) * 256);
    }
}
template <int L>
SYN_DEV void ep_bwd_store(float* __restrict__ lds, int row, int k0, const f32x4& acc) {
    using G = TrainGeom;
    constexpr int SA = G::stride(L);
    const f32x4 act = *reinterpret_cast<const f32x4*>(lds + G::a_off(L) + row * SA + k0);
    f32x4 rr;
#pragma unroll
    for (int r = 0; r < 4; r++) rr[r] = act[r] > 0.0f ? acc[r] : 0.0f;
    *reinterpret_cast<f32x4*>(lds + G::d_off(L) + row * SA + k0) = rr;
}
template <int L>
SYN_DEV void ep_bwd_compute(const EpBwd<L>& F, float* __restrict__ lds, int wave, int lane) {
    using G = TrainGeom;
    constexpr int S4 = TrainImg::T_S4[L], SZ = G::stride(L + 1);
    const EpSlot sl = ep_slot(wave, EP_BWD_BASE[L], EP_BWD_NF[L], EP_BWD_NH[L]);
    if (sl.unit < 0) return;
    const int j = lane & 15, q = lane >> 4;
    const int row = 16 * sl.cb0 + j, k0 = 16 * sl.unit + 4 * q;  // layers 1..4: every input width is a multiple of 16
    const float* dZ = lds + G::d_off(L + 1) + row * SZ + q;
    if (sl.ncb == 2) {
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
#pragma unroll
        for (int s4 = 0; s4 < S4; s4++) {
            float b0[4], b1[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                b0[r] = dZ[16 * s4 + 4 * r];
                b1[r] = dZ[16 * SZ + 16 * s4 + 4 * r];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[s4][r], b0[r], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[s4][r], b1[r], acc1, 0, 0, 0);
            }
        }
        ep_bwd_store<L>(lds, row, k0, acc0);
        ep_bwd_store<L>(lds, row + 16, k0, acc1);
    } else {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s4 = 0; s4 < S4; s4++) {
            float b0[4];
#pragma unroll
            for (int r = 0; r < 4; r++) b0[r] = dZ[16 * s4 + 4 * r];
#pragma unroll
            for (int r = 0; r < 4; r++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[s4][r], b0[r], acc, 0, 0, 0);
        }
        ep_bwd_store<L>(lds, row, k0, acc);
    }
}

// ---- a weight-tile job: tile (ob, s4) of layer L = the forward fragment [s4][ob]; lane (i, q) register r holds
//      W[o = unit(i) of block ob][k = 16 s4 + 4 r + q]
template <int L>
SYN_DEV int ep_unit(int i) { return L == TrainGeom::NL - 1 ? i : 4 * (i & 3) + (i >> 2); }

template <int L, typename F>
SYN_DEV void ep_tile_canonical(int ob, int s4, int lane, F&& f) {  // f(r, canonical index or -1)
    using G = TrainGeom;
    constexpr int K = G::D[L], O = G::D[L + 1];
    const int i = lane & 15, q = lane >> 4;
    const int o = 16 * ob + ep_unit<L>(i);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int k = 16 * s4 + 4 * r + q;
        f(r, (o < O && k < K) ? G::w_off(L) + o * K + k : -1);
    }
}

template <int L>
SYN_DEV void ep_weight_job(const float* __restrict__ lds, int ob, int s4, int lane, float (&wq)[4], float (&mq)[4], float (&vq)[4],
                           const DevTrainHyper& hp, float step_size, float inv_sqrt_bc2, float* img_next, float* timg_next,
                           float* grads_out) {
    using G = TrainGeom;
    constexpr int SA = G::stride(L), SZ = G::stride(L + 1), NOB = MlpGeom::NOB[L];
    const int i = lane & 15, q = lane >> 4;
    // D[m][n] = chain over the samples b ascending of A[m][b] * B[b][n]: row m <-> input 16 s4 + 4 (m & 3) + (m >> 2), so that
    // D register r of lane (n, q) (row 4 q + r) is input 16 s4 + 4 r + q; column n <-> output unit(n) of block ob
    const float* A = lds + G::a_off(L) + q * SA + 16 * s4 + 4 * (i & 3) + (i >> 2);
    const float* dZ = lds + G::d_off(L + 1) + q * SZ + 16 * ob + ep_unit<L>(i);
    float av[G::CHUNK / 4], bv[G::CHUNK / 4];
#pragma unroll
    for (int s = 0; s < G::CHUNK / 4; s++) {
        av[s] = A[4 * s * SA];
        bv[s] = dZ[4 * s * SZ];
    }
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < G::CHUNK / 4; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
    if (grads_out) ep_tile_canonical<L>(ob, s4, lane, [&](int r, int p) { if (p >= 0) grads_out[p] = acc[r]; });
#pragma unroll
    for (int r = 0; r < 4; r++) adam_update(wq[r], mq[r], vq[r], acc[r], hp, step_size, inv_sqrt_bc2);
    *reinterpret_cast<f32x4*>(img_next + MlpGeom::W_OFF[L] + ((s4 * NOB + ob) * 64 + lane) * 4) = f32x4{wq[0], wq[1], wq[2], wq[3]};
    if (L >= 1) {
        // transposed image: [o-group ob][k-block s4][lane' = 16 (c & 3) + k_local][c >> 2], c = unit(i), k_local = 4 r + q
        const int c = ep_unit<L>(i);
        float* tt = timg_next + TrainImg::T_OFF[L] + (ob * TrainImg::T_KB[L] + s4) * 256 + ((c & 3) * 16 + q) * 4 + (c >> 2);
#pragma unroll
        for (int r = 0; r < 4; r++) tt[16 * r] = wq[r];
    }
}

// ---- a bias job: lane owns bias o = 64 half + lane of layer L
template <int L>
SYN_DEV void ep_bias_job(const float* __restrict__ lds, int half, int lane, int nb, float& wb, float& mb, float& vb,
                         const DevTrainHyper& hp, float step_size, float inv_sqrt_bc2, float* img_next, float* grads_out) {
    using G = TrainGeom;
    constexpr int O = G::D[L + 1], SZ = G::stride(L + 1);
    const int o = 64 * half + lane;
    if (o >= O) return;
    const float* dZ = lds + G::d_off(L + 1) + o;
    float acc = 0.0f;
    for (int b = 0; b < nb; b++) acc += dZ[b * SZ];
    if (grads_out) grads_out[G::b_off(L) + o] = acc;
    adam_update(wb, mb, vb, acc, hp, step_size, inv_sqrt_bc2);
    const int ob = o >> 4, c = o & 15;
    const int i = L == G::NL - 1 ? c : (c >> 2) + 4 * (c & 3);
    img_next[MlpGeom::W_FLOATS + MlpGeom::B_OFF[L] + (ob * 4 + (i >> 2)) * 4 + (i & 3)] = wb;
}

// ---- the 16 lanes of a sample's row: lane t's value in every lane (DPP row_newbcast), and the two heads' sums over the
//      entries in ascending order (policy = lanes 0..8, outcome = lanes 9..11) — sequential adds from 0, as the reference order
SYN_DEV void ep_row_gather(float v, float (&out)[12]) {
    out[0] = dpp_f32<0x150>(v); out[1] = dpp_f32<0x151>(v); out[2] = dpp_f32<0x152>(v); out[3] = dpp_f32<0x153>(v);
    out[4] = dpp_f32<0x154>(v); out[5] = dpp_f32<0x155>(v); out[6] = dpp_f32<0x156>(v); out[7] = dpp_f32<0x157>(v);
    out[8] = dpp_f32<0x158>(v); out[9] = dpp_f32<0x159>(v); out[10] = dpp_f32<0x15A>(v); out[11] = dpp_f32<0x15B>(v);
}
SYN_DEV float ep_row_sum(float v, bool pol) {
    float x[12];
    ep_row_gather(v, x);
    float sp = 0.0f, sv = 0.0f;
#pragma unroll
    for (int t = 0; t < 9; t++) sp += x[t];
#pragma unroll
    for (int t = 9; t < 12; t++) sv += x[t];
    return pol ? sp : sv;
}

struct EpJob {
    int kind;  // 0 weight tile (L, a = ob, b = s4), 1 bias group (L, a = half), 2 none
    int L, a, b;
};
__host__ __device__ inline EpJob ep_job_decode(int job) {
    if (job < 32) return EpJob{0, 0, job / 4, job % 4};
    if (job < 80) return EpJob{0, 1, (job - 32) / 8, (job - 32) % 8};
    if (job < 104) return EpJob{0, 2, (job - 80) / 6, (job - 80) % 6};
    if (job < 116) return EpJob{0, 3, (job - 104) / 4, (job - 104) % 4};
    if (job < 119) return EpJob{0, 4, 0, job - 116};
    switch (job) {
        case 119: return EpJob{1, 0, 0, 0};
        case 120: return EpJob{1, 0, 1, 0};
        case 121: return EpJob{1, 1, 0, 0};
        case 122: return EpJob{1, 1, 1, 0};
        case 123: return EpJob{1, 2, 0, 0};
        case 124: return EpJob{1, 3, 0, 0};
        case 125: return EpJob{1, 4, 0, 0};
        default: return EpJob{2, 0, 0, 0};
    }
}

#define EP_LAYER_SWITCH(Lv, CALL)      \
    switch (Lv) {                      \
        case 0: { constexpr int LL = 0; CALL; } break; \
        case 1: { constexpr int LL = 1; CALL; } break; \
        case 2: { constexpr int LL = 2; CALL; } break; \
        case 3: { constexpr int LL = 3; CALL; } break; \
        default: { constexpr int LL = 4; CALL; } break; \
    }

__global__ __launch_bounds__(EP_THREADS) void train_epoch_kernel(EpochParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using G = TrainGeom;
    if (blockIdx.x % EP_XCDS != 0) return;
    const int tid0 = threadIdx.x, wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6), lane0 = tid0 & 63, g = blockIdx.x / EP_XCDS;
    const EpJob job = ep_job_decode(wave0 * EP_WGS + g);
    const int B = P.batch, nb = B;  // B <= CHUNK (host)
    const float bm = 1.0f / (float)B;
    __shared__ unsigned ep_abort, ep_fast;

    // ---- this wave's parameters and moments: registers for the whole epoch
    float wq[4] = {0.0f, 0.0f, 0.0f, 0.0f}, mq[4] = {0.0f, 0.0f, 0.0f, 0.0f}, vq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (job.kind == 0) {
        EP_LAYER_SWITCH(job.L, ep_tile_canonical<LL>(job.a, job.b, lane0, [&](int r, int p) {
            if (p >= 0) { wq[r] = P.w[p]; mq[r] = P.m[p]; vq[r] = P.v[p]; }
        }))
    } else if (job.kind == 1) {
        const int o = 64 * job.a + lane0;
        if (o < G::D[job.L + 1]) {
            const int p = G::b_off(job.L) + o;
            wq[0] = P.w[p]; mq[0] = P.m[p]; vq[0] = P.v[p];
        }
    }
    // the last layer's gradients have 12 real columns in rows of 16: the 4 padding columns feed the matrix cores as exact zeros
    if (tid0 < G::CHUNK * 4) lds[G::d_off(5) + (tid0 >> 2) * G::stride(5) + 12 + (tid0 & 3)] = 0.0f;
    // ---- where did the workers land? Exchanged once with device-scope atomics. If every worker sits on the same XCD the step
    //      barrier below only has to get past the CUs' vector L1s: that XCD's L2 is the coherence point of all of them.
    if (tid0 == 0) {
        ep_abort = 0u;
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) + 1u;  // HW_REG_XCC_ID[3:0]
        __hip_atomic_store(P.sync + 8 + g, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(P.sync + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(P.sync + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)EP_WGS) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) {
                __hip_atomic_store(P.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ep_abort = 1u;
                break;
            }
        }
        bool same = true;
        for (int i = 0; i < EP_WGS; i++) same = same && __hip_atomic_load(P.sync + 8 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xcc;
        ep_fast = (same && !P.force_device_scope) ? 1u : 0u;
        if (g == 0) P.sync[3] = ep_fast;
    }
    __syncthreads();
    if (ep_abort) return;
    const bool one_xcd = ep_fast != 0u;

    // ---- batch of step 0 (later batches are requested one step ahead)
    // features: thread -> sample tid >> 4, features 4 (tid & 15) + {0..3}; heads: the same sample, entry tid & 15
    uint64_t bmy = 0, bop = 0;
    float tgt_next = 0.0f;
    auto request_batch = [&](int s) {
        const size_t base = (size_t)s * (size_t)B;
        const int b = tid0 >> 4, hj = tid0 & 15;
        bmy = b < nb ? P.my_bb[base + b] : 0ull;
        bop = b < nb ? P.op_bb[base + b] : 0ull;
        tgt_next = 0.0f;
        if (b < nb) {
            if (hj < 9) tgt_next = P.tpi[(base + b) * 9 + hj];
            else if (hj < 12) tgt_next = P.tv[(base + b) * 3 + (hj - 9)];
        }
    };
    request_batch(0);

    int pslot = 0;
#define EP_STAMP() do { if (P.prof && tid0 == 0 && s == 2) P.prof[g * 16 + pslot++] = (unsigned long long)__builtin_readcyclecounter(); } while (0)

    for (int s = 0; s < P.n_steps; s++) {
        // per-iteration opaque copies of the thread coordinates: without them the compiler hoists every tile address of every
        // layer variant out of the step loop and keeps ~240 values in scratch for the whole epoch
        int lane = lane0, wave = wave0;
        asm volatile("" : "+v"(lane));
        asm volatile("" : "+s"(wave));
        const int tid = wave * 64 + lane;
        const float* img = (s & 1) ? P.img[1] : P.img[0];
        const float* timg = (s & 1) ? P.timg[1] : P.timg[0];
        float* img_next = (s & 1) ? P.img[0] : P.img[1];
        float* timg_next = (s & 1) ? P.timg[0] : P.timg[1];
        const bool last = s + 1 == P.n_steps;
        if (P.prof && s == 2 && tid0 == 0) {
            // diagnostic probes (SYN_TRAIN_PROFILE): load-to-use latency of a line another workgroup wrote in the last step, and of a
            // line nobody touched since the launch
            const unsigned long long t0 = __builtin_readcyclecounter();
            const float x = __builtin_nontemporal_load(img + MlpGeom::W_OFF[1] + 17 * 256);
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(x) : "memory");
            const unsigned long long t1 = __builtin_readcyclecounter();
            const float y = __builtin_nontemporal_load(P.m + 12345);
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(y) : "memory");
            const unsigned long long t2 = __builtin_readcyclecounter();
            P.prof[g * 16 + 14] = t1 - t0;
            P.prof[g * 16 + 15] = t2 - t1;
        }
        EP_STAMP();
        // ---- fragments: a wave requests what it consumes two layers later, AFTER its own matrix work of the current layer (the
        //      CU's address path takes ~16 cycles per 1 KB wave-load, ~5,000 cycles for the 211 KB of a step: requested up front
        //      they stall the issue of the first layer's matrix instructions by exactly that; measured). Only layers 0 and 1 are
        //      requested at the top. A single load of a line another workgroup wrote in the last step takes ~240 cycles here.
        EpFwd<0> f0; EpFwd<1> f1; EpFwd<2> f2; EpFwd<3> f3; EpFwd<4> f4;
        EpBwd<4> b4; EpBwd<3> b3; EpBwd<2> b2; EpBwd<1> b1;
        ep_fwd_load<0>(img, wave, lane, f0);
        ep_fwd_load<1>(img, wave, lane, f1);
        // ---- features -> A[0] (column 63 of the padded rows and the rows of samples past nb are zero)
        {
            const int b = tid >> 4;
            const uint64_t nf = c4::next_free_cells(bmy | bop);
            f32x4 x4;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int f = (tid & 15) * 4 + it;
                x4[it] = (b < nb && f < 63) ? c4::feature(bmy, bop, nf, f) : 0.0f;
            }
            *reinterpret_cast<f32x4*>(lds + G::a_off(0) + b * G::stride(0) + (tid & 15) * 4) = x4;
        }
        const float tgt = tgt_next;
        const float ltgt = tgt > 0.0f ? det_logf(tgt) : 0.0f;  // for the KL term of the heads, while the first fragments travel
        lds_barrier();
        EP_STAMP();  // features
        // ---- forward: at most one unit per wave and layer
        ep_fwd_compute<0>(f0, lds, wave, lane);
        ep_fwd_load<2>(img, wave, lane, f2);
        ep_fwd_load<3>(img, wave, lane, f3);
        lds_barrier();
        EP_STAMP();
        ep_fwd_compute<1>(f1, lds, wave, lane);
        ep_fwd_load<4>(img, wave, lane, f4);
        ep_bwd_load<4>(timg, wave, lane, b4);
        ep_bwd_load<3>(timg, wave, lane, b3);
        lds_barrier();
        EP_STAMP();
        ep_fwd_compute<2>(f2, lds, wave, lane);
        ep_bwd_load<2>(timg, wave, lane, b2);
        lds_barrier();
        EP_STAMP();
        ep_fwd_compute<3>(f3, lds, wave, lane);
        ep_bwd_load<1>(timg, wave, lane, b1);
        lds_barrier();
        EP_STAMP();
        ep_fwd_compute<4>(f4, lds, wave, lane);
        if (!last) request_batch(s + 1);  // behind the fragments (loads return in order), most of a step ahead of its use
        lds_barrier();
        EP_STAMP();  // forward
        // ---- heads: log_softmax + kl_div and their gradient, 16 lanes per sample, one entry per lane. The sequential sums of the
        //      reference order (entries ascending) are kept: every lane adds the row's values in that order.
        //      A sample's 12 logits / exponentials / targets / KL terms travel between its 16 lanes as DPP row broadcasts.
        {
            const int b = tid >> 4, jx = tid & 15;
            const bool pol = jx < 9;
            const bool live = jx < 12 && b < nb;
            const float xo = live ? lds[G::a_off(5) + b * G::stride(5) + jx] : 0.0f;
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
            const float sc = (pol ? P.hp.policy_weight : P.hp.value_weight) * bm;
            if (jx < 12) lds[G::d_off(5) + b * G::stride(5) + jx] = live ? sc * (det_expf(xo - lse) * tsum - tgt) : 0.0f;
            if (jx == 0 || jx == 9) lds[G::KL_OFF + b * 2 + (pol ? 0 : 1)] = b < nb ? kl : 0.0f;
        }
        lds_barrier();
        EP_STAMP();  // heads
        // the step's losses: sums over the samples in ascending order, by two lanes of a wave that has no tile in the next phases
        if (g == 0 && (tid >> 1) == 5 * 32) {  // wave 5 has no unit in the next two phases
            const int head = tid & 1;
            float kv[G::CHUNK], acc = 0.0f;
#pragma unroll
            for (int b = 0; b < G::CHUNK; b++) kv[b] = lds[G::KL_OFF + b * 2 + head];  // rows past nb hold +0: acc + 0 = acc
#pragma unroll
            for (int b = 0; b < G::CHUNK; b++) acc += kv[b];
            P.losses[2 * (size_t)s + head] = bm * acc;
        }
        // ---- activation gradients for the inputs of layers 4..1
        ep_bwd_compute<4>(b4, lds, wave, lane);
        lds_barrier();
        EP_STAMP();
        ep_bwd_compute<3>(b3, lds, wave, lane);
        lds_barrier();
        EP_STAMP();
        ep_bwd_compute<2>(b2, lds, wave, lane);
        lds_barrier();
        EP_STAMP();
        ep_bwd_compute<1>(b1, lds, wave, lane);
        lds_barrier();
        EP_STAMP();  // activation gradients
        // ---- this wave's parameters: gradient tile, Adam in registers, new fragments into the other image
        {
            const float ss = P.step_size[s], isb = P.inv_sqrt_bc2[s];
            float* gout = last ? P.grads : nullptr;
            if (job.kind == 0) {
                EP_LAYER_SWITCH(job.L, ep_weight_job<LL>(lds, job.a, job.b, lane, wq, mq, vq, P.hp, ss, isb, img_next, timg_next, gout))
            } else if (job.kind == 1) {
                EP_LAYER_SWITCH(job.L, ep_bias_job<LL>(lds, job.a, lane, nb, wq[0], mq[0], vq[0], P.hp, ss, isb, img_next, gout))
            }
        }
        EP_STAMP();  // parameter jobs
        if (!last) {
            // ---- step barrier: release our tiles, arrive, wait for everybody's, acquire.
            //      one XCD:  stores are write-through to the shared L2 (wait for their acknowledgement), readers drop their vector L1
            //                — the workgroup-scope release/acquire of threadgroup-split mode, where a group spans CUs the same way
            //      else:     device scope (L2 write-back + L2/L1 invalidate: everything read afterwards comes from memory)
            if (one_xcd) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            lds_barrier();
            if (tid == 0) {
                __hip_atomic_fetch_add(P.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ordered by the release above
                const unsigned want = (unsigned)EP_WGS * (unsigned)(s + 1);
                unsigned spins = 0;
                for (;;) {
                    const unsigned have = one_xcd ? __hip_atomic_load(P.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                  : __hip_atomic_load(P.sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (have >= want) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 24) || __hip_atomic_load(P.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        __hip_atomic_store(P.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ep_abort = 1u;
                        break;
                    }
                }
            }
            lds_barrier();
            if (ep_abort) return;  // a workgroup of the launch never arrived (not co-resident): the host reports the failure
            // acquire: drop this CU's vector L1 (sc0 = workgroup scope). ASSUMPTION, documented because the memory model only
            // promises it for threadgroup-split mode: outside that mode `buffer_inv sc0` still empties the TCP on gfx950. The
            // agent-scope form (`sc1`) was measured in round 3: it also invalidates the XCD's L2, every fragment of the next step
            // then comes from memory and the step takes 20.6 us instead of 13.4. What guards the assumption: every build's
            // tests/test_gpu_training.py::test_epoch_kernel_modes_agree compares this mode bit for bit with the device-scope
            // barrier and with the queued launches, and syn_trainer_init runs the same comparison on the box it is called on
            // (epoch_barrier_selfcheck): a mismatch switches the engine to the device-scope barrier.
            if (one_xcd) asm volatile("buffer_inv sc0" ::: "memory");
            else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            EP_STAMP();  // step barrier
        }
    }
#undef EP_STAMP
    // ---- canonical parameters and moments for the host, the two-kernel path and the data-parallel learner
    if (job.kind == 0) {
        EP_LAYER_SWITCH(job.L, ep_tile_canonical<LL>(job.a, job.b, lane0, [&](int r, int p) {
            if (p >= 0) { P.w[p] = wq[r]; P.m[p] = mq[r]; P.v[p] = vq[r]; }
        }))
    } else if (job.kind == 1) {
        const int o = 64 * job.a + lane0;
        if (o < G::D[job.L + 1]) {
            const int p = G::b_off(job.L) + o;
            P.w[p] = wq[0]; P.m[p] = mq[0]; P.v[p] = vq[0];
        }
    }
}

}  // namespace syn

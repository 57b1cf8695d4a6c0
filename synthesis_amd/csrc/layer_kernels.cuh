// synthesis_amd — LDS-tiled versions of the stand-alone slimnn layer kernels: features, Linear and
// Conv2d with the reference's arithmetic — separate multiply and add, the reference's accumulation order per output
// (linear.rs:17-25: inputs ascending; conv.rs:45-85: ci -> k1 -> k2, padded taps skipped) — so they stay bit-identical to
// slimnn's loops and to the element-per-thread kernels of engine_kernels.cuh, which remain the fallback for shapes whose tiles
// do not fit 64 KB of LDS.
//
// What bounds them: slimnn's `out += x * w` is two roundings, so it cannot use fma or the matrix cores — a multiply-add costs
// two VALU instructions, and the layers are VALU-bound long before they are HBM-bound (Linear 63 -> 128: 16 KFLOP against 764
// bytes per sample). The tiled kernels keep operands in LDS / registers so that the VALU is the only thing working:
//   linear_tiled_kernel   W^T and a tile of input rows in LDS; a thread owns one output column for TS samples: per input index
//                         one conflict-free weight read + TS broadcast reads feed 2 TS VALU instructions
//   conv2d_tiled_kernel   the weights and the input planes of SB samples in LDS (coalesced staging), one output element per
//                         thread and step, every tap an LDS read instead of a cached global one
//   features4_kernel      four consecutive features per thread, one 16-byte store
#pragma once
#include "device_common.cuh"

namespace syn {

typedef float lk_f32x2 __attribute__((ext_vector_type(2)));

constexpr int LIN_TS = 8;      // samples per thread
constexpr int LIN_TO = 4;      // outputs per thread: 32 multiply-adds per 3 LDS reads of 16 bytes (an LDS read serves 4 SIMDs' VALUs)
constexpr int LIN_SB = 64;     // samples per workgroup tile
constexpr int LIN_XS = 68;     // LDS row stride of the transposed input tile (16-byte aligned rows, 8-way staging conflicts at most)
__host__ __device__ constexpr int linear_tiled_op(int O) { return (O + 3) & ~3; }
__host__ __device__ constexpr size_t linear_tiled_lds_bytes(int I, int O) {
    return ((size_t)I * linear_tiled_op(O) + (size_t)I * LIN_XS) * 4;
}

// LDS: wt[I][OP] (transposed weights, rows padded to 4 outputs), xs[I][LIN_XS] (transposed input tile)
__global__ __launch_bounds__(256) void linear_tiled_kernel(int I, int O, const float* __restrict__ W, const float* __restrict__ b,
                                                           const float* __restrict__ x, int batch, float* __restrict__ y, int relu) {
    extern __shared__ __attribute__((aligned(16))) float lds_lin[];
    const int OP = linear_tiled_op(O);
    float* wt = lds_lin;
    float* xs = lds_lin + (size_t)I * OP;
    for (int idx = threadIdx.x; idx < I * OP; idx += blockDim.x) {
        const int k = idx / OP, o = idx - k * OP;
        wt[idx] = o < O ? W[(size_t)o * I + k] : 0.0f;
    }
    const int ogroups = OP / LIN_TO, items = (LIN_SB / LIN_TS) * ogroups;  // (sample group, output group) pairs of a tile
    for (long long s0 = (long long)blockIdx.x * LIN_SB; s0 < batch; s0 += (long long)gridDim.x * LIN_SB) {
        const int ns = batch - s0 < LIN_SB ? (int)(batch - s0) : LIN_SB;
        __syncthreads();  // the previous tile's readers are done (and, first time, the weights are staged)
        for (int idx = threadIdx.x; idx < LIN_SB * I; idx += blockDim.x) {
            const int smp = idx / I, k = idx - smp * I;
            xs[k * LIN_XS + smp] = smp < ns ? x[(size_t)s0 * I + idx] : 0.0f;
        }
        __syncthreads();
        for (int it = threadIdx.x; it < items; it += blockDim.x) {
            const int sg = it / ogroups, og = it - sg * ogroups, o0 = LIN_TO * og;
            const float* xr = xs + sg * LIN_TS;
            const float* wr = wt + o0;
            lk_f32x2 lo[LIN_TS], hi[LIN_TS];  // outputs (o0, o0 + 1) and (o0 + 2, o0 + 3) of sample s
            {
                const lk_f32x2 b01 = {b[o0 < O ? o0 : 0], b[o0 + 1 < O ? o0 + 1 : 0]};
                const lk_f32x2 b23 = {b[o0 + 2 < O ? o0 + 2 : 0], b[o0 + 3 < O ? o0 + 3 : 0]};
#pragma unroll
                for (int s = 0; s < LIN_TS; s++) { lo[s] = b01; hi[s] = b23; }
            }
            for (int k = 0; k < I; k++) {
                const float4 w4 = *reinterpret_cast<const float4*>(wr + k * OP);
                const float4 x0 = *reinterpret_cast<const float4*>(xr + k * LIN_XS);
                const float4 x1 = *reinterpret_cast<const float4*>(xr + k * LIN_XS + 4);
                const lk_f32x2 w01 = {w4.x, w4.y}, w23 = {w4.z, w4.w};
                const float xv[LIN_TS] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                // two roundings per term, as slimnn (-ffp-contract=off): packed multiply, then packed add
#pragma unroll
                for (int s = 0; s < LIN_TS; s++) {
                    const lk_f32x2 x2 = {xv[s], xv[s]};
                    lo[s] = lo[s] + x2 * w01;
                    hi[s] = hi[s] + x2 * w23;
                }
            }
#pragma unroll
            for (int s = 0; s < LIN_TS; s++) {
                const int smp = sg * LIN_TS + s;
                if (smp < ns) {
                    float* yo = y + ((size_t)s0 + smp) * O + o0;
                    const float r4[4] = {lo[s][0], lo[s][1], hi[s][0], hi[s][1]};
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (o0 + j < O) yo[j] = relu ? __builtin_fmaxf(r4[j], 0.0f) : r4[j];
                }
            }
        }
    }
}

// LDS: ws[COUT*CIN*K*K], xs[SB][CIN*H_IN*W_IN]. A thread owns one output cell (sample, row, col) for 4 output channels at a time:
// the tap's bounds test, address and input value are shared by the 4 channels (their weights are wave-uniform broadcast reads);
// per output element the terms still arrive in slimnn's order ci -> k1 -> k2.
__global__ __launch_bounds__(256) void conv2d_tiled_kernel(int CIN, int COUT, int K, int RP, int CP, int S, int H_IN, int W_IN,
                                                           int H_OUT, int W_OUT, int SB, const float* __restrict__ W,
                                                           const float* __restrict__ b, const float* __restrict__ x, int batch,
                                                           float* __restrict__ y, int relu) {
    extern __shared__ float lds_conv[];
    const int KK = K * K, nW = COUT * CIN * KK, per_in = CIN * H_IN * W_IN, hw = H_OUT * W_OUT, per_out = COUT * hw;
    const int cgroups = (COUT + 3) / 4;
    float* ws = lds_conv;
    float* xs = lds_conv + nW;
    for (int idx = threadIdx.x; idx < nW; idx += blockDim.x) ws[idx] = W[idx];
    for (long long s0 = (long long)blockIdx.x * SB; s0 < batch; s0 += (long long)gridDim.x * SB) {
        const int ns = batch - s0 < SB ? (int)(batch - s0) : SB;
        __syncthreads();
        for (int idx = threadIdx.x; idx < ns * per_in; idx += blockDim.x) xs[idx] = x[(size_t)s0 * per_in + idx];
        __syncthreads();
        // item = (channel group, sample, cell): consecutive threads take consecutive cells of one sample and channel group
        for (int it = threadIdx.x; it < cgroups * ns * hw; it += blockDim.x) {
            const int cg = it / (ns * hw), rem = it - cg * (ns * hw);
            const int smp = rem / hw, rr = rem - smp * hw;
            const int r = rr / W_OUT, c = rr - r * W_OUT;
            const int co0 = 4 * cg;
            lk_f32x2 a01 = {b[co0], co0 + 1 < COUT ? b[co0 + 1] : 0.0f};
            lk_f32x2 a23 = {co0 + 2 < COUT ? b[co0 + 2] : 0.0f, co0 + 3 < COUT ? b[co0 + 3] : 0.0f};
            const float* xb = xs + smp * per_in;
            const int wstride = CIN * KK;
            const float* wc = ws + co0 * wstride;
            const int w1 = co0 + 1 < COUT ? wstride : 0, w2 = co0 + 2 < COUT ? 2 * wstride : 0, w3 = co0 + 3 < COUT ? 3 * wstride : 0;
            for (int ci = 0; ci < CIN; ci++)
                for (int k1 = 0; k1 < K; k1++) {
                    const int in_row = r * S + k1;
                    if (RP <= in_row && in_row < H_IN + RP)
                        for (int k2 = 0; k2 < K; k2++) {
                            const int in_col = c * S + k2;
                            if (CP <= in_col && in_col < W_IN + CP) {
                                const float v = xb[(ci * H_IN + (in_row - RP)) * W_IN + (in_col - CP)];
                                const float* wt = wc + (ci * K + k1) * K + k2;
                                // (channels past COUT alias channel co0: computed, never stored); packed multiply, packed add
                                const lk_f32x2 v2 = {v, v};
                                a01 = a01 + lk_f32x2{wt[0], wt[w1]} * v2;
                                a23 = a23 + lk_f32x2{wt[w2], wt[w3]} * v2;
                            }
                        }
                }
            const float acc[4] = {a01[0], a01[1], a23[0], a23[1]};
            float* yo = y + ((size_t)s0 + smp) * per_out + (size_t)co0 * hw + rr;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (co0 + j < COUT) yo[(size_t)j * hw] = relu ? __builtin_fmaxf(acc[j], 0.0f) : acc[j];
        }
    }
}

// Game::features (connect4.rs:235-258): out[n][63]. A thread writes 4 consecutive floats of the flat output (one 16-byte store;
// 32-bit index arithmetic: n <= 2^25 positions per call, engine.hip falls back to features_kernel beyond); a group usually
// lies inside one position, whose boards are then loaded once.
__global__ void features4_kernel(const unsigned long long* __restrict__ my_bb, const unsigned long long* __restrict__ op_bb, int n,
                                 float* __restrict__ out) {
    const uint32_t total = (uint32_t)n * 63u, groups = (total + 3u) / 4u;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += gridDim.x * blockDim.x) {
        const uint32_t i0 = 4u * g, pos0 = i0 / 63u, f0 = i0 - pos0 * 63u;
        uint64_t my = my_bb[pos0], op = op_bb[pos0];
        uint64_t nf = c4::next_free_cells(my | op);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t f = f0 + j;
            if (f == 63u && i0 + j < total) {  // the group crosses into the next position
                my = my_bb[pos0 + 1];
                op = op_bb[pos0 + 1];
                nf = c4::next_free_cells(my | op);
            }
            if (f >= 63u) f -= 63u;
            v[j] = c4::feature(my, op, nf, (int)f);
        }
        if (i0 + 3u < total) {
            *reinterpret_cast<float4*>(out + i0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (uint32_t j = 0; j < 4u && i0 + j < total; j++) out[i0 + j] = v[j];
        }
    }
}

}  // namespace syn

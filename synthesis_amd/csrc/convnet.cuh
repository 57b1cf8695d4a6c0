// synthesis_amd — Connect4ConvNet leaf evaluation on the f32 matrix cores: the conv policy/value network BASELINE.json's
// north_star words ("slimnn Conv2d over the 2x9x7 bitplane state + Linear policy/value heads").
//
// The reference defines the layers (slimnn/src/conv.rs:45-85 Conv2d::forward, slimnn/src/linear.rs:17-25 Linear::forward) but no
// such network; the architecture is this build's instantiation, fixed in oracle/nn.hpp (Connect4ConvNet):
//     x[2][7][9] bitplanes (mine, theirs)  ->  Conv2d<2, 16, 3, pad 1, stride 1> + ReLU  ->  Linear<1008, 12>
//     logits = out[0..9] raw, value = softmax(out[9..12])            12,412 parameters; 54,592 FLOP per evaluation as slimnn
//     executes it (in-board taps only), 60,480 as the matrix cores do (padded taps multiply zeros)
// Parameter blob: conv.weight[16][2][3][3], conv.bias[16], head.weight[12][1008], head.bias[12].
//
// MI355X mapping — one wave evaluates a tile of 16 positions, everything in its registers, no im2col buffer, no activations
// in LDS. For every board cell p (63 of them):
//   conv    D[channel][position] = bias + sum over the 18 taps:  A = conv weights (16 channels x 20 taps, taps 18, 19 zero) —
//           five registers per lane for the whole tile; B[tap][position] = ONE BIT of the position's bitboards, taken from five
//           per-lane pre-shifted boards (tap (ci, k1, k2) reads plane ci shifted by (k1 - 1) + 7 (k2 - 1), rows that would wrap
//           into the neighbouring column masked out; shifted once more per board row, so that a cell is a compile-time bit of a
//           32-bit half: v_bfe_u32 + v_cvt per tap) — the "LDS-staged board tile" of north_star is a pair of 64-bit registers
//           here. 5 MFMAs, taps in slimnn's order ci -> k1 -> k2; a padded tap is fma(w, 0, acc) = acc.
//   heads   the conv tile's D registers after ReLU ARE the B operands of the head GEMM for that cell (lane (j, q) register r =
//           channel 4 q + r of position j = k-element q of step r): out[o][position] += Wh[o][channel*63 + p] * act, 4 MFMAs
//           with the A fragments ([p][lane][r], one ds_read_b128 per lane and cell) from the 64.5 KB LDS image.
// 567 v_mfma_f32_16x16x4_f32 per tile (Connect4Net: 476). Every output is one k-ordered fma chain, restated by oracle/nn.hpp's
// ACC_FMA mode bit for bit (head inputs position-major, channels 0,4,8,12, 1,5,9,13, ... inside a position); against slimnn's
// own loop order (ACC_SLIMNN) the outputs agree to ~1e-6 (tests: 1e-5). Measured (MI355X): 1.32 G evaluations/s stand-alone (46 % of
// the f32 MFMA peak in algorithmic FLOP; Connect4Net 1.83 G), 61.0k self-play games/s at 262,144 concurrent games (Connect4Net 72.6k).
#pragma once
#include "mlp.cuh"

namespace syn {

struct ConvGeom {
    static constexpr int C = 16, HW = 63, FLAT = C * HW, OUT = 12;
    static constexpr int CONV_W = C * 2 * 3 * 3;                          // 288
    static constexpr int NUM_PARAMS = CONV_W + C + OUT * FLAT + OUT;      // 12,412
    // LDS / global image (floats)
    static constexpr int HEAD_OFF = 0;                  // [p 63][lane 64][r 4]: Wh[o = lane & 15][(4 (lane >> 4) + r) * 63 + p]
    static constexpr int CONVA_OFF = HW * 256;          // [s 5][lane 64]:       Wc[ch = lane & 15][tap 4 s + (lane >> 4)]
    static constexpr int CBIAS_OFF = CONVA_OFF + 5 * 64;  // [q 4][r 4]:         conv bias of channel 4 q + r
    static constexpr int HBIAS_OFF = CBIAS_OFF + 16;    // [q 4][r 4]:           head bias of output 4 q + r (12..15: zero)
    static constexpr int IMG_FLOATS = HBIAS_OFF + 16;   // 16,480 floats = 65,920 B
    static constexpr int FLOP_PER_EVAL = 2 * (C * HW * 18 + OUT * FLAT);  // 60,480 (padded taps counted: what the MFMAs execute)
};
static_assert(ConvGeom::IMG_FLOATS <= MlpGeom::IMG_FLOATS, "the conv image lives in the LDS region of the Connect4Net image");

// canonical blob -> fragment image (host)
inline void build_conv_image(const float* blob, float* img) {
    using G = ConvGeom;
    for (int i = 0; i < G::IMG_FLOATS; i++) img[i] = 0.0f;
    const float* cw = blob;
    const float* cb = blob + G::CONV_W;
    const float* hw = cb + G::C;
    const float* hb = hw + G::OUT * G::FLAT;
    for (int p = 0; p < G::HW; p++)
        for (int lane = 0; lane < 64; lane++)
            for (int r = 0; r < 4; r++) {
                const int o = lane & 15, ch = 4 * (lane >> 4) + r;
                img[G::HEAD_OFF + (p * 64 + lane) * 4 + r] = o < G::OUT ? hw[o * G::FLAT + ch * G::HW + p] : 0.0f;
            }
    for (int s = 0; s < 5; s++)
        for (int lane = 0; lane < 64; lane++) {
            const int ch = lane & 15, t = 4 * s + (lane >> 4);
            img[G::CONVA_OFF + s * 64 + lane] = t < 18 ? cw[ch * 18 + t] : 0.0f;  // W[ch][ci][k1][k2], t = ci*9 + k1*3 + k2
        }
    for (int q = 0; q < 4; q++)
        for (int r = 0; r < 4; r++) {
            img[G::CBIAS_OFF + q * 4 + r] = cb[4 * q + r];
            img[G::HBIAS_OFF + q * 4 + r] = 4 * q + r < G::OUT ? hb[4 * q + r] : 0.0f;
        }
}

SYN_DEV void stage_conv_image(float* __restrict__ lds_img, const float* __restrict__ g_img, int tid, int nthreads) {
    const f32x4* src = reinterpret_cast<const f32x4*>(g_img);
    f32x4* dst = reinterpret_cast<f32x4*>(lds_img);
    for (int i = tid; i < ConvGeom::IMG_FLOATS / 4; i += nthreads) dst[i] = src[i];
}

// The plane a tap reads, shifted so that bit (row + 7 col) of the result is the tap's input for output cell (row, col), zero
// where the tap falls into the padding. t = ci*9 + k1*3 + k2 (slimnn's loop order); t >= 18: the zero taps that pad k to 20.
SYN_DEV uint64_t conv_tap_board(uint64_t my, uint64_t op, int t) {
    if (t >= 18) return 0ull;
    const int ci = t >= 9 ? 1 : 0, u = t - 9 * ci;
    const int k1 = (u * 11) >> 5, k2 = u - 3 * k1;  // u / 3, u % 3 for u < 9
    const int dr = k1 - 1, dc = k2 - 1, sh = dr + 7 * dc;
    const uint64_t plane = ci ? op : my;
    const uint64_t shifted = sh >= 0 ? plane >> sh : plane << (-sh);
    // a row shift must not pull in the neighbouring column's cells: output row 6 has no row 7 above it, row 0 none below
    const uint64_t rows = dr > 0 ? ~(c4::FAB_ROW << 6) : (dr < 0 ? ~c4::FAB_ROW : ~0ull);
    return shifted & rows & c4::FULL;
}

// Evaluates the network for the 16 positions of this wave's tile. Lane l = (j = l & 15, q = l >> 4) passes the bitboards of
// position j. Returns lane (j, q) register r = raw output 4 q + r of position j (0..8 policy logits, 9..11 outcome logits, 12..15
// zero) — the layout mlp_tile16 returns.
SYN_DEV f32x4 conv_tile16(const float* __restrict__ img, int lane, uint64_t my, uint64_t op) {
    using G = ConvGeom;
    const int q = lane >> 4;
    uint64_t S[5];
    float ca[5];
#pragma unroll
    for (int s = 0; s < 5; s++) {
        S[s] = conv_tap_board(my, op, 4 * s + q);
        ca[s] = img[G::CONVA_OFF + s * 64 + lane];
    }
    const f32x4 cb = *reinterpret_cast<const f32x4*>(img + G::CBIAS_OFF + q * 4);
    f32x4 hacc = *reinterpret_cast<const f32x4*>(img + G::HBIAS_OFF + q * 4);
    const float* hw = img + G::HEAD_OFF + lane * 4;
#pragma unroll 1
    for (int r = 0; r < 7; r++) {
        // this row's cells sit at bits r + 7 c: shift the five boards by r once, then every cell is a compile-time bit of a
        // 32-bit half (bit-field extract + convert per tap)
        uint32_t lo[5], hi[5];
#pragma unroll
        for (int s = 0; s < 5; s++) {
            const uint64_t v = S[s] >> r;
            lo[s] = (uint32_t)v;
            hi[s] = (uint32_t)(v >> 32);
        }
        // Software-pipelined over the row's nine cells: the five conv MFMAs of cell c + 1 are issued BEFORE the ReLU and the four
        // head MFMAs of cell c, so the ReLU never reads an accumulator whose chain has just been issued (the compiler filled that
        // dependency with ~20 wait states of s_nop per cell when the cells ran one after the other) and the two chains — conv of the
        // next cell, heads of this one — interleave on the matrix pipe. Same instructions on the same operands: same bits.
        auto conv_cell = [&](int c) {
            f32x4 acc = cb;
#pragma unroll
            for (int s = 0; s < 5; s++) {
                const uint32_t bit = 7 * c < 32 ? __builtin_amdgcn_ubfe(lo[s], 7 * c, 1) : __builtin_amdgcn_ubfe(hi[s], 7 * c - 32, 1);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[s], (float)bit, acc, 0, 0, 0);
            }
            return acc;
        };
        f32x4 acc = conv_cell(0);
#pragma unroll
        for (int c = 0; c < 9; c++) {
            const int p = r * 9 + c;
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(hw + p * 256);
            f32x4 nxt = acc;
            if (c < 8) nxt = conv_cell(c + 1);
#pragma unroll
            for (int k = 0; k < 4; k++) hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[k], relu1(acc[k]), hacc, 0, 0, 0);
            acc = nxt;
        }
    }
    return hacc;
}

// Batched Policy::eval with Connect4ConvNet: n positions -> logits[n][9], value[n][3] (the stand-alone form of the tile)
template <int NT>
__global__ __launch_bounds__(NT) void policy_eval_conv_kernel(const float* __restrict__ g_img,
                                                              const unsigned long long* __restrict__ my_bb,
                                                              const unsigned long long* __restrict__ op_bb, int n,
                                                              float* __restrict__ logits, float* __restrict__ value) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    stage_conv_image(smem, g_img, tid, NT);
    __syncthreads();
    const int ntiles = (n + 15) >> 4;
    const int j = lane & 15, q = lane >> 4;
    for (int tile = blockIdx.x * (NT / 64) + wave; tile < ntiles; tile += gridDim.x * (NT / 64)) {
        uint32_t img_off = 0;  // opaque per iteration: the fragment reads stay next to their MFMAs (see policy_eval_kernel)
        asm volatile("" : "+v"(img_off));
        const int pos = tile * 16 + j;
        const bool valid = pos < n;
        const uint64_t my = valid ? my_bb[pos] : 0ull, op = valid ? op_bb[pos] : 0ull;
        f32x4 o = conv_tile16(smem + img_off, lane, my, op);
        if (valid) {
            if (q < 2) {
#pragma unroll
                for (int r = 0; r < 4; r++) logits[(size_t)pos * 9 + q * 4 + r] = o[r];
            } else if (q == 2) {
                logits[(size_t)pos * 9 + 8] = o[0];
                float v0 = o[1], v1 = o[2], v2 = o[3];
                value_softmax(v0, v1, v2);
                value[(size_t)pos * 3 + 0] = v0;
                value[(size_t)pos * 3 + 1] = v1;
                value[(size_t)pos * 3 + 2] = v2;
            }
        }
    }
}

}  // namespace syn

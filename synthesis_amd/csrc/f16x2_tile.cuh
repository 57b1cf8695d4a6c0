// synthesis_amd — Connect4Net leaf evaluation on the f16 matrix cores with a TWO-TERM SPLIT of every operand ("f16x2").
//
// Same network as mlp.cuh (study-connect4/src/policies.rs:28-59, layers slimnn/src/linear.rs:17-25), other arithmetic:
//   w * 2^t = w_hi + w_lo,   x * 2^s = x_hi + x_lo      (f16 each; hi = RNE(value), lo = RNE(value - hi): 22-23 significand bits)
//   acc = bias * 2^(s+t);  per block of 32 inputs:  acc += W_hi.x_hi;  acc += W_hi.x_lo;  acc += W_lo.x_hi   (f32 accumulator)
// on v_mfma_f32_16x16x32_f16 — 16 matrix-pipe cycles per 32 inputs and 16 outputs x 3 products instead of 8 x 32 cycles on
// v_mfma_f32_16x16x4_f32 (5.3x fewer matrix cycles), and the instruction leaves the SIMD's vector issue port free for half of
// its cycles.  Scales are exact powers of two chosen per checkpoint at load time (f16x2_plan) so that no activation can exceed the
// f16 range (a bound from the weights, not a calibration) and the low terms stay normal numbers.
//
// Fragment layout (v_mfma_f32_16x16x32_f16: lane l holds A[i = l&15][k = 8(l>>4) + jj], B[k = 8(l>>4) + jj][j = l&15], jj = 0..7;
// D[4(l>>4) + r][l&15]): A = weights (natural unit order: row i of output block ob = unit 16 ob + i), B = activations, D[unit][position].
// The D registers of two neighbouring output blocks of layer L ARE one lane's eight B elements of a 32-input block of layer L+1:
// slot jj < 4 = block 2kb register jj, jj >= 4 = block 2kb+1 register jj-4, i.e. input unit(kb, q, jj) = 32 kb + 16 (jj>>2) + 4 q + (jj&3);
// the weight image is laid out for exactly that, so the whole tile runs in one wave's registers with no cross-lane movement.
//
// The arithmetic is a definition of its own ("ACC_F16X2" in oracle/nn_f16x2.hpp restates it on the CPU); it is NOT bit-identical to
// the f32 path (mlp.cuh / ACC_FMA) — the two agree to f32 rounding noise (profiles/r05_f16_split.txt).
#pragma once
#include "device_common.cuh"

namespace syn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct F16Geom {
    static constexpr int NL = 5;
    static constexpr int K[NL] = {63, 128, 96, 64, 48};
    static constexpr int O[NL] = {128, 96, 64, 48, 12};
    static constexpr int NKB[NL] = {2, 4, 3, 2, 2};          // 32-input blocks (inputs padded with zero weights)
    static constexpr int NOB[NL] = {8, 6, 4, 3, 1};          // 16-unit output blocks
    // half offsets of each layer inside ONE part (hi or lo) of the image: [kb][ob][lane 0..63][jj 0..7]
    static constexpr int H_OFF[NL + 1] = {0, 8192, 20480, 26624, 29696, 30720};
    static constexpr int PART_HALFS = 30720;
    static constexpr int PART_WORDS = PART_HALFS / 2;         // 15,360
    // f32 bias image [layer][ob][q][r] (unit 16 ob + 4 q + r), already multiplied by 2^(s+t)
    static constexpr int B_OFF[NL + 1] = {0, 128, 224, 288, 336, 352};
    static constexpr int BIAS_WORD0 = 2 * PART_WORDS;         // 30,720
    static constexpr int SCALE_WORD0 = BIAS_WORD0 + 352;      // f32: 2^cexp[0..3], then 2^out_exp, 3 unused
    static constexpr int IMG_WORDS = SCALE_WORD0 + 8;         // 31,080 words = 124,320 B of LDS
    static constexpr int FEATURE_EXP = 8;                     // features enter as x * 2^8: 256 and 25.6 (+ its low term)
    static constexpr float ACT_MAX = 65504.0f;                // largest finite f16
};

// input unit of layer-(L+1) slot (kb, q, jj), L >= 1
__host__ __device__ constexpr int f16x2_unit_of_slot(int kb, int q, int jj) { return 32 * kb + 16 * (jj >> 2) + 4 * q + (jj & 3); }
// Layer 1: lane q's sixteen inputs are sixteen CONSECUTIVE board bits p = 16 q + 8 kb + jj (bit p = row + 7 col, connect4.rs:108-114),
// so the lane's feature bits are two 16-bit fields of the boards; the flat feature index (connect4.rs:235-258) is row * 9 + col.
// p = 63 is no cell: -1 (padding; its weights are zero).
__host__ __device__ constexpr int f16x2_feature_of_slot(int kb, int q, int jj) {
    const int p = 16 * q + 8 * kb + jj;
    return p < 63 ? (p % 7) * 9 + p / 7 : -1;
}

// ---- layer-1 B operands from the two feature boards (mlp.cuh feature_boards: hi = occupied, lo = mine | next-free) -----------------
// cell values (connect4.rs:235-258) times 2^8: (hi,lo) = (1,1) +256 | (1,0) -256 | (0,1) +25.6 | (0,0) -25.6, the 25.6 as an f16 pair
constexpr uint32_t F16_256 = 0x5C00;       // 256.0
constexpr uint32_t F16_25p6_HI = 0x4E66;   // RNE_f16(0.1f * 256) = 25.59375
constexpr uint32_t F16_25p6_LO = 0x1E66;   // RNE_f16(0.1f * 256 - 25.59375) = 1638 * 2^-18 (the tie goes to even)
// Two features per 32-bit word. H, NL: this lane's 16 occupancy bits / 16 inverted "positive" bits. For the pair (m, m+1):
// t = two bits -> one per half (t * 0x8001 & 0x10001), then   hi word = 25.6|25.6 + occupied * (256 - 25.6) | negative << 15,
// lo word = free * lo(25.6) | (free & negative) << 15   — integer arithmetic on the halves, no carries across them.
template <int M>
SYN_DEV void f16_feature_pair(uint32_t H, uint32_t NL, uint32_t& wh, uint32_t& wl) {
    const uint32_t h2 = __builtin_amdgcn_ubfe(H, M, 2), n2 = __builtin_amdgcn_ubfe(NL, M, 2);
    const uint32_t hs = (h2 * 0x8001u) & 0x10001u, ns = (n2 * 0x8001u) & 0x10001u;     // bit 0 / bit 16
    const uint32_t fs = hs ^ 0x10001u;                                                  // free cells
    wh = (hs * (F16_256 - F16_25p6_HI) + (F16_25p6_HI * 0x10001u)) | (ns << 15);
    wl = (fs * F16_25p6_LO) | ((ns & fs) << 15);
}
template <int KB>
SYN_DEV void f16_feature_block(uint32_t H, uint32_t NL, u32x4& bh, u32x4& bl) {
    uint32_t h, l;
    f16_feature_pair<8 * KB + 0>(H, NL, h, l); bh[0] = h; bl[0] = l;
    f16_feature_pair<8 * KB + 2>(H, NL, h, l); bh[1] = h; bl[1] = l;
    f16_feature_pair<8 * KB + 4>(H, NL, h, l); bh[2] = h; bl[2] = l;
    f16_feature_pair<8 * KB + 6>(H, NL, h, l); bh[3] = h; bl[3] = l;
}
// this lane's two bit fields: H = occupied, NL = NOT positive (positive = mine or the lowest free cell of a column)
SYN_DEV void f16_feature_fields(uint64_t hi, uint64_t lo, int q, uint32_t& H, uint32_t& NL) {
    H = (uint32_t)(hi >> (16 * q)) & 0xFFFFu;
    NL = ~(uint32_t)(lo >> (16 * q)) & 0xFFFFu;
}

SYN_DEV f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }

// ReLU + rescale + split of one D block (4 units of one position per lane): s = med3(acc * c, 0, 65504) (NaN -> 0 like x.max(0.0);
// the upper clamp cannot bind — f16x2_plan bounds every activation below 2^15 — it only keeps a violated bound finite);
// hi = RNE_f16(s), lo = RNE_f16(s - hi) (the subtraction is exact in f32; as fma(hi, -1, s) it is one v_fma_mix_f32).
// s - (f32)half of a packed f16 pair as ONE v_fma_mix_f32 (fma(half, -1, s): exact, the same bits as the subtraction the compiler
// would otherwise build from two conversions and a packed add)
template <int HALF>
SYN_DEV float f16x2_residual(uint32_t hpk, float s) {
    float r;
    if (HALF == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(s));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(s));
    return r;
}
SYN_DEV void f16x2_split_block(f32x4_ v, float c, uint32_t& h01, uint32_t& h23, uint32_t& l01, uint32_t& l23) {
    float s[4];
#pragma unroll
    for (int r = 0; r < 4; r++) s[r] = __builtin_amdgcn_fmed3f(v[r] * c, 0.0f, F16Geom::ACT_MAX);
    h01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{s[0], s[1]}, f16x2));
    h23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{s[2], s[3]}, f16x2));
    const f32x2_ ra = f32x2_{f16x2_residual<0>(h01, s[0]), f16x2_residual<1>(h01, s[1])};
    const f32x2_ rb = f32x2_{f16x2_residual<0>(h23, s[2]), f16x2_residual<1>(h23, s[3])};
    l01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(ra, f16x2));
    l23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(rb, f16x2));
}

// ---- the tile, software-pipelined in 32 groups --------------------------------------------------------------------------------------
// A group = (layer, a part of at most two output blocks, one 32-input block): 3 products x nb MFMAs on nb independent accumulators.
// While a group's MFMAs run, the weight fragments of the next group are on their way from LDS (double buffer), and the activations
// of output blocks that finished EARLIER are rescaled / split for the next layer, one block (16 vector instructions) per group, in
// five stages placed behind successive MFMAs: every stage's inputs are a whole MFMA old, and the two or three vector instructions per
// MFMA fit the half of its 16 cycles in which the instruction leaves the vector issue port free (MI355X guide: an MFMA of this shape
// holds the port for 8 of its 16 cycles). `sched_barrier`s pin that order. The schedule (which block is split in which group) is
// fixed by f16_split_of_group(): a block is split after its part has finished and before the first group that reads it.
struct F16Group { int layer, nob, ob0, nb, kb, nkb; };
constexpr F16Group f16_group(int gi) {
    if (gi < 8) return {0, 8, (gi / 2) * 2, 2, gi % 2, 2};                      // L1: 4 parts x 2 input blocks
    if (gi < 20) return {1, 6, ((gi - 8) / 4) * 2, 2, (gi - 8) % 4, 4};         // L2: 3 parts x 4
    if (gi < 26) return {2, 4, ((gi - 20) / 3) * 2, 2, (gi - 20) % 3, 3};       // L3: 2 parts x 3
    if (gi < 28) return {3, 3, 0, 2, gi - 26, 2};                               // L4: blocks 0-1 x 2
    if (gi < 30) return {3, 3, 2, 1, gi - 28, 2};                               // L4: block 2 x 2
    return {4, 1, 0, 1, gi - 30, 2};                                            // L5: 1 block x 2
}
constexpr int F16_GROUPS = 32;
// output blocks split in group gi: count n and (layer, first block); blocks are first, first + 1, ...
struct F16Split { int n, layer, blk; };
constexpr F16Split f16_split_of_group(int gi) {
    if (gi >= 2 && gi <= 9) return {1, 0, gi - 2};       // L1 blocks 0-7 (parts finish in groups 1, 3, 5, 7; L2 reads from group 8 on)
    if (gi == 12 || gi == 13) return {1, 1, gi - 12};    // L2 blocks 0-1 (finish in 11)
    if (gi == 16 || gi == 17) return {1, 1, gi - 14};    // L2 blocks 2-3 (finish in 15)
    if (gi == 20 || gi == 21) return {1, 1, gi - 16};    // L2 blocks 4-5 (finish in 19; L3's third input block is read in 22)
    if (gi == 23 || gi == 24) return {1, 2, gi - 23};    // L3 blocks 0-1 (finish in 22)
    if (gi == 26) return {2, 2, 2};                      // L3 blocks 2-3 (finish in 25; L4's second input block is read in 27)
    if (gi == 28 || gi == 29) return {1, 3, gi - 28};    // L4 blocks 0-1 (finish in 27)
    if (gi == 30) return {1, 3, 2};                      // L4 block 2 (finishes in 29; L5's second input block is read in 31)
    return {0, 0, 0};
}

template <int PF>
struct F16Regs {
    u32x4 ah[PF + 1][2], al[PF + 1][2];   // weight fragments (hi, lo) of up to 2 blocks: the group in flight + PF groups ahead
    f32x4_ bb[PF + 1][2];          // biases of a part's first group, fetched with its fragments
    f32x4_ acc[2];                 // accumulators of the current part
    f32x4_ pend[4];                // finished accumulators waiting for their split: block b of a layer sits in pend[b & 3]
    f32x4_ sp[2];                  // the split in flight (up to two blocks): scaled / clamped values, then residuals
    uint32_t sh[2][2];             // ... and their hi halves
    float cs[4];                   // the four rescale factors (wave-uniform: scalar registers)
    u32x4 x1h[2], x1l[2], x2h[4], x2l[4], x3h[3], x3l[3], x4h[2], x4l[2], x5h[2], x5l[2];   // B operands per layer (hi, lo) per input block
};

template <int GI, int PF>
SYN_DEV void f16_prefetch(const uint32_t* __restrict__ img, int lane, F16Regs<PF>& R) {
    if constexpr (GI < F16_GROUPS) {
        constexpr F16Group G = f16_group(GI);
        const u32x4* whi = reinterpret_cast<const u32x4*>(img + F16Geom::H_OFF[G.layer] / 2) + lane;
        const u32x4* wlo = reinterpret_cast<const u32x4*>(img + F16Geom::PART_WORDS + F16Geom::H_OFF[G.layer] / 2) + lane;
#pragma unroll
        for (int ob = 0; ob < G.nb; ob++) {
            R.ah[GI % (PF + 1)][ob] = whi[(G.kb * G.nob + G.ob0 + ob) * 64];
            R.al[GI % (PF + 1)][ob] = wlo[(G.kb * G.nob + G.ob0 + ob) * 64];
        }
        if constexpr (G.kb == 0) {
            const float* bimg = reinterpret_cast<const float*>(img + F16Geom::BIAS_WORD0);
            const int q = lane >> 4;
#pragma unroll
            for (int ob = 0; ob < G.nb; ob++)
                R.bb[GI % (PF + 1)][ob] = *reinterpret_cast<const f32x4_*>(bimg + F16Geom::B_OFF[G.layer] + ((G.ob0 + ob) * 4 + q) * 4);
        }
    }
}

// stage ST (0..4) of the splits scheduled in group GI (f16x2_split_block cut into its five dependent steps)
template <int GI, int ST, int PF>
SYN_DEV void f16_split_stage(F16Regs<PF>& R) {
    constexpr F16Split S = f16_split_of_group(GI);
    if constexpr (S.n > 0) {
#pragma unroll
        for (int i = 0; i < S.n; i++) {
            const int blk = S.blk + i;
            if constexpr (ST == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) R.sp[i][r] = R.pend[blk & 3][r] * R.cs[S.layer];
            } else if constexpr (ST == 1) {
#pragma unroll
                for (int r = 0; r < 4; r++) R.sp[i][r] = __builtin_amdgcn_fmed3f(R.sp[i][r], 0.0f, F16Geom::ACT_MAX);
            } else if constexpr (ST == 2) {
                R.sh[i][0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{R.sp[i][0], R.sp[i][1]}, f16x2));
                R.sh[i][1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{R.sp[i][2], R.sp[i][3]}, f16x2));
            } else if constexpr (ST == 3) {
                R.sp[i][0] = f16x2_residual<0>(R.sh[i][0], R.sp[i][0]);
                R.sp[i][1] = f16x2_residual<1>(R.sh[i][0], R.sp[i][1]);
                R.sp[i][2] = f16x2_residual<0>(R.sh[i][1], R.sp[i][2]);
                R.sp[i][3] = f16x2_residual<1>(R.sh[i][1], R.sp[i][3]);
            } else {
                const uint32_t l01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{R.sp[i][0], R.sp[i][1]}, f16x2));
                const uint32_t l23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{R.sp[i][2], R.sp[i][3]}, f16x2));
                const int kb = blk >> 1, at = 2 * (blk & 1);
                u32x4* xh = S.layer == 0 ? R.x2h : S.layer == 1 ? R.x3h : S.layer == 2 ? R.x4h : R.x5h;
                u32x4* xl = S.layer == 0 ? R.x2l : S.layer == 1 ? R.x3l : S.layer == 2 ? R.x4l : R.x5l;
                xh[kb][at] = R.sh[i][0]; xh[kb][at + 1] = R.sh[i][1];
                xl[kb][at] = l01; xl[kb][at + 1] = l23;
            }
        }
    }
}

template <int GI, int NPROD, int PF>
SYN_DEV void f16_group_run(const uint32_t* __restrict__ img, int lane, F16Regs<PF>& R) {
    constexpr F16Group G = f16_group(GI);
    f16_prefetch<GI + PF, PF>(img, lane, R);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G.kb == 0) {
#pragma unroll
        for (int ob = 0; ob < G.nb; ob++) R.acc[ob] = R.bb[GI % (PF + 1)][ob];
    }
    const u32x4* xh = G.layer == 0 ? R.x1h : G.layer == 1 ? R.x2h : G.layer == 2 ? R.x3h : G.layer == 3 ? R.x4h : R.x5h;
    const u32x4* xl = G.layer == 0 ? R.x1l : G.layer == 1 ? R.x2l : G.layer == 2 ? R.x3l : G.layer == 3 ? R.x4l : R.x5l;
    const f16x8 bh = as_f16x8(xh[G.kb]), bl = as_f16x8(xl[G.kb]);
    // MFMA m of the group: product m / nb (hi.hi, hi.lo, lo.hi[, lo.lo]) on accumulator m % nb; the five split stages follow MFMAs
    // 0 .. 4 of a six-MFMA group (two stages per MFMA in the three-MFMA groups of the last layers)
    constexpr int NM = NPROD * G.nb;
    constexpr int PER = NM >= 5 ? 1 : 2;
#pragma unroll
    for (int m = 0; m < NM; m++) {
        const int prod = m / G.nb, ob = m % G.nb;
        const f16x8 a = as_f16x8(prod == 0 || prod == 1 ? R.ah[GI % (PF + 1)][ob] : R.al[GI % (PF + 1)][ob]);
        const f16x8 x = (prod == 0 || prod == 2) ? bh : bl;
        R.acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, x, R.acc[ob], 0, 0, 0);
        if (PER == 1) {
            if (m == 0) f16_split_stage<GI, 0, PF>(R);
            if (m == 1) f16_split_stage<GI, 1, PF>(R);
            if (m == 2) f16_split_stage<GI, 2, PF>(R);
            if (m == 3) f16_split_stage<GI, 3, PF>(R);
            if (m == 4) f16_split_stage<GI, 4, PF>(R);
        } else {
            if (m == 0) { f16_split_stage<GI, 0, PF>(R); f16_split_stage<GI, 1, PF>(R); }
            if (m == 1) { f16_split_stage<GI, 2, PF>(R); f16_split_stage<GI, 3, PF>(R); }
            if (m == 2) f16_split_stage<GI, 4, PF>(R);
        }
        if (m + 1 < NM) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G.kb == G.nkb - 1 && G.layer < 4) {
#pragma unroll
        for (int ob = 0; ob < G.nb; ob++) R.pend[(G.ob0 + ob) & 3] = R.acc[ob];
    }
    if constexpr (GI + 1 < F16_GROUPS) f16_group_run<GI + 1, NPROD, PF>(img, lane, R);
}

// Evaluates the network for the 16 positions of this wave's tile. Lane l = (j = l&15, q = l>>4) passes the feature boards of
// position j (mlp.cuh feature_boards). Returns the last layer's D registers: lane (j,q) register r = raw output 4q + r of
// position j DIVIDED by the image's out scale (multiply by the f32 at word SCALE_WORD0 + 4: an exact power of two).
// PF = how many groups ahead the weight fragments are requested: 1 under the 128-register budget of the 16-wave shapes (other waves
// cover the LDS latency), 2 for a wave that is alone on its SIMD (free_kernel.cuh: +16 registers, -10 % cycles per tile).
template <int NPROD, int PF = 1>
SYN_DEV f32x4_ f16x2_tile16(const uint32_t* __restrict__ img, int lane, uint64_t hi, uint64_t lo) {
    F16Regs<PF> R;
    {   // read once, ahead of everything: a later LDS read of them would wait for the weight fragments in flight behind it
        const f32x4_ c4 = *reinterpret_cast<const f32x4_*>(img + F16Geom::SCALE_WORD0);
#pragma unroll
        for (int l = 0; l < 4; l++) R.cs[l] = bits_f32((uint32_t)__builtin_amdgcn_readfirstlane((int)f32_bits(c4[l])));
    }
    f16_prefetch<0, PF>(img, lane, R);
    if constexpr (PF >= 2) f16_prefetch<1, PF>(img, lane, R);
    uint32_t H, NL;
    f16_feature_fields(hi, lo, lane >> 4, H, NL);
    f16_feature_block<0>(H, NL, R.x1h[0], R.x1l[0]);
    f16_feature_block<1>(H, NL, R.x1h[1], R.x1l[1]);
    // the unused upper half of an odd layer's last input block (L5 reads output blocks 0-2 of L4): zeros against zero weights
    R.x5h[1][2] = 0; R.x5h[1][3] = 0; R.x5l[1][2] = 0; R.x5l[1][3] = 0;
    f16_group_run<0, NPROD, PF>(img, lane, R);
    return R.acc[0];
}

}  // namespace syn

// ====================================================================================================================================
// Host side: the per-checkpoint plan (power-of-two scales) and the LDS image.  Plain C++, no device code.
#include <cmath>
#include <cstring>
#include <vector>

namespace syn {

struct F16Image {
    int s[5];           // activation scale exponent entering layer l (s[0] = FEATURE_EXP)
    int t[5];           // weight scale exponent of layer l
    int cexp[4];        // rescale after layer l: x'_{l+1} = relu(acc * 2^cexp[l])
    int out_exp;        // raw outputs = acc_5 * 2^out_exp
    double bound[5];    // upper bound of layer l's activations (outputs for l = 4), unscaled
    std::vector<uint32_t> words;   // F16Geom::IMG_WORDS
};

// smallest e with 2^e >= v (v > 0)
inline int f16x2_ceil_log2(double v) {
    int e;
    const double m = std::frexp(v, &e);   // v = m 2^e, 0.5 <= m < 1
    return m == 0.5 ? e - 1 : e;
}
inline uint16_t f16x2_bits(float x) { const _Float16 h = (_Float16)x; uint16_t u; std::memcpy(&u, &h, 2); return u; }
inline float f16x2_value(uint16_t u) { _Float16 h; std::memcpy(&h, &u, 2); return (float)h; }

// blob order: l_k.weight[O][I] then l_k.bias[O], k = 1..5 (study-connect4/src/policies.rs:20-24).
// Returns false when the checkpoint cannot be represented (non-finite parameters or scales outside the f32-safe window).
inline bool build_f16x2_image(const float* blob, F16Image& im) {
    im.words.assign(F16Geom::IMG_WORDS, 0u);
    uint16_t* halfs = reinterpret_cast<uint16_t*>(im.words.data());
    float* bimg = reinterpret_cast<float*>(im.words.data() + F16Geom::BIAS_WORD0);
    float* sc = reinterpret_cast<float*>(im.words.data() + F16Geom::SCALE_WORD0);
    std::vector<double> ub(63, 1.0), nb;
    size_t off = 0;
    im.s[0] = F16Geom::FEATURE_EXP;
    for (int l = 0; l < F16Geom::NL; l++) {
        const int K = F16Geom::K[l], O = F16Geom::O[l], NKB = F16Geom::NKB[l], NOB = F16Geom::NOB[l];
        const float* W = blob + off;
        const float* b = W + (size_t)K * O;
        off += (size_t)K * O + O;
        double wmax = 0;
        for (size_t i = 0; i < (size_t)K * O; i++) { if (!std::isfinite(W[i])) return false; wmax = std::fmax(wmax, std::fabs((double)W[i])); }
        for (int o = 0; o < O; o++) if (!std::isfinite(b[o])) return false;
        im.t[l] = wmax > 0 ? 14 - f16x2_ceil_log2(wmax) : 0;
        if (im.t[l] > 40) im.t[l] = 40;
        // bound of this layer's outputs from the bound of its inputs (inputs of layer 1: |x| <= 1; later layers: 0 <= x <= ub)
        nb.assign(O, 0.0);
        double B = 0;
        for (int o = 0; o < O; o++) {
            double acc = (double)b[o];
            for (int i = 0; i < K; i++) {
                const double w = (double)W[(size_t)o * K + i];
                acc += (l == 0 ? std::fabs(w) : (w > 0 ? w : 0.0)) * ub[i];
            }
            nb[o] = acc > 0 ? acc : 0.0;
            B = std::fmax(B, l == 4 ? std::fabs(acc) : nb[o]);
        }
        im.bound[l] = B;
        ub = nb;
        const int e_acc = im.s[l] + im.t[l];                 // acc = true pre-activation * 2^e_acc
        if (e_acc < -60 || e_acc > 60) return false;
        if (l < 4) {
            im.s[l + 1] = 15 - f16x2_ceil_log2(std::fmax(B, 1e-30));
            if (im.s[l + 1] > 24) im.s[l + 1] = 24;
            im.cexp[l] = im.s[l + 1] - e_acc;
            sc[l] = std::ldexp(1.0f, im.cexp[l]);
        } else {
            im.out_exp = -e_acc;
            sc[4] = std::ldexp(1.0f, im.out_exp);
        }
        for (int kb = 0; kb < NKB; kb++)
            for (int ob = 0; ob < NOB; ob++)
                for (int lane = 0; lane < 64; lane++)
                    for (int jj = 0; jj < 8; jj++) {
                        const int i = lane & 15, q = lane >> 4, unit = 16 * ob + i;
                        const int k = l == 0 ? f16x2_feature_of_slot(kb, q, jj) : f16x2_unit_of_slot(kb, q, jj);
                        uint16_t hi = 0, lo = 0;
                        if (unit < O && k >= 0 && k < K) {
                            const float ws = std::ldexp(W[(size_t)unit * K + k], im.t[l]);
                            hi = f16x2_bits(ws);
                            lo = f16x2_bits(ws - f16x2_value(hi));
                        }
                        const size_t at = (size_t)F16Geom::H_OFF[l] + ((size_t)(kb * NOB + ob) * 64 + lane) * 8 + jj;
                        halfs[at] = hi;
                        halfs[(size_t)F16Geom::PART_HALFS + at] = lo;
                    }
        for (int ob = 0; ob < NOB; ob++)
            for (int u = 0; u < 16; u++) {
                const int unit = 16 * ob + u;
                bimg[F16Geom::B_OFF[l] + ob * 16 + u] = unit < O ? std::ldexp(b[unit], e_acc) : 0.0f;
            }
    }
    return true;
}

}  // namespace syn

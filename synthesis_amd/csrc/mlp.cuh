// synthesis_amd — Connect4Net leaf evaluation on the f32 matrix cores.
//
// Replaces study-connect4/src/policies.rs:28-59 (forward + eval): features -> 63->128->96->64->48->12 MLP (ReLU) ->
// 9 raw policy logits + softmax over 3 outcome logits, for a tile of 16 positions per wave.
//
// MI355X mapping (v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[(l>>4)*4+r][l&15]):
//   * A = weights, B = activations, D[unit][position].  The whole network for one 16-position tile runs in ONE wave's
//     registers: the D registers of layer L (after ReLU) ARE the B operands of layer L+1, no LDS round trip and no
//     cross-lane movement.  That works because the weight image is built so that A-row i of output block ob holds
//     unit 16*ob + 4*(i&3) + (i>>2): D register r of block ob on lane (j,q) is then unit 16*ob + 4*r + q, exactly the
//     k = 4*s + q element that step s = 4*ob + r of the next layer consumes.
//   * f32-input MFMA is a k-ordered fused-multiply-add chain (one rounding per term, MI355X guide §3), so every output
//     equals  fma(x[K-1],w[K-1], ... fma(x[1],w[1], fma(x[0],w[0], bias)))  bit for bit — the oracle's ACC_FMA mode.
//   * The 30,144 weights + 348 biases live in LDS for the lifetime of the workgroup in "fragment order": one
//     ds_read_b128 per lane fetches the A operands of 4 consecutive k-steps, conflict-free (lane-linear 16 B).
#pragma once
#include "device_common.cuh"

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier for waves that exchange data through LDS only: waits for this wave's LDS traffic, not for its
// global stores (`__syncthreads()` also drains vmcnt, i.e. every node-pool store still in flight).
SYN_DEV void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct MlpGeom {
    static constexpr int NL = 5;
    static constexpr int K[NL] = {63, 128, 96, 64, 48};      // inputs per layer
    static constexpr int O[NL] = {128, 96, 64, 48, 12};      // outputs per layer
    static constexpr int S4[NL] = {4, 8, 6, 4, 3};           // groups of 4 k-steps (k padded to 16*S4)
    static constexpr int NOB[NL] = {8, 6, 4, 3, 1};          // 16-unit output blocks (O padded to 16*NOB)
    // float offsets of each layer's fragment image: [s4][ob][lane 0..63][r 0..3]
    static constexpr int W_OFF[NL + 1] = {0, 8192, 20480, 26624, 29696, 30464};
    // bias image: [layer][ob][q 0..3][r 0..3]
    static constexpr int B_OFF[NL + 1] = {0, 128, 224, 288, 336, 352};
    static constexpr int W_FLOATS = 30464;
    static constexpr int B_FLOATS = 352;
    static constexpr int IMG_FLOATS = W_FLOATS + B_FLOATS;   // 30,816 floats = 123,264 B of LDS
    static constexpr int NUM_PARAMS = 30492;
};

// Which network unit sits on A-row i of output block ob (see header comment). Last layer: natural order.
__host__ __device__ inline int mlp_unit_of_row(int layer, int ob, int i) {
    return layer == MlpGeom::NL - 1 ? 16 * ob + i : 16 * ob + 4 * (i & 3) + (i >> 2);
}

// ---- layer-1 B operands straight from the bitboards ------------------------------------------------------------------
// Lane (j,q) consumes features f = 4*m + q, m = 0..15 (k = 63 is padding: its weight column is zero, so any finite
// value is fine). Cell values (connect4.rs:235-258) from two derived boards: hi = occupied, lo = mine | next-free:
//   (hi,lo) = (1,1) mine +1.0 | (1,0) theirs -1.0 | (0,1) lowest empty cell +0.1 | (0,0) empty -0.1
// i.e. magnitude = hi ? 1.0 : 0.1, sign = lo ? + : -. Per feature: two 64-bit shifts by a per-lane bit position taken
// from a 16-entry table packed into 4 registers (built once per kernel), instead of ~25 instructions of div/mod/test.
struct FeatureTable { uint32_t t[4]; };
SYN_DEV FeatureTable make_feature_table(int q) {
    FeatureTable T;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        uint32_t packed = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int f = 4 * (4 * w + b) + q;
            int row = f / 9, col = f - 9 * row;
            uint32_t pos = f < 63 ? (uint32_t)(row + 7 * col) : 63u;
            packed |= pos << (8 * b);
        }
        T.t[w] = packed;
    }
    return T;
}
SYN_DEV void feature_boards(uint64_t my, uint64_t op, uint64_t& hi, uint64_t& lo) {
    hi = my | op;
    lo = my | c4::next_free_cells(hi);
}
template <int M>
SYN_DEV float feature_from_boards(const FeatureTable& T, uint64_t hi, uint64_t lo) {
    uint32_t pos = (T.t[M >> 2] >> (8 * (M & 3))) & 0xFFu;
    uint32_t h = (uint32_t)(hi >> pos) & 1u, l = (uint32_t)(lo >> pos) & 1u;
    return bits_f32((h ? 0x3F800000u : 0x3DCCCCCDu) | ((l ^ 1u) << 31));
}
// runtime-indexed form: `packed` = FT.t[w] (4 positions), b = 0..3
SYN_DEV float feature_from_boards_rt(uint32_t packed, int b, uint64_t hi, uint64_t lo) {
    uint32_t pos = (packed >> (8 * b)) & 0xFFu;
    uint32_t h = (uint32_t)(hi >> pos) & 1u, l = (uint32_t)(lo >> pos) & 1u;
    return bits_f32((h ? 0x3F800000u : 0x3DCCCCCDu) | ((l ^ 1u) << 31));
}
template <int S4>
SYN_DEV f32x4 feature_quad(const FeatureTable& T, uint64_t hi, uint64_t lo) {
    f32x4 b;
    b[0] = feature_from_boards<4 * S4 + 0>(T, hi, lo);
    b[1] = feature_from_boards<4 * S4 + 1>(T, hi, lo);
    b[2] = feature_from_boards<4 * S4 + 2>(T, hi, lo);
    b[3] = feature_from_boards<4 * S4 + 3>(T, hi, lo);
    return b;
}

// One layer: acc[ob] (NOB blocks) = bias; for every group of 4 k-steps: 4*NOB MFMAs, interleaved over the blocks
// so consecutive MFMAs never depend on each other (dependent latency 40 cycles > 32-cycle issue).
template <int LAYER, int NOB, int S4, class BOperand>
SYN_DEV void mlp_layer(const float* __restrict__ wimg, const float* __restrict__ bimg, int lane, BOperand bop,
                       f32x4 (&acc)[NOB]) {
    const int q = lane >> 4;
#pragma unroll
    for (int ob = 0; ob < NOB; ob++)
        acc[ob] = *reinterpret_cast<const f32x4*>(bimg + MlpGeom::B_OFF[LAYER] + (ob * 4 + q) * 4);
    const float* wl = wimg + MlpGeom::W_OFF[LAYER] + lane * 4;
#pragma unroll
    for (int s4 = 0; s4 < S4; s4++) {
        f32x4 a[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ob++) a[ob] = *reinterpret_cast<const f32x4*>(wl + (s4 * NOB + ob) * 256);
        f32x4 b = bop(s4);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int ob = 0; ob < NOB; ob++)
                acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][r], b[r], acc[ob], 0, 0, 0);
    }
}

// slimnn's ReLU, x.max(0.0) (activations.rs:32-37; NaN -> 0, -0 -> +0), as ONE instruction: the median of (x, 0, +inf). fmaxf costs
// two — the compiler canonicalises its operand first (a signalling NaN must come out quiet), which a value straight out of an
// fma chain never needs.
// (+inf comes from a scalar register the optimiser cannot see through: with the literal it rewrites the median as that same fmaxf.)
SYN_DEV float relu1(float x) {
    float inf = __builtin_inff();
    asm("" : "+s"(inf));
    return __builtin_amdgcn_fmed3f(x, 0.0f, inf);
}
SYN_DEV f32x4 relu4_(f32x4 v) {
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = relu1(v[r]);
    return v;
}

template <int NOB>
SYN_DEV void relu_inplace(f32x4 (&acc)[NOB]) {
#pragma unroll
    for (int ob = 0; ob < NOB; ob++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[ob][r] = relu1(acc[ob][r]);
}

// Evaluates the network for the 16 positions of this wave's tile. Lane l = (j = l&15, q = l>>4) must pass the
// bitboards of position j. Returns the last layer's D registers: lane (j,q) register r = raw output 4*q + r of
// position j (outputs 0..8 = policy logits, 9..11 = outcome logits, 12..15 = padding).
SYN_DEV f32x4 mlp_tile16(const float* __restrict__ wimg, const float* __restrict__ bimg, int lane,
                         const FeatureTable& FT, uint64_t hi, uint64_t lo) {
    f32x4 h1[8];
    mlp_layer<0, 8, 4>(wimg, bimg, lane, [&](int s4) {
        // s4 is a compile-time constant after unrolling; dispatch to the templated quad
        return s4 == 0 ? feature_quad<0>(FT, hi, lo) : s4 == 1 ? feature_quad<1>(FT, hi, lo)
             : s4 == 2 ? feature_quad<2>(FT, hi, lo) : feature_quad<3>(FT, hi, lo);
    }, h1);
    relu_inplace(h1);

    f32x4 h2[6];
    mlp_layer<1, 6, 8>(wimg, bimg, lane, [&](int s4) { return h1[s4]; }, h2);
    relu_inplace(h2);

    f32x4 h3[4];
    mlp_layer<2, 4, 6>(wimg, bimg, lane, [&](int s4) { return h2[s4]; }, h3);
    relu_inplace(h3);

    f32x4 h4[3];
    mlp_layer<3, 3, 4>(wimg, bimg, lane, [&](int s4) { return h3[s4]; }, h4);
    relu_inplace(h4);

    f32x4 out[1];
    mlp_layer<4, 1, 3>(wimg, bimg, lane, [&](int s4) { return h4[s4]; }, out);
    return out[0];
}

// Output blocks [OB0, OB0 + NB) of a layer with NOB blocks in total: same operands, same per-output fma chains as
// mlp_layer, but only NB accumulators and NB weight fragments are live at a time.
template <int LAYER, int NOB, int OB0, int NB, int S4, class BOperand>
SYN_DEV void mlp_layer_part(const float* __restrict__ wimg, const float* __restrict__ bimg, int lane, BOperand bop,
                            f32x4* acc) {
    const int q = lane >> 4;
#pragma unroll
    for (int ob = 0; ob < NB; ob++)
        acc[ob] = *reinterpret_cast<const f32x4*>(bimg + MlpGeom::B_OFF[LAYER] + ((OB0 + ob) * 4 + q) * 4);
    const float* wl = wimg + MlpGeom::W_OFF[LAYER] + lane * 4;
#pragma unroll
    for (int s4 = 0; s4 < S4; s4++) {
        f32x4 a[NB];
#pragma unroll
        for (int ob = 0; ob < NB; ob++) a[ob] = *reinterpret_cast<const f32x4*>(wl + (s4 * NOB + OB0 + ob) * 256);
        f32x4 b = bop(s4);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int ob = 0; ob < NB; ob++)
                acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][r], b[r], acc[ob], 0, 0, 0);
    }
}

// mlp_tile16 with a smaller register footprint (the two wide layers run in two passes over their output blocks): for the
// 16-waves-per-workgroup launch shape, whose budget is 128 VGPRs. Bit-identical outputs.
SYN_DEV f32x4 mlp_tile16_lowreg(const float* __restrict__ wimg, const float* __restrict__ bimg, int lane,
                                const FeatureTable& FT, uint64_t hi, uint64_t lo) {
    auto feat = [&](int s4) {
        return s4 == 0 ? feature_quad<0>(FT, hi, lo) : s4 == 1 ? feature_quad<1>(FT, hi, lo)
             : s4 == 2 ? feature_quad<2>(FT, hi, lo) : feature_quad<3>(FT, hi, lo);
    };
    f32x4 h1[8];
    mlp_layer_part<0, 8, 0, 4, 4>(wimg, bimg, lane, feat, h1);
    mlp_layer_part<0, 8, 4, 4, 4>(wimg, bimg, lane, feat, h1 + 4);
    relu_inplace(h1);

    f32x4 h2[6];
    mlp_layer_part<1, 6, 0, 3, 8>(wimg, bimg, lane, [&](int s4) { return h1[s4]; }, h2);
    mlp_layer_part<1, 6, 3, 3, 8>(wimg, bimg, lane, [&](int s4) { return h1[s4]; }, h2 + 3);
    relu_inplace(h2);

    f32x4 h3[4];
    mlp_layer_part<2, 4, 0, 2, 6>(wimg, bimg, lane, [&](int s4) { return h2[s4]; }, h3);
    mlp_layer_part<2, 4, 2, 2, 6>(wimg, bimg, lane, [&](int s4) { return h2[s4]; }, h3 + 2);
    relu_inplace(h3);

    f32x4 h4[3];
    mlp_layer<3, 3, 4>(wimg, bimg, lane, [&](int s4) { return h3[s4]; }, h4);
    relu_inplace(h4);

    f32x4 out[1];
    mlp_layer<4, 1, 3>(wimg, bimg, lane, [&](int s4) { return h4[s4]; }, out);
    return out[0];
}

// ---- software-pipelined single-wave tile for the dedicated matrix waves of the producer/consumer kernel ------------------
// Same operands and the same per-output fma chains as mlp_tile16 (bit-identical outputs), but scheduled by hand: a matrix
// wave is ALONE on its SIMD's matrix pipe, so nobody else's MFMAs hide its LDS latency or its dependent-accumulator
// latency. The network is cut into 59 groups of (layer part, 4 k-steps): while the 4*NB MFMAs of group g run (NB
// independent accumulators, interleaved), the NB weight fragments (+ the biases of a part's first group) of group g+1
// are already on their way from LDS into the other half of a register double buffer. `sched_barrier`s pin that order —
// left alone under the 128-VGPR budget, the compiler serialises "one fragment -> wait -> 4 dependent MFMAs".
struct PipeGroup { int layer, nob, ob0, nb, s4, ns4; };
constexpr PipeGroup pipe_group(int gi) {
    if (gi < 16) return {0, 8, (gi / 4) * 2, 2, gi % 4, 4};                   // L1: 4 parts of 2 blocks
    if (gi < 40) return {1, 6, ((gi - 16) / 8) * 2, 2, (gi - 16) % 8, 8};     // L2: 3 parts of 2 blocks
    if (gi < 52) return {2, 4, ((gi - 40) / 6) * 2, 2, (gi - 40) % 6, 6};     // L3: 2 parts of 2 blocks
    if (gi < 56) return {3, 3, 0, 3, gi - 52, 4};                             // L4: 3 blocks
    return {4, 1, 0, 1, gi - 56, 3};                                          // L5: 1 block
}
constexpr int PIPE_GROUPS = 59;

struct PipeRegs {
    f32x4 a[2][3];   // weight fragments: double buffer, up to 3 blocks per group
    f32x4 bb[3];     // biases of the part that starts with the next group
    f32x4 acc[3];    // accumulators of the current part
    f32x4 fq[2];     // layer-1 B operands of this group and the next (4 features per lane each)
    f32x4 h1[8], h2[6], h3[4], h4[3];
};

template <int GI>
SYN_DEV void pipe_prefetch(const float* __restrict__ wimg, const float* __restrict__ bimg, int lane, PipeRegs& R) {
    if constexpr (GI < PIPE_GROUPS) {
        constexpr PipeGroup G = pipe_group(GI);
        const float* wl = wimg + MlpGeom::W_OFF[G.layer] + lane * 4;
#pragma unroll
        for (int ob = 0; ob < G.nb; ob++)
            R.a[GI & 1][ob] = *reinterpret_cast<const f32x4*>(wl + (G.s4 * G.nob + G.ob0 + ob) * 256);
        if constexpr (G.s4 == 0) {
            const int q = lane >> 4;
#pragma unroll
            for (int ob = 0; ob < G.nb; ob++)
                R.bb[ob] = *reinterpret_cast<const f32x4*>(bimg + MlpGeom::B_OFF[G.layer] + ((G.ob0 + ob) * 4 + q) * 4);
        }
    }
}

template <int GI>
SYN_DEV void pipe_features(const FeatureTable& FT, uint64_t hi, uint64_t lo, PipeRegs& R) {
    if constexpr (GI < PIPE_GROUPS) {
        constexpr PipeGroup G = pipe_group(GI);
        if constexpr (G.layer == 0) {
            if constexpr (G.s4 == 0) R.fq[GI & 1] = feature_quad<0>(FT, hi, lo);
            else if constexpr (G.s4 == 1) R.fq[GI & 1] = feature_quad<1>(FT, hi, lo);
            else if constexpr (G.s4 == 2) R.fq[GI & 1] = feature_quad<2>(FT, hi, lo);
            else R.fq[GI & 1] = feature_quad<3>(FT, hi, lo);
        }
    }
}

template <int GI>
SYN_DEV void pipe_group_run(const float* __restrict__ wimg, const float* __restrict__ bimg, int lane, const FeatureTable& FT,
                            uint64_t hi, uint64_t lo, PipeRegs& R) {
    constexpr PipeGroup G = pipe_group(GI);
    pipe_prefetch<GI + 1>(wimg, bimg, lane, R);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G.s4 == 0) {
#pragma unroll
        for (int ob = 0; ob < G.nb; ob++) R.acc[ob] = R.bb[ob];
    }
    f32x4 b;
    if constexpr (G.layer == 0) b = R.fq[GI & 1];
    else if constexpr (G.layer == 1) b = R.h1[G.s4];
    else if constexpr (G.layer == 2) b = R.h2[G.s4];
    else if constexpr (G.layer == 3) b = R.h3[G.s4];
    else b = R.h4[G.s4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int ob = 0; ob < G.nb; ob++)
            R.acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(R.a[GI & 1][ob][r], b[r], R.acc[ob], 0, 0, 0);
    // the next layer-1 group's operands are built from the boards in the shadow of this group's MFMAs
    pipe_features<GI + 1>(FT, hi, lo, R);
    if constexpr (G.s4 == G.ns4 - 1 && G.layer < 4) {
#pragma unroll
        for (int ob = 0; ob < G.nb; ob++) {
            const f32x4 v = relu4_(R.acc[ob]);
            if constexpr (G.layer == 0) R.h1[G.ob0 + ob] = v;
            else if constexpr (G.layer == 1) R.h2[G.ob0 + ob] = v;
            else if constexpr (G.layer == 2) R.h3[G.ob0 + ob] = v;
            else R.h4[G.ob0 + ob] = v;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (GI + 1 < PIPE_GROUPS) pipe_group_run<GI + 1>(wimg, bimg, lane, FT, hi, lo, R);
}

// Lane l = (j = l&15, q = l>>4) passes the two feature boards of position j; returns the last layer's D registers (raw).
SYN_DEV f32x4 mlp_tile16_pipe(const float* __restrict__ wimg, const float* __restrict__ bimg, int lane,
                              const FeatureTable& FT, uint64_t hi, uint64_t lo) {
    PipeRegs R;
    pipe_prefetch<0>(wimg, bimg, lane, R);
    pipe_features<0>(FT, hi, lo, R);
    pipe_group_run<0>(wimg, bimg, lane, FT, hi, lo, R);
    return R.acc[0];
}

// policies.rs:54-57: softmax over the three outcome logits (libtorch: max-subtracted), exp = det_expf,
// sum in index order. v[0..2] in, probabilities out.
SYN_DEV void value_softmax(float& v0, float& v1, float& v2) {
    float m = v0;
    m = v1 > m ? v1 : m;
    m = v2 > m ? v2 : m;
    float e0 = det_expf(v0 - m), e1 = det_expf(v1 - m), e2 = det_expf(v2 - m);
    float total = 0.0f;
    total += e0;
    total += e1;
    total += e2;
    v0 = e0 / total;
    v1 = e1 / total;
    v2 = e2 / total;
}

// value_softmax with its three exponentials and divisions in the packed forms of device_common.cuh (det_expf2_in_range,
// div2_by_shared: the same operations per component, the same bits — lane_kernel.cuh's lane_softmaxes does its value part this way)
// whenever no lane of the wave is outside their exact range; otherwise the plain form. Must be called by whole waves.
SYN_DEV void value_softmax_packed(float& v0, float& v1, float& v2) {
    float m = v0;
    m = v1 > m ? v1 : m;
    m = v2 > m ? v2 : m;
    const float d0 = v0 - m, d1 = v1 - m, d2 = v2 - m;
    const bool in_range = d0 >= -41.0f && d1 >= -41.0f && d2 >= -41.0f;   // (NaN fails)
    if (__ballot(!in_range) == 0ull) {
        const f32x2 ea = det_expf2_in_range(f32x2{d0, d1}), eb = det_expf2_in_range(f32x2{d2, d2});
        float total = 0.0f;
        total += ea[0];
        total += ea[1];
        total += eb[0];
        const float r = rcp_refined_safe_range(total);   // 1 <= total <= 3, every e >= exp(-41) > 2^-60
        const f32x2 qa = div2_by_shared(ea, total, r), qb = div2_by_shared(eb, total, r);
        v0 = qa[0]; v1 = qa[1]; v2 = qb[0];
    } else {
        value_softmax(v0, v1, v2);
    }
}

// Copies the prebuilt weight image (global, fragment order) into LDS. All threads of the block participate.
SYN_DEV void stage_weight_image(float* __restrict__ lds_img, const float* __restrict__ g_img, int tid, int nthreads) {
    const f32x4* src = reinterpret_cast<const f32x4*>(g_img);
    f32x4* dst = reinterpret_cast<f32x4*>(lds_img);
    for (int i = tid; i < MlpGeom::IMG_FLOATS / 4; i += nthreads) dst[i] = src[i];
}


// ====================================================================================================================
// Register-resident, wave-split variant used by the fused self-play kernel.
//
// At 16 trees per CU there is exactly ONE 16-position tile per explore, so a single-wave evaluation (476 dependent-ish
// MFMAs, ~20k cycles) leaves three of the CU's four matrix pipes idle while everybody waits. Here the four waves of a
// workgroup split every layer by 16-unit output blocks and exchange activations through a 14 KB LDS buffer:
//     layer   blocks   wave 0   wave 1   wave 2   wave 3      MFMAs on the critical path
//     L1        8      0,4      1,5      2,6      3,7         32
//     L2        6      0,4      1,5      2        3           64
//     L3        4      0        1        2        3           24
//     L4        3      -        0        1        2           16
//     L5        1      0        -        -        -           12      = 148 instead of 476
// Because an f32 MFMA's A operand is ONE VGPR per k-step, a wave's share of the network is at most 148 registers: the
// weights are loaded once per kernel and then live in VGPRs — no LDS weight image, no per-MFMA ds_read, and the LDS
// budget drops from 123 KB to ~16 KB (which also lets two workgroups share a CU).
// The exchange keeps the fragment trick of the single-wave path: with the permuted unit order a lane's four D registers
// of block ob are exactly the B operands of the next layer's steps 4*ob..4*ob+3 on the SAME lane, so a lane writes one
// 16-byte record ex[ob][lane] and reads ex[s4][lane] back — lane-linear, conflict-free, no transposition.
// Accumulation order per output is unchanged (ascending k, bias first), so results stay bit-identical to mlp_tile16.
struct MlpSplitWeights {
    f32x4 w1[2][4];   // [slot][s4]
    f32x4 w2[2][8];
    f32x4 w3[6];
    f32x4 w4[4];
    f32x4 w5[3];
};

SYN_DEV f32x4 ldg_f32x4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// mw = wave index inside the workgroup (0..3)
template <bool W345_LDS>
SYN_DEV void mlp_split_load_weights(const float* __restrict__ g_img, int mw, int lane, MlpSplitWeights& W) {
    const float* base = g_img + lane * 4;
#pragma unroll
    for (int slot = 0; slot < 2; slot++)
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++)
            W.w1[slot][s4] = ldg_f32x4(base + MlpGeom::W_OFF[0] + (s4 * 8 + (mw + 4 * slot)) * 256);
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
        W.w2[0][s4] = ldg_f32x4(base + MlpGeom::W_OFF[1] + (s4 * 6 + mw) * 256);
        W.w2[1][s4] = ldg_f32x4(base + MlpGeom::W_OFF[1] + (s4 * 6 + (mw < 2 ? 4 + mw : mw)) * 256);
    }
    if (W345_LDS) return;
#pragma unroll
    for (int s4 = 0; s4 < 6; s4++) W.w3[s4] = ldg_f32x4(base + MlpGeom::W_OFF[2] + (s4 * 4 + mw) * 256);
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) W.w4[s4] = ldg_f32x4(base + MlpGeom::W_OFF[3] + (s4 * 3 + (mw > 0 ? mw - 1 : 0)) * 256);
#pragma unroll
    for (int s4 = 0; s4 < 3; s4++) W.w5[s4] = ldg_f32x4(base + MlpGeom::W_OFF[4] + s4 * 256);
}

SYN_DEV f32x4 relu4(f32x4 v) {
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = relu1(v[r]);
    return v;
}

// One 16-position tile, evaluated cooperatively by the 4 waves of the workgroup. `exA`/`exB`: LDS exchange buffers of
// 8 KB and 6 KB (f32x4 per [block][lane]); `bimg`: bias image in LDS (MlpGeom::B_FLOATS floats).
// Contains 4 workgroup barriers; must be called by all 256 threads. Returns the last layer's D registers on wave 0
// (lane (j,q) register r = raw output 4*q + r of position j); other waves return zeros.
//
// W345_LDS: layers 3-5 take their A operands from an LDS copy of the weight image (`w345` = image floats starting at
// MlpGeom::W_OFF[2]) instead of 52 registers. Those layers run one dependent accumulator chain per wave (40-cycle MFMA
// latency), so a ds_read_b128 per 4 MFMAs hides completely; it brings the kernel under 256 VGPRs so that two
// workgroups can share a CU.
template <bool W345_LDS>
SYN_DEV f32x4 mlp_split_tile16(const MlpSplitWeights& W, const float* __restrict__ bimg, const float* __restrict__ w345,
                               f32x4* exA, f32x4* exB, int mw, int lane, const FeatureTable& FT, uint64_t hi,
                               uint64_t lo) {
    const int q = lane >> 4;
    auto wlds = [&](int layer, int nob, int s4, int ob) {
        return *reinterpret_cast<const f32x4*>(w345 + (MlpGeom::W_OFF[layer] - MlpGeom::W_OFF[2]) +
                                               (s4 * nob + ob) * 256 + lane * 4);
    };
    auto bias = [&](int layer, int ob) {
        return *reinterpret_cast<const f32x4*>(bimg + MlpGeom::B_OFF[layer] + (ob * 4 + q) * 4);
    };
    // ---- L1: blocks mw and mw+4, B = features
    {
        f32x4 a0 = bias(0, mw), a1 = bias(0, mw + 4);
        f32x4 bq[4] = {feature_quad<0>(FT, hi, lo), feature_quad<1>(FT, hi, lo), feature_quad<2>(FT, hi, lo),
                       feature_quad<3>(FT, hi, lo)};
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w1[0][s4][r], bq[s4][r], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w1[1][s4][r], bq[s4][r], a1, 0, 0, 0);
            }
        }
        exA[mw * 64 + lane] = relu4(a0);
        exA[(mw + 4) * 64 + lane] = relu4(a1);
    }
    lds_barrier();
    // ---- L2: block mw (all waves) and block 4+mw (waves 0,1); B = exA[0..7]
    {
        const bool two = mw < 2;
        f32x4 a0 = bias(1, mw), a1 = bias(1, two ? 4 + mw : mw);
#pragma unroll
        for (int s4 = 0; s4 < 8; s4++) {
            f32x4 b = exA[s4 * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w2[0][s4][r], b[r], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w2[1][s4][r], b[r], a1, 0, 0, 0);
            }
        }
        exB[mw * 64 + lane] = relu4(a0);
        if (two) exB[(4 + mw) * 64 + lane] = relu4(a1);
    }
    lds_barrier();
    // ---- L3: block mw; B = exB[0..5]
    {
        f32x4 a0 = bias(2, mw);
#pragma unroll
        for (int s4 = 0; s4 < 6; s4++) {
            f32x4 b = exB[s4 * 64 + lane];
            f32x4 w = W345_LDS ? wlds(2, 4, s4, mw) : W.w3[s4];
#pragma unroll
            for (int r = 0; r < 4; r++) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], b[r], a0, 0, 0, 0);
        }
        exA[mw * 64 + lane] = relu4(a0);
    }
    lds_barrier();
    // ---- L4: waves 1..3 compute block mw-1; B = exA[0..3]
    if (mw > 0) {
        f32x4 a0 = bias(3, mw - 1);
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            f32x4 b = exA[s4 * 64 + lane];
            f32x4 w = W345_LDS ? wlds(3, 3, s4, mw - 1) : W.w4[s4];
#pragma unroll
            for (int r = 0; r < 4; r++) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], b[r], a0, 0, 0, 0);
        }
        exB[(mw - 1) * 64 + lane] = relu4(a0);
    }
    lds_barrier();
    // ---- L5: wave 0; B = exB[0..2]
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    if (mw == 0) {
        out = bias(4, 0);
#pragma unroll
        for (int s4 = 0; s4 < 3; s4++) {
            f32x4 b = exB[s4 * 64 + lane];
            f32x4 w = W345_LDS ? wlds(4, 1, s4, 0) : W.w5[s4];
#pragma unroll
            for (int r = 0; r < 4; r++) out = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], b[r], out, 0, 0, 0);
        }
    }
    return out;
}

}  // namespace syn

namespace syn {

// ====================================================================================================================
// LDS-weight, wave-split variant for the quad-asynchronous kernel: same block assignment and accumulation order as
// mlp_split_tile16, but every A operand comes from the shared LDS weight image (one ds_read_b128 per 4 MFMAs) and the
// four waves of the quad synchronise through `sync()` (an LDS spin barrier private to the quad) instead of the
// workgroup barrier, so other quads of the workgroup keep running their own phases.
template <class Sync>
SYN_DEV f32x4 mlp_quad_tile16(const float* __restrict__ wimg, const float* __restrict__ bimg, f32x4* exA, f32x4* exB,
                              int mw, int lane, uint32_t ft_mw /* FeatureTable::t[mw] */, uint64_t hi, uint64_t lo,
                              Sync sync) {
    const int q = lane >> 4;
    auto bias = [&](int layer, int ob) {
        return *reinterpret_cast<const f32x4*>(bimg + MlpGeom::B_OFF[layer] + (ob * 4 + q) * 4);
    };
    // The s4 loops are deliberately NOT unrolled and fetch their LDS operands one step ahead: fully unrolled, the
    // compiler hoists every ds_read of a layer to its top (60+ live registers), which is what pushed the 3- and
    // 4-quad workgroups (168 / 128 VGPR budgets) into scratch spills.
    // ---- features: wave mw builds the B operands of k-steps 4*mw..4*mw+3 (one f32x4 per lane) for the whole quad;
    //      exB is free until the end of L2, so it carries them to the other waves (one extra quad barrier, but 4x less
    //      feature arithmetic per wave and ~40 fewer live registers than computing all 16 features everywhere)
    {
        f32x4 f;
        f[0] = feature_from_boards_rt(ft_mw, 0, hi, lo);
        f[1] = feature_from_boards_rt(ft_mw, 1, hi, lo);
        f[2] = feature_from_boards_rt(ft_mw, 2, hi, lo);
        f[3] = feature_from_boards_rt(ft_mw, 3, hi, lo);
        exB[mw * 64 + lane] = f;
    }
    sync();
    // ---- L1: blocks mw and mw+4; B = features from exB[0..3]
    {
        f32x4 a0 = bias(0, mw), a1 = bias(0, mw + 4);
        const float* w0p = wimg + MlpGeom::W_OFF[0] + mw * 256 + lane * 4;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            f32x4 b = exB[s4 * 64 + lane];
            f32x4 w0 = *reinterpret_cast<const f32x4*>(w0p + s4 * 8 * 256);
            f32x4 w1 = *reinterpret_cast<const f32x4*>(w0p + (s4 * 8 + 4) * 256);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[r], b[r], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r], b[r], a1, 0, 0, 0);
            }
        }
        exA[mw * 64 + lane] = relu4(a0);
        exA[(mw + 4) * 64 + lane] = relu4(a1);
    }
    sync();
    // ---- L2: block mw (all waves) and block 4+mw (waves 0,1); B = exA[0..7]
    {
        const bool two = mw < 2;
        f32x4 a0 = bias(1, mw), a1 = bias(1, two ? 4 + mw : mw);
        const float* w0p = wimg + MlpGeom::W_OFF[1] + mw * 256 + lane * 4;
        const float* w1p = wimg + MlpGeom::W_OFF[1] + (two ? 4 + mw : mw) * 256 + lane * 4;
        f32x4 b = exA[lane];
        f32x4 w0 = *reinterpret_cast<const f32x4*>(w0p), w1 = *reinterpret_cast<const f32x4*>(w1p);
#pragma unroll 1
        for (int s4 = 0; s4 < 8; s4++) {
            int nx = s4 < 7 ? s4 + 1 : 7;
            f32x4 bn = exA[nx * 64 + lane];
            f32x4 w0n = *reinterpret_cast<const f32x4*>(w0p + nx * 6 * 256);
            f32x4 w1n = *reinterpret_cast<const f32x4*>(w1p + nx * 6 * 256);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[r], b[r], a0, 0, 0, 0);
                if (two) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r], b[r], a1, 0, 0, 0);
            }
            b = bn; w0 = w0n; w1 = w1n;
        }
        exB[mw * 64 + lane] = relu4(a0);
        if (two) exB[(4 + mw) * 64 + lane] = relu4(a1);
    }
    sync();
    // ---- L3: block mw; B = exB[0..5]
    {
        f32x4 a0 = bias(2, mw);
        const float* wp = wimg + MlpGeom::W_OFF[2] + mw * 256 + lane * 4;
        f32x4 b = exB[lane];
        f32x4 w = *reinterpret_cast<const f32x4*>(wp);
#pragma unroll 1
        for (int s4 = 0; s4 < 6; s4++) {
            int nx = s4 < 5 ? s4 + 1 : 5;
            f32x4 bn = exB[nx * 64 + lane];
            f32x4 wn = *reinterpret_cast<const f32x4*>(wp + nx * 4 * 256);
#pragma unroll
            for (int r = 0; r < 4; r++) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], b[r], a0, 0, 0, 0);
            b = bn; w = wn;
        }
        exA[mw * 64 + lane] = relu4(a0);
    }
    sync();
    // ---- L4: waves 1..3 compute block mw-1; B = exA[0..3]
    if (mw > 0) {
        f32x4 a0 = bias(3, mw - 1);
        const float* wp = wimg + MlpGeom::W_OFF[3] + (mw - 1) * 256 + lane * 4;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            f32x4 b = exA[s4 * 64 + lane];
            f32x4 w = *reinterpret_cast<const f32x4*>(wp + s4 * 3 * 256);
#pragma unroll
            for (int r = 0; r < 4; r++) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], b[r], a0, 0, 0, 0);
        }
        exB[(mw - 1) * 64 + lane] = relu4(a0);
    }
    sync();
    // ---- L5: wave 0; B = exB[0..2]
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    if (mw == 0) {
        out = bias(4, 0);
        const float* wp = wimg + MlpGeom::W_OFF[4] + lane * 4;
#pragma unroll
        for (int s4 = 0; s4 < 3; s4++) {
            f32x4 b = exB[s4 * 64 + lane];
            f32x4 w = *reinterpret_cast<const f32x4*>(wp + s4 * 256);
#pragma unroll
            for (int r = 0; r < 4; r++) out = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], b[r], out, 0, 0, 0);
        }
    }
    return out;
}

}  // namespace syn

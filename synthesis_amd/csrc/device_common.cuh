// synthesis_amd — device primitives shared by every kernel (gfx950 only; wave = 64 lanes).
//
//  * det_expf            the deterministic f32 exp both sides of every parity test use (algorithm: DESIGN.md §numerics)
//  * row16 primitives    a "group" is one DPP row = 16 consecutive lanes; one MCTS tree lives on one group
//                        (4 trees per wave64, <= 9 lanes active = one lane per Connect4 column / child)
//  * Connect4 bit ops    study-connect4/src/connect4.rs:37-83,221-258 as branch-free u64 arithmetic
//  * StdRng              rand 0.8 StdRng (ChaCha12 + PCG32 seed expansion) — synthesis/src/alpha_zero.rs:189,281,286
//
// Everything here is compiled with -ffp-contract=off and -fhip-fp32-correctly-rounded-divide-sqrt: each f32 operation
// is rounded exactly where the source says, so results are bit-identical to the C++ oracle built the same way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace syn {

#define SYN_DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------------ det_expf
SYN_DEV float bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
SYN_DEV uint32_t f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

SYN_DEV float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return bits_f32(0x7F800000u);
    if (x < -103.97208f) return 0.0f;
    float t = x * 1.44269504f;
    float n = __builtin_rintf(t);
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    int ni = (int)n;
    if (ni >= -125) return bits_f32(f32_bits(y) + ((uint32_t)ni << 23));
    float z = bits_f32(f32_bits(y) + ((uint32_t)(ni + 64) << 23));
    return z * bits_f32((uint32_t)(127 - 64) << 23);
}

// Deterministic f32 natural log (Exploration::Uct); same algorithm as oracle/det_math.hpp det_logf, bit for bit.
SYN_DEV float det_logf(float x) {
    if (x != x || x < 0.0f) return bits_f32(0x7FC00000u);
    if (x == 0.0f) return bits_f32(0xFF800000u);
    uint32_t bits = f32_bits(x);
    if (bits == 0x7F800000u) return x;
    int e = 0;
    if (bits < 0x00800000u) {
        x = x * 8388608.0f;
        bits = f32_bits(x);
        e = -23;
    }
    e += (int)(bits >> 23) - 127;
    float m = bits_f32((bits & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) {
        m = m * 0.5f;
        e += 1;
    }
    float f = m - 1.0f;
    float z = f * f;
    float y = 7.0376836292e-2f;
    y = __builtin_fmaf(y, f, -1.1514610310e-1f);
    y = __builtin_fmaf(y, f, 1.1676998740e-1f);
    y = __builtin_fmaf(y, f, -1.2420140846e-1f);
    y = __builtin_fmaf(y, f, 1.4249322787e-1f);
    y = __builtin_fmaf(y, f, -1.6668057665e-1f);
    y = __builtin_fmaf(y, f, 2.0000714765e-1f);
    y = __builtin_fmaf(y, f, -2.4999993993e-1f);
    y = __builtin_fmaf(y, f, 3.3333331174e-1f);
    y = y * f;
    y = y * z;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = f + y;
    return __builtin_fmaf(fe, 0.693359375f, r);
}

// ------------------------------------------------------------------------------------------------ row16 primitives
SYN_DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// DPP controls (ISA: DPP_CTRL): quad_perm = 0x00-0xFF, row_half_mirror = 0x141, row_mirror = 0x140.
template <int CTRL>
SYN_DEV uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
SYN_DEV float dpp_f32(float v) { return bits_f32(dpp_u32<CTRL>(f32_bits(v))); }

constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i within each 8
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15-i within the row

// All-lanes max over the row. After the two quad steps every lane of a quad holds the quad result, so the mirrors
// (which pair quad 0<->1 and then half 0<->1) complete an all-reduce without LDS traffic.
SYN_DEV uint32_t row_max_u32(uint32_t v) {
    uint32_t o;
    o = dpp_u32<DPP_XOR1>(v); v = v > o ? v : o;
    o = dpp_u32<DPP_XOR2>(v); v = v > o ? v : o;
    o = dpp_u32<DPP_HALF_MIRROR>(v); v = v > o ? v : o;
    o = dpp_u32<DPP_MIRROR>(v); v = v > o ? v : o;
    return v;
}
// max ignoring NaN (Rust f32::max / v_max_f32 semantics); exact, so association order is irrelevant
SYN_DEV float row_max_f32(float v) {
    v = __builtin_fmaxf(v, dpp_f32<DPP_XOR1>(v));
    v = __builtin_fmaxf(v, dpp_f32<DPP_XOR2>(v));
    v = __builtin_fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v));
    v = __builtin_fmaxf(v, dpp_f32<DPP_MIRROR>(v));
    return v;
}

// Argmax with "first index wins ties": returns the winning index on every lane of the row.
// Caller guarantees no NaN in v (see select: NaN is pre-mapped to the sequential-scan semantics).
// (value, index) is packed into one 64-bit key — order-preserving image of the float in the high word, 255 - index
// in the low word — so each butterfly step is two DPP moves, one 64-bit compare and two selects, no branches.
SYN_DEV int row_argmax_first(float v, int idx) {
    v = v + 0.0f;  // -0.0 -> +0.0 so the integer image orders exactly like the float compare
    uint32_t u = f32_bits(v);
    u ^= (uint32_t)((int32_t)u >> 31) | 0x80000000u;
    uint32_t lo = (uint32_t)(255 - idx);
#define SYN_ARGMAX_STEP(CTRL)                                                        \
    {                                                                                \
        uint32_t ou = dpp_u32<CTRL>(u), ol = dpp_u32<CTRL>(lo);                      \
        bool take = (((uint64_t)ou << 32) | ol) > (((uint64_t)u << 32) | lo);        \
        u = take ? ou : u;                                                           \
        lo = take ? ol : lo;                                                         \
    }
    SYN_ARGMAX_STEP(DPP_XOR1)
    SYN_ARGMAX_STEP(DPP_XOR2)
    SYN_ARGMAX_STEP(DPP_HALF_MIRROR)
    SYN_ARGMAX_STEP(DPP_MIRROR)
#undef SYN_ARGMAX_STEP
    return 255 - (int)lo;
}

// value of lane `src` (0..15) of this lane's row
SYN_DEV uint32_t row_bcast_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 16); }
SYN_DEV float row_bcast_f32(float v, int src) { return __shfl(v, src, 16); }

// 16-bit ballot of this lane's row
SYN_DEV uint32_t row_ballot(bool p) {
    unsigned long long b = __ballot(p);
    return (uint32_t)(b >> (lane_id() & 48)) & 0xFFFFu;
}

// Deterministic f32 tanh (slimnn Tanh, activations.rs:39-44); same algorithm as oracle/det_math.hpp det_tanhf, bit for bit.
SYN_DEV float det_tanhf(float x) {
    if (x != x) return x;
    const float z = x < 0.0f ? -x : x;
    if (z > 44.0f) return x < 0.0f ? -1.0f : 1.0f;
    if (z >= 0.625f) {
        const float s = det_expf(z + z);
        const float t = 1.0f - 2.0f / (s + 1.0f);
        return x < 0.0f ? -t : t;
    }
    const float w = x * x;
    float p = -5.70498872745e-3f;
    p = __builtin_fmaf(p, w, 2.06390887954e-2f);
    p = __builtin_fmaf(p, w, -5.37397155531e-2f);
    p = __builtin_fmaf(p, w, 1.33314422036e-1f);
    p = __builtin_fmaf(p, w, -3.33332819422e-1f);
    return __builtin_fmaf(p * w, x, x);
}

// ------------------------------------------------------------------------------------------------ packed division
// Two correctly rounded f32 quotients per instruction stream: the fused-multiply-add core of the hardware's own IEEE division
// (rcp, one Newton step on the reciprocal, quotient, two residual corrections — the sequence hipcc emits between
// v_div_scale and v_div_fixup under -fhip-fp32-correctly-rounded-divide-sqrt), written on 2-vectors so that it issues as
// v_pk_fma_f32 / v_pk_mul_f32: 2 v_rcp + 7 packed instructions for TWO divisions instead of 2 x 12. Scaling and fix-up are
// what the hardware sequence adds for operands near the ends of the exponent range, infinities, NaNs and zero divisors;
// they are the identity for  a = 0 or 2^-60 <= a <= 2^60  and  1 <= b <= 2^16  (no intermediate leaves the normal range),
// which is the ONLY range callers may use this on. tests/test_gpu_parity.py compares it with a / b bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
SYN_DEV f32x2 div2_safe_range(f32x2 a, f32x2 b) {
    f32x2 y = f32x2{__builtin_amdgcn_rcpf(b[0]), __builtin_amdgcn_rcpf(b[1])};
    const f32x2 e = __builtin_elementwise_fma(-b, y, f32x2{1.0f, 1.0f});
    y = __builtin_elementwise_fma(e, y, y);
    f32x2 q = a * y;
    f32x2 r = __builtin_elementwise_fma(-b, q, a);
    q = __builtin_elementwise_fma(r, y, q);
    r = __builtin_elementwise_fma(-b, q, a);
    return __builtin_elementwise_fma(r, y, q);
}
// Two det_expf per instruction stream: exactly det_expf's operations on each component (IEEE mul / fma / add per component, so
// v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 give the same bits) for the arguments that need none of its early-outs and not the
// two-step scaling:  -86 <= x <= 88.72283  (then n >= -124). Callers test that range wave-uniformly and fall back to det_expf.
SYN_DEV f32x2 det_expf2_in_range(f32x2 x) {
    const f32x2 t = x * f32x2{1.44269504f, 1.44269504f};
    const f32x2 n = f32x2{__builtin_rintf(t[0]), __builtin_rintf(t[1])};
    f32x2 r = __builtin_elementwise_fma(n, f32x2{-0.693145751953125f, -0.693145751953125f}, x);
    r = __builtin_elementwise_fma(n, f32x2{-1.42860682030941723212e-6f, -1.42860682030941723212e-6f}, r);
    f32x2 p = f32x2{1.9875691500e-4f, 1.9875691500e-4f};
    p = __builtin_elementwise_fma(p, r, f32x2{1.3981999507e-3f, 1.3981999507e-3f});
    p = __builtin_elementwise_fma(p, r, f32x2{8.3334519073e-3f, 8.3334519073e-3f});
    p = __builtin_elementwise_fma(p, r, f32x2{4.1665795894e-2f, 4.1665795894e-2f});
    p = __builtin_elementwise_fma(p, r, f32x2{1.6666665459e-1f, 1.6666665459e-1f});
    p = __builtin_elementwise_fma(p, r, f32x2{5.0000001201e-1f, 5.0000001201e-1f});
    const f32x2 r2 = r * r;
    const f32x2 y = __builtin_elementwise_fma(p, r2, r) + f32x2{1.0f, 1.0f};
    return f32x2{bits_f32(f32_bits(y[0]) + ((uint32_t)(int)n[0] << 23)), bits_f32(f32_bits(y[1]) + ((uint32_t)(int)n[1] << 23))};
}
// The refined reciprocal of div2_safe_range for a divisor shared by several quotients, and the quotient steps on it: a / b for
// a = 0 or 2^-60 <= a <= 2^60, 1 <= b <= 2^16 (same sequence, same bits as div2_safe_range and as the hardware's IEEE division).
SYN_DEV float rcp_refined_safe_range(float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
SYN_DEV f32x2 div2_by_shared(f32x2 a, float b, float y) {
    const f32x2 nb = f32x2{-b, -b}, yy = f32x2{y, y};
    f32x2 q = a * yy;
    f32x2 r = __builtin_elementwise_fma(nb, q, a);
    q = __builtin_elementwise_fma(r, yy, q);
    r = __builtin_elementwise_fma(nb, q, a);
    return __builtin_elementwise_fma(r, yy, q);
}
// a / b for INTEGER divisors 1 <= b <= 2^16 (explore_value's 1 + n): the same sequence with ONE residual correction. After the Newton
// step y is within an ulp of 1 / b, q0 = RN(a y) within 3 ulp of a / b, its residual exact, and q1 = RN(q0 + r y) rounds a / b perturbed
// by less than 2^-21 ulp — a quotient by a 16-bit divisor is never closer than 2^-17 ulp to a rounding boundary, so q1 is the IEEE
// quotient and div2_safe_range's second correction returns it unchanged.
SYN_DEV f32x2 div2_by_small_int(f32x2 a, f32x2 b) {
    f32x2 y = f32x2{__builtin_amdgcn_rcpf(b[0]), __builtin_amdgcn_rcpf(b[1])};
    const f32x2 e = __builtin_elementwise_fma(-b, y, f32x2{1.0f, 1.0f});
    y = __builtin_elementwise_fma(e, y, y);
    const f32x2 q = a * y;
    const f32x2 r = __builtin_elementwise_fma(-b, q, a);
    return __builtin_elementwise_fma(r, y, q);
}
// sqrtf(x) for 2^-60 <= x <= 2^60 (a visit count): the hardware's own correctly rounded sequence (v_sqrt_f32, then the two
// neighbours tested with exact residuals) without the scaling it wraps around denormal inputs.
SYN_DEV float sqrt_normal_range(float x) {
    const float y = __builtin_amdgcn_sqrtf(x);
    const float yd = bits_f32(f32_bits(y) - 1u), yu = bits_f32(f32_bits(y) + 1u);
    const float rd = __builtin_fmaf(-yd, y, x);
    const float ru = __builtin_fmaf(-yu, y, x);
    float r = rd <= 0.0f ? yd : y;
    r = ru > 0.0f ? yu : r;
    return r;
}
// smallest non-zero prior the packed division accepts; records of smaller (or non-finite) priors carry a flag (sign bit)
constexpr float PRIOR_SAFE_MIN = 0x1p-40f;

// ------------------------------------------------------------------------------------------------ Connect4
namespace c4 {
constexpr int WIDTH = 9, HEIGHT = 7;
constexpr uint64_t fab_row_c() {
    uint64_t r = 0;
    for (int c = 0; c < 9; c++) r |= 1ull << (7 * c);
    return r;
}
constexpr uint64_t FAB_ROW = fab_row_c();
constexpr uint64_t FULL = (1ull << 63) - 1;  // 63 cells
constexpr uint64_t cols0to5_c() {
    uint64_t r = 0;
    for (int c = 0; c < 6; c++) r |= 0x7Full << (7 * c);
    return r;
}
constexpr uint64_t rows_c(int lo, int hi) {
    uint64_t r = 0;
    for (int i = lo; i <= hi; i++) r |= fab_row_c() << i;
    return r;
}
constexpr uint64_t D1_MASK = cols0to5_c() & rows_c(3, 6);
constexpr uint64_t D2_MASK = cols0to5_c() & rows_c(0, 3);
constexpr uint64_t H_MASK = cols0to5_c();
constexpr uint64_t V_MASK = rows_c(0, 3);

// connect4.rs:77-83
SYN_DEV bool won(uint64_t bb) {
    uint64_t d1 = bb & (bb >> 6) & (bb >> 12) & (bb >> 18) & D1_MASK;
    uint64_t d2 = bb & (bb >> 8) & (bb >> 16) & (bb >> 24) & D2_MASK;
    uint64_t h = bb & (bb >> 7) & (bb >> 14) & (bb >> 21) & H_MASK;
    uint64_t v = bb & (bb >> 1) & (bb >> 2) & (bb >> 3) & V_MASK;
    return (v | h | d1 | d2) != 0;
}
SYN_DEV int col_height(uint64_t occ, int col) { return __popcll(occ & (0x7Full << (7 * col))); }

// Legal columns (bit c = column c is not full) from the occupancy board: column c is full iff its top cell, bit 6 + 7c,
// is set. Four top-row bits at stride 7 (x bit 7i) are gathered by ONE full-rate 24-bit multiply: x * (2^18 + 2^12 + 2^6 + 1)
// puts bit 7i on bit 18 + i, every other partial product falls outside bits 18..21 and no two coincide (no carries).
// ~14 full-rate VALU instructions instead of nine 64-bit mask + popcount pairs.
SYN_DEV uint32_t legal_columns(uint64_t occ) {
    constexpr uint32_t PICK = 1u | (1u << 7) | (1u << 14) | (1u << 21), GATHER = (1u << 18) | (1u << 12) | (1u << 6) | 1u;
    const uint32_t free_lo = ~(uint32_t)occ, free_hi = ~(uint32_t)(occ >> 32);
    const uint32_t a = (__umul24((free_lo >> 6) & PICK, GATHER) >> 18) & 0xFu;   // columns 0..3: bits 6, 13, 20, 27
    const uint32_t b = (__umul24((free_hi >> 2) & PICK, GATHER) >> 18) & 0xFu;   // columns 4..7: bits 34, 41, 48, 55
    const uint32_t c = (free_hi >> 30) & 1u;                                      // column 8: bit 62
    return a | (b << 4) | (c << 8);
}

// Cells c (one bit each) such that won(m | c): c completes a four-in-a-row with three stones of m (connect4.rs:70-83, same
// anchor masks, so exactly the geometric lines the reference accepts). For anchor a and shift s the line is
// {a, a+s, a+2s, a+3s}; T_k = anchors whose other three cells are in m; the winning cell is a + k*s.
SYN_DEV uint64_t winning_cells_dir(uint64_t m, int s, uint64_t mask) {
    const uint64_t m1 = m >> s, m2 = m >> (2 * s), m3 = m >> (3 * s);
    const uint64_t p01 = m & m1, p23 = m2 & m3;
    const uint64_t t3 = mask & p01 & m2, t2 = mask & p01 & m3, t1 = mask & m & p23, t0 = mask & m1 & p23;
    return t0 | (t1 << s) | (t2 << (2 * s)) | (t3 << (3 * s));
}
SYN_DEV uint64_t winning_cells(uint64_t m) {
    return winning_cells_dir(m, 6, D1_MASK) | winning_cells_dir(m, 8, D2_MASK) | winning_cells_dir(m, 7, H_MASK) |
           winning_cells_dir(m, 1, V_MASK);
}

// Lowest empty cell of every non-full column (gravity keeps columns contiguous from the bottom):
// a cell is "next free" iff it is empty and (it is in row 0 or the cell below is occupied).
SYN_DEV uint64_t next_free_cells(uint64_t occ) { return ((occ << 1) | FAB_ROW) & ~occ & FULL; }

// connect4.rs:235-258 for one flat feature index f = row*9 + col (f in 0..62)
SYN_DEV float feature(uint64_t my, uint64_t op, uint64_t nextfree, int f) {
    int row = f / 9, col = f - row * 9;
    uint64_t bit = 1ull << (row + 7 * col);
    float v = -0.1f;
    v = (nextfree & bit) ? 0.1f : v;
    v = (op & bit) ? -1.0f : v;
    v = (my & bit) ? 1.0f : v;
    return v;
}
}  // namespace c4

// ------------------------------------------------------------------------------------------------ StdRng (ChaCha12)
struct StdRng {
    uint32_t key[8];
    uint32_t index;  // words consumed so far (word i lives in block i/16)

    SYN_DEV static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

    // rand_core 0.6 seed_from_u64: PCG32 fills the 32-byte key
    SYN_DEV void seed_from_u64(uint64_t state) {
        const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            state = state * MUL + INC;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
        index = 0;
    }

    SYN_DEV uint32_t word(uint32_t i) const {
        uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                          key[4], key[5], key[6], key[7], i >> 4, 0u, 0u, 0u};
        uint32_t x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[k];
#define SYN_QR(a, b, c, d)                          \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 16); \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 12); \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 8);  \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 7);
#pragma unroll
        for (int r = 0; r < 6; r++) {
            SYN_QR(0, 4, 8, 12) SYN_QR(1, 5, 9, 13) SYN_QR(2, 6, 10, 14) SYN_QR(3, 7, 11, 15)
            SYN_QR(0, 5, 10, 15) SYN_QR(1, 6, 11, 12) SYN_QR(2, 7, 8, 13) SYN_QR(3, 4, 9, 14)
        }
#undef SYN_QR
        uint32_t w = i & 15, out = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) out = (w == (uint32_t)k) ? x[k] + s[k] : out;
        return out;
    }

    SYN_DEV uint32_t next_u32() { return word(index++); }

    // all sixteen words of output block `blk` (words 16*blk .. 16*blk+15), for callers that keep a block around
    SYN_DEV void block16(uint32_t blk, uint32_t out[16]) const {
        const uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                                key[4], key[5], key[6], key[7], blk, 0u, 0u, 0u};
        uint32_t x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[k];
#define SYN_QR(a, b, c, d)                          \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 16); \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 12); \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 8);  \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 7);
#pragma unroll 1
        for (int r = 0; r < 6; r++) {
            SYN_QR(0, 4, 8, 12) SYN_QR(1, 5, 9, 13) SYN_QR(2, 6, 10, 14) SYN_QR(3, 7, 11, 15)
            SYN_QR(0, 5, 10, 15) SYN_QR(1, 6, 11, 12) SYN_QR(2, 7, 8, 13) SYN_QR(3, 4, 9, 14)
        }
#undef SYN_QR
#pragma unroll
        for (int k = 0; k < 16; k++) out[k] = x[k] + s[k];
    }

    // Rng::gen_range(0..n) for u8 (rand 0.8.3 UniformInt<u8>::sample_single): u32 widening multiply + modulus zone
    SYN_DEV uint32_t gen_range_u8(uint32_t n) {
        uint32_t ints_to_reject = (0xFFFFFFFFu - n + 1u) % n;
        uint32_t zone = 0xFFFFFFFFu - ints_to_reject;
        for (;;) {
            uint32_t v = next_u32();
            uint64_t m = (uint64_t)v * (uint64_t)n;
            if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
        }
    }

    // Uniform<f32>::new(0, total).sample — rand 0.8.3 UniformFloat<f32>
    SYN_DEV float uniform_0_to(float total) {
        float v12 = bits_f32((next_u32() >> 9) | 0x3F800000u);
        float v01 = v12 - 1.0f;
        return v01 * total + 0.0f;
    }
};

}  // namespace syn

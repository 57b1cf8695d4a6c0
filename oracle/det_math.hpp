// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// Deterministic f32 exp. The reference calls Rust's f32::exp (synthesis/src/mcts.rs:418, slimnn/src/activations.rs:55)
// and libtorch's softmax (study-connect4/src/policies.rs:54-57); both end in a platform libm/vendor expf whose
// last-ulp behaviour is not pinned by anything in the reference. To make "visit-count-exact" parity testable
// across CPU and GPU, the oracle and the HIP engine both implement THIS algorithm (stated in DESIGN.md §numerics)
// with IEEE-exact operations only (mul, fma, rint, integer exponent insertion), so results agree bit for bit:
//   n = rint(x * log2(e));  r = fma(n, -ln2_hi, x);  r = fma(n, -ln2_lo, r)      (Cody-Waite)
//   p = degree-5 Horner in r with fma (Cephes expf coefficients);  y = fma(p, r*r, r) + 1
//   result = y * 2^n by exponent insertion (two-step for n < -125 so subnormals round once)
// |error| vs exact exp is < 1 ulp on the softmax domain x <= 0 (checked in tests/test_oracle_kats.py against float64).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace oracle {

inline float bits_to_float(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline uint32_t float_to_bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

inline float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return bits_to_float(0x7F800000u);
    if (x < -103.97208f) return 0.0f;
    float t = x * 1.44269504f;
    float n = std::rint(t);
    float r = std::fmaf(n, -0.693145751953125f, x);
    r = std::fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = std::fmaf(p, r, 1.3981999507e-3f);
    p = std::fmaf(p, r, 8.3334519073e-3f);
    p = std::fmaf(p, r, 4.1665795894e-2f);
    p = std::fmaf(p, r, 1.6666665459e-1f);
    p = std::fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = std::fmaf(p, r2, r) + 1.0f;
    int ni = (int)n;
    if (ni >= -125) {
        // y in [0.70, 1.42): biased exponent 126 or 127, so the sum stays in 1..254 for -125 <= ni <= 128 (y < 1 there)
        return bits_to_float(float_to_bits(y) + ((uint32_t)ni << 23));
    }
    // subnormal result: scale to a normal first, then one exact-power-of-two multiply (single rounding)
    float z = bits_to_float(float_to_bits(y) + ((uint32_t)(ni + 64) << 23));
    return z * bits_to_float((uint32_t)(127 - 64) << 23);
}

}  // namespace oracle

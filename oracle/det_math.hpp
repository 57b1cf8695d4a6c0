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


// Deterministic f32 natural log for Exploration::Uct (synthesis/src/mcts.rs:364 calls Rust's f32::ln -> platform libm,
// unpinned). Same contract as det_expf: IEEE-exact operations only, identical on CPU and GPU. Cephes logf scheme:
// x = m * 2^e with m in (sqrt(1/2), sqrt(2)], f = m - 1, degree-8 polynomial in f with fma, ln2 split in two parts.
// ln(1) is exactly 0 (a parent with one visit gives a zero exploration term, as in the reference).
inline float det_logf(float x) {
    if (x != x || x < 0.0f) return bits_to_float(0x7FC00000u);
    if (x == 0.0f) return bits_to_float(0xFF800000u);
    uint32_t bits = float_to_bits(x);
    if (bits == 0x7F800000u) return x;
    int e = 0;
    if (bits < 0x00800000u) {  // subnormal: scale into the normal range first
        x = x * 8388608.0f;
        bits = float_to_bits(x);
        e = -23;
    }
    e += (int)(bits >> 23) - 127;
    float m = bits_to_float((bits & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) {
        m = m * 0.5f;
        e += 1;
    }
    float f = m - 1.0f;
    float z = f * f;
    float y = 7.0376836292e-2f;
    y = std::fmaf(y, f, -1.1514610310e-1f);
    y = std::fmaf(y, f, 1.1676998740e-1f);
    y = std::fmaf(y, f, -1.2420140846e-1f);
    y = std::fmaf(y, f, 1.4249322787e-1f);
    y = std::fmaf(y, f, -1.6668057665e-1f);
    y = std::fmaf(y, f, 2.0000714765e-1f);
    y = std::fmaf(y, f, -2.4999993993e-1f);
    y = std::fmaf(y, f, 3.3333331174e-1f);
    y = y * f;
    y = y * z;
    float fe = (float)e;
    y = std::fmaf(fe, -2.12194440e-4f, y);
    y = std::fmaf(-0.5f, z, y);
    float r = f + y;
    return std::fmaf(fe, 0.693359375f, r);
}

// Deterministic f32 tanh for slimnn's Tanh activation (slimnn/src/activations.rs:39-44 calls Rust's f32::tanh -> platform
// libm, unpinned). Cephes tanhf scheme with IEEE-exact operations only (same code on the device): |x| < 0.625 -> odd
// polynomial x + x^3 P(x^2) by fma; otherwise 1 - 2 / (det_expf(2|x|) + 1) with the sign restored; |x| > 44 -> +-1.
inline float det_tanhf(float x) {
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
    p = std::fmaf(p, w, 2.06390887954e-2f);
    p = std::fmaf(p, w, -5.37397155531e-2f);
    p = std::fmaf(p, w, 1.33314422036e-1f);
    p = std::fmaf(p, w, -3.33332819422e-1f);
    return std::fmaf(p * w, x, x);
}

}  // namespace oracle

// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// Restatement of the third-party RNG arithmetic the reference's self-play path consumes. The crates are NOT in
// /root/reference (Cargo.lock is git-ignored, nothing vendored): rand = "0.8.3" (synthesis/Cargo.toml:12),
// which resolves StdRng to rand_chacha 0.3's ChaCha12Rng and SeedableRng::seed_from_u64 to rand_core 0.6's
// PCG32 seed expansion. What is restated here is those crates' published algorithms:
//   * seed_from_u64: PCG32 (MUL 6364136223846793005, INC 11634580027462260723, XSH-RR output) fills the
//     32-byte seed 4 bytes at a time, little-endian.
//   * ChaCha12: RFC 7539 quarter-round, 6 double rounds, 64-bit block counter in words 12-13, 64-bit stream id
//     (0) in words 14-15; the generator buffers 4 consecutive blocks (64 u32 words) per refill.
//   * BlockRng::next_u32 / next_u64 index rules (incl. the straddling case at the buffer edge).
//   * Rng::gen_range(0..n) for u8: widening multiply in u32 with the modulus rejection zone.
//   * WeightedIndex<f32>::new/sample: running f32 sum, Uniform<f32>(0,total) = (u32>>9 -> [1,2)) - 1, times total;
//     index = number of cumulative weights <= draw.
// Reference call sites: synthesis/src/alpha_zero.rs:189 (StdRng::seed_from_u64), :281 (gen_range), :286-287
// (WeightedIndex), synthesis/src/policies/rollout.rs:16 (gen_range), synthesis/src/mcts.rs:692 (test seed).
//
// PARITY UNPINNED for the crate internals: the only in-tree evidence is the three nodes.len() asserts of
// mcts.rs:732,781,830 (see tests/test_oracle_kats.py for which of them this restatement reproduces) plus the
// ChaCha core checked against the RFC 7539 block function test vector with 20 rounds.
#pragma once
#include <cstdint>
#include <cstring>

namespace oracle {

struct ChaChaRng {
    uint32_t key[8];
    uint64_t counter = 0;   // block counter of the next refill
    uint32_t buf[64];
    int index = 64;         // >= 64 means empty
    int rounds = 12;        // StdRng in rand 0.8 = ChaCha12; 20 kept selectable for the RFC vector

    static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    static inline void qr(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
        a += b; d ^= a; d = rotl(d, 16);
        c += d; b ^= c; b = rotl(b, 12);
        a += b; d ^= a; d = rotl(d, 8);
        c += d; b ^= c; b = rotl(b, 7);
    }

    static void block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
        uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                          key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                          (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
        uint32_t x[16];
        std::memcpy(x, s, sizeof(x));
        for (int r = 0; r < rounds; r += 2) {
            qr(x[0], x[4], x[8], x[12]);
            qr(x[1], x[5], x[9], x[13]);
            qr(x[2], x[6], x[10], x[14]);
            qr(x[3], x[7], x[11], x[15]);
            qr(x[0], x[5], x[10], x[15]);
            qr(x[1], x[6], x[11], x[12]);
            qr(x[2], x[7], x[8], x[13]);
            qr(x[3], x[4], x[9], x[14]);
        }
        for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
    }

    static ChaChaRng from_seed(const uint8_t seed[32], int rounds = 12) {
        ChaChaRng r;
        for (int i = 0; i < 8; i++)
            r.key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) |
                       ((uint32_t)seed[4 * i + 3] << 24);
        r.rounds = rounds;
        return r;
    }

    // rand_core 0.6 SeedableRng::seed_from_u64
    static void expand_seed_u64(uint64_t state, uint8_t seed[32]) {
        const uint64_t MUL = 6364136223846793005ull;
        const uint64_t INC = 11634580027462260723ull;
        for (int i = 0; i < 8; i++) {
            state = state * MUL + INC;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
            seed[4 * i + 0] = (uint8_t)x;
            seed[4 * i + 1] = (uint8_t)(x >> 8);
            seed[4 * i + 2] = (uint8_t)(x >> 16);
            seed[4 * i + 3] = (uint8_t)(x >> 24);
        }
    }

    static ChaChaRng seed_from_u64(uint64_t state, int rounds = 12) {
        uint8_t seed[32];
        expand_seed_u64(state, seed);
        return from_seed(seed, rounds);
    }

    void refill() {
        for (int b = 0; b < 4; b++) block(key, counter + (uint64_t)b, 0, rounds, buf + 16 * b);
        counter += 4;
        index = 0;
    }

    uint32_t next_u32() {
        if (index >= 64) refill();
        return buf[index++];
    }

    // rand_core::block::BlockRng::next_u64
    uint64_t next_u64() {
        if (index < 63) {
            uint64_t lo = buf[index], hi = buf[index + 1];
            index += 2;
            return (hi << 32) | lo;
        } else if (index >= 64) {
            refill();
            uint64_t lo = buf[0], hi = buf[1];
            index = 2;
            return (hi << 32) | lo;
        } else {
            uint64_t x = buf[63];
            refill();
            uint64_t y = buf[0];
            index = 1;
            return (y << 32) | x;
        }
    }

    // Rng::gen_range(0..n) with n: u8 (UniformInt<u8>::sample_single -> sample_single_inclusive(0, n-1)):
    // u8 samples in u32; ranges that fit u16 use the exact modulus zone.
    uint8_t gen_range_u8(uint8_t n) {
        uint32_t range = (uint32_t)n;  // high - low + 1 with low = 0, high = n - 1
        uint32_t ints_to_reject = (0xFFFFFFFFu - range + 1u) % range;
        uint32_t zone = 0xFFFFFFFFu - ints_to_reject;
        for (;;) {
            uint32_t v = next_u32();
            uint64_t m = (uint64_t)v * (uint64_t)range;
            uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
            if (lo <= zone) return (uint8_t)hi;
        }
    }

    // Uniform<f32>::new(0, total).sample: value in [1,2) from the top 23 bits, minus 1, times scale (+ low = 0).
    float uniform_f32_0_to(float total) {
        uint32_t bits = (next_u32() >> 9) | 0x3F800000u;
        float v12;
        std::memcpy(&v12, &bits, 4);
        float v01 = v12 - 1.0f;
        return v01 * total + 0.0f;
    }

    // WeightedIndex::<f32>::new(weights).sample(rng); n >= 1, all weights >= 0, sum > 0.
    int weighted_index(const float* w, int n) {
        float cum[64];
        float total = w[0];
        for (int i = 1; i < n; i++) {
            cum[i - 1] = total;
            total += w[i];
        }
        float chosen = uniform_f32_0_to(total);
        int idx = 0;
        for (int i = 0; i < n - 1; i++)
            if (cum[i] <= chosen) idx = i + 1;  // cumulative weights are non-decreasing: partition point
        return idx;
    }
};

}  // namespace oracle

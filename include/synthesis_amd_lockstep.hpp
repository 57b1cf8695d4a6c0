// synthesis_amd — the reference's MCTS with the trees on the HOST and only Policy::eval on the GPU, many trees in lock step.
//
// What this is for. The fused engine (syn_mcts_search / syn_selfplay_run) holds Connect4 and the whole search on the device; a
// caller with a DIFFERENT `Game` impl has no kernel of its own. This header is the other drop-in the reference's API allows
// (BASELINE.json configs[1] as worded: "4096 concurrent games, batched leaf inference"): `MCTS<G, P, N>` (synthesis/src/mcts.rs:
// 7-489) restated over any type with the `Game<N>` surface (synthesis/src/game.rs:65-88), with the one change that makes a GPU
// policy usable at all — `visit()`'s `policy.eval(&game)` (mcts.rs:407) is taken out of the recursion: every tree runs until
// it stands on a leaf that needs the network, the leaves of all trees go through ONE `BatchPolicy::eval_batch` call
// (syn_policy_eval_batch for Connect4), and the trees continue. A tree's explores stay sequential, so each tree is, node for
// node and bit for bit, the tree the reference builds; trees never interact.
//
//   synthesis::Outcome                      synthesis/src/game.rs:9-62 (reversed, value, Ord)
//   synthesis::LockstepTree<G, N>           mcts.rs:103-147 (with_capacity, explore_n), 310-489 (explore, select_best_child,
//                                           exploit_value, explore_value, visit, backprop), 174-225 (target_policy, target_q),
//                                           229-269 (root noise: None / Equal), 273-306 (best_action, solution)
//   synthesis::BatchPolicy<G, N>            policies/traits.rs:4-6 for a batch
//   synthesis::lockstep_search              `MCTS::with_capacity(explores + 1, ..) + explore_n(explores)` for many roots
//   synthesis::lockstep_selfplay            run_n_games (alpha_zero.rs:181-209) over such trees: run_game / sample_action /
//                                           fill_state_info / store_rewards (alpha_zero.rs:229-338), one StdRng per game
//   synthesis::lockstep_*_sharded           the same with one policy per host thread (gather_experience's worker model)
//   synthesis::CombiningPolicy<G, N>        one (GPU) policy shared by the workers of a sharded driver: their batches go out combined
//   synthesis::BatchPolicyWithCache         policies/cache.rs:5-59 in front of a BatchPolicy (Owned...: holding it)
//   synthesis::HipBatchPolicy               BatchPolicy<Connect4, 9> over an evaluation context of the engine (syn_eval_ctx_*)
//
// Numerics: the f32 expression order of mcts.rs, exp / ln through the same deterministic restatements the device and the oracle
// use (det_expf / det_logf below). Compile with -ffp-contract=off for bit parity with syn_mcts_search (tests/test_lockstep.py
// holds this driver to the oracle and to the device search). Fpu::Normal (the reference's own self-play configuration,
// study-connect4/src/main.rs:43-47) draws from the same counter-based per-tree stream as the device path (the draw
// is a pure function of tree seed, scan number and child slot, restated below), PolicyNoise::Dirichlet from the device path's
// per-tree generator and gamma sampler (NoiseRng below): every MCTSConfig of the reference runs here.
#pragma once
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include <unordered_map>

#include "synthesis_amd.hpp"
#include "synthesis_amd_zig_tables.hpp"
#include "synthesis_amd_fpu_normal_table.hpp"

namespace synthesis {

namespace detail {
inline float bits_f32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t f32_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// exp for the prior softmax (mcts.rs:418) — the restatement of csrc/device_common.cuh / oracle/det_math.hpp
inline float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return bits_f32(0x7F800000u);
    if (x < -103.97208f) return 0.0f;
    const float t = x * 1.44269504f;
    const float n = std::nearbyintf(t);
    float r = std::fmaf(n, -0.693145751953125f, x);
    r = std::fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = std::fmaf(p, r, 1.3981999507e-3f);
    p = std::fmaf(p, r, 8.3334519073e-3f);
    p = std::fmaf(p, r, 4.1665795894e-2f);
    p = std::fmaf(p, r, 1.6666665459e-1f);
    p = std::fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = std::fmaf(p, r2, r) + 1.0f;
    const int ni = (int)n;
    if (ni >= -125) return bits_f32(f32_bits(y) + ((uint32_t)ni << 23));
    const float z = bits_f32(f32_bits(y) + ((uint32_t)(ni + 64) << 23));
    return z * bits_f32((uint32_t)(127 - 64) << 23);
}

// ln for Exploration::Uct (mcts.rs:364)
inline float det_logf(float x) {
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
    const float f = m - 1.0f;
    const float z = f * f;
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
    const float fe = (float)e;
    y = std::fmaf(fe, -2.12194440e-4f, y);
    y = std::fmaf(-0.5f, z, y);
    const float r = f + y;
    return std::fmaf(fe, 0.693359375f, r);
}

// ---- Fpu::Normal draws (study-connect4/src/main.rs:43-47 samples thread_rng; this build's reproducible definition, the same
// one csrc/noise.cuh computes on the device) ------------------------------------------------------------------------------------
constexpr uint64_t NOISE_GOLDEN = 0x9E3779B97F4A7C15ull;
constexpr uint64_t NOISE_FPU_TAG = 0xF9C5A7B3E1D20F4Bull;
inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// the seed of one tree's streams: `stream` names the game (or the root of a batched search), `turn` the move within it
inline uint64_t noise_tree_seed(uint64_t stream, uint32_t turn) {
    return mix64((stream ^ 0x5851F42D4C957F2Dull) + (uint64_t)(turn + 1u) * NOISE_GOLDEN);
}
inline uint64_t noise_splitmix64(uint64_t seed, uint32_t index) { return mix64(seed + (uint64_t)(index + 1u) * NOISE_GOLDEN); }
// standard normal from 23 random bits k: u = (2k+1)/2^24, z = Phi^-1(u) by piecewise-linear inversion on the 2,944 cells of
// synthesis_amd_fpu_normal_table.hpp (the table the device reads: csrc/noise.cuh fpu_std_normal): the distance of u to the nearer end
// of (0, 1) as the odd integer m = 2 min(k, 2^23 - 1 - k) + 1; (float)m's exponent and top seven mantissa bits name the cell, its low
// sixteen mantissa bits the position inside it; |z| = fma(frac, T[cell + 1] - T[cell], T[cell])
inline float fpu_std_normal(uint32_t k) {
    const bool neg = k < (1u << 22);
    const uint32_t j = neg ? k : 0x7FFFFFu - k;
    const float f = (float)(2u * j + 1u);   // exact
    uint32_t b;
    std::memcpy(&b, &f, 4);
    const uint32_t cell = (b >> 16) - (127u << 7);
    const float frac = (float)(b & 0xFFFFu) * 0x1p-16f;
    const float t0 = FPU_NORMAL_TABLE[cell];
    const float z = std::fmaf(frac, FPU_NORMAL_TABLE[cell + 1] - t0, t0);
    return neg ? -z : z;
}
// draw number (scan, slot) of a tree: word 0 = SplitMix64 output `scan` of the stream seeded tree_seed ^ NOISE_FPU_TAG, word p + 1 =
// xorshift64(word p); word slot / 2 serves the slot: an even slot takes bits 41..63 of it, an odd slot bits 9..31
inline float noise_fpu_normal(uint64_t tree_seed, uint32_t scan, uint32_t slot, float mean, float std_dev) {
    uint64_t w = noise_splitmix64(tree_seed ^ NOISE_FPU_TAG, scan);
    for (uint32_t p = 0; p < (slot >> 1); p++) { w ^= w << 13; w ^= w >> 7; w ^= w << 17; }   // xorshift64
    const uint32_t k = (slot & 1u) ? ((uint32_t)w >> 9) : (uint32_t)(w >> 41);
    return mean + std_dev * fpu_std_normal(k);
}

// rand 0.8.3 `StdRng` as far as run_game draws from it (alpha_zero.rs:189,281,286-287): ChaCha with 12 rounds keyed by rand_core
// 0.6's `seed_from_u64` (a PCG32 stream fills the 32-byte key), 64-bit block counter, stream 0, words handed out in order;
// `gen_range(0..n as u8)` = UniformInt<u8>::sample_single (32-bit widening multiply with a rejection zone);
// `WeightedIndex::<f32>::new(w).sample()` = one f32 in [0, total) from the top 23 bits of one word, partition point of the
// cumulative weights. Same stream as csrc/device_common.cuh::StdRng (checked word for word in tests/test_lockstep.py).
class StdRng {
public:
    explicit StdRng(uint64_t seed) {
        uint64_t state = seed;
        for (int i = 0; i < 8; i++) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            const uint32_t rot = (uint32_t)(state >> 59);
            key_[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
    }
    uint32_t next_u32() {
        if (pos_ == 16) refill();
        return buf_[pos_++];
    }
    uint64_t next_u64() {   // rand_core BlockRng::next_u64: the next two words of the stream, low word first
        const uint64_t lo = next_u32();
        return lo | ((uint64_t)next_u32() << 32);
    }
    uint64_t words_drawn() const { return drawn_blocks_ * 16 - (uint64_t)(16 - pos_); }
    uint32_t gen_range_u8(uint32_t n) {
        const uint32_t zone = 0xFFFFFFFFu - (0xFFFFFFFFu - n + 1u) % n;
        for (;;) {
            const uint64_t m = (uint64_t)next_u32() * (uint64_t)n;
            if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
        }
    }
    template <size_t N>
    int weighted_index(const std::array<float, N>& w) {
        float cum[N > 1 ? N - 1 : 1];
        float total = w[0];
        for (size_t c = 1; c < N; c++) {
            cum[c - 1] = total;
            total += w[c];
        }
        const float unit = bits_f32((next_u32() >> 9) | 0x3F800000u) - 1.0f;   // [0, 1)
        const float chosen = unit * total + 0.0f;                            // Uniform::new(0, total).sample
        int idx = 0;
        for (size_t c = 0; c + 1 < N; c++)
            if (cum[c] <= chosen) idx = (int)c + 1;
        return idx;
    }

private:
    static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    void refill() {
        const uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key_[0], key_[1], key_[2], key_[3],
                                 key_[4], key_[5], key_[6], key_[7], (uint32_t)drawn_blocks_, (uint32_t)(drawn_blocks_ >> 32), 0u, 0u};
        uint32_t x[16];
        for (int k = 0; k < 16; k++) x[k] = in[k];
        auto quarter = [&x](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        };
        for (int r = 0; r < 6; r++) {
            quarter(0, 4, 8, 12); quarter(1, 5, 9, 13); quarter(2, 6, 10, 14); quarter(3, 7, 11, 15);
            quarter(0, 5, 10, 15); quarter(1, 6, 11, 12); quarter(2, 7, 8, 13); quarter(3, 4, 9, 14);
        }
        for (int k = 0; k < 16; k++) buf_[k] = x[k] + in[k];
        drawn_blocks_++;
        pos_ = 0;
    }
    uint32_t key_[8];
    uint32_t buf_[16];
    uint64_t drawn_blocks_ = 0;
    int pos_ = 16;
};

// ---- PolicyNoise::Dirichlet (mcts.rs:241-256: Dirichlet::new_with_size(alpha, n).sample(&mut thread_rng)) ---------------------
// rand_distr 0.4's algorithms — Dirichlet = normalised Gamma(alpha, 1) draws, Gamma by Marsaglia-Tsang (shape < 1 boosted by
// U^(1/shape), shape = 1 by inversion), its standard normal by the 256-layer ziggurat — on a StdRng per tree seeded
// tree_seed ^ NOISE_DIRICHLET_TAG instead of thread_rng; exp / ln through deterministic restatements. The same definition as
// csrc/noise.cuh: the device path and the host trees draw the same sample for the same tree.
constexpr uint64_t NOISE_DIRICHLET_TAG = 0xD1A1C4E7D1A1C4E7ull;
inline double bits_f64(uint64_t u) { double f; std::memcpy(&f, &u, 8); return f; }
inline uint64_t f64_bits(double f) { uint64_t u; std::memcpy(&u, &f, 8); return u; }
inline double det_exp64(double x) {
    if (x != x) return x;
    if (x < -700.0) return 0.0;
    if (x > 700.0) return bits_f64(0x7FF0000000000000ull);
    const double k = std::rint(x * 1.4426950408889634);
    double r = std::fma(k, -6.93147180369123816490e-01, x);
    r = std::fma(k, -1.90821492927058770002e-10, r);
    // exp(r) on |r| <= ln2 / 2: Taylor to r^13, Horner
    static const double inv_fact[12] = {1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
                                        1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0};
    double p = 1.0 / 6227020800.0;
    for (int i = 0; i < 12; i++) p = std::fma(p, r, inv_fact[i]);
    p = std::fma(p, r, 1.0);
    const long long ki = (long long)k;
    const double s1 = bits_f64((uint64_t)(1023 + ki / 2) << 52), s2 = bits_f64((uint64_t)(1023 + (ki - ki / 2)) << 52);
    return p * s1 * s2;
}
inline double det_log64(double x) {
    if (x != x || x < 0.0) return bits_f64(0x7FF8000000000000ull);
    if (x == 0.0) return bits_f64(0xFFF0000000000000ull);
    uint64_t b = f64_bits(x);
    if (b == 0x7FF0000000000000ull) return x;
    int e = 0;
    if (b < 0x0010000000000000ull) {
        x = x * 4503599627370496.0;
        b = f64_bits(x);
        e = -52;
    }
    e += (int)(b >> 52) - 1023;
    double m = bits_f64((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e += 1;
    }
    // ln m = 2 atanh(s), s = (m - 1) / (m + 1): odd series to s^25
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 25.0;
    for (int d = 23; d >= 3; d -= 2) p = std::fma(p, z, 1.0 / (double)d);
    p = std::fma(p, z, 1.0);
    const double lm = 2.0 * s * p;
    const double fe = (double)e;
    return std::fma(fe, 6.93147180369123816490e-01, std::fma(fe, 1.90821492927058770002e-10, lm));
}
class NoiseRng {
public:
    explicit NoiseRng(uint64_t seed) : g_(seed) {}
    float gamma(float shape) {   // gamma.rs Gamma::new(shape, 1.0).sample
        if (shape == 1.0f) return (float)(-det_log64(open01_f64())) * 1.0f;
        if (shape < 1.0f) {
            const float inv_shape = 1.0f / shape;
            const float s1 = shape + 1.0f;
            const float d = s1 - 0.33333334f, c = 1.0f / std::sqrt(9.0f * d);
            const float u = open01_f32();
            return gamma_large(d, c) * det_expf(inv_shape * det_logf(u));
        }
        const float d = shape - 0.33333334f, c = 1.0f / std::sqrt(9.0f * d);
        return gamma_large(d, c);
    }

private:
    double gen_f64() { return (double)(g_.next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    double open01_f64() { return bits_f64((g_.next_u64() >> 12) | 0x3FF0000000000000ull) - (1.0 - 1.1102230246251565e-16); }
    float open01_f32() { return bits_f32((g_.next_u32() >> 9) | 0x3F800000u) - (1.0f - 5.9604645e-08f); }
    double standard_normal() {   // normal.rs StandardNormal + utils.rs ziggurat (symmetric)
        for (;;) {
            const uint64_t bits = g_.next_u64();
            const uint32_t i = (uint32_t)bits & 0xFFu;
            const double u = bits_f64((bits >> 12) | 0x4000000000000000ull) - 3.0;
            const double x = u * ZIG_NORM_X[i];
            const double ax = x < 0.0 ? -x : x;
            if (ax < ZIG_NORM_X[i + 1]) return x;
            if (i == 0u) {   // the tail beyond R
                double tx = 1.0, ty = 0.0;
                while (-2.0 * ty < tx * tx) {
                    const double a = open01_f64();
                    const double b = open01_f64();
                    tx = det_log64(a) / ZIG_NORM_R;
                    ty = det_log64(b);
                }
                return u < 0.0 ? tx - ZIG_NORM_R : ZIG_NORM_R - tx;
            }
            if (ZIG_NORM_F[i + 1] + (ZIG_NORM_F[i] - ZIG_NORM_F[i + 1]) * gen_f64() < det_exp64(-x * x / 2.0)) return x;
        }
    }
    float gamma_large(float d, float c) {   // GammaLargeShape::sample, scale 1
        for (;;) {
            const float x = (float)standard_normal();
            const float v_cbrt = 1.0f + c * x;
            if (v_cbrt <= 0.0f) continue;
            const float v = v_cbrt * v_cbrt * v_cbrt;
            const float u = open01_f32();
            const float x_sqr = x * x;
            if (u < 1.0f - 0.0331f * x_sqr * x_sqr || det_logf(u) < 0.5f * x_sqr + d * (1.0f - v + det_logf(v))) return d * v * 1.0f;
        }
    }
    StdRng g_;
};
// the root's noise vector: n Gamma(alpha) draws divided by their sum (dirichlet.rs), in child order
inline void noise_dirichlet(uint64_t tree_seed, float alpha, uint32_t n, float* out) {
    NoiseRng g(tree_seed ^ NOISE_DIRICHLET_TAG);
    float sum = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        out[i] = g.gamma(alpha);
        sum += out[i];
    }
    const float invacc = 1.0f / sum;
    for (uint32_t i = 0; i < n; i++) out[i] = out[i] * invacc;
}

// Host threads this process may actually run at once: the hardware concurrency, cut to a cgroup CPU quota if there is one (a
// container with 16 CPUs' worth of time on a 256-thread host: more runnable threads than quota get throttled in the middle of a
// round and every other thread waits for them at the round's end).
inline int usable_host_threads() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
        long long quota = 0, period = 0;
        if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
            n = std::min(n, (unsigned)std::max(1ll, (quota + period - 1) / period));
        std::fclose(f);
    } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
        long long quota = 0, period = 100000;
        const bool have = std::fscanf(g, "%lld", &quota) == 1;
        std::fclose(g);
        if (FILE* pf = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(pf, "%lld", &period) != 1) period = 100000;
            std::fclose(pf);
        }
        if (have && quota > 0 && period > 0) n = std::min(n, (unsigned)std::max(1ll, (quota + period - 1) / period));
    }
    return (int)std::min(32u, n);
}

// A fixed set of host threads that run fn(i) for i in [0, n) (trees are independent): the indices are handed out in small blocks
// from a shared counter, so a thread that drew cheap trees takes more of them. One pool lives for a whole search: its phase per
// round would otherwise start and join `threads` threads explores + 1 times.
class WorkerPool {
public:
    explicit WorkerPool(int threads) : nthreads_((size_t)(threads < 1 ? 1 : threads)) {
        for (size_t k = 1; k < nthreads_; k++) workers_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            generation_++;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;

    template <class F>
    void run(size_t n, F&& fn) {
        if (nthreads_ <= 1 || n < 64) {
            for (size_t i = 0; i < n; i++) fn(i);
            return;
        }
        std::function<void(size_t)> f = std::ref(fn);
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &f;
            n_ = n;
            grain_ = std::max<size_t>(1, n / (nthreads_ * 8));
            next_.store(0, std::memory_order_relaxed);
            pending_ = nthreads_ - 1;
            generation_++;
        }
        cv_.notify_all();
        std::exception_ptr mine = nullptr;
        try {
            drain(f, n, grain_);
        } catch (...) {
            mine = std::current_exception();   // (the workers still hold a pointer to f: wait for them before unwinding)
        }
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
        if (mine && !error_) error_ = mine;
        job_ = nullptr;
        if (error_) {
            std::exception_ptr e = error_;
            error_ = nullptr;
            std::rethrow_exception(e);
        }
    }

private:
    void drain(const std::function<void(size_t)>& f, size_t n, size_t grain) {
        for (;;) {
            const size_t b = next_.fetch_add(grain, std::memory_order_relaxed);
            if (b >= n) return;
            const size_t e = std::min(n, b + grain);
            for (size_t i = b; i < e; i++) f(i);
        }
    }
    void loop() {
        size_t seen = 0;
        for (;;) {
            const std::function<void(size_t)>* f;
            size_t n, grain;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                f = job_;
                n = n_;
                grain = grain_;
            }
            try {
                drain(*f, n, grain);
            } catch (...) {
                next_.store(n, std::memory_order_relaxed);   // nobody starts another block of a failed phase
                std::lock_guard<std::mutex> lk(mu_);
                if (!error_) error_ = std::current_exception();
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                pending_--;
            }
            done_.notify_one();
        }
    }
    const size_t nthreads_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)>* job_ = nullptr;
    size_t n_ = 0, grain_ = 1, pending_ = 0, generation_ = 0;
    std::atomic<size_t> next_{0};
    bool stop_ = false;
    std::exception_ptr error_ = nullptr;
};

}  // namespace detail

// ---- game.rs:9-62 ---------------------------------------------------------------------------------------------------------
struct Outcome {
    enum Kind : uint8_t { Lose = 0, Draw = 1, Win = 2 };  // (the index of Into<usize>, mcts.rs:10-18)
    Kind kind = Draw;
    uint16_t turns = 0;   // (usize in the reference; a game is over after Game::MAX_TURNS plies — 16 bits keep a tree node in one cache line)

    static Outcome from_reward(float value) {  // impl From<f32>
        return Outcome{value > 0.0f ? Win : (value < 0.0f ? Lose : Draw), 0};
    }
    Outcome reversed() const { return Outcome{kind == Win ? Lose : (kind == Lose ? Win : Draw), (uint16_t)(turns + 1)}; }
    float value() const { return kind == Win ? 1.0f : (kind == Draw ? 0.0f : -1.0f); }
    // impl Ord: a win in fewer turns is greater; draws and losses in more turns are greater; Win > Draw > Lose
    static int cmp(const Outcome& a, const Outcome& b) {
        if (a.kind != b.kind) return a.kind < b.kind ? -1 : 1;
        if (a.turns == b.turns) return 0;
        if (a.kind == Win) return b.turns < a.turns ? -1 : 1;
        return a.turns < b.turns ? -1 : 1;
    }
    bool operator==(const Outcome& o) const { return kind == o.kind && turns == o.turns; }
};

struct Solution {  // Option<Outcome>; None < Some(_)
    bool some = false;
    Outcome outcome;
    static Solution max(const Solution& a, const Solution& b) {
        if (!a.some) return b;
        if (!b.some) return a;
        return Outcome::cmp(b.outcome, a.outcome) >= 0 ? b : a;
    }
};

// ---- policies/traits.rs:4-6, for a batch ------------------------------------------------------------------------------------
template <class G, int N>
struct BatchPolicy {
    virtual ~BatchPolicy() = default;
    // logits[i][0..N), value[i][0..3) = policy.eval(*games[i])
    virtual void eval_batch(const std::vector<const G*>& games, float* logits, float* value) = 0;
    // The same call in two parts, for a policy that computes elsewhere (a GPU): begin() may return before the answers exist,
    // end() returns when logits / value of the last begin() are filled. One batch in flight at a time. The drivers evaluate one half
    // of a thread's trees this way while the thread advances the other half. (Default: begin() does it all.)
    virtual void eval_batch_begin(const std::vector<const G*>& games, float* logits, float* value) { eval_batch(games, logits, value); }
    virtual void eval_batch_end() {}
};

// ---- one tree ---------------------------------------------------------------------------------------------------------------
template <class G, int N>
class LockstepTree {
public:
    struct Node {  // mcts.rs:28-39 (field order: no padding — 64 bytes for Connect4, a node per cache line's worth)
        uint32_t parent = 0, first_child = 0;
        G game;
        float outcome_probs[3] = {0.0f, 0.0f, 0.0f};
        float num_visits = 0.0f;
        float action_prob = 0.0f;
        Solution solution;
        uint8_t num_children = 0;
        uint8_t action = 0;

        float q() const { return (outcome_probs[2] - outcome_probs[0]) / num_visits; }
        bool is_unvisited() const { return num_children == 0 && !solution.some; }
        uint32_t last_child() const { return first_child + num_children; }
    };

    // MCTS::with_capacity(explores + 1, cfg, policy, game) followed by explore_n(explores): nothing runs before advance()
    // noise_seed: detail::noise_tree_seed(stream, turn) of this tree — read by Fpu::Normal and PolicyNoise::Dirichlet
    LockstepTree(const MCTSConfig& cfg, const G& game, int explores, uint64_t noise_seed = 0)
        : cfg_(cfg), explores_(explores), noise_seed_(noise_seed) {
        if (cfg.root_policy_noise == PolicyNoise::Dirichlet && !(cfg.noise_alpha > 0.0f))
            throw Error(SYN_ERR_INVALID_ARGUMENT, "PolicyNoise::Dirichlet needs alpha > 0");
        nodes_.reserve((size_t)explores + 1);
        Node root;
        root.game = game;
        nodes_.push_back(root);
    }

    // The next move's tree in the same storage (run_game builds a fresh MCTS per move, alpha_zero.rs:240-241): the node arena keeps
    // its capacity, so a game allocates once.
    void reset(const G& game, uint64_t noise_seed = 0) {
        noise_seed_ = noise_seed;
        fpu_scans_ = 0;
        nodes_.clear();
        Node root;
        root.game = game;
        nodes_.push_back(root);
        done_ = 0;
        constructed_ = false;
        pending_ = 0;
        pending_any_solved_ = false;
    }

    // Runs this tree until it stands on a leaf whose position the policy has to evaluate — returns that position; supply() must be
    // called before the next advance() — or until the search is over (nullptr).
    const G* advance() {
        for (;;) {
            uint32_t node_id;
            if (!constructed_) {
                node_id = 0;  // with_capacity: visit(root) (mcts.rs:133)
            } else {
                // explore_n (mcts.rs:139-147): a solved root ends the search
                if (done_ >= explores_ || nodes_[0].solution.some) return nullptr;
                done_++;
                // explore (mcts.rs:310-325)
                node_id = 0;
                bool handled = false;
                for (;;) {
                    const Node& node = nodes_[node_id];
                    if (node.solution.some) {
                        float probs[3] = {0.0f, 0.0f, 0.0f};
                        probs[node.solution.outcome.kind] = 1.0f;
                        backprop(node_id, probs, true);
                        handled = true;
                        break;
                    }
                    if (node.is_unvisited()) break;
                    node_id = select_best_child(node);
                }
                if (handled) continue;
            }
            // visit (mcts.rs:374-430) up to the policy call
            for (;;) {
                const uint32_t first_child = (uint32_t)nodes_.size();
                if (nodes_[node_id].solution.some) {
                    float probs[3] = {0.0f, 0.0f, 0.0f};
                    probs[nodes_[node_id].solution.outcome.kind] = 1.0f;
                    backprop(node_id, probs, true);
                    finish_construction();
                    node_id = UINT32_MAX;
                    break;
                }
                const G game = nodes_[node_id].game;
                uint8_t num_children = 0;
                bool any_solved = false;
                for (int action : game.iter_actions()) {
                    Node child;
                    child.parent = node_id;
                    child.game = game;
                    const bool is_over = child.game.step(action);
                    if (is_over) {
                        any_solved = true;
                        child.solution.some = true;
                        child.solution.outcome = Outcome::from_reward(child.game.reward(child.game.player()));
                    }
                    child.action = (uint8_t)action;
                    child.action_prob = 1.0f;
                    nodes_.push_back(child);
                    num_children++;
                }
                nodes_[node_id].first_child = first_child;
                nodes_[node_id].num_children = num_children;
                if (cfg_.auto_extend && num_children == 1) {
                    node_id = first_child;  // `return self.visit(first_child)`: the outer call's any_solved is dropped with it
                    continue;
                }
                pending_ = node_id;
                pending_any_solved_ = any_solved;
                return &nodes_[node_id].game;
            }
            (void)node_id;
        }
    }

    // The rest of visit() for the position advance() returned: softmax of the children's logits (mcts.rs:407-427), backprop.
    void supply(const float* logits, const float* value) {
        Node& node = nodes_[pending_];
        const uint32_t first = node.first_child, last = node.last_child();
        float max_logit = -std::numeric_limits<float>::infinity();
        for (uint32_t c = first; c < last; c++) {
            const float logit = logits[nodes_[c].action];
            max_logit = std::fmax(max_logit, logit);  // f32::max: a NaN operand is ignored
            nodes_[c].action_prob = logit;
        }
        float total = 0.0f;
        for (uint32_t c = first; c < last; c++) {
            nodes_[c].action_prob = detail::det_expf(nodes_[c].action_prob - max_logit);
            total += nodes_[c].action_prob;
        }
        for (uint32_t c = first; c < last; c++) nodes_[c].action_prob /= total;
        float probs[3] = {value[0], value[1], value[2]};
        backprop(pending_, probs, pending_any_solved_);
        finish_construction();
    }

    // ---- what a caller reads off the finished tree (mcts.rs:174-306) ----
    size_t num_nodes() const { return nodes_.size(); }
    const Node& root() const { return nodes_[0]; }
    const Node* children_begin() const { return nodes_.data() + nodes_[0].first_child; }
    const Node* children_end() const { return nodes_.data() + nodes_[0].last_child(); }

    int best_action(ActionSelection sel) const {
        bool have = false;
        float b0 = 0.0f, b1 = 0.0f;
        int best = -1;
        for (const Node* c = children_begin(); c != children_end(); ++c) {
            float v0, v1;
            if (c->solution.some && c->solution.outcome.kind == Outcome::Win) { v0 = 0.0f; v1 = (float)c->solution.outcome.turns; }
            else if (!c->solution.some) { v0 = 1.0f; v1 = sel == ActionSelection::Q ? -c->q() : c->num_visits; }
            else if (c->solution.outcome.kind == Outcome::Draw) { v0 = 2.0f; v1 = -(float)c->solution.outcome.turns; }
            else { v0 = 3.0f; v1 = -(float)c->solution.outcome.turns; }
            // Some((v0, v1)) > best_value: lexicographic partial order of the pair, None below everything
            const bool greater = !have || (v0 != b0 ? v0 > b0 : v1 > b1);
            if (greater) { have = true; b0 = v0; b1 = v1; best = c->action; }
        }
        return best;
    }

    Solution solution(int action) const {
        for (const Node* c = children_begin(); c != children_end(); ++c)
            if (c->action == action) return c->solution;
        return Solution{};
    }

    std::array<float, N> target_policy() const {
        std::array<float, N> pi{};
        float total = 0.0f;
        const Node& r = nodes_[0];
        if (r.num_visits == 1.0f) {
            const bool win = r.solution.some && r.solution.outcome.kind == Outcome::Win;
            for (const Node* c = children_begin(); c != children_end(); ++c) {
                const float v = win ? ((c->solution.some && c->solution.outcome.kind == Outcome::Lose) ? 1.0f : 0.0f) : 1.0f;
                pi[c->action] = v;
                total += v;
            }
        } else {
            for (const Node* c = children_begin(); c != children_end(); ++c) {
                pi[c->action] = c->num_visits;
                total += c->num_visits;
            }
        }
        for (int i = 0; i < N; i++) pi[i] /= total;
        return pi;
    }

    std::array<float, 3> target_q() const {
        const Node& r = nodes_[0];
        std::array<float, 3> q{};
        if (r.solution.some) q[r.solution.outcome.kind] = 1.0f;
        else
            for (int i = 0; i < 3; i++) q[i] = r.outcome_probs[i] / r.num_visits;
        return q;
    }

private:
    void finish_construction() {
        if (constructed_) return;
        constructed_ = true;
        // add_root_noise (mcts.rs:229-269)
        if (cfg_.root_policy_noise == PolicyNoise::Equal && nodes_[0].num_children >= 2) {
            const float w = cfg_.noise_weight, noise = 1.0f / (float)nodes_[0].num_children;
            for (uint32_t c = nodes_[0].first_child; c < nodes_[0].last_child(); c++)
                nodes_[c].action_prob = nodes_[c].action_prob * (1.0f - w) + w * noise;
        }
        if (cfg_.root_policy_noise == PolicyNoise::Dirichlet && nodes_[0].num_children >= 2) {   // mcts.rs:241-256
            float noise[N];
            detail::noise_dirichlet(noise_seed_, cfg_.noise_alpha, nodes_[0].num_children, noise);
            uint32_t k = 0;
            for (uint32_t c = nodes_[0].first_child; c < nodes_[0].last_child(); c++, k++)
                nodes_[c].action_prob = nodes_[c].action_prob * (1.0f - cfg_.noise_weight) + cfg_.noise_weight * noise[k];
        }
    }

    uint32_t select_best_child(const Node& parent) {  // mcts.rs:327-341: the first maximum wins, NaN never replaces
        uint32_t best = 0;
        bool have = false, drew = false;
        float best_value = 0.0f;
        // the parent's share of explore_value (mcts.rs:364, 368) is the same for every child: once per scan
        const float visits = cfg_.exploration == Exploration::Uct ? std::sqrt(cfg_.c * detail::det_logf(parent.num_visits))
                                                                  : std::sqrt(parent.num_visits);
        for (uint32_t id = parent.first_child; id < parent.last_child(); id++) {
            const Node& child = nodes_[id];
            const float value = exploit_value(parent, child, id - parent.first_child, drew) + explore_value(visits, child);
            if (!have || value > best_value) {
                have = true;
                best = id;
                best_value = value;
            }
        }
        if (drew) fpu_scans_++;  // a scan that took at least one Fpu::Normal draw uses up one scan number of the tree's stream
        return best;
    }

    float exploit_value(const Node& parent, const Node& child, uint32_t slot, bool& drew) const {  // mcts.rs:343-359
        if (child.solution.some)
            return cfg_.select_solved_nodes ? child.solution.outcome.reversed().value() : -std::numeric_limits<float>::infinity();
        if (child.num_children == 0) {
            if (cfg_.fpu == Fpu::Const) return cfg_.fpu_value;
            if (cfg_.fpu == Fpu::ParentQ) return parent.q();
            if (cfg_.fpu == Fpu::Func) return cfg_.fpu_fn();  // Fpu::Func(fpu_fn) => fpu_fn() (mcts.rs:354): the caller's function, its own state
            drew = true;  // Fpu::Func(|| Normal(mean, std)) (main.rs:43-47)
            return detail::noise_fpu_normal(noise_seed_, fpu_scans_, slot, cfg_.fpu_value, cfg_.fpu_std);
        }
        return -child.q();
    }

    float explore_value(float visits, const Node& child) const {  // mcts.rs:361-372
        if (cfg_.exploration == Exploration::Uct) return visits / std::sqrt(child.num_visits);
        return cfg_.c * child.action_prob * visits / (1.0f + child.num_visits);
    }

    void backprop(uint32_t leaf, float (&outcome_probs)[3], bool solved) {  // mcts.rs:432-488
        uint32_t node_id = leaf;
        for (;;) {
            Node& node = nodes_[node_id];
            const uint32_t parent = node.parent;
            if (cfg_.solve && solved) {
                bool all_solved = true;
                Solution best = node.solution;
                for (uint32_t c = node.first_child; c < node.last_child(); c++) {
                    Solution s = nodes_[c].solution;
                    if (s.some) s.outcome = s.outcome.reversed();
                    all_solved = all_solved && s.some;
                    best = Solution::max(best, s);
                }
                if (best.some && best.outcome.kind == Outcome::Win) {
                    node.solution = best;
                    if (cfg_.correct_values_on_solve) {
                        for (int i = 0; i < 3; i++) outcome_probs[i] = -node.outcome_probs[i];
                        outcome_probs[2] += node.num_visits + 1.0f;
                    }
                } else if (best.some && all_solved) {
                    node.solution = best;
                    if (cfg_.correct_values_on_solve) {
                        for (int i = 0; i < 3; i++) outcome_probs[i] = -node.outcome_probs[i];
                        outcome_probs[best.outcome.kind == Outcome::Draw ? 1 : 0] += node.num_visits + 1.0f;
                    }
                } else {
                    solved = false;
                }
            }
            for (int i = 0; i < 3; i++) node.outcome_probs[i] += outcome_probs[i];
            node.num_visits += 1.0f;
            if (node_id == 0) break;
            std::swap(outcome_probs[0], outcome_probs[2]);
            node_id = parent;
        }
    }

    MCTSConfig cfg_;
    int explores_ = 0, done_ = 0;
    uint64_t noise_seed_ = 0;  // Fpu::Normal: the tree's seed and the number of scans that drew so far
    uint32_t fpu_scans_ = 0;
    bool constructed_ = false;
    uint32_t pending_ = 0;
    bool pending_any_solved_ = false;
    std::vector<Node> nodes_;
};

namespace detail {
// The round structure of every driver below. Units [first, first + count) are trees (or game slots): step_unit(u, logits, value)
// hands unit u the answer for its last leaf (nullptr: it had none) and runs it to its next leaf (returned) or to its end
// (nullptr). With a pool: per round one phase over all live units, then one eval_batch with their leaves. Without (one thread):
// the units form two halves that take turns — while one half's leaves are with the policy (eval_batch_begin ... _end), the
// thread runs the other half. Units never interact, so the schedule changes no result.
template <class G, int N, class StepUnit>
void run_rounds(BatchPolicy<G, N>& policy, uint32_t first, uint32_t count, WorkerPool* pool, StepUnit&& step_unit, size_t& rounds,
                size_t& evals) {
    struct Half {
        std::vector<uint32_t> live;   // units still running; after step(): the ones standing on a leaf, in batch order
        std::vector<const G*> batch;
        std::vector<float> logits, value;
        bool have_results = false;
    };
    Half halves[2];
    const uint32_t split = (pool == nullptr && count >= 16) ? count / 2 : count;
    for (uint32_t i = 0; i < count; i++) halves[i < split ? 0 : 1].live.push_back(first + i);
    std::vector<const G*> want(count, nullptr);
    auto step = [&](Half& h) {
        auto body = [&](size_t k) {
            const uint32_t u = h.live[k];
            want[u - first] = step_unit(u, h.have_results ? &h.logits[k * (size_t)N] : nullptr, h.have_results ? &h.value[k * 3] : nullptr);
        };
        if (pool) pool->run(h.live.size(), body);
        else
            for (size_t k = 0; k < h.live.size(); k++) body(k);
        size_t kept = 0;
        h.batch.clear();
        for (uint32_t u : h.live)
            if (want[u - first]) {
                h.batch.push_back(want[u - first]);
                h.live[kept++] = u;
            }
        h.live.resize(kept);
        h.logits.resize(kept * (size_t)N);
        h.value.resize(kept * 3);
        h.have_results = false;
    };
    step(halves[0]);
    if (halves[1].live.empty()) {
        Half& h = halves[0];
        while (!h.live.empty()) {
            rounds++;
            evals += h.batch.size();
            policy.eval_batch(h.batch, h.logits.data(), h.value.data());
            h.have_results = true;
            step(h);
        }
        return;
    }
    step(halves[1]);
    bool in_flight = false;
    auto begin = [&](Half& h) {
        rounds++;
        evals += h.batch.size();
        policy.eval_batch_begin(h.batch, h.logits.data(), h.value.data());
        in_flight = true;
    };
    auto end = [&](Half& h) {
        in_flight = false;
        policy.eval_batch_end();
        h.have_results = true;
    };
    try {
        if (!halves[0].live.empty()) begin(halves[0]);
        for (int cur = 0;; cur ^= 1) {
            Half& a = halves[cur];        // with the policy (if it has leaves at all)
            Half& b = halves[cur ^ 1];    // its leaves are ready
            if (!a.live.empty()) end(a);
            if (!b.live.empty()) begin(b);
            if (!a.live.empty()) step(a);  // beside b's evaluation
            if (a.live.empty() && b.live.empty()) break;
        }
    } catch (...) {
        if (in_flight) try { policy.eval_batch_end(); } catch (...) {}   // leave the policy with nothing in flight
        throw;
    }
}

// fn(s) for s in [0, shards), every shard on its own thread (shard 0 on the caller's); the first exception is rethrown after all
// have returned.
template <class F>
void run_shards(size_t shards, F&& fn) {
    std::vector<std::exception_ptr> errors(shards, nullptr);
    std::vector<std::thread> threads;
    auto guarded = [&](size_t s) {
        try {
            fn(s);
        } catch (...) {
            errors[s] = std::current_exception();
        }
    };
    for (size_t s = 1; s < shards; s++) threads.emplace_back(guarded, s);
    if (shards > 0) guarded(0);
    for (auto& t : threads) t.join();
    for (auto& e : errors)
        if (e) std::rethrow_exception(e);
}
}  // namespace detail

// `explores` explores from every root: MCTS::with_capacity + explore_n (mcts.rs:123-147) for all of them, the leaves batched.
// threads = host threads for the trees (0: what this process may use, at most 32): with more than one, a round is one phase over
// all trees on a pool and one eval_batch with their leaves; with one, the trees take turns in two halves (run_rounds above).
// rounds_out / evals_out: eval_batch calls and positions evaluated. noise_stream: root i's Fpu::Normal draws come from stream
// noise_stream + i, turn 0 — syn_mcts_search's numbering with noise_stream = 0.
template <class G, int N>
std::vector<LockstepTree<G, N>> lockstep_search(BatchPolicy<G, N>& policy, const MCTSConfig& cfg, const std::vector<G>& roots,
                                                int explores, int threads = 0, size_t* rounds_out = nullptr,
                                                size_t* evals_out = nullptr, uint64_t noise_stream = 0) {
    if (threads <= 0) threads = detail::usable_host_threads();
    std::vector<LockstepTree<G, N>> trees;
    trees.reserve(roots.size());
    for (const G& g : roots) trees.emplace_back(cfg, g, explores, detail::noise_tree_seed(noise_stream + (uint64_t)trees.size(), 0u));
    size_t rounds = 0, evals = 0;
    auto step_tree = [&trees](uint32_t t, const float* logits, const float* value) {
        if (logits) trees[t].supply(logits, value);   // the rest of visit(): softmax, backprop
        return trees[t].advance();
    };
    if (threads > 1) {
        detail::WorkerPool pool(threads);
        detail::run_rounds<G, N>(policy, 0u, (uint32_t)trees.size(), &pool, step_tree, rounds, evals);
    } else {
        detail::run_rounds<G, N>(policy, 0u, (uint32_t)trees.size(), nullptr, step_tree, rounds, evals);
    }
    if (rounds_out) *rounds_out = rounds;
    if (evals_out) *evals_out = evals;
    return trees;
}

// The same with one policy object PER HOST THREAD — the reference's worker model (alpha_zero.rs:192-198: every worker thread owns
// its policy): the roots are split into policies.size() contiguous shards, every shard runs on its own thread over its own policy
// (two halves taking turns), nothing is shared and no thread waits for another. Results as lockstep_search's.
template <class G, int N>
std::vector<LockstepTree<G, N>> lockstep_search_sharded(const std::vector<BatchPolicy<G, N>*>& policies, const MCTSConfig& cfg,
                                                        const std::vector<G>& roots, int explores, size_t* rounds_out = nullptr,
                                                        size_t* evals_out = nullptr, uint64_t noise_stream = 0) {
    if (policies.empty()) throw Error(SYN_ERR_INVALID_ARGUMENT, "lockstep_search_sharded: no policies");
    std::vector<LockstepTree<G, N>> trees;
    trees.reserve(roots.size());
    for (const G& g : roots) trees.emplace_back(cfg, g, explores, detail::noise_tree_seed(noise_stream + (uint64_t)trees.size(), 0u));
    const size_t shards = policies.size(), n = trees.size();
    std::vector<size_t> rounds(shards, 0), evals(shards, 0);
    auto step_tree = [&trees](uint32_t t, const float* logits, const float* value) {
        if (logits) trees[t].supply(logits, value);
        return trees[t].advance();
    };
    detail::run_shards(shards, [&](size_t s) {
        const size_t lo = n * s / shards, hi = n * (s + 1) / shards;
        detail::run_rounds<G, N>(*policies[s], (uint32_t)lo, (uint32_t)(hi - lo), nullptr, step_tree, rounds[s], evals[s]);
    });
    if (rounds_out) *rounds_out = std::accumulate(rounds.begin(), rounds.end(), (size_t)0);
    if (evals_out) *evals_out = std::accumulate(evals.begin(), evals.end(), (size_t)0);
    return trees;
}

// ---- run_n_games over host trees (alpha_zero.rs:181-338) --------------------------------------------------------------------
// What ReplayBuffer::add stores for one game (data.rs:151-158) plus the bookkeeping syn_selfplay_run reports.
template <class G, int N>
struct LockstepGameRecord {
    std::vector<G> states;
    std::vector<std::array<float, N>> pis;
    std::vector<std::array<float, 3>> vs;       // after store_rewards
    std::vector<uint8_t> actions;
    std::vector<uint32_t> root_nodes;            // nodes.len() of each move's tree
    Outcome final_outcome;                       // for the side to move in the final position (alpha_zero.rs:258)
};

namespace detail {
// run_n_games for one worker (alpha_zero.rs:181-209) over host trees: `slots` games in flight, each on its own tree; a game whose
// search is over plays its move (run_game's loop body, alpha_zero.rs:246-264), starts the next move's tree and goes on until it
// stands on a leaf; a game that ends hands its slot to the next game index of the run (a counter shared by all workers).
template <class G, int N>
class SelfplayWorker {
public:
    struct Shared {   // one run: the games [first_game, first_game + num_games), their records, the next index to start
        const RolloutConfig& cfg;
        uint64_t seed, first_game;
        size_t num_games;
        std::vector<LockstepGameRecord<G, N>>& out;
        std::atomic<size_t> next{0};
    };
    // The reference's own discipline (alpha_zero.rs:140,189,201-205; gather_experience_host_trees below): this worker plays the games
    // [start, start + count) of the run ONE AFTER ANOTHER on one StdRng::seed_from_u64(worker_seed) that runs through all of them.
    SelfplayWorker(Shared& shared, size_t start, size_t count, uint64_t worker_seed)
        : shared_(shared), cfg_(shared.cfg), sequential_(true), seq_next_(start), seq_end_(start + count), seq_count_(count),
          worker_seed_(worker_seed) {
        if (count == 0) return;
        plays_.emplace_back(cfg_.mcts_cfg, cfg_.num_explores);
        plays_.back().rng = StdRng(worker_seed);
        plays_.back().start_keeping_rng(seq_next_, (worker_seed << 32) + 0u);
        seq_next_++;
    }
    SelfplayWorker(Shared& shared, size_t slots) : shared_(shared), cfg_(shared.cfg) {
        plays_.reserve(slots);
        for (size_t i = 0; i < slots; i++) {
            const size_t g = shared_.next.fetch_add(1, std::memory_order_relaxed);
            if (g >= shared_.num_games) break;
            plays_.emplace_back(cfg_.mcts_cfg, cfg_.num_explores);
            plays_.back().start(g, shared_.seed + shared_.first_game + (uint64_t)g);
        }
    }
    size_t slots() const { return plays_.size(); }
    // run_rounds' step_unit: slot u takes the answer for its leaf and runs to its next leaf; nullptr: the slot has no game left
    const G* step(uint32_t u, const float* logits, const float* value) {
        Play& p = plays_[u];
        if (logits) p.tree.supply(logits, value);
        for (;;) {
            if (const G* leaf = p.tree.advance()) return leaf;
            play_move(p, shared_.out[p.index]);
            if (!p.over) {
                p.next_tree();
                continue;
            }
            if (sequential_) {
                if (seq_next_ >= seq_end_) return nullptr;
                p.start_keeping_rng(seq_next_, (worker_seed_ << 32) + (uint64_t)(seq_next_ - (seq_end_ - seq_count())));
                seq_next_++;
                continue;
            }
            const size_t g = shared_.next.fetch_add(1, std::memory_order_relaxed);
            if (g >= shared_.num_games) return nullptr;
            p.start(g, shared_.seed + shared_.first_game + (uint64_t)g);
        }
    }

private:
    struct StateInfo { int turn; float t; std::array<float, 3> q, z; };   // alpha_zero.rs:211-227
    struct Play {
        G game;
        size_t index = 0;     // which game of the run
        uint64_t stream = 0;  // seed + game index: its StdRng (alpha_zero.rs:189 per game) and its trees' noise streams
        StdRng rng;
        LockstepTree<G, N> tree;
        int num_turns = 0;
        bool over = false;
        std::vector<StateInfo> infos;
        Play(const MCTSConfig& m, int explores) : game(), rng(0), tree(m, G(), explores) {}
        void start(size_t g, uint64_t s) {
            game = G();
            index = g;
            stream = s;
            rng = StdRng(s);
            num_turns = 0;
            over = false;
            infos.clear();
            next_tree();
        }
        // the worker's generator runs on from the previous game (run_n_games passes ONE &mut rng to every run_game)
        void start_keeping_rng(size_t g, uint64_t noise_stream) {
            game = G();
            index = g;
            stream = noise_stream;
            num_turns = 0;
            over = false;
            infos.clear();
            next_tree();
        }
        // the next move's MCTS (run_game builds a fresh one per move, alpha_zero.rs:240-241) in the same arena
        void next_tree() { tree.reset(game, noise_tree_seed(stream, (uint32_t)num_turns)); }
    };

    // the part of run_game's loop body behind explore_n (alpha_zero.rs:246-264), then the game's end (266-267)
    void play_move(Play& p, LockstepGameRecord<G, N>& rec) const {
        const std::array<float, N> search_policy = p.tree.target_policy();
        rec.states.push_back(p.game);
        rec.pis.push_back(search_policy);
        rec.vs.push_back({0.0f, 0.0f, 0.0f});
        rec.root_nodes.push_back((uint32_t)p.tree.num_nodes());
        p.infos.push_back(StateInfo{p.num_turns + 1, 0.0f, p.tree.target_q(), {0.0f, 0.0f, 0.0f}});
        // sample_action (alpha_zero.rs:270-294)
        const int best = p.tree.best_action(cfg_.action);
        const Solution best_solution = p.tree.solution(best);
        int action;
        if (p.num_turns < cfg_.random_actions_until) {
            const std::vector<int> legal = p.game.iter_actions();
            action = legal[p.rng.gen_range_u8((uint32_t)legal.size())];
        } else if (p.num_turns < cfg_.sample_actions_until && (!best_solution.some || !cfg_.stop_games_when_solved)) {
            action = p.rng.weighted_index(search_policy);
        } else {
            action = best;
        }
        rec.actions.push_back((uint8_t)action);
        Solution solution = p.tree.solution(action);
        const bool is_over = p.game.step(action);
        if (is_over) {
            solution.some = true;
            solution.outcome = Outcome::from_reward(p.game.reward(p.game.player()));
        } else if (!cfg_.stop_games_when_solved) {
            solution.some = false;
        }
        p.num_turns++;
        if (!solution.some) return;
        p.over = true;
        rec.final_outcome = solution.outcome;
        // fill_state_info (alpha_zero.rs:296-307)
        const int num_turns = (int)p.infos.size();
        Outcome outcome = solution.outcome.reversed();
        for (int i = num_turns - 1; i >= 0; i--) {
            p.infos[(size_t)i].z[outcome.kind] = 1.0f;
            p.infos[(size_t)i].t = (float)p.infos[(size_t)i].turn / (float)num_turns;
            outcome = outcome.reversed();
        }
        // store_rewards (alpha_zero.rs:309-338)
        for (int i = 0; i < num_turns; i++) {
            const StateInfo& st = p.infos[(size_t)i];
            std::array<float, 3> v{};
            switch (cfg_.value_target) {
                case ValueTarget::Q: v = st.q; break;
                case ValueTarget::Z: v = st.z; break;
                case ValueTarget::QZaverage:
                    for (int k = 0; k < 3; k++) v[k] = st.q[k] * cfg_.value_target_p + st.z[k] * (1.0f - cfg_.value_target_p);
                    break;
                case ValueTarget::QtoZ: {
                    const float pp = (1.0f - st.t) * cfg_.value_target_from + st.t * cfg_.value_target_to;
                    for (int k = 0; k < 3; k++) v[k] = st.q[k] * (1.0f - pp) + st.z[k] * pp;
                    break;
                }
            }
            rec.vs[(size_t)i] = v;
        }
    }

    size_t seq_count() const { return seq_count_; }
    Shared& shared_;
    const RolloutConfig& cfg_;
    std::vector<Play> plays_;
    bool sequential_ = false;
    size_t seq_next_ = 0, seq_end_ = 0, seq_count_ = 0;
    uint64_t worker_seed_ = 0;
};
}  // namespace detail

// `num_games` games [first_game, first_game + num_games), `concurrent` of them in flight at a time (0: all): every game searches
// its current position on its own host tree, the leaves go to the policy in batches, a game whose search is over plays its move
// (sample_action on its own StdRng::seed_from_u64(seed + game index): the per-game seeding of syn_selfplay_run; the
// Fpu::Normal draws of move `turn` from the tree stream (seed + game index, turn)), starts the next move's tree and keeps going
// until it, too, stands on a leaf; a finished game's slot takes the next game of the run. threads as lockstep_search's.
// Identical, game for game and float for float, to syn_selfplay_run on the same policy (tests/test_lockstep.py) — a game depends
// on its index only, not on what runs beside it.
template <class G, int N>
std::vector<LockstepGameRecord<G, N>> lockstep_selfplay(BatchPolicy<G, N>& policy, const RolloutConfig& cfg, size_t num_games,
                                                        uint64_t seed, uint64_t first_game = 0, int threads = 0,
                                                        size_t* rounds_out = nullptr, size_t* evals_out = nullptr,
                                                        size_t concurrent = 0) {
    if (threads <= 0) threads = detail::usable_host_threads();
    std::vector<LockstepGameRecord<G, N>> out(num_games);
    typename detail::SelfplayWorker<G, N>::Shared shared{cfg, seed, first_game, num_games, out};
    detail::SelfplayWorker<G, N> worker(shared, concurrent == 0 ? num_games : std::min(concurrent, num_games));
    size_t rounds = 0, evals = 0;
    auto step = [&worker](uint32_t u, const float* logits, const float* value) { return worker.step(u, logits, value); };
    if (threads > 1) {
        detail::WorkerPool pool(threads);
        detail::run_rounds<G, N>(policy, 0u, (uint32_t)worker.slots(), &pool, step, rounds, evals);
    } else {
        detail::run_rounds<G, N>(policy, 0u, (uint32_t)worker.slots(), nullptr, step, rounds, evals);
    }
    if (rounds_out) *rounds_out = rounds;
    if (evals_out) *evals_out = evals;
    return out;
}

// The same with one policy object per host thread — gather_experience's worker model (alpha_zero.rs:132-154, 192-198): the
// `concurrent` slots are divided among policies.size() workers, every worker runs its slots on its own thread over its own policy
// (two halves taking turns) and draws the next game index from the run's shared counter when one of its games ends. No thread
// waits for another; the records are lockstep_selfplay's.
template <class G, int N>
std::vector<LockstepGameRecord<G, N>> lockstep_selfplay_sharded(const std::vector<BatchPolicy<G, N>*>& policies, const RolloutConfig& cfg,
                                                                size_t num_games, uint64_t seed, uint64_t first_game = 0,
                                                                size_t concurrent = 0, size_t* rounds_out = nullptr,
                                                                size_t* evals_out = nullptr) {
    if (policies.empty()) throw Error(SYN_ERR_INVALID_ARGUMENT, "lockstep_selfplay_sharded: no policies");
    std::vector<LockstepGameRecord<G, N>> out(num_games);
    typename detail::SelfplayWorker<G, N>::Shared shared{cfg, seed, first_game, num_games, out};
    const size_t shards = policies.size(), slots = concurrent == 0 ? num_games : std::min(concurrent, num_games);
    std::vector<size_t> rounds(shards, 0), evals(shards, 0);
    detail::run_shards(shards, [&](size_t s) {
        detail::SelfplayWorker<G, N> worker(shared, slots * (s + 1) / shards - slots * s / shards);
        auto step = [&worker](uint32_t u, const float* logits, const float* value) { return worker.step(u, logits, value); };
        detail::run_rounds<G, N>(*policies[s], 0u, (uint32_t)worker.slots(), nullptr, step, rounds[s], evals[s]);
    });
    if (rounds_out) *rounds_out = std::accumulate(rounds.begin(), rounds.end(), (size_t)0);
    if (evals_out) *evals_out = std::accumulate(evals.begin(), evals.end(), (size_t)0);
    return out;
}

// gather_experience as the reference runs it (alpha_zero.rs:120-209) over host trees: policies.size() = num_workers + 1 workers,
// worker i plays games_to_schedule / workers_left games ONE AFTER ANOTHER on its own thread, its own policy and ONE
// StdRng::seed_from_u64(seed * (num_workers + 1) + i) that runs through all of its games (run_n_games: alpha_zero.rs:189,201-205) —
// so, as in the reference, the games depend on the number of workers, and a worker has one game (one leaf) in flight: this is the
// driver that can be compared with a run of the reference, not the one to play many games with (lockstep_selfplay[_sharded] gives
// every game a generator of its own). Records in worker order (buffer.extend per worker, alpha_zero.rs:165-168). The trees' Fpu::Func /
// Dirichlet draws (thread_rng in the reference) come from the tree stream ((worker seed << 32) + the game's number under its worker, turn).
template <class G, int N>
std::vector<LockstepGameRecord<G, N>> gather_experience_host_trees(const std::vector<BatchPolicy<G, N>*>& policies, const RolloutConfig& cfg,
                                                                   size_t games_per_train, uint64_t seed, size_t* rounds_out = nullptr,
                                                                   size_t* evals_out = nullptr) {
    if (policies.empty()) throw Error(SYN_ERR_INVALID_ARGUMENT, "gather_experience_host_trees: no policies");
    std::vector<LockstepGameRecord<G, N>> out(games_per_train);
    typename detail::SelfplayWorker<G, N>::Shared shared{cfg, seed, 0, games_per_train, out};
    const size_t workers = policies.size();
    std::vector<size_t> start(workers), count(workers), rounds(workers, 0), evals(workers, 0);
    size_t games_to_schedule = games_per_train, workers_left = workers, at = 0;
    for (size_t i = 0; i < workers; i++) {   // alpha_zero.rs:132-154
        count[i] = games_to_schedule / workers_left;
        start[i] = at;
        at += count[i];
        games_to_schedule -= count[i];
        workers_left--;
    }
    detail::run_shards(workers, [&](size_t i) {
        if (count[i] == 0) return;
        detail::SelfplayWorker<G, N> worker(shared, start[i], count[i], seed * (uint64_t)workers + (uint64_t)i);
        auto step = [&worker](uint32_t u, const float* logits, const float* value) { return worker.step(u, logits, value); };
        detail::run_rounds<G, N>(*policies[i], 0u, (uint32_t)worker.slots(), nullptr, step, rounds[i], evals[i]);
    });
    if (rounds_out) *rounds_out = std::accumulate(rounds.begin(), rounds.end(), (size_t)0);
    if (evals_out) *evals_out = std::accumulate(evals.begin(), evals.end(), (size_t)0);
    return out;
}

// One policy object shared by several host threads, each of which sees a BatchPolicy of its own (worker(i)): the batches the
// workers hand in are COMBINED — whatever has been handed in while the policy was busy goes to it as one batch, so the policy
// (a GPU) sees few large calls instead of many small ones, and no worker waits for a worker, only for its own answers.
// (gather_experience gives every worker its own policy, alpha_zero.rs:192-198; a GPU policy is better shared: a launch costs the
// same for 100 positions as for 4,000.) The inner policy is only ever called by one thread at a time.
template <class G, int N>
class CombiningPolicy {
public:
    CombiningPolicy(BatchPolicy<G, N>& inner, size_t workers) : inner_(inner) {
        for (size_t i = 0; i < workers; i++) workers_.emplace_back(new Worker(*this));
    }
    BatchPolicy<G, N>& worker(size_t i) { return *workers_[i]; }
    std::vector<BatchPolicy<G, N>*> workers() {
        std::vector<BatchPolicy<G, N>*> v;
        for (auto& w : workers_) v.push_back(w.get());
        return v;
    }
    size_t combined_calls() const { return calls_; }   // calls of the inner policy

private:
    struct Request {
        const std::vector<const G*>* games = nullptr;
        float* logits = nullptr;
        float* value = nullptr;
        bool done = true;
        std::exception_ptr error = nullptr;
    };
    struct Worker : BatchPolicy<G, N> {
        CombiningPolicy& owner;
        Request req;
        explicit Worker(CombiningPolicy& o) : owner(o) {}
        void eval_batch(const std::vector<const G*>& games, float* logits, float* value) override {
            eval_batch_begin(games, logits, value);
            eval_batch_end();
        }
        void eval_batch_begin(const std::vector<const G*>& games, float* logits, float* value) override {
            req.games = &games;
            req.logits = logits;
            req.value = value;
            req.done = false;
            req.error = nullptr;
            owner.hand_in(req);
        }
        void eval_batch_end() override { owner.collect(req); }
    };

    // (mu_ held) everything handed in so far goes to the inner policy as one batch
    void launch_locked() {
        in_flight_.swap(queue_);
        all_games_.clear();
        for (const Request* r : in_flight_) all_games_.insert(all_games_.end(), r->games->begin(), r->games->end());
        all_logits_.resize(all_games_.size() * (size_t)N);
        all_value_.resize(all_games_.size() * 3);
        calls_++;
        try {
            inner_.eval_batch_begin(all_games_, all_logits_.data(), all_value_.data());
        } catch (...) {
            launch_error_ = std::current_exception();   // reported to the batch's owners when it is collected
        }
    }
    void hand_in(Request& r) {
        std::lock_guard<std::mutex> lk(mu_);
        queue_.push_back(&r);
        if (in_flight_.empty() && !finishing_) launch_locked();
    }
    void collect(Request& r) {
        std::unique_lock<std::mutex> lk(mu_);
        while (!r.done) {
            if (in_flight_.empty() || finishing_) {   // (r is in the queue behind a batch somebody else is finishing)
                cv_.wait(lk);
                continue;
            }
            // this thread finishes the batch in flight: waits for the inner policy, hands the answers out, launches what has queued up
            finishing_ = true;
            std::exception_ptr err = launch_error_;
            launch_error_ = nullptr;
            lk.unlock();
            if (!err) {
                try {
                    inner_.eval_batch_end();
                } catch (...) {
                    err = std::current_exception();
                }
            }
            size_t at = 0;
            for (Request* q : in_flight_) {
                const size_t n = q->games->size();
                if (!err) {
                    std::memcpy(q->logits, &all_logits_[at * (size_t)N], n * (size_t)N * sizeof(float));
                    std::memcpy(q->value, &all_value_[at * 3], n * 3 * sizeof(float));
                }
                at += n;
            }
            lk.lock();
            for (Request* q : in_flight_) {
                q->error = err;
                q->done = true;
            }
            in_flight_.clear();
            finishing_ = false;
            if (!queue_.empty()) launch_locked();
            cv_.notify_all();
        }
        if (r.error) std::rethrow_exception(r.error);
    }

    BatchPolicy<G, N>& inner_;
    std::vector<std::unique_ptr<Worker>> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Request*> queue_, in_flight_;
    bool finishing_ = false;
    std::exception_ptr launch_error_ = nullptr;
    std::vector<const G*> all_games_;
    std::vector<float> all_logits_, all_value_;
    size_t calls_ = 0;
};

// policies/cache.rs:5-59 for a batch policy. A batch is answered position by position in batch order, as the reference's sequential
// eval calls would be: a position in the map (or seen EARLIER IN THE SAME BATCH — by then the reference would have inserted it) is
// a hit; the first occurrences of the others go to the inner policy as ONE batch and are remembered. A deterministic policy gives
// the same answers with or without the wrapper, so trees do not change — only the number of positions the policy sees.
// One wrapper per worker (it is not thread-safe, like the reference's HashMap); `Policy` is BatchPolicy<G, N>& to borrow
// (PolicyWithCache) or a BatchPolicy type to own (OwnedPolicyWithCache, constructed from the arguments after the capacity).
template <class G, int N, class Hash = typename DefaultGameHash<G>::type>
class BatchPolicyWithCache : public BatchPolicy<G, N> {
public:
    BatchPolicyWithCache(size_t capacity, BatchPolicy<G, N>& policy) : policy_(policy) { cache_.reserve(capacity); }   // with_capacity
    void eval_batch(const std::vector<const G*>& games, float* logits, float* value) override {
        const size_t n = games.size();
        miss_games_.clear();
        from_.assign(n, nullptr);
        miss_slot_.assign(n, (size_t)-1);
        pending_.clear();
        for (size_t i = 0; i < n; i++) {
            const auto it = cache_.find(*games[i]);
            if (it != cache_.end()) {
                from_[i] = &it->second;   // (unordered_map: references stay valid across the inserts below)
                hits_++;
                continue;
            }
            const auto ins = pending_.emplace(*games[i], miss_games_.size());
            if (ins.second) miss_games_.push_back(games[i]);
            else hits_++;   // same position earlier in this batch
            miss_slot_[i] = ins.first->second;
        }
        if (!miss_games_.empty()) {
            miss_logits_.resize(miss_games_.size() * (size_t)N);
            miss_value_.resize(miss_games_.size() * 3);
            policy_.eval_batch(miss_games_, miss_logits_.data(), miss_value_.data());
            misses_ += miss_games_.size();
            for (size_t k = 0; k < miss_games_.size(); k++) {
                Answer a;
                std::memcpy(a.logits, &miss_logits_[k * (size_t)N], sizeof(a.logits));
                std::memcpy(a.value, &miss_value_[k * 3], sizeof(a.value));
                cache_.emplace(*miss_games_[k], a);
            }
        }
        for (size_t i = 0; i < n; i++) {
            const float* lg = from_[i] ? from_[i]->logits : &miss_logits_[miss_slot_[i] * (size_t)N];
            const float* vv = from_[i] ? from_[i]->value : &miss_value_[miss_slot_[i] * 3];
            std::memcpy(logits + i * (size_t)N, lg, (size_t)N * sizeof(float));
            std::memcpy(value + i * 3, vv, 3 * sizeof(float));
        }
    }
    size_t hits() const { return hits_; }
    size_t misses() const { return misses_; }     // positions the inner policy was asked for
    size_t size() const { return cache_.size(); }
    void clear() { cache_.clear(); }

private:
    struct Answer { float logits[N]; float value[3]; };
    BatchPolicy<G, N>& policy_;
    std::unordered_map<G, Answer, Hash> cache_;
    std::unordered_map<G, size_t, Hash> pending_;
    std::vector<const G*> miss_games_;
    std::vector<const Answer*> from_;
    std::vector<size_t> miss_slot_;
    std::vector<float> miss_logits_, miss_value_;
    size_t hits_ = 0, misses_ = 0;
};
template <class G, int N, class P, class Hash = typename DefaultGameHash<G>::type>
class OwnedBatchPolicyWithCache : public BatchPolicy<G, N> {
public:
    template <class... Args>
    explicit OwnedBatchPolicyWithCache(size_t capacity, Args&&... policy_args)
        : policy(std::forward<Args>(policy_args)...), cached_(capacity, policy) {}
    void eval_batch(const std::vector<const G*>& games, float* logits, float* value) override { cached_.eval_batch(games, logits, value); }
    size_t hits() const { return cached_.hits(); }
    size_t misses() const { return cached_.misses(); }
    P policy;

private:
    BatchPolicyWithCache<G, N, Hash> cached_;
};

// BatchPolicy<Connect4, 9> on the GPU: one worker's policy = one evaluation context of the engine (syn_eval_ctx: its own stream
// and staging, the engine's weights). Several of them on one engine serve several host threads at once, one thread each.
class HipBatchPolicy : public BatchPolicy<Connect4, 9> {
public:
    explicit HipBatchPolicy(syn_engine* h) {
        const int rc = syn_eval_ctx_create(h, &ctx_);
        if (rc != SYN_OK) throw Error(rc, syn_last_error(h));
    }
    explicit HipBatchPolicy(Engine& e) : HipBatchPolicy(e.handle()) {}
    ~HipBatchPolicy() override { syn_eval_ctx_destroy(ctx_); }
    HipBatchPolicy(const HipBatchPolicy&) = delete;
    HipBatchPolicy& operator=(const HipBatchPolicy&) = delete;

    void eval_batch(const std::vector<const Connect4*>& games, float* logits, float* value) override {
        eval_batch_begin(games, logits, value);
        eval_batch_end();
    }
    void eval_batch_begin(const std::vector<const Connect4*>& games, float* logits, float* value) override {
        my_.resize(games.size());
        op_.resize(games.size());
        for (size_t i = 0; i < games.size(); i++) { my_[i] = games[i]->my_bb(); op_[i] = games[i]->op_bb(); }
        logits_ = logits;
        value_ = value;
        const int rc = syn_eval_ctx_submit(ctx_, my_.data(), op_.data(), (int)games.size());
        if (rc != SYN_OK) throw Error(rc, syn_eval_ctx_last_error(ctx_));
    }
    void eval_batch_end() override {
        const int rc = syn_eval_ctx_wait(ctx_, logits_, value_);
        if (rc != SYN_OK) throw Error(rc, syn_eval_ctx_last_error(ctx_));
    }

private:
    syn_eval_ctx* ctx_ = nullptr;
    std::vector<uint64_t> my_, op_;
    float* logits_ = nullptr;
    float* value_ = nullptr;
};

}  // namespace synthesis

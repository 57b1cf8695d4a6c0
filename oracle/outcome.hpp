// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// CPU restatement of the solver lattice of the reference:
//   synthesis/src/game.rs:9-66   (Outcome, From<f32>, reversed, value, Ord)
//   synthesis/src/mcts.rs:10-26  (Into<usize>, Into<[f32;3]>)
// Parity: pinned by the reference's own tests synthesis/src/game.rs:94-141 (replayed in tests/test_oracle_kats.py).
#pragma once
#include <cstdint>

namespace oracle {

// Index order is the reference's Into<usize> (mcts.rs:10-18): Lose=0, Draw=1, Win=2.
enum OutcomeKind : uint8_t { LOSE = 0, DRAW = 1, WIN = 2 };

struct Outcome {
    OutcomeKind kind;
    uint32_t turns;
    bool operator==(const Outcome& o) const { return kind == o.kind && turns == o.turns; }
};

// game.rs:16-26
inline Outcome outcome_from_reward(float value) {
    if (value > 0.0f) return {WIN, 0};
    if (value < 0.0f) return {LOSE, 0};
    return {DRAW, 0};
}

// game.rs:29-35
inline Outcome reversed(const Outcome& o) {
    switch (o.kind) {
        case WIN: return {LOSE, o.turns + 1};
        case LOSE: return {WIN, o.turns + 1};
        default: return {DRAW, o.turns + 1};
    }
}

// game.rs:37-43
inline float value(const Outcome& o) {
    switch (o.kind) {
        case WIN: return 1.0f;
        case DRAW: return 0.0f;
        default: return -1.0f;
    }
}

inline int cmp_u(uint32_t a, uint32_t b) { return a < b ? -1 : (a > b ? 1 : 0); }

// game.rs:46-60. Returns -1 / 0 / +1 for Less / Equal / Greater.
inline int cmp(const Outcome& a, const Outcome& b) {
    if (a.kind == WIN && b.kind == WIN) return cmp_u(b.turns, a.turns);  // reversed: fewer turns is greater
    if (a.kind == WIN) return 1;
    if (b.kind == WIN) return -1;
    if (a.kind == DRAW && b.kind == DRAW) return cmp_u(a.turns, b.turns);
    if (a.kind == DRAW) return 1;   // Draw > Lose
    if (b.kind == DRAW) return -1;  // Lose < Draw
    return cmp_u(a.turns, b.turns);  // Lose vs Lose
}

// Option<Outcome> with Rust's derived ordering: None < Some(_).
struct OptOutcome {
    bool some = false;
    Outcome o{LOSE, 0};
    static OptOutcome none() { return {}; }
    static OptOutcome of(Outcome x) { return {true, x}; }
};

inline int cmp(const OptOutcome& a, const OptOutcome& b) {
    if (!a.some && !b.some) return 0;
    if (!a.some) return -1;
    if (!b.some) return 1;
    return cmp(a.o, b.o);
}

// std::cmp::Ord::max(a, b): returns b unless a is strictly Greater.
inline OptOutcome max_opt(const OptOutcome& a, const OptOutcome& b) { return cmp(a, b) > 0 ? a : b; }

// mcts.rs:20-26
inline void onehot(const Outcome& o, float dist[3]) {
    dist[0] = dist[1] = dist[2] = 0.0f;
    dist[(int)o.kind] = 1.0f;
}

}  // namespace oracle

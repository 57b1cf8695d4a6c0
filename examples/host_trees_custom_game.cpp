// Example: the reference's MCTS<G, P, N> for a game that has no kernel of its own — trees on the host, the policy batched.
//
// A caller's own `Game` (here: a subtraction game — take 1, 2 or 3 stones, whoever takes the last stone wins) only needs the
// surface of synthesis/src/game.rs:68-88 that MCTS uses: player(), is_over(), reward(player), iter_actions(), step(action).
// A `BatchPolicy<G, N>` evaluates the leaves of many trees in one call (policies/traits.rs:4-6 for a batch) — on a GPU that is
// one launch (synthesis::HipBatchPolicy does it for Connect4 through syn_eval_ctx_*); here a uniform CPU policy stands in, so
// the example needs no GPU:
//
//   g++ -std=c++17 -O2 -ffp-contract=off -pthread -I include examples/host_trees_custom_game.cpp -o host_trees && ./host_trees
//
// With several host threads and ONE policy object, give every thread a worker of a CombiningPolicy and use the *_sharded drivers:
// the workers' leaves go to the policy as one combined batch, no thread waits for another.
#include <cstdio>

#include "synthesis_amd_lockstep.hpp"

using namespace synthesis;

struct Subtraction {                       // Game<3>
    enum PlayerId { First = 0, Second = 1 };
    int stones = 0;
    PlayerId to_move = First;
    PlayerId player() const { return to_move; }
    bool is_over() const { return stones == 0; }
    float reward(PlayerId p) const { return stones != 0 ? 0.0f : (p == to_move ? -1.0f : 1.0f); }   // the side to move has lost
    std::vector<int> iter_actions() const {
        std::vector<int> a;
        for (int k = 0; k < 3; k++)
            if (k + 1 <= stones) a.push_back(k);
        return a;
    }
    bool step(int action) {
        stones -= action + 1;
        to_move = to_move == First ? Second : First;
        return is_over();
    }
};

struct UniformPolicy : BatchPolicy<Subtraction, 3> {
    void eval_batch(const std::vector<const Subtraction*>& games, float* logits, float* value) override {
        for (size_t i = 0; i < games.size(); i++) {
            for (int k = 0; k < 3; k++) logits[i * 3 + k] = 0.0f;
            value[i * 3 + 0] = value[i * 3 + 1] = value[i * 3 + 2] = 1.0f / 3.0f;
        }
    }
};

int main() {
    std::vector<Subtraction> roots(12);
    for (size_t i = 0; i < roots.size(); i++) roots[i].stones = 2 + (int)i;

    // one policy, one thread: the trees take turns in two halves (one half's leaves with the policy while the other half advances)
    UniformPolicy policy;
    size_t calls = 0, leaves = 0;
    const auto trees = lockstep_search<Subtraction, 3>(policy, MCTSConfig{}, roots, /* explores */ 400, /* threads */ 1, &calls, &leaves);

    // the same roots on three worker threads that share the policy
    CombiningPolicy<Subtraction, 3> shared(policy, 3);
    const auto again = lockstep_search_sharded<Subtraction, 3>(shared.workers(), MCTSConfig{}, roots, 400);

    for (size_t i = 0; i < trees.size(); i++) {
        const auto& root = trees[i].root();
        const int best = trees[i].best_action(ActionSelection::NumVisits);
        std::printf("%2d stones: %s, take %d  (%zu nodes; sharded run: take %d)\n", roots[i].stones,
                    !root.solution.some ? "unsolved" : (root.solution.outcome.kind == Outcome::Win ? "won " : "lost"), best + 1,
                    trees[i].num_nodes(), again[i].best_action(ActionSelection::NumVisits) + 1);
    }
    std::printf("%zu policy calls for %zu leaves\n", calls, leaves);
    return 0;
}

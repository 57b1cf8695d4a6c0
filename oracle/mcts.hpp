// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of the reference's single-tree MCTS, function for function and in the same f32 operation
// order (build with -ffp-contract=off, no fast-math):
//   synthesis/src/mcts.rs:28-100   Node
//   synthesis/src/mcts.rs:123-147  with_capacity / explore_n
//   synthesis/src/mcts.rs:174-225  target_policy / target_q
//   synthesis/src/mcts.rs:229-269  root noise (None / Equal / Dirichlet: rand_distr's sampler restated in noise.hpp)
//   synthesis/src/mcts.rs:273-306  best_action / solution
//   synthesis/src/mcts.rs:310-372  explore / select_best_child / exploit_value / explore_value
//   synthesis/src/mcts.rs:374-427  visit (expansion, auto-extend, legal-move softmax)
//   synthesis/src/mcts.rs:429-488  backprop (MCTS-Solver marking, value correction)
//   synthesis/src/config.rs:9-44   Exploration / ActionSelection / Fpu / MCTSConfig / PolicyNoise
// Transcendentals: the reference calls Rust's f32::exp / sqrt / ln (platform libm for exp/ln — unpinned). sqrt
// and division are IEEE-exact; exp is replaced by the deterministic oracle::det_expf (det_math.hpp) so that a
// device implementation can match bit for bit; ln (Uct only) is the deterministic oracle::det_logf likewise.
// Parity: pinned by the reference's KATs mcts.rs:691-868 as far as they are reproducible (tests/test_oracle_kats.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "det_math.hpp"
#include "noise.hpp"
#include "outcome.hpp"

namespace oracle {

enum ExplorationKind : int { UCT = 0, POLYNOMIAL_UCT = 1 };  // config.rs:9-13
enum ActionSelection : int { SELECT_Q = 0, SELECT_NUM_VISITS = 1 };  // config.rs:15-19
enum FpuKind : int { FPU_CONST = 0, FPU_PARENT_Q = 1, FPU_FUNC = 2 };  // config.rs:21-26
enum NoiseKind : int { NOISE_NONE = 0, NOISE_EQUAL = 1, NOISE_DIRICHLET = 2 };  // config.rs:39-44

// config.rs:28-37
struct MCTSConfig {
    int exploration = POLYNOMIAL_UCT;
    float c = 3.0f;
    bool solve = true;
    bool correct_values_on_solve = true;
    bool select_solved_nodes = true;
    bool auto_extend = true;
    int fpu = FPU_CONST;
    float fpu_value = 1.0f;
    float (*fpu_fn)() = nullptr;   // Fpu::Func: an arbitrary closure, or (nullptr) the reference's own Normal(fpu_value, fpu_std)
    float fpu_std = 0.0f;          // (study-connect4/src/main.rs:43-47) drawn from the tree's noise stream (noise.hpp)
    int noise = NOISE_NONE;
    float noise_alpha = 0.0f;
    float noise_weight = 0.0f;
};

// Event counters for the bench's algorithmic-bytes accounting; not part of the reference.
struct MCTSCounters {
    uint64_t explores = 0 /* explore() calls + root visits */, select_levels = 0, children_scanned = 0, expansions = 0, new_nodes = 0;
    uint64_t policy_evals = 0, backprop_levels = 0, solver_children = 0, solved_hits = 0;
    uint64_t max_depth = 0;  // deepest backprop chain (levels) seen
};

template <class G>
struct Node {
    uint32_t parent;
    uint32_t first_child;
    uint8_t num_children;
    G game;
    OptOutcome solution;
    uint8_t action;
    float action_prob;
    float outcome_probs[3];
    float num_visits;

    float q() const { return (outcome_probs[2] - outcome_probs[0]) / num_visits; }  // mcts.rs:42-44
    bool is_unvisited() const { return num_children == 0 && !solution.some; }       // mcts.rs:71-73
    uint32_t last_child() const { return first_child + num_children; }              // mcts.rs:86-88
};

// Option<f32> / Option<(f32,f32)> comparisons follow Rust's derived PartialOrd: None < Some(_); Some(a) > Some(b)
// iff a > b (false when unordered); tuples compare lexicographically with partial_cmp.
inline bool opt_gt(bool a_some, float a, bool b_some, float b) {
    if (!a_some) return false;
    if (!b_some) return true;
    return a > b;
}
inline bool tuple_gt(float a0, float a1, float b0, float b1) {
    if (a0 < b0) return false;
    if (a0 > b0) return true;
    if (a0 == b0) return a1 > b1;
    return false;  // unordered first element
}

template <class G, class P>
struct MCTS {
    static constexpr int N = G::N;
    uint32_t root = 0;
    std::vector<Node<G>> nodes;
    P* policy;
    MCTSConfig cfg;
    MCTSCounters* ctr = nullptr;
    uint64_t noise_seed = 0;          // this tree's stream for Fpu::Func Normal draws / the Dirichlet sample (noise.hpp)
    mutable uint32_t fpu_scans = 0;   // select_best_child calls of this tree that took at least one Normal draw (noise.hpp)
    mutable bool fpu_scan_used = false;   // ... the one in progress did (exploit_value is const in the reference too)

    // mcts.rs:123-137
    // `storage`: an empty vector whose capacity is reused (the reference allocates a fresh Vec::with_capacity per
    // move, mcts.rs:124; reusing the allocation changes no result and keeps the multi-threaded CPU baseline from
    // serialising on mmap/munmap of ~700 KB vectors).
    // `noise_seed_`: seed of this tree's stream for the Fpu::Func Normal draws and the root's Dirichlet sample (the reference
    // draws both from thread_rng, mcts.rs:236 / main.rs:46 — unreproducible; see noise.hpp for the discipline used here)
    MCTS(size_t capacity, const MCTSConfig& cfg_, P* policy_, const G& game, MCTSCounters* ctr_ = nullptr,
         std::vector<Node<G>>* storage = nullptr, uint64_t noise_seed_ = 0)
        : policy(policy_), cfg(cfg_), ctr(ctr_), noise_seed(noise_seed_) {
        if (storage) { nodes = std::move(*storage); nodes.clear(); }
        nodes.reserve(capacity);
        nodes.push_back(unvisited(0, game, OptOutcome::none(), 0, 0.0f));
        if (ctr) ctr->explores++;  // the root visit is one select/expand/backprop pass like any explore
        float probs[3];
        bool any_solved;
        uint32_t node_id = visit(root, probs, any_solved);
        backprop(node_id, probs, any_solved);
        add_root_noise();
    }

    static Node<G> unvisited(uint32_t parent, const G& game, OptOutcome solution, uint8_t action, float prob) {
        Node<G> n;
        n.parent = parent;
        n.first_child = 0;
        n.num_children = 0;
        n.game = game;
        n.solution = solution;
        n.action = action;
        n.action_prob = prob;
        n.outcome_probs[0] = n.outcome_probs[1] = n.outcome_probs[2] = 0.0f;
        n.num_visits = 0.0f;
        return n;
    }

    // mcts.rs:139-147
    void explore_n(size_t n) {
        for (size_t i = 0; i < n; i++) {
            if (nodes[root].solution.some) break;
            explore();
        }
    }

    // mcts.rs:174-211
    void target_policy(float* search_policy) const {
        for (int i = 0; i < N; i++) search_policy[i] = 0.0f;
        float total = 0.0f;
        const Node<G>& r = nodes[root];
        if (r.num_visits == 1.0f) {
            if (r.solution.some && r.solution.o.kind == WIN) {
                for (uint32_t c = r.first_child; c < r.last_child(); c++) {
                    float v = (nodes[c].solution.some && nodes[c].solution.o.kind == LOSE) ? 1.0f : 0.0f;
                    search_policy[nodes[c].action] = v;
                    total += v;
                }
            } else {
                for (uint32_t c = r.first_child; c < r.last_child(); c++) {
                    search_policy[nodes[c].action] = 1.0f;
                    total += 1.0f;
                }
            }
        } else {
            for (uint32_t c = r.first_child; c < r.last_child(); c++) {
                float v = nodes[c].num_visits;
                search_policy[nodes[c].action] = v;
                total += v;
            }
        }
        for (int i = 0; i < N; i++) search_policy[i] /= total;
    }

    // mcts.rs:213-225
    void target_q(float out[3]) const {
        const Node<G>& r = nodes[root];
        if (r.solution.some) {
            onehot(r.solution.o, out);
        } else {
            for (int i = 0; i < 3; i++) out[i] = r.outcome_probs[i] / r.num_visits;
        }
    }

    // mcts.rs:229-269
    void add_root_noise() {
        if (cfg.noise == NOISE_EQUAL) {
            Node<G>& r = nodes[root];
            if (r.num_children < 2) return;
            float noise = 1.0f / (float)r.num_children;
            for (uint32_t c = r.first_child; c < r.last_child(); c++)
                nodes[c].action_prob = nodes[c].action_prob * (1.0f - cfg.noise_weight) + cfg.noise_weight * noise;
        } else if (cfg.noise == NOISE_DIRICHLET) {  // mcts.rs:241-256
            Node<G>& r = nodes[root];
            if (r.num_children < 2) return;
            float noise_probs[64];
            noise_dirichlet(noise_seed, cfg.noise_alpha, (uint32_t)r.num_children, noise_probs);
            uint32_t k = 0;
            for (uint32_t c = r.first_child; c < r.last_child(); c++, k++)
                nodes[c].action_prob = nodes[c].action_prob * (1.0f - cfg.noise_weight) + cfg.noise_weight * noise_probs[k];
        }
    }

    // mcts.rs:273-294
    int best_action(int action_selection) const {
        const Node<G>& r = nodes[root];
        int best = -1;
        bool have = false;
        float b0 = 0.0f, b1 = 0.0f;
        for (uint32_t c = r.first_child; c < r.last_child(); c++) {
            const Node<G>& ch = nodes[c];
            float v0, v1;
            if (ch.solution.some && ch.solution.o.kind == WIN) {
                v0 = 0.0f; v1 = (float)ch.solution.o.turns;
            } else if (!ch.solution.some) {
                v0 = 1.0f; v1 = action_selection == SELECT_Q ? -ch.q() : ch.num_visits;
            } else if (ch.solution.o.kind == DRAW) {
                v0 = 2.0f; v1 = -(float)ch.solution.o.turns;
            } else {
                v0 = 3.0f; v1 = -(float)ch.solution.o.turns;
            }
            if (!have || tuple_gt(v0, v1, b0, b1)) {
                have = true;
                b0 = v0; b1 = v1;
                best = ch.action;
            }
        }
        return best;  // the reference unwraps: a root without children panics there
    }

    // mcts.rs:296-306
    OptOutcome solution(int action) const {
        const Node<G>& r = nodes[root];
        for (uint32_t c = r.first_child; c < r.last_child(); c++)
            if (nodes[c].action == (uint8_t)action) return nodes[c].solution;
        return OptOutcome::none();
    }

    // mcts.rs:310-325
    void explore() {
        if (ctr) ctr->explores++;
        uint32_t node_id = root;
        for (;;) {
            const Node<G>& node = nodes[node_id];
            if (node.solution.some) {
                float probs[3];
                onehot(node.solution.o, probs);
                if (ctr) ctr->solved_hits++;
                backprop(node_id, probs, true);
                return;
            } else if (node.is_unvisited()) {
                float probs[3];
                bool any_solved;
                uint32_t leaf = visit(node_id, probs, any_solved);
                backprop(leaf, probs, any_solved);
                return;
            } else {
                node_id = select_best_child(node);
            }
        }
    }

    // mcts.rs:327-341
    uint32_t select_best_child(const Node<G>& parent) {
        uint32_t best_child = 0;
        bool have = false;
        float best_value = 0.0f;
        if (ctr) { ctr->select_levels++; ctr->children_scanned += parent.num_children; }
        fpu_scan_used = false;
        for (uint32_t child_id = parent.first_child; child_id < parent.last_child(); child_id++) {
            const Node<G>& child = nodes[child_id];
            float q = exploit_value(parent, child);
            float u = explore_value(parent, child);
            float value = q + u;
            if (opt_gt(true, value, have, best_value)) {
                best_child = child_id;
                best_value = value;
                have = true;
            }
        }
        if (fpu_scan_used) fpu_scans++;
        return best_child;
    }

    // mcts.rs:343-359
    float exploit_value(const Node<G>& parent, const Node<G>& child) const {
        if (child.solution.some) {
            if (cfg.select_solved_nodes) return value(reversed(child.solution.o));
            return -std::numeric_limits<float>::infinity();
        } else if (child.num_children == 0) {
            switch (cfg.fpu) {
                case FPU_CONST: return cfg.fpu_value;
                case FPU_PARENT_Q: return parent.q();
                default:
                    if (cfg.fpu_fn) return cfg.fpu_fn();
                    fpu_scan_used = true;
                    return noise_fpu_normal(noise_seed, fpu_scans, (uint32_t)(&child - &nodes[parent.first_child]), cfg.fpu_value, cfg.fpu_std);
            }
        } else {
            return -child.q();
        }
    }

    // mcts.rs:361-372
    float explore_value(const Node<G>& parent, const Node<G>& child) const {
        if (cfg.exploration == UCT) {
            float visits = std::sqrt(cfg.c * det_logf(parent.num_visits));
            return visits / std::sqrt(child.num_visits);
        } else {
            float visits = std::sqrt(parent.num_visits);
            return cfg.c * child.action_prob * visits / (1.0f + child.num_visits);
        }
    }

    // mcts.rs:374-427
    uint32_t visit(uint32_t node_id, float outcome_probs[3], bool& any_solved_out) {
        uint32_t first_child = (uint32_t)nodes.size();
        if (nodes[node_id].solution.some) {
            onehot(nodes[node_id].solution.o, outcome_probs);
            any_solved_out = true;
            return node_id;
        }
        G game = nodes[node_id].game;
        uint8_t num_children = 0;
        bool any_solved = false;
        int actions[G::N];
        int n_act = game.legal_actions(actions);
        for (int i = 0; i < n_act; i++) {
            G child_game = game;
            bool is_over = child_game.step(actions[i]);
            OptOutcome sol = OptOutcome::none();
            if (is_over) {
                any_solved = true;
                sol = OptOutcome::of(outcome_from_reward(child_game.reward(child_game.player_id())));
            }
            nodes.push_back(unvisited(node_id, child_game, sol, (uint8_t)actions[i], 1.0f));
            num_children++;
        }
        if (ctr) { ctr->expansions++; ctr->new_nodes += num_children; }
        nodes[node_id].first_child = first_child;
        nodes[node_id].num_children = num_children;
        uint32_t last_child = first_child + num_children;

        if (cfg.auto_extend && num_children == 1) {
            return visit(first_child, outcome_probs, any_solved_out);
        }
        float logits[G::N];
        policy->eval(game, logits, outcome_probs);
        if (ctr) ctr->policy_evals++;

        // softmax restricted to the legal children (max-subtracted), summed in child order
        float max_logit = -std::numeric_limits<float>::infinity();
        for (uint32_t c = first_child; c < last_child; c++) {
            float logit = logits[nodes[c].action];
            max_logit = rust_max(max_logit, logit);
            nodes[c].action_prob = logit;
        }
        float total = 0.0f;
        for (uint32_t c = first_child; c < last_child; c++) {
            nodes[c].action_prob = det_expf(nodes[c].action_prob - max_logit);
            total += nodes[c].action_prob;
        }
        for (uint32_t c = first_child; c < last_child; c++) nodes[c].action_prob /= total;
        any_solved_out = any_solved;
        return node_id;
    }

    // Rust's f32::max: if one argument is NaN the other is returned.
    static float rust_max(float a, float b) {
        if (a != a) return b;
        if (b != b) return a;
        return a > b ? a : b;
    }

    // mcts.rs:429-488
    void backprop(uint32_t leaf_node_id, float outcome_probs[3], bool solved) {
        uint32_t node_id = leaf_node_id;
        uint64_t levels = 0;
        for (;;) {
            uint32_t parent = nodes[node_id].parent;
            levels++;
            if (ctr) { ctr->backprop_levels++; if (levels > ctr->max_depth) ctr->max_depth = levels; }
            if (cfg.solve && solved) {
                bool all_solved = true;
                OptOutcome best = nodes[node_id].solution;
                const Node<G>& nd = nodes[node_id];
                if (ctr) ctr->solver_children += nd.num_children;
                for (uint32_t c = nd.first_child; c < nd.last_child(); c++) {
                    OptOutcome soln = nodes[c].solution.some ? OptOutcome::of(reversed(nodes[c].solution.o))
                                                             : OptOutcome::none();
                    all_solved = all_solved && soln.some;
                    best = max_opt(best, soln);
                }
                Node<G>& node = nodes[node_id];
                if (best.some && best.o.kind == WIN) {
                    node.solution = OptOutcome::of({WIN, best.o.turns});
                    if (cfg.correct_values_on_solve) {
                        for (int i = 0; i < 3; i++) outcome_probs[i] = -node.outcome_probs[i];
                        outcome_probs[2] += node.num_visits + 1.0f;
                    }
                } else if (best.some && all_solved) {
                    node.solution = best;
                    if (cfg.correct_values_on_solve) {
                        for (int i = 0; i < 3; i++) outcome_probs[i] = -node.outcome_probs[i];
                        if (best.o.kind == DRAW) outcome_probs[1] += node.num_visits + 1.0f;
                        else outcome_probs[0] += node.num_visits + 1.0f;
                    }
                } else {
                    solved = false;
                }
            }
            Node<G>& node = nodes[node_id];
            for (int i = 0; i < 3; i++) node.outcome_probs[i] += outcome_probs[i];
            node.num_visits += 1.0f;
            if (node_id == root) break;
            float t = outcome_probs[0];
            outcome_probs[0] = outcome_probs[2];
            outcome_probs[2] = t;
            node_id = parent;
        }
    }
};

}  // namespace oracle

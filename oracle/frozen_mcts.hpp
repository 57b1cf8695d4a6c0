// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of the evaluator's baseline tree, the scalar-value `FrozenMCTS` (synthesis/src/evaluator.rs:230-534):
//   Node (233-300): parent, first_child, num_children, game, solution, action, action_prob, cum_value, num_visits
//   exploit / with_capacity (311-338), best_action (364-389), explore (391-406), select_best_child (408-437),
//   visit (439-483), backprop (485-527), explore_n (529-533)
// It differs from `MCTS` (mcts.hpp): one scalar value per node, the policy is called BEFORE the children are created,
// no auto-extend / noise / value-distribution targets, unvisited children are scored `fpu + prior` without an exploration
// term, only Uct exploration, its own solver rule (a lost child proves Win(0)), explore_n never stops early.
// No reference test exercises it -> PARITY UNPINNED against the reference (this file restates the source line by line).
// Transcendentals as in mcts.hpp: det_expf / det_logf instead of libm.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "det_math.hpp"
#include "mcts.hpp"
#include "outcome.hpp"

namespace oracle {

template <class G>
struct FrozenNode {
    uint32_t parent, first_child;
    uint8_t num_children;
    G game;
    OptOutcome solution;
    uint8_t action;
    float action_prob, cum_value, num_visits;
    bool is_unvisited() const { return num_children == 0 && !solution.some; }
    bool is_visited() const { return num_children != 0; }
    bool is_unsolved() const { return !solution.some; }
    uint32_t last_child() const { return first_child + num_children; }
};

template <class G, class P>
struct FrozenMCTS {
    static constexpr int N = G::N;
    uint32_t root = 0;
    std::vector<FrozenNode<G>> nodes;
    P* policy;
    MCTSConfig cfg;

    // evaluator.rs:325-338
    FrozenMCTS(size_t capacity, const MCTSConfig& cfg_, P* policy_, const G& game) : policy(policy_), cfg(cfg_) {
        nodes.reserve(capacity);
        nodes.push_back(unvisited(0, game, OptOutcome::none(), 0, 0.0f));
        bool any_solved;
        float v = visit(root, any_solved);
        backprop(root, v, any_solved);
    }
    static FrozenNode<G> unvisited(uint32_t parent, const G& game, OptOutcome sol, uint8_t action, float prob) {
        FrozenNode<G> n;
        n.parent = parent; n.first_child = 0; n.num_children = 0; n.game = game; n.solution = sol; n.action = action;
        n.action_prob = prob; n.cum_value = 0.0f; n.num_visits = 0.0f;
        return n;
    }
    void explore_n(size_t n) {  // 529-533: never stops early
        for (size_t i = 0; i < n; i++) explore();
    }
    // 364-389
    int best_action(int action_selection) const {
        const auto& r = nodes[root];
        int best = -1;
        float best_value = -std::numeric_limits<float>::infinity();
        for (uint32_t c = r.first_child; c < r.last_child(); c++) {
            const auto& ch = nodes[c];
            if (ch.is_unvisited()) continue;
            float v;
            if (ch.solution.some) {
                v = ch.solution.o.kind == WIN ? -std::numeric_limits<float>::infinity()
                  : ch.solution.o.kind == DRAW ? 1e6f : std::numeric_limits<float>::infinity();
            } else {
                v = action_selection == SELECT_Q ? -ch.cum_value / ch.num_visits : ch.num_visits;
            }
            if (best < 0 || v > best_value) { best_value = v; best = ch.action; }
        }
        return best;  // the reference unwraps
    }
    // 391-406
    void explore() {
        uint32_t id = root;
        for (;;) {
            const auto& node = nodes[id];
            if (node.solution.some) { backprop(id, value(node.solution.o), true); return; }
            if (node.is_unvisited()) {
                bool any_solved;
                float v = visit(id, any_solved);
                backprop(id, v, any_solved);
                return;
            }
            id = select_best_child(id);
        }
    }
    // 408-437
    uint32_t select_best_child(uint32_t id) const {
        const auto& node = nodes[id];
        bool have = false;
        uint32_t best = 0;
        float best_value = -std::numeric_limits<float>::infinity();
        for (uint32_t c = node.first_child; c < node.last_child(); c++) {
            const auto& ch = nodes[c];
            float v;
            if (ch.is_unvisited()) {
                v = cfg.fpu_value + ch.action_prob;  // Fpu::Const only ("Unsupported fpu in baseline" otherwise)
            } else {
                float q = ch.solution.some ? value(reversed(ch.solution.o)) : -ch.cum_value / ch.num_visits;
                float visits = std::sqrt(cfg.c * det_logf(node.num_visits));  // Exploration::Uct only
                float u = visits / std::sqrt(ch.num_visits);
                v = q + u;
            }
            if (!have || v > best_value) { have = true; best = c; best_value = v; }
        }
        return best;
    }
    // 439-483
    float visit(uint32_t id, bool& any_solved) {
        uint32_t first_child = (uint32_t)nodes.size();
        G game = nodes[id].game;
        float logits[G::N], dist[3];
        policy->eval(game, logits, dist);
        uint8_t num_children = 0;
        any_solved = false;
        float max_logit = -std::numeric_limits<float>::infinity();
        int acts[G::N];
        int n_act = game.legal_actions(acts);
        for (int i = 0; i < n_act; i++) {
            G child = game;
            bool over = child.step(acts[i]);
            OptOutcome sol = OptOutcome::none();
            if (over) {
                any_solved = true;
                sol = OptOutcome::of(outcome_from_reward(child.reward(child.player_id())));
            }
            float logit = logits[acts[i]];
            max_logit = MCTS<G, P>::rust_max(max_logit, logit);
            nodes.push_back(unvisited(id, child, sol, (uint8_t)acts[i], logit));
            num_children++;
        }
        nodes[id].first_child = first_child;
        nodes[id].num_children = num_children;
        float total = 0.0f;
        for (uint32_t c = first_child; c < first_child + num_children; c++) {
            nodes[c].action_prob = det_expf(nodes[c].action_prob - max_logit);
            total += nodes[c].action_prob;
        }
        for (uint32_t c = first_child; c < first_child + num_children; c++) nodes[c].action_prob /= total;
        return dist[2] - dist[0];
    }
    // 485-527
    void backprop(uint32_t leaf, float v, bool solved) {
        uint32_t id = leaf;
        for (;;) {
            uint32_t parent = nodes[id].parent;
            if (cfg.solve && solved && nodes[id].is_unsolved()) {
                bool all_solved = true;
                OptOutcome worst = OptOutcome::none();
                const auto& nd = nodes[id];
                for (uint32_t c = nd.first_child; c < nd.last_child(); c++) {
                    const auto& ch = nodes[c];
                    if (ch.is_unvisited() || ch.is_unsolved()) all_solved = false;
                    else if (!worst.some || cmp(ch.solution, worst) < 0) worst = ch.solution;
                }
                auto& node = nodes[id];
                if (worst.some && worst.o.kind == LOSE) {
                    node.solution = OptOutcome::of({WIN, 0});  // at least one child is lost for the opponent
                    v = -node.cum_value + (node.num_visits + 1.0f);
                } else if (node.is_visited() && all_solved) {
                    Outcome best_for_me = reversed(worst.o);
                    node.solution = OptOutcome::of(best_for_me);
                    if (best_for_me.kind == DRAW) v = -node.cum_value;
                    else v = -node.cum_value - (node.num_visits + 1.0f);
                } else {
                    solved = false;
                }
            }
            auto& node = nodes[id];
            node.cum_value += v;
            node.num_visits += 1.0f;
            v = -v;
            if (id == root) break;
            id = parent;
        }
    }
};

}  // namespace oracle

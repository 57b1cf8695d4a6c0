// synthesis_amd — syn_mcts_search_lockstep: the host-tree, GPU-policy form of the search (include/synthesis_amd_lockstep.hpp)
// behind the C ABI. Host code only; it sits ABOVE the boundary and uses nothing but public entry points (syn_policy_eval_batch)
// plus the library's error slot. Compiled with -ffp-contract=off like the rest: every f32 operation of the tree arithmetic rounds
// where mcts.rs's expression order says.
#include <chrono>
#include <exception>

#include "../../include/synthesis_amd_lockstep.hpp"

extern "C" int syn_internal_fail(syn_engine* h, int code, const char* msg);  // engine.hip: fills syn_last_error

namespace {
struct TimedPolicy : synthesis::BatchPolicy<synthesis::Connect4, 9> {
    synthesis::HipBatchPolicy inner;
    double seconds = 0.0;
    explicit TimedPolicy(syn_engine* h) : inner(h) {}
    void eval_batch(const std::vector<const synthesis::Connect4*>& games, float* logits, float* value) override {
        const auto t0 = std::chrono::steady_clock::now();
        inner.eval_batch(games, logits, value);
        seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
};
}  // namespace

extern "C" int syn_mcts_search_lockstep(syn_engine* h, const syn_mcts_config* cfg, const uint64_t* my_bb, const uint64_t* op_bb,
                                        int n, int explores, int action_selection, int host_threads,
                                        syn_search_result* results, syn_lockstep_stats* stats) {
    using namespace synthesis;
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!cfg || n < 0 || explores < 0 || (n > 0 && (!my_bb || !op_bb || !results)))
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_mcts_search_lockstep");
    if (action_selection != SYN_ACTION_Q && action_selection != SYN_ACTION_NUM_VISITS)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown action selection");
    if (cfg->exploration != SYN_EXPLORATION_UCT && cfg->exploration != SYN_EXPLORATION_POLYNOMIAL_UCT)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown exploration");
    if (cfg->fpu == SYN_FPU_NORMAL || cfg->root_policy_noise == SYN_NOISE_DIRICHLET)
        return syn_internal_fail(h, SYN_ERR_UNSUPPORTED,
                                 "syn_mcts_search_lockstep: SYN_FPU_NORMAL / SYN_NOISE_DIRICHLET draw from the device path's per-tree "
                                 "streams (syn_mcts_search)");
    if (cfg->fpu != SYN_FPU_CONST && cfg->fpu != SYN_FPU_PARENT_Q)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown fpu");
    if (cfg->root_policy_noise != SYN_NOISE_NONE && cfg->root_policy_noise != SYN_NOISE_EQUAL)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown root policy noise");
    MCTSConfig m;
    m.exploration = (Exploration)cfg->exploration;
    m.c = cfg->c;
    m.solve = cfg->solve != 0;
    m.correct_values_on_solve = cfg->correct_values_on_solve != 0;
    m.select_solved_nodes = cfg->select_solved_nodes != 0;
    m.auto_extend = cfg->auto_extend != 0;
    m.fpu = (Fpu)cfg->fpu;
    m.fpu_value = cfg->fpu_value;
    m.root_policy_noise = (PolicyNoise)cfg->root_policy_noise;
    m.noise_alpha = cfg->noise_alpha;
    m.noise_weight = cfg->noise_weight;
    m.fpu_std = cfg->fpu_std;
    try {
        std::vector<Connect4> roots;
        roots.reserve((size_t)n);
        for (int i = 0; i < n; i++) {
            if ((my_bb[i] & op_bb[i]) != 0 || ((my_bb[i] | op_bb[i]) >> 63) != 0)
                return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "a root is not a Connect4 position");
            roots.push_back(Connect4::from_bitboards(my_bb[i], op_bb[i]));
            if (roots.back().is_over()) return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "a root is not a searchable Connect4 position");
        }
        TimedPolicy policy(h);
        size_t rounds = 0, evals = 0;
        const auto t0 = std::chrono::steady_clock::now();
        const auto trees = lockstep_search<Connect4, 9>(policy, m, roots, explores, host_threads, &rounds, &evals);
        const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int i = 0; i < n; i++) {
            const auto& t = trees[(size_t)i];
            syn_search_result r{};
            for (auto c = t.children_begin(); c != t.children_end(); ++c) {
                const int a = c->action;
                r.child_N[a] = c->num_visits;
                for (int k = 0; k < 3; k++) r.child_W[a][k] = c->outcome_probs[k];
                r.child_P[a] = c->action_prob;
                r.child_sol[a][0] = c->solution.some ? 1 : 0;
                r.child_sol[a][1] = c->solution.some ? (int32_t)c->solution.outcome.kind : 0;
                r.child_sol[a][2] = c->solution.some ? (int32_t)c->solution.outcome.turns : 0;
            }
            const auto& root = t.root();
            r.root_N = root.num_visits;
            for (int k = 0; k < 3; k++) r.root_W[k] = root.outcome_probs[k];
            r.root_sol[0] = root.solution.some ? 1 : 0;
            r.root_sol[1] = root.solution.some ? (int32_t)root.solution.outcome.kind : 0;
            r.root_sol[2] = root.solution.some ? (int32_t)root.solution.outcome.turns : 0;
            r.num_nodes = (uint32_t)t.num_nodes();
            r.best_action = t.best_action((ActionSelection)action_selection);
            const auto pi = t.target_policy();
            for (int k = 0; k < 9; k++) r.target_pi[k] = pi[(size_t)k];
            const auto q = t.target_q();
            for (int k = 0; k < 3; k++) r.target_q[k] = q[(size_t)k];
            results[i] = r;
        }
        if (stats) {
            stats->rounds = rounds;
            stats->positions_evaluated = evals;
            stats->seconds_total = total;
            stats->seconds_policy = policy.seconds;
        }
    } catch (const Error& e) {
        return e.code;  // (the failed C-ABI call left its text in syn_last_error)
    } catch (const std::exception& e) {
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, e.what());
    }
    return SYN_OK;
}

// synthesis_amd — syn_mcts_search_lockstep: the host-tree, GPU-policy form of the search (include/synthesis_amd_lockstep.hpp)
// behind the C ABI. Host code only; it sits ABOVE the boundary and uses nothing but public entry points (syn_eval_ctx_*: one
// evaluation context per host thread) plus the library's error slot and the engine's slot count. Compiled with -ffp-contract=off like the rest: every f32 operation of the tree arithmetic rounds
// where mcts.rs's expression order says.
#include <chrono>
#include <exception>
#include <memory>

#include "../../include/synthesis_amd_lockstep.hpp"

extern "C" int syn_internal_fail(syn_engine* h, int code, const char* msg);  // engine.hip: fills syn_last_error
extern "C" int syn_internal_concurrent_games(const syn_engine* h);           // engine.hip: syn_engine_config::concurrent_games

namespace {
// one worker's policy: an evaluation context of the engine, with the time its thread spent blocked in it
struct TimedPolicy : synthesis::BatchPolicy<synthesis::Connect4, 9> {
    synthesis::HipBatchPolicy inner;
    double seconds = 0.0;
    explicit TimedPolicy(syn_engine* h) : inner(h) {}
    template <class F>
    void timed(F&& f) {
        const auto t0 = std::chrono::steady_clock::now();
        f();
        seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    void eval_batch(const std::vector<const synthesis::Connect4*>& games, float* logits, float* value) override {
        timed([&] { inner.eval_batch(games, logits, value); });
    }
    void eval_batch_begin(const std::vector<const synthesis::Connect4*>& games, float* logits, float* value) override {
        timed([&] { inner.eval_batch_begin(games, logits, value); });
    }
    void eval_batch_end() override {
        timed([&] { inner.eval_batch_end(); });
    }
};

// host_threads workers (0: what the process may use) over ONE evaluation context: their batches go to the GPU combined
struct Workers {
    TimedPolicy gpu;
    synthesis::CombiningPolicy<synthesis::Connect4, 9> combined;
    std::vector<synthesis::BatchPolicy<synthesis::Connect4, 9>*> policies;
    static size_t count(int host_threads, size_t units) {
        const size_t t = (size_t)(host_threads > 0 ? host_threads : synthesis::detail::usable_host_threads());
        return std::max<size_t>(1, std::min(t, (units + 31) / 32));   // a worker wants a few dozen trees for its two halves
    }
    Workers(syn_engine* h, int host_threads, size_t units) : gpu(h), combined(gpu, count(host_threads, units)), policies(combined.workers()) {}
    double seconds_policy() const { return gpu.seconds; }   // host time inside the context's calls (launch + waiting for the GPU)
};
}  // namespace

namespace {
// syn_mcts_config -> MCTSConfig for the host trees; 0 or the error code left in syn_last_error
int host_mcts_config(syn_engine* h, const syn_mcts_config* cfg, synthesis::MCTSConfig& m) {
    using namespace synthesis;
    if (cfg->exploration != SYN_EXPLORATION_UCT && cfg->exploration != SYN_EXPLORATION_POLYNOMIAL_UCT)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown exploration");
    if (cfg->fpu != SYN_FPU_CONST && cfg->fpu != SYN_FPU_PARENT_Q && cfg->fpu != SYN_FPU_NORMAL && cfg->fpu != SYN_FPU_FUNC)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown fpu");
    if (cfg->fpu == SYN_FPU_FUNC && cfg->fpu_fn == nullptr)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "SYN_FPU_FUNC needs fpu_fn (Fpu::Func(fn() -> f32), config.rs:25)");
    if (cfg->fpu == SYN_FPU_NORMAL && !(cfg->fpu_std >= 0.0f && cfg->fpu_std < 1e30f && cfg->fpu_value == cfg->fpu_value))
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "SYN_FPU_NORMAL needs a finite mean and 0 <= std < 1e30");
    if (cfg->root_policy_noise != SYN_NOISE_NONE && cfg->root_policy_noise != SYN_NOISE_EQUAL && cfg->root_policy_noise != SYN_NOISE_DIRICHLET)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown root policy noise");
    if (cfg->root_policy_noise != SYN_NOISE_NONE && !(cfg->noise_weight >= 0.0f))
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "PolicyNoise weight must be >= 0");
    if (cfg->root_policy_noise == SYN_NOISE_DIRICHLET && !(cfg->noise_alpha > 0.0f && cfg->noise_alpha < 1e30f))
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "SYN_NOISE_DIRICHLET needs 0 < alpha < 1e30 (Dirichlet::new_with_size)");
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
    m.fpu_fn = cfg->fpu == SYN_FPU_FUNC ? cfg->fpu_fn : nullptr;
    return SYN_OK;
}
}  // namespace

extern "C" int syn_selfplay_run_lockstep(syn_engine* h, const syn_rollout_config* cfg, uint64_t base_seed, uint64_t first_game,
                                         int n_games, int host_threads, int32_t* plies, uint64_t* states_bb, float* pis, float* vs,
                                         uint8_t* actions, uint32_t* root_nodes, uint8_t* final_kind, syn_lockstep_stats* stats) {
    using namespace synthesis;
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!cfg || n_games < 0) return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_selfplay_run_lockstep");
    if (cfg->value_target < SYN_VALUE_Z || cfg->value_target > SYN_VALUE_Q_TO_Z)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown value target");
    if (cfg->action != SYN_ACTION_Q && cfg->action != SYN_ACTION_NUM_VISITS)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown action selection");
    if (cfg->num_explores < 0 || cfg->random_actions_until < 0 || cfg->sample_actions_until < 0)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "num_explores / random_actions_until / sample_actions_until must be >= 0");
    RolloutConfig rc;
    const int cr = host_mcts_config(h, &cfg->mcts_cfg, rc.mcts_cfg);
    if (cr != SYN_OK) return cr;
    rc.num_explores = cfg->num_explores;
    rc.random_actions_until = cfg->random_actions_until;
    rc.sample_actions_until = cfg->sample_actions_until;
    rc.stop_games_when_solved = cfg->stop_games_when_solved != 0;
    rc.value_target = (ValueTarget)cfg->value_target;
    rc.value_target_p = cfg->value_target_p;
    rc.value_target_from = cfg->value_target_from;
    rc.value_target_to = cfg->value_target_to;
    rc.action = (ActionSelection)cfg->action;
    try {
        // syn_selfplay_run's shape: n_games jobs over the engine's concurrent_games slots — here divided among the host workers
        const size_t concurrent = std::min<size_t>((size_t)n_games, (size_t)std::max(1, syn_internal_concurrent_games(h)));
        Workers workers(h, host_threads, concurrent);
        size_t rounds = 0, evals = 0;
        const auto t0 = std::chrono::steady_clock::now();
        const auto games = lockstep_selfplay_sharded<Connect4, 9>(workers.policies, rc, (size_t)n_games, base_seed, first_game, concurrent,
                                                                  &rounds, &evals);
        const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        constexpr size_t T = Connect4::MAX_TURNS;
        for (size_t g = 0; g < games.size(); g++) {
            const auto& r = games[g];
            const size_t n = r.states.size();
            if (plies) plies[g] = (int32_t)n;
            if (final_kind) final_kind[g] = (uint8_t)r.final_outcome.kind;
            for (size_t k = 0; k < n; k++) {
                const size_t p = g * T + k;
                if (states_bb) { states_bb[p * 2] = r.states[k].my_bb(); states_bb[p * 2 + 1] = r.states[k].op_bb(); }
                if (pis) for (int c = 0; c < 9; c++) pis[p * 9 + (size_t)c] = r.pis[k][(size_t)c];
                if (vs) for (int c = 0; c < 3; c++) vs[p * 3 + (size_t)c] = r.vs[k][(size_t)c];
                if (actions) actions[p] = r.actions[k];
                if (root_nodes) root_nodes[p] = r.root_nodes[k];
            }
        }
        if (stats) {
            stats->rounds = workers.combined.combined_calls();
            stats->positions_evaluated = evals;
            stats->seconds_total = total;
            stats->seconds_policy = workers.seconds_policy();
        }
    } catch (const Error& e) {
        return syn_internal_fail(h, e.code, e.what());
    } catch (const std::exception& e) {
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, e.what());
    }
    return SYN_OK;
}

extern "C" int syn_mcts_search_lockstep(syn_engine* h, const syn_mcts_config* cfg, const uint64_t* my_bb, const uint64_t* op_bb,
                                        int n, int explores, int action_selection, int host_threads,
                                        syn_search_result* results, syn_lockstep_stats* stats) {
    using namespace synthesis;
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!cfg || n < 0 || explores < 0 || (n > 0 && (!my_bb || !op_bb || !results)))
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_mcts_search_lockstep");
    if (action_selection != SYN_ACTION_Q && action_selection != SYN_ACTION_NUM_VISITS)
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown action selection");
    MCTSConfig m;
    if (const int rc = host_mcts_config(h, cfg, m)) return rc;
    try {
        std::vector<Connect4> roots;
        roots.reserve((size_t)n);
        for (int i = 0; i < n; i++) {
            if ((my_bb[i] & op_bb[i]) != 0 || ((my_bb[i] | op_bb[i]) >> 63) != 0)
                return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "a root is not a Connect4 position");
            roots.push_back(Connect4::from_bitboards(my_bb[i], op_bb[i]));
            if (roots.back().is_over()) return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, "a root is not a searchable Connect4 position");
        }
        Workers workers(h, host_threads, roots.size());
        size_t rounds = 0, evals = 0;
        const auto t0 = std::chrono::steady_clock::now();
        const auto trees = lockstep_search_sharded<Connect4, 9>(workers.policies, m, roots, explores, &rounds, &evals);
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
            stats->rounds = workers.combined.combined_calls();
            stats->positions_evaluated = evals;
            stats->seconds_total = total;
            stats->seconds_policy = workers.seconds_policy();
        }
    } catch (const Error& e) {
        return syn_internal_fail(h, e.code, e.what());
    } catch (const std::exception& e) {
        return syn_internal_fail(h, SYN_ERR_INVALID_ARGUMENT, e.what());
    }
    return SYN_OK;
}

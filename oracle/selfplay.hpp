// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of the reference's self-play driver and policy wrappers:
//   synthesis/src/alpha_zero.rs:211-268  StateInfo, run_game
//   synthesis/src/alpha_zero.rs:270-294  sample_action
//   synthesis/src/alpha_zero.rs:296-338  fill_state_info, store_rewards
//   synthesis/src/alpha_zero.rs:181-209  run_n_games (one RNG + one PolicyWithCache per worker)
//   synthesis/src/config.rs:1-7,46-56    ValueTarget, RolloutConfig
//   synthesis/src/policies/cache.rs:19-32    PolicyWithCache::eval
//   synthesis/src/policies/rollout.rs:8-31   RolloutPolicy::eval
// RNG: the reference seeds ONE StdRng per worker thread (alpha_zero.rs:140,189) and threads it through all of
// that worker's games. The engine runs thousands of games concurrently, so the build defines one StdRng PER GAME,
// seeded seed_from_u64(base_seed + game_index) — i.e. a reference worker that plays exactly one game. Trajectories
// are therefore comparable oracle<->engine game for game; they equal a Rust run only under that seeding.
#pragma once
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "mcts.hpp"
#include "nn.hpp"
#include "rng.hpp"

namespace oracle {

enum ValueTargetKind : int { VT_Z = 0, VT_Q = 1, VT_QZ_AVERAGE = 2, VT_Q_TO_Z = 3 };  // config.rs:1-7

// config.rs:46-56
struct RolloutConfig {
    int num_explores = 800;
    int random_actions_until = 1;
    int sample_actions_until = 30;
    bool stop_games_when_solved = false;
    int value_target = VT_Q;
    float vt_p = 0.0f;     // QZaverage { p }
    float vt_from = 0.0f;  // QtoZ { from, to }
    float vt_to = 0.0f;
    int action = SELECT_NUM_VISITS;
    MCTSConfig mcts_cfg;
};

// cache.rs:5-32 (HashMap keyed by the game; Connect4 hashes its two bitboards)
template <class P>
struct PolicyWithCache {
    struct Entry { float logits[9]; float value[3]; };
    P* policy;
    std::unordered_map<Connect4, Entry, Connect4Hash> cache;
    uint64_t hits = 0, misses = 0;
    PolicyWithCache(size_t capacity, P* p) : policy(p) { cache.reserve(capacity); }
    void eval(const Connect4& game, float logits[9], float value[3]) {
        auto it = cache.find(game);
        if (it != cache.end()) {
            for (int i = 0; i < 9; i++) logits[i] = it->second.logits[i];
            for (int i = 0; i < 3; i++) value[i] = it->second.value[i];
            hits++;
            return;
        }
        policy->eval(game, logits, value);
        Entry e;
        for (int i = 0; i < 9; i++) e.logits[i] = logits[i];
        for (int i = 0; i < 3; i++) e.value[i] = value[i];
        cache.emplace(game, e);
        misses++;
    }
};

// rollout.rs:8-31
template <class G>
struct RolloutPolicy {
    ChaChaRng* rng;
    void eval(const G& game, float logits[G::N], float value[3]) {
        int player = game.player_id();
        G rollout = game;
        bool is_over = game.is_over();
        while (!is_over) {
            int acts[G::N];
            int n = rollout.legal_actions(acts);
            uint8_t i = rng->gen_range_u8((uint8_t)n);
            is_over = rollout.step(acts[i]);
        }
        float r = rollout.reward(player);
        for (int i = 0; i < G::N; i++) logits[i] = 0.0f;
        value[0] = value[1] = value[2] = 0.0f;
        if (r == 0.0f) value[1] = 1.0f;
        else if (r < 0.0f) value[0] = 1.0f;
        else value[2] = 1.0f;
    }
};

// One recorded position of a finished game: what ReplayBuffer::add stores (data.rs:151-158) plus bookkeeping.
struct GameRecord {
    int plies = 0;
    uint64_t my_bb[Connect4::MAX_TURNS];
    uint64_t op_bb[Connect4::MAX_TURNS];
    float pi[Connect4::MAX_TURNS][9];
    float v[Connect4::MAX_TURNS][3];
    uint8_t action[Connect4::MAX_TURNS];
    uint32_t root_nodes[Connect4::MAX_TURNS];  // nodes.len() of each move's tree
    uint8_t final_kind = 0;  // outcome for the side to move in the final position (alpha_zero.rs:258)
};

struct StateInfo {
    int turn;
    float t;
    float q[3];
    float z[3];
};

// alpha_zero.rs:270-294
template <class P>
int sample_action(const RolloutConfig& cfg, MCTS<Connect4, P>& mcts, const Connect4& game, const float* search_policy,
                  ChaChaRng& rng, int num_turns) {
    int best = mcts.best_action(cfg.action);
    OptOutcome solution = mcts.solution(best);
    if (num_turns < cfg.random_actions_until) {
        int acts[9];
        int n_legal = game.legal_actions(acts);
        int n = rng.gen_range_u8((uint8_t)n_legal);
        return acts[n];
    } else if (num_turns < cfg.sample_actions_until && (!solution.some || !cfg.stop_games_when_solved)) {
        return rng.weighted_index(search_policy, 9);
    }
    return best;
}

// alpha_zero.rs:229-268 + 296-338
// `noise_stream` = base_seed + game index: with the turn it seeds each move's tree for Fpu::Func / Dirichlet draws (noise.hpp)
template <class P>
void run_game(const RolloutConfig& cfg, P& policy, ChaChaRng& rng, GameRecord& rec, MCTSCounters* ctr = nullptr,
              uint64_t noise_stream = 0) {
    Connect4 game = Connect4::new_game();
    OptOutcome solution = OptOutcome::none();
    float search_policy[9];
    int num_turns = 0;
    std::vector<StateInfo> infos;
    infos.reserve(Connect4::MAX_TURNS);
    rec.plies = 0;

    std::vector<Node<Connect4>> storage;
    while (!solution.some) {
        MCTS<Connect4, P> mcts((size_t)cfg.num_explores + 1, cfg.mcts_cfg, &policy, game, ctr, &storage,
                               noise_tree_seed(noise_stream, (uint32_t)num_turns));
        mcts.explore_n((size_t)cfg.num_explores);

        mcts.target_policy(search_policy);
        int k = rec.plies++;
        rec.my_bb[k] = game.my_bb;
        rec.op_bb[k] = game.op_bb;
        for (int i = 0; i < 9; i++) rec.pi[k][i] = search_policy[i];
        for (int i = 0; i < 3; i++) rec.v[k][i] = 0.0f;
        rec.root_nodes[k] = (uint32_t)mcts.nodes.size();
        StateInfo si;
        si.turn = num_turns + 1;
        si.t = 0.0f;
        mcts.target_q(si.q);
        si.z[0] = si.z[1] = si.z[2] = 0.0f;
        infos.push_back(si);

        int action = sample_action(cfg, mcts, game, search_policy, rng, num_turns);
        rec.action[k] = (uint8_t)action;
        solution = mcts.solution(action);

        bool is_over = game.step(action);
        if (is_over) solution = OptOutcome::of(outcome_from_reward(game.reward(game.player_id())));
        else if (!cfg.stop_games_when_solved) solution = OptOutcome::none();
        num_turns++;
        storage = std::move(mcts.nodes);
    }
    rec.final_kind = (uint8_t)solution.o.kind;

    // fill_state_info(state_infos, solution.reversed()) — alpha_zero.rs:296-307
    Outcome outcome = reversed(solution.o);
    int n = (int)infos.size();
    for (int i = n - 1; i >= 0; i--) {
        infos[i].z[(int)outcome.kind] = 1.0f;
        infos[i].t = (float)infos[i].turn / (float)n;
        outcome = reversed(outcome);
    }
    // store_rewards — alpha_zero.rs:309-338
    for (int i = 0; i < n; i++) {
        const StateInfo& s = infos[i];
        float* v = rec.v[i];
        switch (cfg.value_target) {
            case VT_Q:
                for (int j = 0; j < 3; j++) v[j] = s.q[j];
                break;
            case VT_Z:
                for (int j = 0; j < 3; j++) v[j] = s.z[j];
                break;
            case VT_QZ_AVERAGE:
                for (int j = 0; j < 3; j++) v[j] = s.q[j] * cfg.vt_p + s.z[j] * (1.0f - cfg.vt_p);
                break;
            default: {
                float p = (1.0f - s.t) * cfg.vt_from + s.t * cfg.vt_to;
                for (int j = 0; j < 3; j++) v[j] = s.q[j] * (1.0f - p) + s.z[j] * p;
            }
        }
    }
}

}  // namespace oracle

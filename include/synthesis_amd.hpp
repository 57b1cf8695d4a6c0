// synthesis_amd — C++ host side above the C ABI: the reference's plug-in surface for this path, same names and meaning.
//
// The reference is Rust (no toolchain in this image), so the host mirror is written in C++ (header-only, C++17, links
// libsynthesis_amd.so). What mirrors what:
//   synthesis::Exploration / ActionSelection / Fpu / PolicyNoise / ValueTarget / MCTSConfig / RolloutConfig
//                                                    synthesis/src/config.rs:1-56
//   synthesis::Connect4                              study-connect4/src/connect4.rs:108-258 (Game<9>: new, player, is_over,
//                                                    reward, iter_actions, step, features)
//   synthesis::Policy<G, N>::eval                    synthesis/src/policies/traits.rs:4-6
//   synthesis::PolicyWithCache / OwnedPolicyWithCache policies/cache.rs:5-59 (host map in front of any Policy)
//   synthesis::HipPolicy                             the `impl Policy<Connect4, 9>` a maintainer adds;
//                                                    eval = batch of one, eval_batch = the throughput form
//   synthesis::ReplayBuffer                          synthesis/src/data.rs:107-235 (new_game, add, extend,
//                                                    keep_last_n_games, deduplicate, the counters)
//   synthesis::run_n_games                           synthesis/src/alpha_zero.rs:181-209
//   synthesis::vanilla_mcts_search                   MCTS over RolloutPolicy (policies/rollout.rs:8-31; mcts.rs:691-868)
//   synthesis::frozen_mcts_exploit / mcts_vs_mcts /  the evaluator's baseline and its three match loops,
//     eval_against_rollout_mcts / eval_against_old   synthesis/src/evaluator.rs:129-228, 308-319 (FrozenMCTS over RolloutPolicy)
//   synthesis::Learner                               the optimiser half of alpha_zero.rs:28-36,72-94
// Errors: the reference panics (unwrap / assert!) on this path; here every failed C-ABI call throws synthesis::Error
// carrying the status code and syn_last_error's text. Nothing is computed on the host: without the library or without
// an MI355X, Engine's constructor throws (SYN_ERR_NO_DEVICE).
#pragma once
#include <array>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "synthesis_amd.h"

namespace synthesis {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// ---- config.rs ------------------------------------------------------------------------------------------------------
enum class Exploration { Uct = SYN_EXPLORATION_UCT, PolynomialUct = SYN_EXPLORATION_POLYNOMIAL_UCT };
enum class ActionSelection { Q = SYN_ACTION_Q, NumVisits = SYN_ACTION_NUM_VISITS };
enum class Fpu { Const = SYN_FPU_CONST, ParentQ = SYN_FPU_PARENT_Q, Normal = SYN_FPU_NORMAL /* Func(|| Normal(mean, std)) */,
                 Func = SYN_FPU_FUNC /* Func(fn() -> f32), config.rs:25: host trees only */ };
enum class PolicyNoise { None = SYN_NOISE_NONE, Equal = SYN_NOISE_EQUAL, Dirichlet = SYN_NOISE_DIRICHLET };
enum class ValueTarget { Z = SYN_VALUE_Z, Q = SYN_VALUE_Q, QZaverage = SYN_VALUE_QZ_AVERAGE, QtoZ = SYN_VALUE_Q_TO_Z };

struct MCTSConfig {  // defaults: policy_mcts_cfg of study-connect4/src/main.rs:58-66
    Exploration exploration = Exploration::PolynomialUct;
    float c = 3.0f;
    bool solve = true;
    bool correct_values_on_solve = true;
    bool select_solved_nodes = true;
    bool auto_extend = true;
    Fpu fpu = Fpu::Const;
    float fpu_value = 1.0f;
    PolicyNoise root_policy_noise = PolicyNoise::None;
    float noise_alpha = 0.0f;
    float noise_weight = 0.0f;
    float fpu_std = 0.0f;  // Fpu::Normal: Func(|| Normal::new(fpu_value, fpu_std).sample(..)), study-connect4/src/main.rs:43-47
    float (*fpu_fn)() = nullptr;  // Fpu::Func(fn() -> f32) (config.rs:25), called at mcts.rs:354 by the host trees (thread-safe, as in the reference)

    syn_mcts_config to_c() const {
        syn_mcts_config m{};
        m.exploration = (int32_t)exploration; m.c = c; m.solve = solve; m.correct_values_on_solve = correct_values_on_solve;
        m.select_solved_nodes = select_solved_nodes; m.auto_extend = auto_extend; m.fpu = (int32_t)fpu;
        m.fpu_value = fpu_value; m.root_policy_noise = (int32_t)root_policy_noise; m.noise_alpha = noise_alpha;
        m.noise_weight = noise_weight; m.fpu_std = fpu_std; m.fpu_fn = fpu_fn;
        return m;
    }
};

struct RolloutConfig {  // defaults: rollout_cfg of main.rs:28-36 with explores per BASELINE.json
    int num_explores = 800;
    int random_actions_until = 1;
    int sample_actions_until = 30;
    bool stop_games_when_solved = false;
    ValueTarget value_target = ValueTarget::Q;
    float value_target_p = 0.0f, value_target_from = 0.0f, value_target_to = 0.0f;
    ActionSelection action = ActionSelection::NumVisits;
    MCTSConfig mcts_cfg;

    syn_rollout_config to_c() const {
        syn_rollout_config r{};
        r.num_explores = num_explores; r.random_actions_until = random_actions_until;
        r.sample_actions_until = sample_actions_until; r.stop_games_when_solved = stop_games_when_solved;
        r.value_target = (int32_t)value_target; r.value_target_p = value_target_p;
        r.value_target_from = value_target_from; r.value_target_to = value_target_to; r.action = (int32_t)action;
        r.mcts_cfg = mcts_cfg.to_c();
        return r;
    }
};

// ---- Game<9> for Connect4 ---------------------------------------------------------------------------------------------
// Column-major bitboards, bit = row + 7 * col (connect4.rs:3-13); my_bb belongs to the side to move.
class Connect4 {
public:
    static constexpr int MAX_NUM_ACTIONS = 9, MAX_TURNS = 63, WIDTH = 9, HEIGHT = 7, NUM_PLAYERS = 2;
    enum PlayerId { Red = 0, Black = 1 };

    Connect4() = default;                                   // Game::new: empty board, Red to move
    static Connect4 from_bitboards(uint64_t my_bb, uint64_t op_bb) {
        Connect4 g;
        g.my_ = my_bb;
        g.op_ = op_bb;
        int stones = 0;
        for (uint64_t o = my_bb | op_bb; o; o &= o - 1) stones++;
        g.player_ = (stones & 1) ? Black : Red;
        return g;
    }
    uint64_t my_bb() const { return my_; }
    uint64_t op_bb() const { return op_; }
    PlayerId player() const { return player_; }
    int height(int col) const {
        int h = 0;
        for (uint64_t c = ((my_ | op_) >> (HEIGHT * col)) & 0x7Full; c; c &= c - 1) h++;
        return h;
    }
    bool is_over() const { return four_in_a_row(op_) || full(); }
    // +1 if `p` has won, -1 if it has lost, 0 otherwise (only the side that just moved can have four in a row)
    float reward(PlayerId p) const {
        if (!four_in_a_row(op_)) return 0.0f;
        return p == player_ ? -1.0f : 1.0f;
    }
    std::vector<int> iter_actions() const {  // legal columns, ascending
        std::vector<int> a;
        a.reserve(WIDTH);
        for (int c = 0; c < WIDTH; c++)
            if (height(c) < HEIGHT) a.push_back(c);
        return a;
    }
    bool step(int col) {
        if (col < 0 || col >= WIDTH) throw Error(SYN_ERR_INVALID_ARGUMENT, "illegal Connect4 move");  // (before height(): it shifts by 7 * col)
        const int h = height(col);
        if (h >= HEIGHT) throw Error(SYN_ERR_INVALID_ARGUMENT, "illegal Connect4 move");
        const uint64_t mine = my_ | (1ull << (h + HEIGHT * col));
        my_ = op_;
        op_ = mine;
        player_ = player_ == Red ? Black : Red;
        return is_over();
    }
    // Game::features: 1x7x9 plane, index row * 9 + col; mine +1, theirs -1, empty -0.1, lowest empty cell +0.1
    std::array<float, 63> features() const {
        std::array<float, 63> f{};
        for (int col = 0; col < WIDTH; col++) {
            const int h = height(col);
            for (int row = 0; row < HEIGHT; row++) {
                const uint64_t bit = 1ull << (row + HEIGHT * col);
                float v = -0.1f;
                if (my_ & bit) v = 1.0f;
                else if (op_ & bit) v = -1.0f;
                else if (row == h) v = 0.1f;
                f[(size_t)row * WIDTH + col] = v;
            }
        }
        return f;
    }
    bool operator==(const Connect4& o) const { return my_ == o.my_ && op_ == o.op_; }

private:
    // connect4.rs:70-83: four shift-AND chains (up-left diagonal >> 6, up-right >> 8, horizontal >> 7, vertical >> 1), each masked
    // to the cells a line of that direction can start on
    static bool four_in_a_row(uint64_t bb) {
        constexpr uint64_t ROW0 = 0x0040810204081ull;                    // bit 7c, columns 0..6 (a line's first of four columns)
        constexpr uint64_t ROWS_0_3 = ROW0 * 0xFull, ROWS_3_6 = ROW0 * 0x78ull;
        constexpr uint64_t ALL_COLS_ROWS_0_3 = ROWS_0_3 | (0xFull << 49) | (0xFull << 56);
        const uint64_t d1 = bb & (bb >> 6) & (bb >> 12) & (bb >> 18) & (ROWS_3_6 & COLS_0_5);
        const uint64_t d2 = bb & (bb >> 8) & (bb >> 16) & (bb >> 24) & (ROWS_0_3 & COLS_0_5);
        const uint64_t h = bb & (bb >> 7) & (bb >> 14) & (bb >> 21) & COLS_0_5;
        const uint64_t v = bb & (bb >> 1) & (bb >> 2) & (bb >> 3) & ALL_COLS_ROWS_0_3;
        return (d1 | d2 | h | v) != 0;
    }
    static constexpr uint64_t COLS_0_5 = (1ull << 42) - 1;
    bool full() const { return (my_ | op_) == ((1ull << 63) - 1); }
    uint64_t my_ = 0, op_ = 0;
    PlayerId player_ = Red;
};

// ---- Policy ------------------------------------------------------------------------------------------------------------
template <class G, int N>
struct Policy {  // policies/traits.rs:4-6
    virtual ~Policy() = default;
    virtual std::pair<std::array<float, N>, std::array<float, 3>> eval(const G& game) = 0;
};

// ---- PolicyWithCache / OwnedPolicyWithCache (policies/cache.rs:5-59) ------------------------------------------------------
// `HashMap<G, ([f32; N], [f32; 3])>` in front of a policy: a position seen before is answered from the map, a new one goes to the
// policy and is remembered. G needs operator== and a hash (default std::hash<G>; Connect4's is below). PolicyWithCache borrows the
// policy (the reference's `&'a mut P`: run_game wraps its worker's policy for the length of one game), OwnedPolicyWithCache holds it.
// (On the device the same thing is Engine's policy_cache_log2: a lock-free table shared by all trees of a launch.)
struct Connect4Hash {
    size_t operator()(const Connect4& g) const {
        uint64_t x = g.my_bb() * 0x9E3779B97F4A7C15ull ^ (g.op_bb() + 0x7F4A7C15F39CC060ull);
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        return (size_t)x;
    }
};
template <class G> struct DefaultGameHash { using type = std::hash<G>; };
template <> struct DefaultGameHash<Connect4> { using type = Connect4Hash; };

template <class G, int N, class Hash = typename DefaultGameHash<G>::type>
class PolicyWithCache : public Policy<G, N> {
public:
    using Answer = std::pair<std::array<float, N>, std::array<float, 3>>;
    PolicyWithCache(size_t capacity, Policy<G, N>& policy) : policy(policy) { cache.reserve(capacity); }   // with_capacity
    Answer eval(const G& game) override {
        const auto it = cache.find(game);
        if (it != cache.end()) return it->second;
        const Answer pi_v = policy.eval(game);
        cache.emplace(game, pi_v);
        return pi_v;
    }
    Policy<G, N>& policy;
    std::unordered_map<G, Answer, Hash> cache;
};

template <class G, int N, class P, class Hash = typename DefaultGameHash<G>::type>
class OwnedPolicyWithCache : public Policy<G, N> {
public:
    using Answer = std::pair<std::array<float, N>, std::array<float, 3>>;
    template <class... Args>
    explicit OwnedPolicyWithCache(size_t capacity, Args&&... policy_args) : policy(std::forward<Args>(policy_args)...) { cache.reserve(capacity); }
    Answer eval(const G& game) override {
        const auto it = cache.find(game);
        if (it != cache.end()) return it->second;
        const Answer pi_v = policy.eval(game);
        cache.emplace(game, pi_v);
        return pi_v;
    }
    P policy;
    std::unordered_map<G, Answer, Hash> cache;
};

// One engine handle = one GPU + one stream + a device node pool (the reference's "one policy per worker thread").
class Engine {
public:
    // policy_cache_log2 > 0: PolicyWithCache (policies/cache.rs) on the device, a table of 2^n 64-byte entries
    Engine(int concurrent_games, int max_explores, int device = 0, int policy_cache_log2 = 0) {
        syn_engine_config cfg{};
        cfg.concurrent_games = concurrent_games;
        cfg.max_explores = max_explores;
        cfg.policy_cache_log2 = policy_cache_log2;
        const int rc = syn_engine_create(&cfg, device, &h_);
        if (rc != SYN_OK) throw Error(rc, syn_last_error(nullptr));
    }
    ~Engine() { if (h_) syn_engine_destroy(h_); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    // vs.load(model_i.ot): the VarStore's tensors in name order as one flat f32 blob
    void load_weights(const std::vector<float>& blob) { check(syn_load_weights(h_, blob.data(), blob.size())); }
    syn_engine* handle() const { return h_; }
    void check(int rc) const {
        if (rc != SYN_OK) throw Error(rc, syn_last_error(h_));
    }

private:
    syn_engine* h_ = nullptr;
};

// One worker's policy (alpha_zero.rs:192-198: every worker thread owns its policy): an evaluation context of the engine — its own
// stream and pinned staging, the engine's weights — so several HipPolicy objects on one Engine serve several host threads at once.
class HipPolicy : public Policy<Connect4, 9> {
public:
    explicit HipPolicy(Engine& e) : e_(e) { e_.check(syn_eval_ctx_create(e_.handle(), &ctx_)); }
    ~HipPolicy() override { syn_eval_ctx_destroy(ctx_); }
    HipPolicy(const HipPolicy&) = delete;
    HipPolicy& operator=(const HipPolicy&) = delete;
    std::pair<std::array<float, 9>, std::array<float, 3>> eval(const Connect4& game) override {
        const uint64_t my = game.my_bb(), op = game.op_bb();
        std::pair<std::array<float, 9>, std::array<float, 3>> out;
        const int rc = syn_eval_ctx_eval(ctx_, &my, &op, 1, out.first.data(), out.second.data());
        if (rc != SYN_OK) throw Error(rc, syn_eval_ctx_last_error(ctx_));
        return out;
    }
    // n positions per call: logits[n][9], value[n][3]
    void eval_batch(const std::vector<Connect4>& games, std::vector<std::array<float, 9>>& logits,
                    std::vector<std::array<float, 3>>& value) {
        std::vector<uint64_t> my(games.size()), op(games.size());
        for (size_t i = 0; i < games.size(); i++) { my[i] = games[i].my_bb(); op[i] = games[i].op_bb(); }
        logits.resize(games.size());
        value.resize(games.size());
        if (games.empty()) return;
        // on this object's own context (stream, staging, error slot): two HipPolicy objects of one Engine may batch from two threads
        const int rc = syn_eval_ctx_eval(ctx_, my.data(), op.data(), (int)games.size(), logits[0].data(), value[0].data());
        if (rc != SYN_OK) throw Error(rc, syn_eval_ctx_last_error(ctx_));
    }

private:
    Engine& e_;
    syn_eval_ctx* ctx_ = nullptr;
};

// ---- ReplayBuffer (data.rs:107-235) --------------------------------------------------------------------------------------
struct FlatBatch {
    std::vector<Connect4> games;
    std::vector<std::array<float, 63>> states;
    std::vector<std::array<float, 9>> pis;
    std::vector<std::array<float, 3>> vs;
};

class ReplayBuffer {
public:
    explicit ReplayBuffer(size_t n = 0) {
        game_ids_.reserve(n); games.reserve(n); pis.reserve(n); vs.reserve(n);
    }
    void new_game() { game_id_ += 1; }
    size_t total_games_played() const { return game_id_; }
    size_t curr_games() const {
        size_t n = 0;
        for (size_t i = 0; i < game_ids_.size(); i++) n += (i == 0 || game_ids_[i] != game_ids_[i - 1]);
        return n;
    }
    size_t total_steps() const { return steps_; }
    size_t curr_steps() const { return vs.size(); }
    void add(const Connect4& game, const std::array<float, 9>& pi, const std::array<float, 3>& v) {
        game_ids_.push_back(game_id_);
        steps_ += 1;
        games.push_back(game);
        pis.push_back(pi);
        vs.push_back(v);
    }
    void extend(ReplayBuffer& other) {
        steps_ += other.steps_;
        const size_t start = game_id_;
        for (size_t g : other.game_ids_) game_ids_.push_back(g + start);
        game_id_ += other.game_id_;
        games.insert(games.end(), other.games.begin(), other.games.end());
        pis.insert(pis.end(), other.pis.begin(), other.pis.end());
        vs.insert(vs.end(), other.vs.begin(), other.vs.end());
        other.games.clear(); other.pis.clear(); other.vs.clear(); other.game_ids_.clear();
    }
    void keep_last_n_games(size_t n) {
        if (game_id_ <= n) return;
        const size_t min_game_id = game_id_ - n;
        size_t drop = 0;
        while (drop < game_ids_.size() && game_ids_[drop] < min_game_id) drop++;
        game_ids_.erase(game_ids_.begin(), game_ids_.begin() + (long)drop);
        games.erase(games.begin(), games.begin() + (long)drop);
        pis.erase(pis.begin(), pis.begin() + (long)drop);
        vs.erase(vs.begin(), vs.begin() + (long)drop);
    }
    // average the targets of identical states — on the GPU (sort + segmented reduce, sums in buffer order); output in
    // ascending (my_bb, op_bb) order (the reference's HashMap order is unspecified)
    FlatBatch deduplicate(Engine& e) const {
        const size_t n = games.size();
        std::vector<uint64_t> my(n), op(n), umy(n), uop(n);
        for (size_t i = 0; i < n; i++) { my[i] = games[i].my_bb(); op[i] = games[i].op_bb(); }
        FlatBatch out;
        out.pis.resize(n);
        out.vs.resize(n);
        std::vector<uint32_t> num(n);
        size_t count = 0;
        e.check(syn_replay_deduplicate(e.handle(), my.data(), op.data(), n ? pis[0].data() : nullptr,
                                       n ? vs[0].data() : nullptr, n, umy.data(), uop.data(),
                                       n ? out.pis[0].data() : nullptr, n ? out.vs[0].data() : nullptr, num.data(), &count));
        out.pis.resize(count);
        out.vs.resize(count);
        for (size_t i = 0; i < count; i++) {
            out.games.push_back(Connect4::from_bitboards(umy[i], uop[i]));
            out.states.push_back(out.games.back().features());
        }
        return out;
    }

    std::vector<Connect4> games;
    std::vector<std::array<float, 9>> pis;
    std::vector<std::array<float, 3>> vs;

private:
    size_t game_id_ = 0, steps_ = 0;
    std::vector<size_t> game_ids_;
};

// run_n_games (alpha_zero.rs:181-209): games [first_game, first_game + num_games) played on the GPU; game g draws from
// its own StdRng::seed_from_u64(seed + g).
inline ReplayBuffer run_n_games(Engine& e, const RolloutConfig& cfg, size_t num_games, uint64_t seed,
                                uint64_t first_game = 0, syn_counters* counters = nullptr) {
    const syn_rollout_config rc = cfg.to_c();
    const size_t n = num_games, T = Connect4::MAX_TURNS;
    std::vector<int32_t> plies(n);
    std::vector<uint64_t> states(n * T * 2);
    std::vector<float> pis(n * T * 9), vs(n * T * 3);
    e.check(syn_selfplay_run(e.handle(), &rc, seed, first_game, (int)n, plies.data(), states.data(), pis.data(), vs.data(),
                             nullptr, nullptr, nullptr, counters));
    ReplayBuffer buffer(T * n);
    for (size_t g = 0; g < n; g++) {
        buffer.new_game();
        for (size_t k = 0; k < (size_t)plies[g]; k++) {
            const size_t p = g * T + k;
            std::array<float, 9> pi;
            std::array<float, 3> v;
            for (int c = 0; c < 9; c++) pi[c] = pis[p * 9 + c];
            for (int c = 0; c < 3; c++) v[c] = vs[p * 3 + c];
            buffer.add(Connect4::from_bitboards(states[p * 2], states[p * 2 + 1]), pi, v);
        }
    }
    return buffer;
}

// MCTS::with_capacity + explore_n for a batch of roots (mcts.rs:123-147); results indexed by action (column)
inline std::vector<syn_search_result> mcts_search(Engine& e, const MCTSConfig& cfg, const std::vector<Connect4>& roots,
                                                  int explores, ActionSelection action = ActionSelection::NumVisits) {
    const syn_mcts_config mc = cfg.to_c();
    std::vector<uint64_t> my(roots.size()), op(roots.size());
    for (size_t i = 0; i < roots.size(); i++) { my[i] = roots[i].my_bb(); op[i] = roots[i].op_bb(); }
    std::vector<syn_search_result> out(roots.size());
    e.check(syn_mcts_search(e.handle(), &mc, my.data(), op.data(), (int)roots.size(), explores, (int)action, out.data()));
    return out;
}

// MCTS over RolloutPolicy (policies/rollout.rs:8-31), the pairing of the reference's MCTS tests: leaf
// evaluations; root i plays its random playouts from StdRng::seed_from_u64(seed + i). Needs no network weights.
inline std::vector<syn_search_result> vanilla_mcts_search(Engine& e, const MCTSConfig& cfg, uint64_t seed,
                                                          const std::vector<Connect4>& roots, int explores,
                                                          ActionSelection action = ActionSelection::NumVisits) {
    const syn_mcts_config mc = cfg.to_c();
    std::vector<uint64_t> my(roots.size()), op(roots.size());
    for (size_t i = 0; i < roots.size(); i++) { my[i] = roots[i].my_bb(); op[i] = roots[i].op_bb(); }
    std::vector<syn_search_result> out(roots.size());
    e.check(syn_mcts_search_rollout(e.handle(), &mc, seed, my.data(), op.data(), (int)roots.size(), explores, (int)action,
                                    out.data()));
    return out;
}

// ---- evaluator.rs: the rollout baseline ("VanillaMCTS<n>") and its match loops -------------------------------------------
// The `StdRng::seed_from_u64(seed)` behind a match's `RolloutPolicy { rng }` (evaluator.rs:171-172, 207-208): the seed plus
// how many 32-bit words the match has drawn so far. The generator itself lives on the device.
struct RolloutRng {
    uint64_t seed = 0;
    uint64_t words = 0;
};

inline MCTSConfig rollout_mcts_cfg() {  // study-connect4/src/main.rs:74-82
    MCTSConfig c;
    c.exploration = Exploration::Uct;
    c.c = 2.0f;
    c.auto_extend = false;
    c.fpu = Fpu::Const;
    c.fpu_value = std::numeric_limits<float>::infinity();
    return c;
}

// FrozenMCTS::exploit(explores, cfg, &mut RolloutPolicy { rng }, game, action_selection) (evaluator.rs:308-319) for a batch
// of independent (game, generator) pairs; rngs[i] advances past the words search i used.
inline std::vector<syn_frozen_result> frozen_mcts_exploit(Engine& e, const MCTSConfig& cfg, std::vector<RolloutRng>& rngs,
                                                          const std::vector<Connect4>& roots, const std::vector<int32_t>& explores,
                                                          ActionSelection action = ActionSelection::NumVisits) {
    if (rngs.size() != roots.size() || explores.size() != roots.size())
        throw Error(SYN_ERR_INVALID_ARGUMENT, "frozen_mcts_exploit: one generator and one explore count per root");
    const syn_mcts_config mc = cfg.to_c();
    const size_t n = roots.size();
    std::vector<uint64_t> my(n), op(n), seeds(n), words(n);
    for (size_t i = 0; i < n; i++) {
        my[i] = roots[i].my_bb(); op[i] = roots[i].op_bb(); seeds[i] = rngs[i].seed; words[i] = rngs[i].words;
    }
    std::vector<syn_frozen_result> out(n);
    e.check(syn_frozen_search_rollout(e.handle(), &mc, seeds.data(), words.data(), my.data(), op.data(), explores.data(), (int)n,
                                      (int)action, out.data()));
    for (size_t i = 0; i < n; i++) rngs[i].words = words[i];
    return out;
}

// The two match loops of the evaluator, all matches of a batch in lockstep (one search call per ply and side).
// `policy_explores` < 0: both sides are baselines (mcts_vs_mcts, evaluator.rs:200-228: `player` searches with p1_explores, the
// other side with p2_explores). Otherwise eval_against_rollout_mcts (evaluator.rs:163-198): `player` is the engine's network
// under MCTS::exploit(policy_explores, policy_cfg, .., policy_action), the other side the baseline with p2_explores.
// Returns game.reward(first_player) per match (seed).
namespace detail {
inline std::vector<float> rollout_matches(Engine& e, const MCTSConfig& rollout_cfg, ActionSelection rollout_action,
                                          Connect4::PlayerId player, int p1_explores, int p2_explores,
                                          const std::vector<uint64_t>& seeds, int policy_explores,
                                          const MCTSConfig& policy_cfg, ActionSelection policy_action) {
    const size_t n = seeds.size();
    std::vector<Connect4> games(n);
    std::vector<RolloutRng> rngs(n);
    for (size_t i = 0; i < n; i++) rngs[i].seed = seeds[i];
    std::vector<float> reward(n, 0.0f);
    std::vector<char> over(n, 0);
    const Connect4::PlayerId first_player = Connect4().player();
    for (;;) {
        std::vector<size_t> idx;
        for (size_t i = 0; i < n; i++)
            if (!over[i]) idx.push_back(i);
        if (idx.empty()) break;
        // lockstep: every live match is at the same ply, so the same side is to move in all of them
        const bool players_turn = games[idx[0]].player() == player;
        std::vector<Connect4> roots;
        for (size_t i : idx) roots.push_back(games[i]);
        std::vector<int> actions(idx.size());
        if (players_turn && policy_explores >= 0) {
            auto res = mcts_search(e, policy_cfg, roots, policy_explores, policy_action);
            for (size_t k = 0; k < idx.size(); k++) actions[k] = res[k].best_action;
        } else {
            std::vector<RolloutRng> r;
            for (size_t i : idx) r.push_back(rngs[i]);
            std::vector<int32_t> ex(idx.size(), players_turn ? p1_explores : p2_explores);
            auto res = frozen_mcts_exploit(e, rollout_cfg, r, roots, ex, rollout_action);
            for (size_t k = 0; k < idx.size(); k++) { actions[k] = res[k].best_action; rngs[idx[k]] = r[k]; }
        }
        for (size_t k = 0; k < idx.size(); k++) {
            Connect4& g = games[idx[k]];
            if (g.step(actions[k])) { over[idx[k]] = 1; reward[idx[k]] = g.reward(first_player); }
        }
    }
    return reward;
}
}  // namespace detail

// evaluator.rs:129-160 eval_against_old: MCTS::exploit with network `p1` for the first player and `p2` for the second, from the
// empty board (no randomness: one game per pairing). The two checkpoints take turns on the one engine (122 KB per switch).
inline float eval_against_old(Engine& e, const MCTSConfig& policy_cfg, int policy_explores, ActionSelection policy_action,
                              const std::vector<float>& p1, const std::vector<float>& p2) {
    Connect4 game;
    const Connect4::PlayerId first_player = game.player();
    for (;;) {
        e.load_weights(game.player() == first_player ? p1 : p2);
        const int action = mcts_search(e, policy_cfg, {game}, policy_explores, policy_action)[0].best_action;
        if (game.step(action)) break;
    }
    return game.reward(first_player);
}

inline std::vector<float> mcts_vs_mcts(Engine& e, const MCTSConfig& rollout_cfg, ActionSelection rollout_action,
                                       Connect4::PlayerId player, int p1_explores, int p2_explores,
                                       const std::vector<uint64_t>& seeds) {
    return detail::rollout_matches(e, rollout_cfg, rollout_action, player, p1_explores, p2_explores, seeds, -1, MCTSConfig(),
                                   ActionSelection::NumVisits);
}

inline std::vector<float> eval_against_rollout_mcts(Engine& e, const MCTSConfig& policy_cfg, int policy_explores,
                                                    ActionSelection policy_action, const MCTSConfig& rollout_cfg,
                                                    ActionSelection rollout_action, Connect4::PlayerId player,
                                                    int opponent_explores, const std::vector<uint64_t>& seeds) {
    return detail::rollout_matches(e, rollout_cfg, rollout_action, player, 0, opponent_explores, seeds, policy_explores,
                                   policy_cfg, policy_action);
}

// The optimiser half of alpha_zero.rs:28-36,72-94 (Adam::default + weight decay; kl_div losses) on the device.
class Learner {
public:
    Learner(Engine& e, const std::vector<float>& blob, float weight_decay, float policy_weight, float value_weight)
        : e_(e) {
        syn_train_config c{weight_decay, policy_weight, value_weight, 0.9f, 0.999f, 1e-8f};
        e_.check(syn_trainer_init(e_.handle(), blob.data(), blob.size(), &c));
    }
    // one optimiser step on a batch given by positions; returns (pi_loss, v_loss) of alpha_zero.rs:79-80
    std::array<float, 2> step(const std::vector<Connect4>& games, const std::vector<std::array<float, 9>>& target_pi,
                              const std::vector<std::array<float, 3>>& target_v, float lr) {
        std::vector<uint64_t> my(games.size()), op(games.size());
        for (size_t i = 0; i < games.size(); i++) { my[i] = games[i].my_bb(); op[i] = games[i].op_bb(); }
        std::array<float, 2> losses{};
        e_.check(syn_train_step(e_.handle(), my.data(), op.data(), target_pi[0].data(), target_v[0].data(),
                                (int)games.size(), lr, losses.data()));
        return losses;
    }
    // vs.save + reload in the workers (alpha_zero.rs:97,194): the engine's self-play network becomes the trained one
    void publish() { e_.check(syn_trainer_publish_weights(e_.handle())); }

private:
    Engine& e_;
};

}  // namespace synthesis

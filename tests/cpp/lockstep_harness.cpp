// Test harness (CPU): include/synthesis_amd_lockstep.hpp driven without a GPU. The batch policy is the ORACLE's Connect4Net
// (liboracle.so, test infrastructure) — so every tree must come out identical to oracle/mcts.hpp's sequential MCTS over the same
// network — and a second, unrelated Game (a three-action subtraction game) checks that the driver is generic over Game<N>.
//   lockstep_harness c4 <blob.f32> <roots.u64> <explores> <variant> <threads> <out.bin>
//   lockstep_harness nim
//   lockstep_harness selfplay <blob.f32> <games> <explores> <variant> <threads> <seed> <first_game> <out.bin>
//   lockstep_harness rng <seed> <words>
//   lockstep_harness policythrow [end]
//   lockstep_harness threads
// <threads> < 0: the sharded drivers with -threads policies (one per host thread). Environment: LS_CONCURRENT = games in flight
// (self-play; default all), LS_ASYNC = 1: a policy whose eval_batch_begin() computes on another thread until eval_batch_end(),
// LS_COMBINE = 1: the sharded drivers' workers share ONE policy through a CombiningPolicy.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <stdexcept>

#include "synthesis_amd_lockstep.hpp"

extern "C" void orc_c4net_eval(const float* blob, const uint64_t* my_bb, const uint64_t* op_bb, int n, float* logits, float* value,
                               int mode);

using namespace synthesis;

struct OraclePolicy : BatchPolicy<Connect4, 9> {
    const float* blob;
    size_t calls = 0, positions = 0;
    void eval_batch(const std::vector<const Connect4*>& games, float* logits, float* value) override {
        std::vector<uint64_t> my(games.size()), op(games.size());
        for (size_t i = 0; i < games.size(); i++) { my[i] = games[i]->my_bb(); op[i] = games[i]->op_bb(); }
        orc_c4net_eval(blob, my.data(), op.data(), (int)games.size(), logits, value, /* ACC_FMA */ 1);
        calls++;
        positions += games.size();
    }
};

// the split form really split: begin() starts the evaluation on another thread, end() joins it — what a GPU policy does
struct AsyncOraclePolicy : OraclePolicy {
    std::thread worker;
    void eval_batch_begin(const std::vector<const Connect4*>& games, float* logits, float* value) override {
        worker = std::thread([this, &games, logits, value] { OraclePolicy::eval_batch(games, logits, value); });
    }
    void eval_batch_end() override { worker.join(); }
};
// n policies for n workers — or (LS_COMBINE = 1) one policy behind a CombiningPolicy with n workers
struct Policies {
    std::vector<std::unique_ptr<OraclePolicy>> owned;
    std::unique_ptr<CombiningPolicy<Connect4, 9>> combined;
    std::vector<std::unique_ptr<BatchPolicyWithCache<Connect4, 9>>> cached;   // LS_CACHE = 1: policies/cache.rs in front of each
    std::vector<BatchPolicy<Connect4, 9>*> list;
    Policies(size_t n, const float* blob) {
        const char* a = std::getenv("LS_ASYNC");
        const char* c = std::getenv("LS_COMBINE");
        const char* k = std::getenv("LS_CACHE");
        const bool combine = c && c[0] == '1';
        for (size_t i = 0; i < (combine ? 1 : n); i++) {
            owned.emplace_back(a && a[0] == '1' ? new AsyncOraclePolicy : new OraclePolicy);
            owned.back()->blob = blob;
            list.push_back(owned.back().get());
        }
        if (combine) {
            combined.reset(new CombiningPolicy<Connect4, 9>(*owned[0], n));
            list = combined->workers();
        }
        if (k && k[0] == '1') {
            for (auto& p : list) {
                cached.emplace_back(new BatchPolicyWithCache<Connect4, 9>(1024, *p));
                p = cached.back().get();
            }
        }
    }
    size_t calls() const { size_t c = 0; for (const auto& p : owned) c += p->calls; return c; }
    size_t positions() const { size_t c = 0; for (const auto& p : owned) c += p->positions; return c; }
    size_t hits() const { size_t c = 0; for (const auto& p : cached) c += p->hits(); return c; }
    size_t misses() const { size_t c = 0; for (const auto& p : cached) c += p->misses(); return c; }
};

// Take 1, 2 or 3 stones; whoever takes the last stone wins. Game<3>.
struct Nim {
    enum PlayerId { First = 0, Second = 1 };
    int stones = 0;
    PlayerId to_move = First;
    bool last_taken = false;
    PlayerId player() const { return to_move; }
    bool is_over() const { return stones == 0; }
    float reward(PlayerId p) const { return stones != 0 ? 0.0f : (p == to_move ? -1.0f : 1.0f); }  // the side to move has lost
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

struct UniformPolicy : BatchPolicy<Nim, 3> {
    void eval_batch(const std::vector<const Nim*>& games, float* logits, float* value) override {
        for (size_t i = 0; i < games.size(); i++) {
            for (int k = 0; k < 3; k++) logits[i * 3 + k] = 0.0f;
            value[i * 3 + 0] = value[i * 3 + 1] = value[i * 3 + 2] = 1.0f / 3.0f;
        }
    }
};

// the same game, but a position with exactly seven stones cannot be constructed: step() throws (a worker thread of the pool)
struct ThrowingNim : Nim {
    bool step(int action) {
        if (stones - (action + 1) == 7) throw std::runtime_error("seven stones");
        return Nim::step(action);
    }
};
struct UniformThrowingPolicy : BatchPolicy<ThrowingNim, 3> {
    void eval_batch(const std::vector<const ThrowingNim*>& games, float* logits, float* value) override {
        for (size_t i = 0; i < games.size(); i++) {
            for (int k = 0; k < 3; k++) logits[i * 3 + k] = 0.0f;
            value[i * 3 + 0] = value[i * 3 + 1] = value[i * 3 + 2] = 1.0f / 3.0f;
        }
    }
};

// policies/cache.rs for single positions: a policy that counts its calls, behind the borrowing and the owning wrapper
struct CountingNimPolicy : Policy<Nim, 3> {
    size_t calls = 0;
    std::pair<std::array<float, 3>, std::array<float, 3>> eval(const Nim& g) override {
        calls++;
        return {{(float)g.stones, 0.5f, -1.0f}, {0.25f, 0.5f, 0.25f}};
    }
};
struct NimHash { size_t operator()(const Nim& g) const { return (size_t)g.stones * 2u + (size_t)g.to_move; } };
inline bool operator==(const Nim& a, const Nim& b) { return a.stones == b.stones && a.to_move == b.to_move; }

// the counter-based Fpu::Func of variant 10 (tests/test_lockstep.py computes the same numbers): call k returns
// 0.5 + (k * 2654435761 mod 2^32 >> 16) / 65536, exact in f32
static uint32_t g_fpu_calls = 0;
static float counter_fpu() {
    const uint32_t h = (g_fpu_calls++) * 2654435761u;
    return 0.5f + (float)(h >> 16) * (1.0f / 65536.0f);
}

int main(int argc, char** argv) {
    if (argc >= 2 && std::string(argv[1]) == "cachepolicy") {
        CountingNimPolicy inner;
        PolicyWithCache<Nim, 3, NimHash> borrowed(16, inner);
        OwnedPolicyWithCache<Nim, 3, CountingNimPolicy, NimHash> owned(16);
        int wrong = 0;
        for (int round = 0; round < 3; round++)
            for (int st = 1; st <= 10; st++) {
                Nim g; g.stones = st; g.to_move = (st & 1) ? Nim::First : Nim::Second;
                const auto a = borrowed.eval(g), b = owned.eval(g);
                wrong += a.first[0] != (float)st || b.first[0] != (float)st || a.second[1] != 0.5f || b.second[1] != 0.5f;
            }
        Nim other; other.stones = 3; other.to_move = Nim::Second;   // same stones as an entry, other side to move: a different key
        borrowed.eval(other);
        std::printf("borrowed calls %zu entries %zu owned calls %zu entries %zu wrong %d\n", inner.calls, borrowed.cache.size(),
                    owned.policy.calls, owned.cache.size(), wrong);
        // Connect4's default hash: two positions with the stones swapped are different keys
        struct C4Count : Policy<Connect4, 9> {
            size_t calls = 0;
            std::pair<std::array<float, 9>, std::array<float, 3>> eval(const Connect4& g) override {
                calls++;
                std::pair<std::array<float, 9>, std::array<float, 3>> o{};
                o.first[0] = (float)(g.my_bb() & 0xFFFF);
                return o;
            }
        } c4;
        PolicyWithCache<Connect4, 9> c4c(8, c4);
        const Connect4 a = Connect4::from_bitboards(1, 128), b = Connect4::from_bitboards(128, 1);
        for (int i = 0; i < 4; i++) { c4c.eval(a); c4c.eval(b); }
        std::printf("connect4 calls %zu entries %zu first %g %g\n", c4.calls, c4c.cache.size(), c4c.eval(a).first[0], c4c.eval(b).first[0]);
        return 0;
    }
    if (argc >= 2 && std::string(argv[1]) == "nimthrow") {
        // 256 trees on 4 threads; the roots with 8..10 stones reach the forbidden position in their first expansion
        UniformThrowingPolicy policy;
        std::vector<ThrowingNim> roots(256);
        for (size_t i = 0; i < roots.size(); i++) roots[i].stones = 3 + (int)(i % 9);
        try {
            lockstep_search<ThrowingNim, 3>(policy, MCTSConfig{}, roots, 50, 4);
            std::printf("no exception\n");
            return 1;
        } catch (const std::runtime_error& e) {
            std::printf("caught %s\n", e.what());
        }
        // and the pool is gone with the failed search: a second search on the same thread count works
        std::vector<ThrowingNim> ok(128);
        for (auto& g : ok) g.stones = 5;
        const auto trees = lockstep_search<ThrowingNim, 3>(policy, MCTSConfig{}, ok, 50, 4);
        std::printf("second search %zu trees, root solved %d\n", trees.size(), trees[0].root().solution.some ? 1 : 0);
        return 0;
    }
    if (argc >= 2 && std::string(argv[1]) == "threads") {   // what the drivers take for threads = 0
        std::printf("%d\n", detail::usable_host_threads());
        return 0;
    }
    if (argc >= 2 && std::string(argv[1]) == "policythrow") {
        // a policy that fails in its 40th call — in begin() or, with a second argument, in end() — under every driver shape: the
        // error reaches the caller (no worker left waiting for answers that never come), and the policy can be used again
        struct FailingPolicy : BatchPolicy<Nim, 3> {
            size_t calls = 0, fail_at = 40;
            bool in_end = false, armed = false;
            void fill(const std::vector<const Nim*>& games, float* logits, float* value) {
                for (size_t i = 0; i < games.size(); i++) {
                    for (int k = 0; k < 3; k++) logits[i * 3 + k] = 0.0f;
                    value[i * 3 + 0] = value[i * 3 + 1] = value[i * 3 + 2] = 1.0f / 3.0f;
                }
            }
            void eval_batch(const std::vector<const Nim*>& games, float* logits, float* value) override {
                eval_batch_begin(games, logits, value);
                eval_batch_end();
            }
            void eval_batch_begin(const std::vector<const Nim*>& games, float* logits, float* value) override {
                fill(games, logits, value);
                if (++calls == fail_at) {
                    if (!in_end) throw std::runtime_error("policy failed in begin");
                    armed = true;
                }
            }
            void eval_batch_end() override {
                if (armed) {
                    armed = false;
                    throw std::runtime_error("policy failed in end");
                }
            }
        };
        const bool in_end = argc >= 3;
        std::vector<Nim> roots(600);
        for (size_t i = 0; i < roots.size(); i++) roots[i].stones = 9 + (int)(i % 7);
        for (int shape = 0; shape < 4; shape++) {   // pool; one thread, two halves; sharded; sharded over one combined policy
            FailingPolicy policy;
            policy.in_end = in_end;
            FailingPolicy p2, p3;
            p2.fail_at = p3.fail_at = 1u << 30;
            std::vector<BatchPolicy<Nim, 3>*> three{&policy, &p2, &p3};
            CombiningPolicy<Nim, 3> combined(policy, 5);
            try {
                if (shape == 0) lockstep_search<Nim, 3>(policy, MCTSConfig{}, roots, 80, 4);
                if (shape == 1) lockstep_search<Nim, 3>(policy, MCTSConfig{}, roots, 80, 1);
                if (shape == 2) lockstep_search_sharded<Nim, 3>(three, MCTSConfig{}, roots, 80);
                if (shape == 3) lockstep_search_sharded<Nim, 3>(combined.workers(), MCTSConfig{}, roots, 80);
                std::printf("shape %d: no exception\n", shape);
                return 1;
            } catch (const std::runtime_error& e) {
                std::printf("shape %d: caught %s\n", shape, e.what());
            }
            policy.fail_at = 1u << 30;
            const auto trees = shape == 3 ? lockstep_search_sharded<Nim, 3>(combined.workers(), MCTSConfig{}, roots, 20)
                                          : lockstep_search<Nim, 3>(policy, MCTSConfig{}, roots, 20, shape == 0 ? 4 : 1);
            std::printf("shape %d: again %zu trees\n", shape, trees.size());
        }
        return 0;
    }
    if (argc >= 2 && std::string(argv[1]) == "nim") {
        UniformPolicy policy;
        MCTSConfig cfg;
        std::vector<Nim> roots;
        for (int s = 1; s <= 14; s++) {
            Nim g;
            g.stones = s;
            roots.push_back(g);
        }
        const auto trees = lockstep_search<Nim, 3>(policy, cfg, roots, 600, 1);
        for (size_t i = 0; i < trees.size(); i++) {
            const auto& r = trees[i].root();
            std::printf("nim %d %d %d %u %d\n", roots[i].stones, r.solution.some ? 1 : 0, (int)r.solution.outcome.kind,
                        r.solution.outcome.turns, trees[i].best_action(ActionSelection::NumVisits));
        }
        return 0;
    }
    if (argc == 4 && std::string(argv[1]) == "rng") {   // the first <words> outputs of StdRng::seed_from_u64(seed)
        detail::StdRng rng(std::strtoull(argv[2], nullptr, 10));
        for (int i = 0; i < std::atoi(argv[3]); i++) std::printf("%u\n", rng.next_u32());
        return 0;
    }
    if (argc == 10 && (std::string(argv[1]) == "selfplay" || std::string(argv[1]) == "gather")) {
        const bool gather = std::string(argv[1]) == "gather";   // gather_experience with the reference's per-worker StdRng (threads = -(num_workers + 1))
        // run_n_games over host trees with the oracle's network: every game must equal oracle/selfplay.hpp's sequential run_game
        std::ifstream bf(argv[2], std::ios::binary);
        std::vector<float> blob((size_t)30492);
        bf.read(reinterpret_cast<char*>(blob.data()), (std::streamsize)(blob.size() * 4));
        const size_t games = (size_t)std::atoi(argv[3]);
        const int variant = std::atoi(argv[5]);
        RolloutConfig rc;   // variant 0: the parity rollout configuration
        rc.num_explores = std::atoi(argv[4]);
        if (variant == 1) { rc.value_target = ValueTarget::Z; rc.stop_games_when_solved = true; rc.action = ActionSelection::Q; }
        if (variant == 2) { rc.value_target = ValueTarget::QZaverage; rc.value_target_p = 0.25f; rc.random_actions_until = 3; rc.sample_actions_until = 10;
                            rc.mcts_cfg.exploration = Exploration::Uct; rc.mcts_cfg.c = 1.4f; rc.mcts_cfg.fpu = Fpu::ParentQ; }
        if (variant == 3) { rc.value_target = ValueTarget::QtoZ; rc.value_target_from = 0.1f; rc.value_target_to = 0.9f;
                            rc.mcts_cfg.root_policy_noise = PolicyNoise::Equal; rc.mcts_cfg.noise_weight = 0.25f; }
        if (variant == 4) { rc.mcts_cfg.fpu = Fpu::Normal; rc.mcts_cfg.fpu_value = 1.0f; rc.mcts_cfg.fpu_std = 0.1f; }
        if (variant == 5) { rc.mcts_cfg.root_policy_noise = PolicyNoise::Dirichlet; rc.mcts_cfg.noise_alpha = 0.3f; rc.mcts_cfg.noise_weight = 0.25f;
                            rc.mcts_cfg.fpu = Fpu::Normal; rc.mcts_cfg.fpu_value = 1.0f; rc.mcts_cfg.fpu_std = 0.1f; }
        const int threads = std::atoi(argv[6]);
        const char* conc = std::getenv("LS_CONCURRENT");
        const size_t concurrent = conc ? (size_t)std::atoll(conc) : 0;
        Policies policies(threads < 0 ? (size_t)-threads : 1, blob.data());
        size_t rounds = 0, evals = 0;
        const uint64_t seed = std::strtoull(argv[7], nullptr, 10), first_game = std::strtoull(argv[8], nullptr, 10);
        const auto recs = gather ? gather_experience_host_trees<Connect4, 9>(policies.list, rc, games, seed, &rounds, &evals)
                          : threads < 0 ? lockstep_selfplay_sharded<Connect4, 9>(policies.list, rc, games, seed, first_game, concurrent, &rounds, &evals)
                                      : lockstep_selfplay<Connect4, 9>(*policies.list[0], rc, games, seed, first_game, threads, &rounds, &evals,
                                                                       concurrent);
        // out.bin: per game [plies i32][final_kind i32] then 63 x {my u64, op u64, pi f32[9], v f32[3], action u32, root_nodes u32}
        std::ofstream of(argv[9], std::ios::binary);
        for (const auto& r : recs) {
            const int32_t head[2] = {(int32_t)r.states.size(), (int32_t)r.final_outcome.kind};
            of.write(reinterpret_cast<const char*>(head), 8);
            for (size_t k = 0; k < 63; k++) {
                uint64_t bb[2] = {0, 0};
                float f[12] = {0};
                uint32_t u[2] = {0, 0};
                if (k < r.states.size()) {
                    bb[0] = r.states[k].my_bb(); bb[1] = r.states[k].op_bb();
                    for (int c = 0; c < 9; c++) f[c] = r.pis[k][(size_t)c];
                    for (int c = 0; c < 3; c++) f[9 + c] = r.vs[k][(size_t)c];
                    u[0] = r.actions[k]; u[1] = r.root_nodes[k];
                }
                of.write(reinterpret_cast<const char*>(bb), 16);
                of.write(reinterpret_cast<const char*>(f), 48);
                of.write(reinterpret_cast<const char*>(u), 8);
            }
        }
        std::printf("rounds %zu evals %zu calls %zu\n", rounds, evals, policies.calls());
        return 0;
    }
    if (argc != 8 || std::string(argv[1]) != "c4") return 2;
    std::ifstream bf(argv[2], std::ios::binary);
    std::vector<float> blob((size_t)30492);
    bf.read(reinterpret_cast<char*>(blob.data()), (std::streamsize)(blob.size() * 4));
    std::ifstream rf(argv[3], std::ios::binary | std::ios::ate);
    const size_t n = (size_t)rf.tellg() / 16;
    rf.seekg(0);
    std::vector<uint64_t> bb(2 * n);
    rf.read(reinterpret_cast<char*>(bb.data()), (std::streamsize)(bb.size() * 8));
    const int explores = std::atoi(argv[4]), variant = std::atoi(argv[5]), threads = std::atoi(argv[6]);
    MCTSConfig cfg;  // variant 0: the parity configuration
    if (variant == 1) { cfg.exploration = Exploration::Uct; cfg.c = 1.4f; cfg.fpu = Fpu::ParentQ; }
    if (variant == 2) { cfg.root_policy_noise = PolicyNoise::Equal; cfg.noise_weight = 0.25f; cfg.auto_extend = false; cfg.select_solved_nodes = false; }
    if (variant == 3) { cfg.solve = false; cfg.fpu_value = 0.5f; }
    if (variant == 4) { cfg.correct_values_on_solve = false; cfg.c = 1.5f; }
    if (variant == 5) { cfg.fpu = Fpu::Normal; cfg.fpu_value = 1.0f; cfg.fpu_std = 0.1f; }   // study-connect4/src/main.rs:43-47
    if (variant == 6) { cfg.exploration = Exploration::Uct; cfg.c = 1.4f; cfg.fpu = Fpu::Normal; cfg.fpu_value = 0.5f; cfg.fpu_std = 0.3f; }
    if (variant == 7) { cfg.root_policy_noise = PolicyNoise::Dirichlet; cfg.noise_alpha = 0.3f; cfg.noise_weight = 0.25f; }   // mcts.rs:241-256
    if (variant == 8) { cfg.root_policy_noise = PolicyNoise::Dirichlet; cfg.noise_alpha = 1.0f; cfg.noise_weight = 0.5f;
                        cfg.fpu = Fpu::Normal; cfg.fpu_value = 1.0f; cfg.fpu_std = 0.1f; }
    if (variant == 9) { cfg.root_policy_noise = PolicyNoise::Dirichlet; cfg.noise_alpha = 2.5f; cfg.noise_weight = 0.25f; cfg.auto_extend = false; }
    // Fpu::Func(fn() -> f32) (config.rs:25, called at mcts.rs:354): a function with a state of its own. Its values depend on the order
    // of its calls, so the test searches ONE root on one thread and compares with the sequential oracle calling the same function.
    if (variant == 10) { cfg.fpu = Fpu::Func; cfg.fpu_fn = counter_fpu; g_fpu_calls = 0; }
    std::vector<Connect4> roots;
    for (size_t i = 0; i < n; i++) roots.push_back(Connect4::from_bitboards(bb[i], bb[n + i]));
    Policies policies(threads < 0 ? (size_t)-threads : 1, blob.data());
    size_t rounds = 0, evals = 0;
    const auto trees = threads < 0 ? lockstep_search_sharded<Connect4, 9>(policies.list, cfg, roots, explores, &rounds, &evals)
                                   : lockstep_search<Connect4, 9>(*policies.list[0], cfg, roots, explores, threads, &rounds, &evals);
    std::vector<syn_search_result> out(n);
    for (size_t i = 0; i < n; i++) {
        const auto& t = trees[i];
        syn_search_result r{};
        for (auto c = t.children_begin(); c != t.children_end(); ++c) {
            const int a = c->action;
            r.child_N[a] = c->num_visits;
            for (int k = 0; k < 3; k++) r.child_W[a][k] = c->outcome_probs[k];
            r.child_P[a] = c->action_prob;
            r.child_sol[a][0] = c->solution.some;
            r.child_sol[a][1] = c->solution.some ? (int)c->solution.outcome.kind : 0;
            r.child_sol[a][2] = c->solution.some ? (int)c->solution.outcome.turns : 0;
        }
        r.root_N = t.root().num_visits;
        for (int k = 0; k < 3; k++) r.root_W[k] = t.root().outcome_probs[k];
        r.root_sol[0] = t.root().solution.some;
        r.root_sol[1] = t.root().solution.some ? (int)t.root().solution.outcome.kind : 0;
        r.root_sol[2] = t.root().solution.some ? (int)t.root().solution.outcome.turns : 0;
        r.num_nodes = (uint32_t)t.num_nodes();
        r.best_action = t.best_action(ActionSelection::NumVisits);
        const auto pi = t.target_policy();
        for (int k = 0; k < 9; k++) r.target_pi[k] = pi[(size_t)k];
        const auto q = t.target_q();
        for (int k = 0; k < 3; k++) r.target_q[k] = q[(size_t)k];
        out[i] = r;
    }
    std::ofstream of(argv[7], std::ios::binary);
    of.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)(out.size() * sizeof(syn_search_result)));
    std::printf("rounds %zu evals %zu calls %zu positions %zu hits %zu misses %zu fpu_calls %u\n", rounds, evals, policies.calls(), policies.positions(),
                policies.hits(), policies.misses(), g_fpu_calls);
    return 0;
}

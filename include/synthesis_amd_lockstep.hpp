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
//   synthesis::HipBatchPolicy               BatchPolicy<Connect4, 9> over syn_policy_eval_batch
//
// Numerics: the f32 expression order of mcts.rs, exp / ln through the same deterministic restatements the device and the oracle
// use (det_expf / det_logf below). Compile with -ffp-contract=off for bit parity with syn_mcts_search (tests/test_gpu_lockstep.py
// holds this driver to the oracle and to the device search). Fpu::Normal and PolicyNoise::Dirichlet need the per-tree random
// streams of the device path and are not offered here: Error(SYN_ERR_UNSUPPORTED).
#pragma once
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>

#include "synthesis_amd.hpp"

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

// A fixed set of host threads that run fn(i) for i in [0, n), contiguous chunks (trees are independent). One pool lives for a whole
// search: its two phases per round would otherwise start and join `threads` threads 2 x (explores + 1) times.
class WorkerPool {
public:
    explicit WorkerPool(int threads) : nthreads_((size_t)(threads < 1 ? 1 : threads)) {
        for (size_t k = 1; k < nthreads_; k++) workers_.emplace_back([this, k] { loop(k); });
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
            pending_ = nthreads_ - 1;
            generation_++;
        }
        cv_.notify_all();
        std::exception_ptr mine = nullptr;
        try {
            chunk(0, f, n);
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
    void chunk(size_t k, const std::function<void(size_t)>& f, size_t n) const {
        for (size_t i = n * k / nthreads_; i < n * (k + 1) / nthreads_; i++) f(i);
    }
    void loop(size_t k) {
        size_t seen = 0;
        for (;;) {
            const std::function<void(size_t)>* f;
            size_t n;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                f = job_;
                n = n_;
            }
            try {
                chunk(k, *f, n);
            } catch (...) {
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
    size_t n_ = 0, pending_ = 0, generation_ = 0;
    bool stop_ = false;
    std::exception_ptr error_ = nullptr;
};
}  // namespace detail

// ---- game.rs:9-62 ---------------------------------------------------------------------------------------------------------
struct Outcome {
    enum Kind : uint8_t { Lose = 0, Draw = 1, Win = 2 };  // (the index of Into<usize>, mcts.rs:10-18)
    Kind kind = Draw;
    uint32_t turns = 0;

    static Outcome from_reward(float value) {  // impl From<f32>
        return Outcome{value > 0.0f ? Win : (value < 0.0f ? Lose : Draw), 0};
    }
    Outcome reversed() const { return Outcome{kind == Win ? Lose : (kind == Lose ? Win : Draw), turns + 1}; }
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
};

// ---- one tree ---------------------------------------------------------------------------------------------------------------
template <class G, int N>
class LockstepTree {
public:
    struct Node {  // mcts.rs:28-39
        uint32_t parent = 0, first_child = 0;
        uint8_t num_children = 0;
        G game;
        Solution solution;
        uint8_t action = 0;
        float action_prob = 0.0f;
        float outcome_probs[3] = {0.0f, 0.0f, 0.0f};
        float num_visits = 0.0f;

        float q() const { return (outcome_probs[2] - outcome_probs[0]) / num_visits; }
        bool is_unvisited() const { return num_children == 0 && !solution.some; }
        uint32_t last_child() const { return first_child + num_children; }
    };

    // MCTS::with_capacity(explores + 1, cfg, policy, game) followed by explore_n(explores): nothing runs before advance()
    LockstepTree(const MCTSConfig& cfg, const G& game, int explores) : cfg_(cfg), explores_(explores) {
        if (cfg.fpu == Fpu::Normal || cfg.root_policy_noise == PolicyNoise::Dirichlet)
            throw Error(SYN_ERR_UNSUPPORTED, "lockstep MCTS: Fpu::Normal / PolicyNoise::Dirichlet draw from the device path's per-tree streams");
        nodes_.reserve((size_t)explores + 1);
        Node root;
        root.game = game;
        nodes_.push_back(root);
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
    }

    uint32_t select_best_child(const Node& parent) const {  // mcts.rs:327-341: the first maximum wins, NaN never replaces
        uint32_t best = 0;
        bool have = false;
        float best_value = 0.0f;
        for (uint32_t id = parent.first_child; id < parent.last_child(); id++) {
            const Node& child = nodes_[id];
            const float value = exploit_value(parent, child) + explore_value(parent, child);
            if (!have || value > best_value) {
                have = true;
                best = id;
                best_value = value;
            }
        }
        return best;
    }

    float exploit_value(const Node& parent, const Node& child) const {  // mcts.rs:343-359
        if (child.solution.some)
            return cfg_.select_solved_nodes ? child.solution.outcome.reversed().value() : -std::numeric_limits<float>::infinity();
        if (child.num_children == 0) return cfg_.fpu == Fpu::Const ? cfg_.fpu_value : parent.q();
        return -child.q();
    }

    float explore_value(const Node& parent, const Node& child) const {  // mcts.rs:361-372
        if (cfg_.exploration == Exploration::Uct) {
            const float visits = std::sqrt(cfg_.c * detail::det_logf(parent.num_visits));
            return visits / std::sqrt(child.num_visits);
        }
        const float visits = std::sqrt(parent.num_visits);
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
    bool constructed_ = false;
    uint32_t pending_ = 0;
    bool pending_any_solved_ = false;
    std::vector<Node> nodes_;
};

// `explores` explores from every root, all trees advancing together: per round one eval_batch call with the leaves of every tree
// that still needs one. threads = host threads for the tree phases (0: hardware concurrency, at most 32). rounds_out / evals_out:
// eval_batch calls and positions evaluated.
template <class G, int N>
std::vector<LockstepTree<G, N>> lockstep_search(BatchPolicy<G, N>& policy, const MCTSConfig& cfg, const std::vector<G>& roots,
                                                int explores, int threads = 0, size_t* rounds_out = nullptr,
                                                size_t* evals_out = nullptr) {
    if (threads <= 0) threads = (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<LockstepTree<G, N>> trees;
    trees.reserve(roots.size());
    for (const G& g : roots) trees.emplace_back(cfg, g, explores);
    std::vector<const G*> want(trees.size(), nullptr);
    std::vector<const G*> batch;
    std::vector<uint32_t> owner;
    std::vector<float> logits, value;
    size_t rounds = 0, evals = 0;
    std::vector<uint32_t> live(trees.size());
    for (size_t i = 0; i < trees.size(); i++) live[i] = (uint32_t)i;
    detail::WorkerPool pool(threads);
    while (!live.empty()) {
        pool.run(live.size(), [&](size_t k) { want[live[k]] = trees[live[k]].advance(); });
        batch.clear();
        owner.clear();
        for (uint32_t t : live)
            if (want[t]) {
                batch.push_back(want[t]);
                owner.push_back(t);
            }
        if (batch.empty()) break;
        logits.resize(batch.size() * (size_t)N);
        value.resize(batch.size() * 3);
        policy.eval_batch(batch, logits.data(), value.data());
        rounds++;
        evals += batch.size();
        pool.run(owner.size(), [&](size_t k) { trees[owner[k]].supply(&logits[k * (size_t)N], &value[k * 3]); });
        live = owner;  // a tree that returned nullptr is finished
    }
    if (rounds_out) *rounds_out = rounds;
    if (evals_out) *evals_out = evals;
    return trees;
}

// BatchPolicy<Connect4, 9> on the GPU: one syn_policy_eval_batch call per round
class HipBatchPolicy : public BatchPolicy<Connect4, 9> {
public:
    explicit HipBatchPolicy(syn_engine* h) : h_(h) {}
    explicit HipBatchPolicy(Engine& e) : h_(e.handle()) {}
    void eval_batch(const std::vector<const Connect4*>& games, float* logits, float* value) override {
        my_.resize(games.size());
        op_.resize(games.size());
        for (size_t i = 0; i < games.size(); i++) { my_[i] = games[i]->my_bb(); op_[i] = games[i]->op_bb(); }
        const int rc = syn_policy_eval_batch(h_, my_.data(), op_.data(), (int)games.size(), logits, value);
        if (rc != SYN_OK) throw Error(rc, syn_last_error(h_));
    }

private:
    syn_engine* h_;
    std::vector<uint64_t> my_, op_;
};

}  // namespace synthesis

// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// extern "C" surface over the CPU restatement so that tests/ (ctypes), __graft_entry__.smoke() and bench.py's
// cpu_baseline leg can drive it. Nothing under synthesis_amd/ links, loads or calls this library.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "connect4.hpp"
#include "det_math.hpp"
#include "frozen_mcts.hpp"
#include "mcts.hpp"
#include "nn.hpp"
#include "outcome.hpp"
#include "rng.hpp"
#include "selfplay.hpp"
#include "tictactoe.hpp"
#include "train.hpp"

using namespace oracle;

extern "C" {

// ---------------------------------------------------------------- Outcome lattice (game.rs:9-66)
// kind: 0 Lose, 1 Draw, 2 Win; some: 0 = None
int orc_outcome_cmp(int a_some, int a_kind, unsigned a_turns, int b_some, int b_kind, unsigned b_turns) {
    OptOutcome a = a_some ? OptOutcome::of({(OutcomeKind)a_kind, a_turns}) : OptOutcome::none();
    OptOutcome b = b_some ? OptOutcome::of({(OutcomeKind)b_kind, b_turns}) : OptOutcome::none();
    return cmp(a, b);
}
void orc_outcome_reversed(int kind, unsigned turns, int* out_kind, unsigned* out_turns) {
    Outcome r = reversed({(OutcomeKind)kind, turns});
    *out_kind = r.kind;
    *out_turns = r.turns;
}
int orc_outcome_from_reward(float v) { return outcome_from_reward(v).kind; }
float orc_outcome_value(int kind) { return value({(OutcomeKind)kind, 0}); }

// ---------------------------------------------------------------- Connect4 (connect4.rs)
int orc_c4_won(uint64_t bb) { return Connect4::won(bb) ? 1 : 0; }

// Plays `moves` from the empty board. Per move: over[i] = return value of step. Final state written out.
// legal_before[i] / legal_after[i]: whether column moves[i] is offered by iter_actions before / after the move.
void orc_c4_play(const uint8_t* moves, int n, uint8_t* over, uint8_t* legal_before, uint8_t* legal_after,
                 uint64_t* my_bb, uint64_t* op_bb, int* player, int* winner, float* reward_red, float* reward_black,
                 float* reward_to_move) {
    Connect4 g = Connect4::new_game();
    for (int i = 0; i < n; i++) {
        int acts[9];
        int na = g.legal_actions(acts);
        legal_before[i] = 0;
        for (int j = 0; j < na; j++) legal_before[i] |= (acts[j] == moves[i]);
        over[i] = g.step(moves[i]) ? 1 : 0;
        na = g.legal_actions(acts);
        legal_after[i] = 0;
        for (int j = 0; j < na; j++) legal_after[i] |= (acts[j] == moves[i]);
    }
    *my_bb = g.my_bb;
    *op_bb = g.op_bb;
    *player = g.player;
    *winner = g.winner();
    *reward_red = g.reward(0);
    *reward_black = g.reward(1);
    *reward_to_move = g.reward(g.player_id());
}

void orc_c4_features(const uint64_t* my_bb, const uint64_t* op_bb, int n, float* out) {
    for (int i = 0; i < n; i++) Connect4::from_bitboards(my_bb[i], op_bb[i]).features(out + (size_t)i * 63);
}

// ---------------------------------------------------------------- slimnn layers
void orc_linear_forward(int I, int O, const float* W, const float* b, const float* x, int batch, float* y, int mode) {
    for (int n = 0; n < batch; n++) linear_forward(I, O, W, b, x + (size_t)n * I, y + (size_t)n * O, mode);
}
int orc_conv2d_forward(int CIN, int COUT, int K, int RP, int CP, int S, int H_IN, int W_IN, int H_OUT, int W_OUT,
                       const float* W, const float* b, const float* x, int batch, float* y, int mode) {
    for (int n = 0; n < batch; n++)
        if (!conv2d_forward(CIN, COUT, K, RP, CP, S, H_IN, W_IN, H_OUT, W_OUT, W, b,
                            x + (size_t)n * CIN * H_IN * W_IN, y + (size_t)n * COUT * H_OUT * W_OUT, mode))
            return 1;
    return 0;
}
void orc_relu(float* x, int n) { relu_inplace(x, n); }
void orc_tanh(float* x, int n) { tanh_inplace(x, n); }
// Fpu::Func draws of n_scans consecutive scans of one tree, nine child slots each (oracle/noise.hpp): out[scan][slot]
void orc_std_normal_from_bits(const uint32_t* k, int n, float* out) {
    for (int i = 0; i < n; i++) out[i] = fpu_std_normal(k[i]);   // the Fpu::Func draw's standard normal (table inversion, round 6)
}
void orc_noise_fpu_normals(uint64_t tree_seed, int n_scans, float mean, float std_dev, float* out) {
    for (int s = 0; s < n_scans; s++)
        for (int i = 0; i < 9; i++) out[(size_t)s * 9 + i] = noise_fpu_normal(tree_seed, (uint32_t)s, (uint32_t)i, mean, std_dev);
}
void orc_softmax_slimnn(const float* x, float* y, int n) { softmax_slimnn(x, y, n); }
void orc_softmax_stable(const float* x, float* y, int n) { softmax_stable(x, y, n); }
void orc_det_expf(const float* x, float* y, int n) {
    for (int i = 0; i < n; i++) y[i] = det_expf(x[i]);
}
void orc_det_logf(const float* x, float* y, int n) {
    for (int i = 0; i < n; i++) y[i] = det_logf(x[i]);
}

// ---------------------------------------------------------------- Connect4Net (policies.rs)
size_t orc_c4net_num_params() { return Connect4Net::NUM_PARAMS; }
void orc_c4net_eval(const float* blob, const uint64_t* my_bb, const uint64_t* op_bb, int n, float* logits,
                    float* value, int mode) {
    Connect4Net net;
    net.blob = blob;
    net.mode = mode;
    for (int i = 0; i < n; i++)
        net.eval(Connect4::from_bitboards(my_bb[i], op_bb[i]), logits + (size_t)i * 9, value + (size_t)i * 3);
}
size_t orc_c4conv_num_params() { return Connect4ConvNet::NUM_PARAMS; }
// Connect4ConvNet (oracle/nn.hpp): logits[n][9] raw, value[n][3] softmaxed; raw12 (optional) = the 12 raw outputs
void orc_c4conv_eval(const float* blob, const uint64_t* my_bb, const uint64_t* op_bb, int n, float* logits, float* value,
                     float* raw12, int mode) {
    Connect4ConvNet net;
    net.blob = blob;
    net.mode = mode;
    for (int i = 0; i < n; i++) {
        Connect4 g = Connect4::from_bitboards(my_bb[i], op_bb[i]);
        net.eval(g, logits + (size_t)i * 9, value + (size_t)i * 3);
        if (raw12) {
            float x[2 * Connect4ConvNet::HW];
            Connect4ConvNet::planes(g, x);
            net.forward(x, raw12 + (size_t)i * 12);
        }
    }
}
// one v_mfma_f32_16x16x32_f16 output per (a[32], b[32], c): the accumulation model of nn_f16x2.hpp (test replay of device vectors)
void orc_mfma_f16_k32(const uint16_t* a, const uint16_t* b, const float* c, int n, float* out) {
    for (int i = 0; i < n; i++) out[i] = mfma_f16_k32(c[i], a + (size_t)i * 32, b + (size_t)i * 32);
}
void orc_f16_round_trip(const float* x, int n, uint16_t* bits, float* back) {
    for (int i = 0; i < n; i++) { bits[i] = f16_bits_rne(x[i]); back[i] = f16_value(bits[i]); }
}
// the f16x2 plan of a Connect4Net blob: exponents [s0..s4, t0..t4, cexp0..3, out_exp] and the five bounds; returns 1 if the blob has one
int orc_f16x2_plan(const float* blob, int* exps15, double* bounds5) {
    F16x2Net net(blob);
    for (int l = 0; l < 5; l++) { exps15[l] = net.s[l]; exps15[5 + l] = net.t[l]; bounds5[l] = net.bound[l]; }
    for (int l = 0; l < 4; l++) exps15[10 + l] = net.cexp[l];
    exps15[14] = net.out_exp;
    return net.ok ? 1 : 0;
}
void orc_c4net_forward_raw(const float* blob, const float* x63, int n, float* out12, int mode) {
    Connect4Net net;
    net.blob = blob;
    net.mode = mode;
    for (int i = 0; i < n; i++) net.forward(x63 + (size_t)i * 63, out12 + (size_t)i * 12);
}

// ---------------------------------------------------------------- RNG (rand 0.8 StdRng restatement)
void orc_chacha_block(const uint32_t* key8, uint64_t counter, uint64_t stream, int rounds, uint32_t* out16) {
    ChaChaRng::block(key8, counter, stream, rounds, out16);
}
void orc_stdrng_from_seed_u64s(const uint8_t* seed32, int rounds, int n, uint64_t* out) {
    ChaChaRng r = ChaChaRng::from_seed(seed32, rounds);
    for (int i = 0; i < n; i++) out[i] = r.next_u64();
}
void orc_stdrng_seed_from_u64_u32s(uint64_t seed, int rounds, int n, uint32_t* out) {
    ChaChaRng r = ChaChaRng::seed_from_u64(seed, rounds);
    for (int i = 0; i < n; i++) out[i] = r.next_u32();
}
void orc_stdrng_seed_from_u64_u64s(uint64_t seed, int rounds, int n, uint64_t* out) {
    ChaChaRng r = ChaChaRng::seed_from_u64(seed, rounds);
    for (int i = 0; i < n; i++) out[i] = r.next_u64();
}
void orc_stdrng_gen_range_u8(uint64_t seed, uint8_t range, int n, uint8_t* out) {
    ChaChaRng r = ChaChaRng::seed_from_u64(seed);
    for (int i = 0; i < n; i++) out[i] = r.gen_range_u8(range);
}
void orc_stdrng_weighted_index(uint64_t seed, const float* w, int nw, int n, int* out) {
    ChaChaRng r = ChaChaRng::seed_from_u64(seed);
    for (int i = 0; i < n; i++) out[i] = r.weighted_index(w, nw);
}

// ---------------------------------------------------------------- MCTS config marshalling
struct orc_mcts_config {
    int exploration;  // 0 Uct, 1 PolynomialUct
    float c;
    int solve, correct_values_on_solve, select_solved_nodes, auto_extend;
    int fpu;  // 0 Const, 1 ParentQ, 2 Func = Normal(fpu_value, fpu_std), 3 Func(fn) with the function orc_set_fpu_fn installed
    float fpu_value;
    int noise;  // 0 None, 1 Equal, 2 Dirichlet
    float noise_alpha, noise_weight;
    float fpu_std;
};
struct orc_rollout_config {
    int num_explores, random_actions_until, sample_actions_until, stop_games_when_solved;
    int value_target;  // 0 Z, 1 Q, 2 QZaverage, 3 QtoZ
    float vt_p, vt_from, vt_to;
    int action;  // 0 Q, 1 NumVisits
    orc_mcts_config mcts;
};

// Fpu::Func(fn() -> f32) (config.rs:25): the function a test installs for configurations with fpu = 3
static float (*g_fpu_fn)() = nullptr;
extern "C" void orc_set_fpu_fn(float (*fn)()) { g_fpu_fn = fn; }

static MCTSConfig to_cfg(const orc_mcts_config& c) {
    MCTSConfig m;
    m.exploration = c.exploration;
    m.c = c.c;
    m.solve = c.solve != 0;
    m.correct_values_on_solve = c.correct_values_on_solve != 0;
    m.select_solved_nodes = c.select_solved_nodes != 0;
    m.auto_extend = c.auto_extend != 0;
    m.fpu = c.fpu == 3 ? (int)FPU_FUNC : c.fpu;
    m.fpu_fn = c.fpu == 3 ? g_fpu_fn : nullptr;
    m.fpu_value = c.fpu_value;
    m.noise = c.noise;
    m.noise_alpha = c.noise_alpha;
    m.noise_weight = c.noise_weight;
    m.fpu_std = c.fpu_std;
    return m;
}
static RolloutConfig to_rollout(const orc_rollout_config& c) {
    RolloutConfig r;
    r.num_explores = c.num_explores;
    r.random_actions_until = c.random_actions_until;
    r.sample_actions_until = c.sample_actions_until;
    r.stop_games_when_solved = c.stop_games_when_solved != 0;
    r.value_target = c.value_target;
    r.vt_p = c.vt_p;
    r.vt_from = c.vt_from;
    r.vt_to = c.vt_to;
    r.action = c.action;
    r.mcts_cfg = to_cfg(c.mcts);
    return r;
}

// ---------------------------------------------------------------- TicTacToe KATs (mcts.rs:691-868)
// which: 0 = test_solve_win, 1 = test_solve_loss, 2 = test_solve_draw. rounds selects the ChaCha variant of the
// rollout RNG (12 = rand 0.8 StdRng). Outputs: per-action child solution (some/kind/turns), best_action(Q),
// nodes.len(), root solution, target policy.
void orc_ttt_kat(int which, uint64_t seed, int rounds, int max_explores, int* child_some, int* child_kind,
                 unsigned* child_turns, int* best_action_q, unsigned* num_nodes, int* root_some, int* root_kind,
                 unsigned* root_turns, float* search_policy, float* target_q) {
    ChaChaRng rng = ChaChaRng::seed_from_u64(seed, rounds);
    RolloutPolicy<TicTacToe> policy{&rng};
    TicTacToe game = TicTacToe::new_game();
    if (which == 0) { game.step(0); game.step(2); }
    else if (which == 1) { game.step(0); game.step(2); game.step(6); }
    else { game.step(0); game.step(4); }
    MCTSConfig cfg;
    cfg.exploration = POLYNOMIAL_UCT;
    cfg.c = 2.0f;
    cfg.solve = true;
    cfg.correct_values_on_solve = true;
    cfg.select_solved_nodes = true;
    cfg.auto_extend = true;
    cfg.fpu = FPU_CONST;
    cfg.fpu_value = std::numeric_limits<float>::infinity();
    MCTS<TicTacToe, RolloutPolicy<TicTacToe>> mcts(1601, cfg, &policy, game);
    int it = 0;
    while (!mcts.nodes[mcts.root].solution.some && it < max_explores) {
        mcts.explore();
        it++;
    }
    for (int a = 0; a < 9; a++) {
        OptOutcome s = mcts.solution(a);
        child_some[a] = s.some;
        child_kind[a] = s.o.kind;
        child_turns[a] = s.o.turns;
    }
    *best_action_q = mcts.best_action(SELECT_Q);
    *num_nodes = (unsigned)mcts.nodes.size();
    *root_some = mcts.nodes[mcts.root].solution.some;
    *root_kind = mcts.nodes[mcts.root].solution.o.kind;
    *root_turns = mcts.nodes[mcts.root].solution.o.turns;
    mcts.target_policy(search_policy);
    mcts.target_q(target_q);
}

// test_add_noise's first half (mcts.rs:834-859): priors of the root's children after construction.
void orc_ttt_root_priors(uint64_t seed, float* priors9, int* n_children) {
    ChaChaRng rng = ChaChaRng::seed_from_u64(seed);
    RolloutPolicy<TicTacToe> policy{&rng};
    MCTSConfig cfg;
    cfg.exploration = POLYNOMIAL_UCT;
    cfg.c = 2.0f;
    cfg.select_solved_nodes = false;
    cfg.auto_extend = false;
    cfg.fpu_value = std::numeric_limits<float>::infinity();
    MCTS<TicTacToe, RolloutPolicy<TicTacToe>> mcts(1601, cfg, &policy, TicTacToe::new_game());
    const auto& r = mcts.nodes[mcts.root];
    *n_children = r.num_children;
    for (uint32_t c = r.first_child; c < r.last_child(); c++) priors9[c - r.first_child] = mcts.nodes[c].action_prob;
}

// ---------------------------------------------------------------- Connect4 MCTS search on given roots
// For each root i: builds MCTS::with_capacity(explores+1, cfg, policy, root) and runs explore_n(explores).
// Outputs per root: child_* arrays are indexed by ACTION (column), zero where the column is not a child.
}  // extern "C"
template <class P>
static void c4_search_collect(MCTS<Connect4, P>& mcts, int i, int action_selection, float* child_N, float* child_W,
                              float* child_P, int* child_sol, float* root_stat, int* root_sol, unsigned* num_nodes,
                              int* best_action, float* target_pi, float* target_q) {
    const auto& r = mcts.nodes[mcts.root];
    for (int a = 0; a < 9; a++) {
        child_N[i * 9 + a] = 0;
        child_P[i * 9 + a] = 0;
        for (int j = 0; j < 3; j++) child_W[(i * 9 + a) * 3 + j] = 0;
        for (int j = 0; j < 3; j++) child_sol[(i * 9 + a) * 3 + j] = 0;
    }
    for (uint32_t c = r.first_child; c < r.last_child(); c++) {
        const auto& ch = mcts.nodes[c];
        int a = ch.action;
        child_N[i * 9 + a] = ch.num_visits;
        child_P[i * 9 + a] = ch.action_prob;
        for (int j = 0; j < 3; j++) child_W[(i * 9 + a) * 3 + j] = ch.outcome_probs[j];
        child_sol[(i * 9 + a) * 3 + 0] = ch.solution.some;
        child_sol[(i * 9 + a) * 3 + 1] = ch.solution.o.kind;
        child_sol[(i * 9 + a) * 3 + 2] = (int)ch.solution.o.turns;
    }
    root_stat[i * 4 + 0] = r.num_visits;
    for (int j = 0; j < 3; j++) root_stat[i * 4 + 1 + j] = r.outcome_probs[j];
    root_sol[i * 3 + 0] = r.solution.some;
    root_sol[i * 3 + 1] = r.solution.o.kind;
    root_sol[i * 3 + 2] = (int)r.solution.o.turns;
    num_nodes[i] = (unsigned)mcts.nodes.size();
    best_action[i] = mcts.best_action(action_selection);
    mcts.target_policy(target_pi + i * 9);
    mcts.target_q(target_q + i * 3);
}
template <class Net>
static void c4_mcts_search_impl(const orc_mcts_config* cfg_in, const float* blob, int nn_mode, const uint64_t* my_bb,
                                const uint64_t* op_bb, int n, int explores, int action_selection, float* child_N,
                                float* child_W, float* child_P, int* child_sol, float* root_stat, int* root_sol,
                                unsigned* num_nodes, int* best_action, float* target_pi, float* target_q) {
    MCTSConfig cfg = to_cfg(*cfg_in);
    Net net;
    net.blob = blob;
    net.mode = nn_mode;
    for (int i = 0; i < n; i++) {
        Connect4 root = Connect4::from_bitboards(my_bb[i], op_bb[i]);
        // root i of a bare search: noise stream = i (the device: base_seed 0 + root index), turn 0
        MCTS<Connect4, Net> mcts((size_t)explores + 1, cfg, &net, root, nullptr, nullptr, noise_tree_seed((uint64_t)i, 0));
        mcts.explore_n((size_t)explores);
        c4_search_collect(mcts, i, action_selection, child_N, child_W, child_P, child_sol, root_stat, root_sol, num_nodes,
                          best_action, target_pi, target_q);
    }
}
extern "C" {

void orc_c4_mcts_search(const orc_mcts_config* cfg_in, const float* blob, int nn_mode, const uint64_t* my_bb,
                        const uint64_t* op_bb, int n, int explores, int action_selection, float* child_N,
                        float* child_W, float* child_P, int* child_sol, float* root_stat, int* root_sol,
                        unsigned* num_nodes, int* best_action, float* target_pi, float* target_q) {
    c4_mcts_search_impl<Connect4Net>(cfg_in, blob, nn_mode, my_bb, op_bb, n, explores, action_selection, child_N, child_W, child_P,
                                     child_sol, root_stat, root_sol, num_nodes, best_action, target_pi, target_q);
}
// the same search with Connect4ConvNet (oracle/nn.hpp) as the policy
void orc_c4conv_mcts_search(const orc_mcts_config* cfg_in, const float* blob, int nn_mode, const uint64_t* my_bb,
                            const uint64_t* op_bb, int n, int explores, int action_selection, float* child_N,
                            float* child_W, float* child_P, int* child_sol, float* root_stat, int* root_sol,
                            unsigned* num_nodes, int* best_action, float* target_pi, float* target_q) {
    c4_mcts_search_impl<Connect4ConvNet>(cfg_in, blob, nn_mode, my_bb, op_bb, n, explores, action_selection, child_N, child_W,
                                         child_P, child_sol, root_stat, root_sol, num_nodes, best_action, target_pi, target_q);
}

// MCTS<Connect4, RolloutPolicy> — the pairing the reference's own MCTS tests use (mcts.rs:691-868, there with
// TicTacToe); RolloutPolicy = rollout.rs:8-31. (The evaluator's rollout players use the separate FrozenMCTS tree.) root i draws
// its playouts from its own StdRng::seed_from_u64(seed + i), consumed in explore order.
void orc_c4_mcts_search_rollout(const orc_mcts_config* cfg_in, uint64_t seed, const uint64_t* my_bb, const uint64_t* op_bb,
                                int n, int explores, int action_selection, float* child_N, float* child_W,
                                float* child_P, int* child_sol, float* root_stat, int* root_sol, unsigned* num_nodes,
                                int* best_action, float* target_pi, float* target_q) {
    MCTSConfig cfg = to_cfg(*cfg_in);
    for (int i = 0; i < n; i++) {
        ChaChaRng rng = ChaChaRng::seed_from_u64(seed + (uint64_t)i);
        RolloutPolicy<Connect4> policy{&rng};
        Connect4 root = Connect4::from_bitboards(my_bb[i], op_bb[i]);
        MCTS<Connect4, RolloutPolicy<Connect4>> mcts((size_t)explores + 1, cfg, &policy, root);
        mcts.explore_n((size_t)explores);
        c4_search_collect(mcts, i, action_selection, child_N, child_W, child_P, child_sol, root_stat, root_sol, num_nodes,
                          best_action, target_pi, target_q);
    }
}

}  // extern "C"

// 32-bit words a generator has handed out so far (the buffer holds 64 words = four blocks per refill)
static uint64_t words_consumed(const ChaChaRng& r) { return r.counter == 0 ? 0 : r.counter * 16 - 64 + (uint64_t)r.index; }

// FrozenMCTS over RolloutPolicy (evaluator.rs:230-534 + policies/rollout.rs:8-31) on Connect4. Root i plays out on
// StdRng::seed_from_u64(seeds[i]) after skipping rng_words[i] output words; rng_words[i] returns the words consumed in total.
// policy_kind 0 = Connect4Net instead (restatement check only; the reference never pairs the baseline with a network).
// Outputs per root: child_N / child_cum / child_P [9] by action, child_sol [9][3], root {N, cum}, root_sol [3],
// num_nodes, best_action.
template <class M>
static void frozen_collect(const M& mcts, int i, int action_selection, float* child_N, float* child_cum, float* child_P,
                           int* child_sol, float* root_stat, int* root_sol, unsigned* num_nodes, int* best_action) {
    const auto& r = mcts.nodes[mcts.root];
    for (int a = 0; a < 9; a++) {
        child_N[i * 9 + a] = child_cum[i * 9 + a] = child_P[i * 9 + a] = 0;
        for (int j = 0; j < 3; j++) child_sol[(i * 9 + a) * 3 + j] = 0;
    }
    for (uint32_t c = r.first_child; c < r.last_child(); c++) {
        const auto& ch = mcts.nodes[c];
        int a = ch.action;
        child_N[i * 9 + a] = ch.num_visits;
        child_cum[i * 9 + a] = ch.cum_value;
        child_P[i * 9 + a] = ch.action_prob;
        child_sol[(i * 9 + a) * 3 + 0] = ch.solution.some;
        child_sol[(i * 9 + a) * 3 + 1] = ch.solution.some ? ch.solution.o.kind : 0;
        child_sol[(i * 9 + a) * 3 + 2] = ch.solution.some ? (int)ch.solution.o.turns : 0;
    }
    root_stat[i * 2 + 0] = r.num_visits;
    root_stat[i * 2 + 1] = r.cum_value;
    root_sol[i * 3 + 0] = r.solution.some;
    root_sol[i * 3 + 1] = r.solution.some ? r.solution.o.kind : 0;
    root_sol[i * 3 + 2] = r.solution.some ? (int)r.solution.o.turns : 0;
    num_nodes[i] = (unsigned)mcts.nodes.size();
    best_action[i] = mcts.best_action(action_selection);
}

extern "C" {

void orc_c4_frozen_search(const orc_mcts_config* cfg_in, int policy_kind, const float* blob, int nn_mode,
                          const uint64_t* seeds, uint64_t* rng_words, const uint64_t* my_bb, const uint64_t* op_bb, int n,
                          const int* explores, int action_selection, float* child_N, float* child_cum, float* child_P,
                          int* child_sol, float* root_stat, int* root_sol, unsigned* num_nodes, int* best_action) {
    MCTSConfig cfg = to_cfg(*cfg_in);
    for (int i = 0; i < n; i++) {
        Connect4 root = Connect4::from_bitboards(my_bb[i], op_bb[i]);
        if (policy_kind == 0) {
            Connect4Net net;
            net.blob = blob;
            net.mode = nn_mode;
            FrozenMCTS<Connect4, Connect4Net> mcts((size_t)explores[i] + 1, cfg, &net, root);
            mcts.explore_n((size_t)explores[i]);
            frozen_collect(mcts, i, action_selection, child_N, child_cum, child_P, child_sol, root_stat, root_sol, num_nodes,
                           best_action);
        } else {
            ChaChaRng rng = ChaChaRng::seed_from_u64(seeds[i]);
            for (uint64_t w = 0; w < rng_words[i]; w++) (void)rng.next_u32();
            RolloutPolicy<Connect4> policy{&rng};
            FrozenMCTS<Connect4, RolloutPolicy<Connect4>> mcts((size_t)explores[i] + 1, cfg, &policy, root);
            mcts.explore_n((size_t)explores[i]);
            frozen_collect(mcts, i, action_selection, child_N, child_cum, child_P, child_sol, root_stat, root_sol, num_nodes,
                           best_action);
            rng_words[i] = words_consumed(rng);
        }
    }
}

// evaluator.rs:200-228 mcts_vs_mcts: two rollout baselines on ONE generator. `player` (0 = the side that moves first) searches
// with p1_explores, the other side with p2_explores. Returns game.reward(first_player); moves[] / words_after[] per ply.
float orc_c4_mcts_vs_mcts(const orc_mcts_config* rollout_cfg, int rollout_action, int player, int p1_explores,
                          int p2_explores, uint64_t seed, uint8_t* moves, int* n_moves, uint64_t* words_after) {
    MCTSConfig cfg = to_cfg(*rollout_cfg);
    ChaChaRng rng = ChaChaRng::seed_from_u64(seed);
    RolloutPolicy<Connect4> policy{&rng};
    Connect4 game;
    const int first_player = game.player_id();
    int n = 0;
    for (;;) {
        const int explores = game.player_id() == player ? p1_explores : p2_explores;
        FrozenMCTS<Connect4, RolloutPolicy<Connect4>> mcts((size_t)explores + 1, cfg, &policy, game);
        mcts.explore_n((size_t)explores);
        const int action = mcts.best_action(rollout_action);
        moves[n] = (uint8_t)action;
        words_after[n] = words_consumed(rng);
        n++;
        if (game.step(action)) break;
    }
    *n_moves = n;
    return game.reward(first_player);
}

// evaluator.rs:163-198 eval_against_rollout_mcts: the network's MCTS::exploit as `player`, the rollout baseline as the other side.
float orc_c4_eval_against_rollout(const orc_mcts_config* policy_cfg, int policy_explores, int policy_action,
                                  const float* blob, int nn_mode, const orc_mcts_config* rollout_cfg, int rollout_action,
                                  int player, int opponent_explores, uint64_t seed, uint8_t* moves, int* n_moves,
                                  uint64_t* words_after) {
    MCTSConfig pcfg = to_cfg(*policy_cfg), rcfg = to_cfg(*rollout_cfg);
    Connect4Net net;
    net.blob = blob;
    net.mode = nn_mode;
    ChaChaRng rng = ChaChaRng::seed_from_u64(seed);
    RolloutPolicy<Connect4> policy{&rng};
    Connect4 game;
    const int first_player = game.player_id();
    int n = 0;
    for (;;) {
        int action;
        if (game.player_id() == player) {
            MCTS<Connect4, Connect4Net> mcts((size_t)policy_explores + 1, pcfg, &net, game);  // mcts.rs:110-121 exploit
            mcts.explore_n((size_t)policy_explores);
            action = mcts.best_action(policy_action);
        } else {
            FrozenMCTS<Connect4, RolloutPolicy<Connect4>> mcts((size_t)opponent_explores + 1, rcfg, &policy, game);
            mcts.explore_n((size_t)opponent_explores);
            action = mcts.best_action(rollout_action);
        }
        moves[n] = (uint8_t)action;
        words_after[n] = words_consumed(rng);
        n++;
        if (game.step(action)) break;
    }
    *n_moves = n;
    return game.reward(first_player);
}

// evaluator.rs:129-160 eval_against_old: MCTS::exploit with network p1 for the first player, p2 for the second, from the
// empty board. Returns game.reward(first_player).
float orc_c4_eval_against_old(const orc_mcts_config* policy_cfg, int policy_explores, int policy_action, const float* blob1,
                              const float* blob2, int nn_mode, uint8_t* moves, int* n_moves) {
    MCTSConfig cfg = to_cfg(*policy_cfg);
    Connect4Net p1, p2;
    p1.blob = blob1; p1.mode = nn_mode;
    p2.blob = blob2; p2.mode = nn_mode;
    Connect4 game;
    const int first_player = game.player_id();
    int n = 0;
    for (;;) {
        MCTS<Connect4, Connect4Net> mcts((size_t)policy_explores + 1, cfg, game.player_id() == first_player ? &p1 : &p2, game);
        mcts.explore_n((size_t)policy_explores);
        const int action = mcts.best_action(policy_action);
        moves[n++] = (uint8_t)action;
        if (game.step(action)) break;
    }
    *n_moves = n;
    return game.reward(first_player);
}

// ---------------------------------------------------------------- self-play (run_n_games / gather_experience)
// Plays games [first_game, first_game + n_games) with per-game RNG seed_from_u64(base_seed + game_index), split over
// `threads` OS threads the way gather_experience splits workers (alpha_zero.rs:132-154): each thread owns its
// policy copy and, if use_cache, its own PolicyWithCache (alpha_zero.rs:196-198).
// Outputs are indexed by (game - first_game): plies[n], states_bb[n][63][2], pis[n][63][9], vs[n][63][3],
// actions[n][63], root_nodes[n][63], final_kind[n]. counters[9] = MCTSCounters fields summed; counters[9..10] =
// cache hits, misses; counters[11] = max backprop depth. Returns wall seconds.
}  // extern "C"
template <class Net>
static double c4_selfplay_impl(const orc_rollout_config* cfg_in, const float* blob, int nn_mode, uint64_t base_seed,
                       uint64_t first_game, int n_games, int threads, int use_cache, int* plies, uint64_t* states_bb,
                       float* pis, float* vs, uint8_t* actions, uint32_t* root_nodes, uint8_t* final_kind,
                       uint64_t* counters, bool worker_rng = false) {
    RolloutConfig cfg = to_rollout(*cfg_in);
    if (threads < 1) threads = 1;
    std::vector<MCTSCounters> ctrs(threads);
    std::vector<uint64_t> hits(threads, 0), misses(threads, 0);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int w) {
        // games_to_schedule / workers_left split, alpha_zero.rs:138
        int start = 0, count = 0, left = n_games;
        for (int i = 0; i <= w; i++) {
            start += count;
            count = left / (threads - i);
            left -= count;
        }
        Net net;
        net.blob = blob;
        net.mode = nn_mode;
        PolicyWithCache<Net> cached((size_t)Connect4::MAX_TURNS * (size_t)(count > 0 ? count : 1), &net);
        // counters live on the worker's own stack while it runs: neighbouring elements of `ctrs` share cache lines, and
        // hundreds of threads incrementing them would time false sharing instead of the search
        MCTSCounters local;
        // worker_rng: the reference's own discipline (alpha_zero.rs:140,189,201-205) — ONE StdRng per worker, seeded
        // seed * (num_workers + 1) + i_worker, running through all of that worker's games one after another; `base_seed` is the
        // iteration's `seed`, `threads` = num_workers + 1. (Otherwise every game has a generator of its own: DESIGN.md §7.)
        const uint64_t worker_seed = base_seed * (uint64_t)threads + (uint64_t)w;
        ChaChaRng wrng = ChaChaRng::seed_from_u64(worker_seed);
        for (int g = start; g < start + count; g++) {
            ChaChaRng grng = ChaChaRng::seed_from_u64(base_seed + first_game + (uint64_t)g);
            ChaChaRng& rng = worker_rng ? wrng : grng;
            // the trees' noise streams (Fpu::Func / Dirichlet; thread_rng in the reference): per game in both disciplines
            const uint64_t stream = worker_rng ? (worker_seed << 32) + (uint64_t)(g - start) : base_seed + first_game + (uint64_t)g;
            GameRecord rec;
            if (use_cache) run_game(cfg, cached, rng, rec, &local, stream);
            else run_game(cfg, net, rng, rec, &local, stream);
            if (plies) plies[g] = rec.plies;
            if (final_kind) final_kind[g] = rec.final_kind;
            for (int k = 0; k < rec.plies; k++) {
                size_t p = (size_t)g * 63 + k;
                if (states_bb) { states_bb[p * 2] = rec.my_bb[k]; states_bb[p * 2 + 1] = rec.op_bb[k]; }
                if (pis) for (int j = 0; j < 9; j++) pis[p * 9 + j] = rec.pi[k][j];
                if (vs) for (int j = 0; j < 3; j++) vs[p * 3 + j] = rec.v[k][j];
                if (actions) actions[p] = rec.action[k];
                if (root_nodes) root_nodes[p] = rec.root_nodes[k];
            }
        }
        ctrs[w] = local;
        hits[w] = cached.hits;
        misses[w] = cached.misses;
    };
    std::vector<std::thread> pool;
    for (int w = 1; w < threads; w++) pool.emplace_back(worker, w);
    worker(0);
    for (auto& t : pool) t.join();
    auto t1 = std::chrono::steady_clock::now();
    if (counters) {
        for (int i = 0; i < 12; i++) counters[i] = 0;
        for (int w = 0; w < threads; w++) {
            const MCTSCounters& c = ctrs[w];
            uint64_t v[9] = {c.explores, c.select_levels, c.children_scanned, c.expansions, c.new_nodes,
                             c.policy_evals, c.backprop_levels, c.solver_children, c.solved_hits};
            for (int i = 0; i < 9; i++) counters[i] += v[i];
            counters[9] += hits[w];
            counters[10] += misses[w];
            if (c.max_depth > counters[11]) counters[11] = c.max_depth;
        }
    }
    return std::chrono::duration<double>(t1 - t0).count();
}

extern "C" {

double orc_c4_selfplay(const orc_rollout_config* cfg_in, const float* blob, int nn_mode, uint64_t base_seed,
                       uint64_t first_game, int n_games, int threads, int use_cache, int* plies, uint64_t* states_bb,
                       float* pis, float* vs, uint8_t* actions, uint32_t* root_nodes, uint8_t* final_kind,
                       uint64_t* counters) {
    return c4_selfplay_impl<Connect4Net>(cfg_in, blob, nn_mode, base_seed, first_game, n_games, threads, use_cache, plies, states_bb,
                                         pis, vs, actions, root_nodes, final_kind, counters);
}
// gather_experience as the reference runs it (alpha_zero.rs:120-209): num_workers + 1 workers, worker i plays games_to_schedule /
// workers_left games ONE AFTER ANOTHER on one StdRng::seed_from_u64(seed * (num_workers + 1) + i_worker) — the games depend on the
// number of workers, as they do in the reference. Outputs in worker order (buffer.extend per worker, alpha_zero.rs:165-168).
double orc_c4_gather_experience(const orc_rollout_config* cfg_in, const float* blob, int nn_mode, uint64_t seed, int n_games,
                                int workers_plus_1, int use_cache, int* plies, uint64_t* states_bb, float* pis, float* vs,
                                uint8_t* actions, uint32_t* root_nodes, uint8_t* final_kind, uint64_t* counters) {
    return c4_selfplay_impl<Connect4Net>(cfg_in, blob, nn_mode, seed, 0, n_games, workers_plus_1, use_cache, plies, states_bb, pis, vs,
                                         actions, root_nodes, final_kind, counters, /*worker_rng=*/true);
}
// the same games with Connect4ConvNet (oracle/nn.hpp) as the policy
double orc_c4conv_selfplay(const orc_rollout_config* cfg_in, const float* blob, int nn_mode, uint64_t base_seed,
                           uint64_t first_game, int n_games, int threads, int use_cache, int* plies, uint64_t* states_bb,
                           float* pis, float* vs, uint8_t* actions, uint32_t* root_nodes, uint8_t* final_kind,
                           uint64_t* counters) {
    return c4_selfplay_impl<Connect4ConvNet>(cfg_in, blob, nn_mode, base_seed, first_game, n_games, threads, use_cache, plies,
                                             states_bb, pis, vs, actions, root_nodes, final_kind, counters);
}

// ---------------------------------------------------------------- training step + replay de-duplication
struct orc_train_hyper {
    float weight_decay, policy_weight, value_weight, beta1, beta2, eps;
};
static TrainHyper to_hyper(const orc_train_hyper& h) {
    TrainHyper t;
    t.weight_decay = h.weight_decay;
    t.policy_weight = h.policy_weight;
    t.value_weight = h.value_weight;
    t.beta1 = h.beta1;
    t.beta2 = h.beta2;
    t.eps = h.eps;
    return t;
}
// Gradients of one minibatch (features X[B][63]); grad_out[30492]; losses[2].
void orc_train_gradients(const float* blob, const orc_train_hyper* hp, const float* X, const float* tpi,
                         const float* tv, int B, float* grad_out, float* losses) {
    Trainer t(blob, to_hyper(*hp));
    t.gradients(X, tpi, tv, B, losses);
    std::memcpy(grad_out, t.grad.data(), t.grad.size() * sizeof(float));
}
// n_steps optimiser steps on consecutive minibatches X[n_steps][B][63] ...; state (m, v, step) starts at zero unless
// m_io / v_io / step_io are given (then they are read and written back). losses[n_steps][2].
void orc_train_steps(float* blob_io, const orc_train_hyper* hp, const float* X, const float* tpi, const float* tv,
                     int B, int n_steps, const float* lrs, float* m_io, float* v_io, long long* step_io,
                     float* losses) {
    Trainer t(blob_io, to_hyper(*hp));
    if (m_io) std::memcpy(t.m.data(), m_io, t.m.size() * sizeof(float));
    if (v_io) std::memcpy(t.v.data(), v_io, t.v.size() * sizeof(float));
    if (step_io) t.step = *step_io;
    for (int s = 0; s < n_steps; s++)
        t.train_step(X + (size_t)s * B * 63, tpi + (size_t)s * B * 9, tv + (size_t)s * B * 3, B, lrs[s], losses + 2 * s);
    std::memcpy(blob_io, t.w.data(), t.w.size() * sizeof(float));
    if (m_io) std::memcpy(m_io, t.m.data(), t.m.size() * sizeof(float));
    if (v_io) std::memcpy(v_io, t.v.data(), t.v.size() * sizeof(float));
    if (step_io) *step_io = t.step;
}
// The same two entry points for Connect4ConvNet (train.hpp ConvTrainer); positions as bitboards, blob / grads 12,412 floats.
void orc_convtrain_gradients(const float* blob, const orc_train_hyper* hp, const uint64_t* my_bb, const uint64_t* op_bb,
                             const float* tpi, const float* tv, int B, float* grad_out, float* losses) {
    ConvTrainer t(blob, to_hyper(*hp));
    t.gradients(my_bb, op_bb, tpi, tv, B, losses);
    std::memcpy(grad_out, t.grad.data(), t.grad.size() * sizeof(float));
}
void orc_convtrain_steps(float* blob_io, const orc_train_hyper* hp, const uint64_t* my_bb, const uint64_t* op_bb, const float* tpi,
                         const float* tv, int B, int n_steps, const float* lrs, float* m_io, float* v_io, long long* step_io,
                         float* losses) {
    ConvTrainer t(blob_io, to_hyper(*hp));
    if (m_io) std::memcpy(t.m.data(), m_io, t.m.size() * sizeof(float));
    if (v_io) std::memcpy(t.v.data(), v_io, t.v.size() * sizeof(float));
    if (step_io) t.step = *step_io;
    for (int s = 0; s < n_steps; s++) {
        t.gradients(my_bb + (size_t)s * B, op_bb + (size_t)s * B, tpi + (size_t)s * B * 9, tv + (size_t)s * B * 3, B, losses + 2 * s);
        t.adam(lrs[s]);
    }
    std::memcpy(blob_io, t.w.data(), t.w.size() * sizeof(float));
    if (m_io) std::memcpy(m_io, t.m.data(), t.m.size() * sizeof(float));
    if (v_io) std::memcpy(v_io, t.v.data(), t.v.size() * sizeof(float));
    if (step_io) *step_io = t.step;
}
// Adam only (after an external gradient all-reduce)
void orc_train_adam(float* blob_io, const orc_train_hyper* hp, const float* grad, float lr, float* m_io, float* v_io,
                    long long* step_io) {
    Trainer t(blob_io, to_hyper(*hp));
    std::memcpy(t.m.data(), m_io, t.m.size() * sizeof(float));
    std::memcpy(t.v.data(), v_io, t.v.size() * sizeof(float));
    std::memcpy(t.grad.data(), grad, t.grad.size() * sizeof(float));
    t.step = *step_io;
    t.adam(lr);
    std::memcpy(blob_io, t.w.data(), t.w.size() * sizeof(float));
    std::memcpy(m_io, t.m.data(), t.m.size() * sizeof(float));
    std::memcpy(v_io, t.v.data(), t.v.size() * sizeof(float));
    *step_io = t.step;
}
// data.rs:196-235. Outputs sized for n entries; returns the number of unique states (ascending (my, op)).
size_t orc_dedup(const uint64_t* my_bb, const uint64_t* op_bb, const float* pis, const float* vs, size_t n,
                 uint64_t* out_my, uint64_t* out_op, float* out_pi, float* out_v, uint32_t* out_num) {
    std::vector<DedupEntry> d = deduplicate(my_bb, op_bb, pis, vs, n);
    for (size_t i = 0; i < d.size(); i++) {
        out_my[i] = d[i].my_bb;
        out_op[i] = d[i].op_bb;
        for (int j = 0; j < 9; j++) out_pi[i * 9 + j] = d[i].pi[j];
        for (int j = 0; j < 3; j++) out_v[i * 3 + j] = d[i].v[j];
        out_num[i] = d[i].num;
    }
    return d.size();
}

}  // extern "C"

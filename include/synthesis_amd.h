/* synthesis_amd — MI355X-native batched self-play engine: C ABI (drop-in boundary).
 *
 * The reference (coreylowman/synthesis) has no FFI; its plug-in surface is Rust generics:
 *   trait Game<N>            synthesis/src/game.rs:68-88
 *   trait Policy<G,N>        synthesis/src/policies/traits.rs:4-6   fn eval(&mut self,&G)->([f32;N],[f32;3])
 *   trait NNPolicy<G,N>      synthesis/src/policies/traits.rs:8-11
 *   MCTS<G,P,N>              synthesis/src/mcts.rs:102-147 (private module, reached through alpha_zero/evaluator)
 *   run_n_games / run_game   synthesis/src/alpha_zero.rs:181-268
 * Every entry point below is the batched, plain-C form of one of those; the doc comment on each names the reference
 * interface it replaces. INTEGRATION.md shows the Rust `extern "C"` block + `impl Policy<Connect4, 9>` a maintainer
 * would add on the reference side.
 *
 * Conventions: all functions return SYN_OK (0) or a negative syn_status; syn_last_error() gives the message of the
 * last failure on that handle (or of the last failed syn_engine_create when handle == NULL). Nothing throws or unwinds
 * across this boundary. Host pointers unless a parameter is documented as a device pointer. The caller owns every
 * output buffer; the engine keeps no pointer past the call. One handle = one GPU + one stream; calls on a handle are
 * not re-entrant; different handles are independent (the reference's one-policy-per-thread rule, alpha_zero.rs:192-198).
 */
#ifndef SYNTHESIS_AMD_H
#define SYNTHESIS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum syn_status {
    SYN_OK = 0,
    SYN_ERR_INVALID_ARGUMENT = -1,
    SYN_ERR_NO_DEVICE = -2,       /* no usable gfx950 device / HIP runtime failure at create */
    SYN_ERR_HIP = -3,             /* a HIP call failed; see syn_last_error */
    SYN_ERR_NO_WEIGHTS = -4,      /* policy/value network weights not loaded yet */
    SYN_ERR_UNSUPPORTED = -5,     /* config variant not implemented on the device */
    SYN_ERR_CAPACITY = -6,        /* node pool too small for the requested explores */
    SYN_ERR_CANCELLED = -7        /* syn_cancel was called while the call ran: results of the games that finished are valid */
} syn_status;

/* ---- plain-data mirrors of synthesis/src/config.rs ---------------------------------------------------------- */

/* config.rs:9-13 Exploration */
enum { SYN_EXPLORATION_UCT = 0, SYN_EXPLORATION_POLYNOMIAL_UCT = 1 };
/* config.rs:15-19 ActionSelection */
enum { SYN_ACTION_Q = 0, SYN_ACTION_NUM_VISITS = 1 };
/* config.rs:21-26 Fpu. Func(fn() -> f32) is a plain function pointer (config.rs:25), called once per unexpanded child per
 * select_best_child scan (mcts.rs:351-356):
 *   SYN_FPU_FUNC {fpu_fn}: that pointer as `float (*)(void)`. A host function — it is called by the HOST trees of
 *     syn_mcts_search_lockstep / syn_selfplay_run_lockstep (from their worker threads, as the reference calls it from its
 *     workers: it must be thread-safe, like the reference's thread_rng closure); the device entry points (syn_mcts_search,
 *     syn_selfplay_run, ...) refuse it with SYN_ERR_UNSUPPORTED — a kernel cannot call into the host.
 *   SYN_FPU_NORMAL {fpu_value = mean, fpu_std = std}: the closure the reference itself configures —
 *     Normal::new(mean, std).sample(&mut thread_rng()) (study-connect4/src/main.rs:43-47) — as a function the device can
 *     compute too: the draw comes from a per-tree counter-based stream instead of thread_rng (reproducible; csrc/noise.cuh),
 *     identical on the device and on the host trees. */
enum { SYN_FPU_CONST = 0, SYN_FPU_PARENT_Q = 1, SYN_FPU_NORMAL = 2, SYN_FPU_FUNC = 3 };
/* config.rs:39-44 PolicyNoise. Dirichlet{alpha, weight} (mcts.rs:241-256) samples rand_distr's Dirichlet on the device from
 * the tree's own stream (csrc/noise.cuh). */
enum { SYN_NOISE_NONE = 0, SYN_NOISE_EQUAL = 1, SYN_NOISE_DIRICHLET = 2 };
/* config.rs:1-7 ValueTarget */
enum { SYN_VALUE_Z = 0, SYN_VALUE_Q = 1, SYN_VALUE_QZ_AVERAGE = 2, SYN_VALUE_Q_TO_Z = 3 };
/* Outcome kind, index order of mcts.rs:10-18 (Into<usize>) */
enum { SYN_OUTCOME_LOSE = 0, SYN_OUTCOME_DRAW = 1, SYN_OUTCOME_WIN = 2 };

/* config.rs:28-37 MCTSConfig */
typedef struct syn_mcts_config {
    int32_t exploration;              /* SYN_EXPLORATION_* */
    float c;                          /* Uct{c} / PolynomialUct{c} */
    int32_t solve;
    int32_t correct_values_on_solve;
    int32_t select_solved_nodes;
    int32_t auto_extend;
    int32_t fpu;                      /* SYN_FPU_* */
    float fpu_value;                  /* Fpu::Const(value); mean of SYN_FPU_NORMAL */
    int32_t root_policy_noise;        /* SYN_NOISE_* */
    float noise_alpha;                /* Dirichlet{alpha,..} */
    float noise_weight;               /* Equal{weight} / Dirichlet{..,weight} */
    float fpu_std;                    /* standard deviation of SYN_FPU_NORMAL (>= 0) */
    float (*fpu_fn)(void);            /* SYN_FPU_FUNC: Fpu::Func's fn() -> f32 (config.rs:25); NULL otherwise */
} syn_mcts_config;

/* config.rs:46-56 RolloutConfig (num_workers has no meaning here: concurrency is syn_engine_config.concurrent_games) */
typedef struct syn_rollout_config {
    int32_t num_explores;
    int32_t random_actions_until;
    int32_t sample_actions_until;
    int32_t stop_games_when_solved;
    int32_t value_target;             /* SYN_VALUE_* */
    float value_target_p;             /* QZaverage{p} */
    float value_target_from;          /* QtoZ{from,..} */
    float value_target_to;            /* QtoZ{..,to} */
    int32_t action;                   /* SYN_ACTION_* */
    syn_mcts_config mcts_cfg;
} syn_rollout_config;

typedef struct syn_engine_config {
    int32_t concurrent_games;   /* trees resident on the GPU at once (BASELINE: 4096); rounded up to a multiple of 16 */
    int32_t max_explores;       /* node-pool slab per tree = 1 + 9*(max_explores+1) nodes */
    int32_t policy_cache_log2;  /* PolicyWithCache (policies/cache.rs:19-32) on the device: 0 = off, else a table of
                                 * 2^policy_cache_log2 entries of 64 bytes (10..30) shared by all games of the engine;
                                 * semantics-neutral (the network is deterministic), used by the lane-per-tree kernel */
    int32_t reserved1;
} syn_engine_config;

typedef struct syn_engine syn_engine;

/* Fills the deterministic parity configuration: policy_mcts_cfg of study-connect4/src/main.rs:58-66 inside the
 * rollout_cfg of main.rs:28-36 (explores 800 per BASELINE.json; the reference default is 1600). */
void syn_default_rollout_config(syn_rollout_config* cfg);

/* ---- lifecycle ------------------------------------------------------------------------------------------------- */

/* Replaces: VarStore::new + P::new(&vs) per worker (alpha_zero.rs:192-193). Allocates the device node pool. */
int syn_engine_create(const syn_engine_config* cfg, int device, syn_engine** out);
int syn_engine_destroy(syn_engine* h);
const char* syn_last_error(const syn_engine* h);

/* Replaces: vs.load(models/model_i.ot) (alpha_zero.rs:194). blob = l_1.weight[128x63], l_1.bias[128],
 * l_2.weight[96x128], l_2.bias[96], l_3.weight[64x96], l_3.bias[64], l_4.weight[48x64], l_4.bias[48],
 * l_5.weight[12x48], l_5.bias[12] (VarStore names, study-connect4/src/policies.rs:20-24), row-major [out][in],
 * n_floats must be 30492. Empties the policy cache (its entries belong to the previous network). */
int syn_load_weights(syn_engine* h, const float* blob, size_t n_floats);
/* The conv policy/value network BASELINE.json's north_star words — slimnn Conv2d (slimnn/src/conv.rs:45-85) over the 2x7x9
 * bitplane state + Linear policy/value heads (slimnn/src/linear.rs:17-25) — behind the same Policy::eval surface
 * (policies/traits.rs:4-6): x[2][7][9] = (mine, theirs) -> Conv2d<2, 16, 3, pad 1, stride 1> + ReLU -> Linear<1008, 12>, logits =
 * out[0..9], value = softmax(out[9..12]). The reference ships the layers but no such network, so the architecture is this
 * library's (oracle/nn.hpp Connect4ConvNet restates it). blob = conv.weight[16][2][3][3], conv.bias[16], head.weight[12][1008],
 * head.bias[12]; n_floats must be 12412. Replaces the engine's network: syn_policy_eval_batch*, syn_mcts_search and
 * syn_selfplay_run then evaluate this network (lane-per-tree kernels: max_explores <= 7280, else SYN_ERR_UNSUPPORTED);
 * syn_load_weights / syn_trainer_publish_weights switch back to Connect4Net. Empties the policy cache. */
int syn_load_weights_conv(syn_engine* h, const float* blob, size_t n_floats);

/* The arithmetic Connect4Net is evaluated in. The reference's own is f32 (libtorch `y = x W^T + b`, study-connect4/src/policies.rs:
 * 28-44; slimnn/src/linear.rs:17-25 in its Rust-only form); both choices below compute that function and differ in rounding only.
 *   SYN_NET_ARITH_F32   (default) every product and sum in f32 on v_mfma_f32_16x16x4_f32, ascending input order, fused
 *                       multiply-add per term: bit for bit oracle/nn.hpp's ACC_FMA, within 1e-5 of slimnn's order.
 *   SYN_NET_ARITH_F16X2 every weight and activation as a pair of f16 numbers (hi + lo, 22-23 significand bits, exact power-of-two
 *                       scales chosen per checkpoint from a bound on the activations), products hi.hi + hi.lo + lo.hi on
 *                       v_mfma_f32_16x16x32_f16 with f32 accumulation: 5.3x fewer matrix-pipe cycles per evaluation. Bit for bit
 *                       oracle/nn_f16x2.hpp (which restates the instruction's accumulation, identified on MI355X); as close to
 *                       an f64 evaluation as the f32 arithmetic is (profiles/r05_f16_split.txt: 1.0e-7 against 1.0e-7 on the
 *                       random-init network, 1.8e-4 against 2.3e-4 on logits of magnitude 258 of a trained one).
 * The choice holds for syn_policy_eval_batch*, syn_eval_ctx_*, syn_mcts_search and syn_selfplay_run until changed; it empties the
 * policy cache. F16X2 needs Connect4Net (not Connect4ConvNet: SYN_ERR_UNSUPPORTED), max_explores <= 7280 (lane-per-tree kernels) and
 * finite parameters. Results under the two arithmetics differ in the last bits of the logits, so searches may differ where two
 * moves are within rounding of each other — which is why the choice is the caller's and never made silently. */
enum { SYN_NET_ARITH_F32 = 0, SYN_NET_ARITH_F16X2 = 1 };
int syn_set_network_arithmetic(syn_engine* h, int arithmetic);
/* The scales of the f16x2 plan of the engine's current Connect4Net (valid = 0 when there is none): layer l's inputs are multiplied
 * by 2^activation_exp[l], its weights by 2^weight_exp[l]; bound[l] = the bound on layer l's outputs the next exponent was chosen
 * from; raw outputs = accumulators * 2^out_exp. arithmetic / plan may be NULL. */
typedef struct syn_f16x2_plan {
    int valid;
    int activation_exp[5];
    int weight_exp[5];
    int out_exp;
    double bound[5];
} syn_f16x2_plan;
int syn_get_network_arithmetic(syn_engine* h, int* arithmetic, syn_f16x2_plan* plan);
/* The same plan for a parameter blob without an engine (pure host code, no GPU needed): what syn_set_network_arithmetic would choose for
 * these parameters. SYN_OK with plan->valid = 0 when the blob has no plan (non-finite parameters). */
int syn_f16x2_plan_of_blob(const float* blob, size_t n_floats, syn_f16x2_plan* plan);

/* ---- leaf evaluation ------------------------------------------------------------------------------------------- */

/* Replaces: Policy::eval (study-connect4/src/policies.rs:47-59) for n states at once. State i is the position with
 * bitboards (my_bb[i], op_bb[i]) in the layout of connect4.rs:3-13 (my_bb = side to move). Outputs: logits[n*9] raw
 * policy logits, value[n*3] = softmax over [lose, draw, win]. Up to 32,768 positions the call runs on the engine's own
 * evaluation context (below): 14 us for n <= 16, 27 us for 4,096; beyond, pageable transfers around the throughput kernel.
 * Side effects differ by size: a batch of up to 32,768 positions runs on that context's own non-blocking stream — it does NOT wait
 * for work queued on the engine stream (e.g. a preceding syn_policy_eval_batch_device(..., sync = 0)) and syn_last_timing reports 0
 * for it; larger batches run on the engine stream, drain it and are timed. Not re-entrant: the engine's context is one per engine
 * (two threads evaluating at once take one syn_eval_ctx each). */
int syn_policy_eval_batch(syn_engine* h, const uint64_t* my_bb, const uint64_t* op_bb, int n, float* logits,
                          float* value);
/* Same with all four pointers resident in device memory (no PCIe in the call); asynchronous on the engine stream
 * unless sync != 0. */
int syn_policy_eval_batch_device(syn_engine* h, const uint64_t* d_my_bb, const uint64_t* d_op_bb, int n,
                                 float* d_logits, float* d_value, int sync);

/* Replaces: one worker's policy object. gather_experience gives every worker thread its own policy (alpha_zero.rs:192-198:
 * VarStore, P::new, cache — "one policy per thread, &mut self eval"); an evaluation context is that object on the GPU: its own
 * stream, pinned staging and device scratch, reading the engine's weight image (loaded once, syn_load_weights*). Contexts of one
 * engine may be used at the same time from different host threads — one thread per context at a time; the engine handle's
 * own calls stay single-threaded as before. Loading other weights while a context has a batch in flight is the caller's race.
 *   submit: copies the n positions and launches their evaluation; returns without waiting (one batch in flight per context).
 *   wait:   blocks until the submitted batch is done and copies logits[n*9] / value[n*3] out (syn_policy_eval_batch's outputs,
 *           bit for bit).  eval = submit + wait.
 * A call is latency (n = 1 from a Rust `impl Policy`; the leaves of a few hundred trees from a self-play worker): the positions
 * are read and small results written across the host link in place, no transfer commands. Errors: the usual codes, the text in
 * syn_eval_ctx_last_error(ctx) (not in the engine's slot; a text starting with "note:" beside SYN_OK reports a completion word that
 * went missing — the answers were complete, the context has switched to waiting on its stream). Destroy every context before
 * syn_engine_destroy. */
typedef struct syn_eval_ctx syn_eval_ctx;
int syn_eval_ctx_create(syn_engine* h, syn_eval_ctx** out);
int syn_eval_ctx_submit(syn_eval_ctx* ctx, const uint64_t* my_bb, const uint64_t* op_bb, int n);
int syn_eval_ctx_wait(syn_eval_ctx* ctx, float* logits, float* value);
int syn_eval_ctx_eval(syn_eval_ctx* ctx, const uint64_t* my_bb, const uint64_t* op_bb, int n, float* logits, float* value);
const char* syn_eval_ctx_last_error(const syn_eval_ctx* ctx);
int syn_eval_ctx_destroy(syn_eval_ctx* ctx);

/* Replaces: Game::features (connect4.rs:235-258) for n states: out[n*63], index row*9 + col. */
int syn_features_batch(syn_engine* h, const uint64_t* my_bb, const uint64_t* op_bb, int n, float* out);

/* ---- slimnn layer semantics (slimnn/src/linear.rs:17-25, conv.rs:45-85, activations.rs) ------------------------- */
/* y[batch][O] = b + sum_i x[batch][i] * W[o][i], accumulated in ascending i with separate multiply and add. */
int syn_linear_forward(syn_engine* h, int I, int O, const float* W, const float* b, const float* x, int batch,
                       float* y, int relu);
/* slimnn activations as layers on x[batch][n] (slimnn/src/activations.rs:31-63): SYN_ACT_RELU = x.max(0.0); SYN_ACT_TANH =
 * x.tanh() (a deterministic tanh shared with the oracle: Rust's f32::tanh is libm, unpinned); SYN_ACT_SOFTMAX =
 * Softmax::apply_1d per row — exp of every element with NO max subtraction, summed in index order, divided by the total. */
enum { SYN_ACT_RELU = 0, SYN_ACT_TANH = 1, SYN_ACT_SOFTMAX = 2 };
int syn_activation_forward(syn_engine* h, int kind, const float* x, int batch, int n, float* y);
/* NCHW cross-correlation; W[COUT][CIN][K][K]; x[batch][CIN][H_IN][W_IN]; y[batch][COUT][H_OUT][W_OUT]; accumulation
 * order ci -> k1 -> k2. H_OUT/W_OUT must satisfy conv.rs:50-51 or SYN_ERR_INVALID_ARGUMENT is returned. */
int syn_conv2d_forward(syn_engine* h, int CIN, int COUT, int K, int ROW_PAD, int COL_PAD, int STRIDE, int H_IN,
                       int W_IN, int H_OUT, int W_OUT, const float* W, const float* b, const float* x, int batch,
                       float* y, int relu);

/* ---- search ---------------------------------------------------------------------------------------------------- */

/* Per-root result of a search; child_* arrays are indexed by ACTION (column); entries of columns that are not
 * children of the root are zero. */
typedef struct syn_search_result {
    float child_N[9];          /* Node::num_visits                     (mcts.rs:38) */
    float child_W[9][3];       /* Node::outcome_probs [lose,draw,win]  (mcts.rs:37) */
    float child_P[9];          /* Node::action_prob                    (mcts.rs:36) */
    int32_t child_sol[9][3];   /* {is_some, kind, turns} of Node::solution (mcts.rs:34) */
    float root_N;
    float root_W[3];
    int32_t root_sol[3];
    uint32_t num_nodes;        /* nodes.len() */
    int32_t best_action;       /* MCTS::best_action (mcts.rs:273-294) */
    float target_pi[9];        /* MCTS::target_policy (mcts.rs:174-211) */
    float target_q[3];         /* MCTS::target_q (mcts.rs:213-225) */
} syn_search_result;

/* Replaces: MCTS::with_capacity(explores+1, cfg, policy, root) + explore_n(explores) (mcts.rs:123-147) for n
 * independent roots, each on its own device-resident tree. */
int syn_mcts_search(syn_engine* h, const syn_mcts_config* cfg, const uint64_t* my_bb, const uint64_t* op_bb, int n,
                    int explores, int action_selection, syn_search_result* results);

/* Replaces: MCTS<G, RolloutPolicy> (the pairing of the reference's MCTS tests, mcts.rs:691-868): the same search with
 * RolloutPolicy (policies/rollout.rs:8-31) as the leaf evaluation — uniformly random playouts, zero logits, one-hot outcome. Root i
 * draws its playouts from its own StdRng::seed_from_u64(seed + i), in explore order. No network weights are needed. */
int syn_mcts_search_rollout(syn_engine* h, const syn_mcts_config* cfg, uint64_t seed, const uint64_t* my_bb,
                            const uint64_t* op_bb, int n, int explores, int action_selection, syn_search_result* results);

/* Replaces: the same reference call as syn_mcts_search — MCTS::with_capacity + explore_n for n roots (mcts.rs:123-147) — in the
 * reference's own division of labour (BASELINE.json configs[1] as worded: concurrent games, batched leaf inference): the trees
 * live on the HOST (include/synthesis_amd_lockstep.hpp: MCTS<G, P, N> restated over any Game, Policy::eval taken out of visit()),
 * and the leaves go to the GPU in batches: the roots are divided among host_threads workers (gather_experience's worker model,
 * alpha_zero.rs:132-154); a worker runs its trees in two halves that take turns — one half's leaves are being evaluated while it
 * advances the other — and the batches the workers hand in are combined into one launch on one evaluation context
 * (syn_eval_ctx_*). This is the driver a caller with a different Game impl instantiates; for Connect4 the fused syn_mcts_search
 * is the fast path and this entry point exists to hold the driver to it: results are identical, field for field. host_threads:
 * 0 = what the process may use (hardware concurrency cut to a cgroup CPU quota), at most 32. cfg: every configuration syn_mcts_search takes and SYN_FPU_FUNC (a host function pointer: this is where it can be called) — SYN_FPU_NORMAL and SYN_NOISE_DIRICHLET
 * draw what syn_mcts_search draws (root i: tree stream (i, turn 0)). stats may be NULL. */
typedef struct syn_lockstep_stats {
    uint64_t rounds;               /* evaluation launches (combined batches) */
    uint64_t positions_evaluated;  /* leaves over all launches */
    double seconds_total;
    double seconds_policy;         /* host time inside the evaluation context's calls (staging, launch, waiting for the GPU) */
} syn_lockstep_stats;
int syn_mcts_search_lockstep(syn_engine* h, const syn_mcts_config* cfg, const uint64_t* my_bb, const uint64_t* op_bb, int n,
                             int explores, int action_selection, int host_threads, syn_search_result* results,
                             syn_lockstep_stats* stats);
/* Replaces: run_n_games (alpha_zero.rs:181-209) in the same division of labour — BASELINE.json configs[1] as worded: concurrent
 * games (syn_selfplay_run's shape: n_games jobs over the engine's concurrent_games slots, a finished game's slot takes the next
 * game), every game's MCTS on the HOST (one tree per move, alpha_zero.rs:240-244), the leaves batched to the GPU as above,
 * run_game / sample_action / fill_state_info / store_rewards (alpha_zero.rs:229-338) on the host with game g's own
 * StdRng::seed_from_u64(base_seed + g) (include/synthesis_amd_lockstep.hpp::lockstep_selfplay_sharded).
 * Arguments and outputs are syn_selfplay_run's; the games are identical to that call's, move for move and float for float.
 * host_threads as above; SYN_FPU_FUNC, SYN_FPU_NORMAL (the reference's own self-play configuration) and SYN_NOISE_DIRICHLET included — the
 * host trees take the draws of syn_selfplay_run's trees; stats may be NULL. */
int syn_selfplay_run_lockstep(syn_engine* h, const syn_rollout_config* cfg, uint64_t base_seed, uint64_t first_game, int n_games,
                              int host_threads, int32_t* plies, uint64_t* states_bb, float* pis, float* vs, uint8_t* actions,
                              uint32_t* root_nodes, uint8_t* final_kind, syn_lockstep_stats* stats);

/* ---- evaluator baseline ---------------------------------------------------------------------------------------- */

/* One root of the evaluator's baseline tree after the search (evaluator.rs:233-243 Node fields of the root's children,
 * indexed by action; children of illegal actions are all-zero). */
typedef struct syn_frozen_result {
    float child_N[9];          /* Node::num_visits  */
    float child_cum[9];        /* Node::cum_value   */
    float child_P[9];          /* Node::action_prob */
    int32_t child_sol[9][3];   /* {is_some, kind (0 Lose, 1 Draw, 2 Win), turns} of Node::solution */
    float root_N;
    float root_cum;
    int32_t root_sol[3];
    uint32_t num_nodes;        /* nodes.len() */
    int32_t best_action;       /* FrozenMCTS::best_action (evaluator.rs:364-389) */
} syn_frozen_result;

/* Replaces: FrozenMCTS::exploit(explores, cfg, &mut RolloutPolicy { rng }, game, action_selection) (evaluator.rs:308-319
 * with policies/rollout.rs:8-31) — the "VanillaMCTS<explores>" baseline of eval_against_rollout_mcts / mcts_vs_mcts
 * (evaluator.rs:163-228) — for n independent roots, one device-resident tree each. Root i draws its playouts from
 * StdRng::seed_from_u64(seeds[i]) starting at 32-bit output word rng_words[i]; on return rng_words[i] is the first word the
 * search did not use, so a caller that replays a match move by move keeps ONE generator per match as the reference does
 * (pass 0 for the first move, hand the value back for the next). explores[i] is per root (mcts_vs_mcts gives each side its own).
 * The trees live in the engine's node pool, re-partitioned per call for the largest explores[i] of the batch (1 + 9 records of
 * 16 bytes per visit), so searches far deeper than the engine's max_explores run, fewer at a time (up to 233,014 explores:
 * the reference's largest baseline is VanillaMCTS204800); SYN_ERR_CAPACITY if not even one such tree fits.
 * cfg: Exploration::Uct and Fpu::Const only — the reference panics on anything else (SYN_ERR_UNSUPPORTED here); the
 * baseline ignores every other field except `solve`. No network weights are needed. */
int syn_frozen_search_rollout(syn_engine* h, const syn_mcts_config* cfg, const uint64_t* seeds, uint64_t* rng_words,
                              const uint64_t* my_bb, const uint64_t* op_bb, const int32_t* explores, int n,
                              int action_selection, syn_frozen_result* results);

/* ---- self-play ------------------------------------------------------------------------------------------------- */

/* Event counters summed over all games of a run (definitions: SURVEY.md §8d; used for algorithmic-bytes accounting) */
typedef struct syn_counters {
    uint64_t explores;          /* MCTS::explore calls + root visits */
    uint64_t select_levels;     /* select_best_child calls */
    uint64_t children_scanned;  /* children examined by select_best_child */
    uint64_t expansions;        /* visit() bodies that created children */
    uint64_t new_nodes;         /* nodes created */
    uint64_t policy_evals;      /* Policy::eval calls = leaf evaluations */
    uint64_t backprop_levels;   /* nodes updated by backprop */
    uint64_t solver_children;   /* child solutions read by the solver branch of backprop */
    uint64_t solved_hits;       /* explores that ended on an already solved node */
    uint64_t moves;             /* plies played */
    uint64_t games;             /* games finished */
    uint64_t max_depth;         /* longest root-to-leaf chain (levels) any backprop walked */
} syn_counters;

/* Replaces: run_n_games (alpha_zero.rs:181-209) for games [first_game, first_game + n_games). Game g uses its own
 * StdRng::seed_from_u64(base_seed + g). Output slot j = g - first_game. Any output pointer may be
 * NULL.  plies[n]; states_bb[n][63][2] = (my_bb, op_bb) of every recorded position (ReplayBuffer.games, data.rs:151-158);
 * pis[n][63][9]; vs[n][63][3] (after store_rewards, alpha_zero.rs:309-338); actions[n][63]; root_nodes[n][63] =
 * nodes.len() of each move's tree; final_kind[n] = outcome kind for the side to move in the final position.
 * counters may be NULL. */
int syn_selfplay_run(syn_engine* h, const syn_rollout_config* cfg, uint64_t base_seed, uint64_t first_game,
                     int n_games, int32_t* plies, uint64_t* states_bb, float* pis, float* vs, uint8_t* actions,
                     uint32_t* root_nodes, uint8_t* final_kind, syn_counters* counters);

/* ---- learner step and replay de-duplication (the step right after the self-play path) ---------- */

/* LearningConfig's optimiser fields (config.rs:76-94) + Adam::default() (alpha_zero.rs:33-36) */
typedef struct syn_train_config {
    float weight_decay;    /* LearningConfig::weight_decay (study-connect4/src/main.rs:20: 1e-6) */
    float policy_weight;   /* LearningConfig::policy_weight */
    float value_weight;    /* LearningConfig::value_weight */
    float beta1;           /* 0.9 */
    float beta2;           /* 0.999 */
    float eps;             /* 1e-8 */
} syn_train_config;

/* Replaces: P::new(&vs) + Adam::default().build(&vs, lr) (alpha_zero.rs:31-36): parameters (same blob order as
 * syn_load_weights) and zeroed Adam moments on the device. The first call on an engine also runs a ~1 ms self-check of the epoch
 * kernel's step barrier (eight steps on a synthetic batch through the fast and the device-scope barrier, compared bit for bit; a
 * mismatch makes this engine use the device-scope barrier) and thereby discards a data set uploaded with syn_train_set_data
 * before it: upload after initialising. */
int syn_trainer_init(syn_engine* h, const float* blob, size_t n_floats, const syn_train_config* cfg);
/* The same for Connect4ConvNet (syn_load_weights_conv's network and blob order, 12412 floats): the trainer then runs that
 * network through syn_train_step / syn_train_gradients_device + syn_train_apply_device / syn_train_set_data + syn_train_epoch
 * (minibatches of at most 32 positions, else SYN_ERR_UNSUPPORTED), syn_trainer_get_state copies 12412 floats per array, and
 * syn_trainer_publish_weights makes the trained conv network the engine's policy. f32, the arithmetic order of
 * oracle/train.hpp::ConvTrainer (checked against torch float64 goldens); an engine trains one network at a time. Like
 * syn_trainer_init, the first call on an engine runs a self-check (eight steps on a synthetic batch through the four-workgroup
 * epoch kernel and through the one-workgroup kernel, compared bit for bit) and thereby discards a data set uploaded before it. */
int syn_trainer_init_conv(syn_engine* h, const float* blob, size_t n_floats, const syn_train_config* cfg);
/* Replaces: one iteration of the minibatch loop alpha_zero.rs:76-92 (forward, log_softmax, kl_div(Sum)/batch for both
 * heads, loss, backward_step). States are given as bitboards; losses[2] = {pi_loss, v_loss} (may be NULL). */
int syn_train_step(syn_engine* h, const uint64_t* my_bb, const uint64_t* op_bb, const float* target_pi,
                   const float* target_v, int batch, float lr, float* losses);
/* Data-parallel form (BASELINE configs[4]): gradients of this rank's minibatch into a caller-owned DEVICE buffer of
 * 30,492 floats (all other pointers are device pointers too), to be all-reduced by the caller (RCCL), then applied with
 * syn_train_apply_device(grad_scale = 1/ranks). losses is a host pointer (may be NULL). */
int syn_train_gradients_device(syn_engine* h, const uint64_t* d_my_bb, const uint64_t* d_op_bb, const float* d_target_pi,
                               const float* d_target_v, int batch, float* d_grads, float* losses);
int syn_train_apply_device(syn_engine* h, const float* d_grads, float lr, float grad_scale);
/* The stream-ordered form of the same two calls (one optimiser step of alpha_zero.rs:76-92 split around the RCCL all-reduce, no host
 * in between): both enqueue on `stream` (a hipStream_t of the engine's device: the stream the batch was prepared on and the all-reduce
 * runs on; NULL is the device's null stream — PyTorch's default stream — not the engine's own) and return without synchronising. d_losses: DEVICE pointer to 2 floats {pi-loss sum, v-loss sum}
 * of this rank's minibatch (may be NULL) — typically the two words behind the 30,492 gradients, so they ride in the same message.
 * While a caller drives the trainer this way it keeps the engine's other trainer entry points (which use the engine's stream) idle. */
int syn_train_gradients_enqueue(syn_engine* h, void* stream, const uint64_t* d_my_bb, const uint64_t* d_op_bb, const float* d_target_pi,
                                const float* d_target_v, int batch, float* d_grads, float* d_losses);
int syn_train_apply_enqueue(syn_engine* h, void* stream, const float* d_grads, float lr, float grad_scale);
/* Arithmetic of the Connect4ConvNet learner's gradient step (forward, dZ, backward; BASELINE configs[4] words the on-node training
 * step "bf16 conv"): SYN_TRAIN_F32 (default after every syn_trainer_init*) = f32 matrix cores, bit-identical to the oracle;
 * SYN_TRAIN_BF16 = every matrix operand rounded to bf16 and multiplied on the bf16 matrix cores with f32 accumulation — master
 * weights, Adam moments, the softmax / KL head and Adam itself stay f32. Training only: inference never runs in bf16 (it cannot
 * hold north_star's 1e-5). SYN_ERR_UNSUPPORTED for the Connect4Net learner. No counterpart in the reference (its learner is
 * libtorch f32, alpha_zero.rs:28-37). */
enum { SYN_TRAIN_F32 = 0, SYN_TRAIN_BF16 = 1 };
int syn_trainer_set_precision(syn_engine* h, int precision);
/* Copies out parameters / Adam moments / last gradient (each may be NULL) and the optimiser step count. */
int syn_trainer_get_state(syn_engine* h, float* blob, float* m, float* v, long long* step, float* grads);
/* Replaces: vs.save(model_{i+1}.ot) + the workers' vs.load (alpha_zero.rs:97,194): the trained parameters become the
 * engine's policy for syn_policy_eval_batch / syn_mcts_search / syn_selfplay_run. */
int syn_trainer_publish_weights(syn_engine* h);
/* Epochs without the host in the loop. syn_train_set_data uploads the de-duplicated buffer once per iteration (the
 * tensors `states / target_pis / target_vs` of alpha_zero.rs:52-58, positions as bitboards); syn_train_epoch then runs
 * n_steps optimiser steps in one call: step s trains on the states perm[s*batch .. (s+1)*batch) — the BatchRandSampler's
 * index_select (data.rs:41-62; the caller draws the permutation and applies drop_last) — and step_losses[s][0..1]
 * receives its (pi_loss, v_loss). Bit-identical to n_steps calls of syn_train_step on the gathered batches. For batch <= 32
 * (the reference's batch_size) the whole epoch is ONE persistent kernel launch (csrc/train_epoch.cuh: 16 workgroups, the
 * gradient of the LAST step is what syn_trainer_get_state reports). That kernel needs its 16 workgroups resident together; the
 * learner's state is snapshotted before the launch, and if the workgroups never become co-resident (another kernel holds the CUs:
 * it gives up after ~10 s) the snapshot is restored and the epoch runs through the queued per-step launches instead — same bits,
 * the call still returns SYN_OK. Larger batches queue two launches per step. The Connect4ConvNet learner (syn_trainer_init_conv;
 * minibatches of at most 32) runs an epoch as ONE persistent launch as well (csrc/train_conv_mfma.cuh), in f32 or — after
 * syn_trainer_set_precision(SYN_TRAIN_BF16) — on the bf16 matrix cores: four workgroups of one XCD share every step (same
 * chains, same bits as one workgroup), with the same snapshot; if they never become co-resident, or if the start-up self-check of
 * syn_trainer_init_conv found this engine's four-workgroup kernel disagreeing with the one-workgroup kernel, the epoch is one
 * launch of a single persistent workgroup instead (no co-residency requirement, 2.3x slower). */
int syn_train_set_data(syn_engine* h, const uint64_t* my_bb, const uint64_t* op_bb, const float* target_pi,
                       const float* target_v, size_t n);
int syn_train_epoch(syn_engine* h, const int32_t* perm, size_t n_steps, int batch, float lr, float* step_losses);

/* Replaces: ReplayBuffer::deduplicate (data.rs:196-235): identical states are merged, their targets summed in buffer
 * order and divided by the count. Outputs are sized for n entries; *out_count = number of unique states, emitted in
 * ascending (my_bb, op_bb) order (the reference's order is HashMap iteration order, i.e. unspecified). */
int syn_replay_deduplicate(syn_engine* h, const uint64_t* my_bb, const uint64_t* op_bb, const float* pis,
                           const float* vs, size_t n, uint64_t* out_my, uint64_t* out_op, float* out_pi, float* out_v,
                           uint32_t* out_num, size_t* out_count);

/* Timing of the last syn_selfplay_run / syn_mcts_search / *_device call on this handle, measured with HIP events on
 * the engine stream: kernel_ms = device time of the dominant kernel launch(es), n_launches = how many. */
int syn_last_timing(const syn_engine* h, float* kernel_ms, int* n_launches);

/* PolicyWithCache statistics of the last syn_selfplay_run / syn_mcts_search: Policy::eval calls answered from the table and
 * calls that ran the network (both 0 when the cache is off or the row-per-tree kernels ran). */
int syn_last_cache_stats(const syn_engine* h, uint64_t* hits, uint64_t* misses);

/* Launch shape the last syn_selfplay_run / syn_mcts_search used (the engine picks it from the number of concurrent games,
 * DESIGN.md §6.1): *shape = 1 row-per-tree kernel with the weights in registers (16 trees per workgroup), 2 = the same
 * with two workgroups per CU, 3 = quad-async row kernel (several 16-tree quads per workgroup), 4 = lane-per-tree kernel
 * (one tree per lane), 5 = the evaluator baseline's lane-per-tree kernel / the producer-consumer debug shape, 6 = the lane-per-tree kernel
 * with two trees per lane, 7 = the free-running row kernel of the f16x2 arithmetic (at most 16 trees per CU: four waves of four trees, every
 * wave evaluating its own leaves); grid / threads = workgroups and threads per workgroup. Diagnostics and tests only. */
int syn_last_launch_shape(const syn_engine* h, int* shape, int* grid, int* threads);

/* A self-play launch plays its whole batch inside ONE kernel (seconds to tens of seconds). These two entry points may be called
 * from ANOTHER host thread while syn_selfplay_run / syn_mcts_search / syn_mcts_search_rollout runs on the handle (they use a stream
 * of their own and never touch the handle's error string):
 * syn_progress: *started = jobs (games / roots) handed to tree slots so far, capped at the call's job count; after a cancel, the
 *   number handed out when the cancel was applied (a lower bound of what finally ran). *finished = self-play games completed.
 * syn_cancel: no further job is handed out; the running call returns once the jobs already started have finished (at most one
 *   game's duration) — with SYN_ERR_CANCELLED if anything was left out: self-play games that never started have plies[g] = 0,
 *   roots that were not searched have all-zero results (num_nodes == 0); everything else is valid. A cancel that arrives after
 *   the last job was handed out cancels nothing and the call returns SYN_OK. A cancel that arrives while the call is still
 *   setting up is applied before the first job is handed out. Returns SYN_ERR_INVALID_ARGUMENT when no call is in flight.
 *   syn_frozen_search_rollout takes its roots in a grid-stride loop: a cancel during it is accepted and has no effect.
 * The reference has no counterpart (its workers are joined at the end of gather_experience, alpha_zero.rs:156-170). */
int syn_progress(syn_engine* h, int* started, int* finished);
int syn_cancel(syn_engine* h);

/* Device-side RNG / math primitives exposed for parity tests against the oracle (no reference counterpart):
 * out[i] = i-th u32 of StdRng::seed_from_u64(seed) as generated on the GPU. */
int syn_debug_stdrng_u32(syn_engine* h, uint64_t seed, int n, uint32_t* out);
/* y[i] = device det_expf(a[i]); q[i] = a[i] / b[i]; s[i] = sqrtf(a[i]), or det_logf(a[i]) where b[i] < 0
 * (IEEE-exactness checks) */
int syn_debug_math(syn_engine* h, const float* a, const float* b, int n, float* out_exp_a, float* out_div,
                   float* out_sqrt_a);

/* fast[i] = the packed division of the lane / producer-consumer kernels' descent (device_common.cuh div2_by_small_int) on
 * a[i] / b[i]; full[i] = the device's IEEE division. Equal bit for bit for a = 0 or 2^-60 <= a <= 2^60 and INTEGER 1 <= b <= 2^16. */
int syn_debug_fast_div(syn_engine* h, const float* a, const float* b, int n, float* out_fast, float* out_full);
/* Enumeration behind the descent's shortened sequences: every f32 significand (2^23) at three exponents against every integer
 * divisor b_lo .. b_hi, and every significand at four exponents under the square root. mismatches3[0] = quotients where
 * div2_by_small_int differs from the IEEE division, [1] = the same for div2_safe_range, [2] = square roots where sqrt_normal_range
 * differs from sqrtf. All three must be 0. */
int syn_debug_small_int_math(syn_engine* h, int b_lo, int b_hi, unsigned long long* mismatches3);

/* PMC calibration probe (tools/calibrate_pmc.py): gathers n_spans 288-byte sibling spans of 32-byte node records at
 * record offsets d_span_off[i] from d_base (device pointers), optionally rewriting 16 bytes per touched record. */
int syn_debug_calibrate(syn_engine* h, const void* d_base, const uint32_t* d_span_off, int n_spans, int do_write);

#ifdef __cplusplus
}
#endif
#endif /* SYNTHESIS_AMD_H */

// synthesis_amd — device-resident MCTS over a flat structure-of-arrays node pool.
//
// Replaces, with identical results (same f32 operations in the same order), the reference's
//   synthesis/src/mcts.rs:310-325   explore            -> tree_select_expand + tree_finish
//   synthesis/src/mcts.rs:327-372   select_best_child / exploit_value / explore_value
//   synthesis/src/mcts.rs:374-427   visit (expansion, auto-extend, legal-move softmax)
//   synthesis/src/mcts.rs:429-488   backprop (MCTS-Solver + value correction)
//   synthesis/src/mcts.rs:174-225,273-306  target_policy / target_q / best_action / solution
//   synthesis/src/alpha_zero.rs:229-338    run_game / sample_action / fill_state_info / store_rewards
//
// MI355X mapping
//   * One tree = one DPP row (16 lanes, 4 trees per wave64); lane c owns column c / child c (<= 9 active).
//     select = one 2x16-byte gather per level + a 4-step DPP arg-max (first index wins ties); no LDS, no atomics.
//     A tree has exactly one leaf in flight (the reference is sequential per tree, mcts.rs:139-147), so its f32
//     sums are order-exact; throughput comes from thousands of independent trees, never from intra-tree races.
//   * Node pool (HBM): one 32-byte record per node = two 16-byte halves, slab of `cap` nodes per tree slot:
//       stat half = {N, W_lose, W_draw, W_win}            read by select, read-modify-written by backprop
//       edge half = {first_child, meta, P, parent}        written at expansion; meta rewritten when solved
//     meta = num_children[0:3] | action[4:7] | sol_some[8] | sol_kind[9:10] | sol_turns[16:31].
//     Children of a node are contiguous (ids first_child..first_child+n), so a row's 9 lanes read one 288-byte
//     span (two 16-byte loads per lane). The reference also keeps the Game in every node (mcts.rs:33); here positions are re-derived
//     in registers while descending (my' = op, op' = my | bit), so no board array is read or written on the hot path.
//   * The descent path is kept in registers (lane L&15 of register L>>4 holds level L), so backprop touches all
//     its nodes with ONE parallel gather/scatter; only the (rare) solver branch walks level by level.
#pragma once
#include "device_common.cuh"

namespace syn {

// ---------------------------------------------------------------------------------------------- configuration (POD)
struct DevMctsCfg {
    int exploration;   // 0 Uct, 1 PolynomialUct
    float c;
    int solve, correct_values, select_solved, auto_extend;
    int fpu;           // 0 Const, 1 ParentQ, 2 Func = Normal(fpu_value, fpu_std) (noise.cuh; lane-per-tree kernels only)
    float fpu_value;
    int noise;         // 0 None, 1 Equal, 2 Dirichlet(noise_alpha) (noise.cuh; lane-per-tree kernels only)
    float noise_weight;
    float fpu_std, noise_alpha;
    int fast_div;      // lane / producer-consumer kernels: c is inside the range the packed division is exact on (host-checked)
};
struct DevRolloutCfg {
    int num_explores, random_until, sample_until, stop_when_solved;
    int value_target;
    float vt_p, vt_from, vt_to;
    int action;  // 0 Q, 1 NumVisits
};

// Compile-time view of the MCTS config. FAST = the reference's own configuration family (PolynomialUct, Fpu::Const,
// solver + value correction + select_solved_nodes + auto_extend, study-connect4/src/main.rs:37-66): every switch folds
// away and only `c` and `fpu_value` stay runtime. The generic view serves every other combination with the same code.
// FAST (an int since round 4): 0 = every switch at run time; 1 = the parity configuration family above; 2 = the reference's OWN
// self-play configuration (study-connect4/src/main.rs:37-49): the same switches with Fpu::Func(|| Normal(mean, std)) instead of
// the constant — its draws (noise.cuh) exist in the lane-per-tree kernels only.
template <int FAST>
struct CfgView {
    const DevMctsCfg& c;
    SYN_DEV bool puct() const { return FAST ? true : c.exploration == 1; }
    SYN_DEV bool fpu_const() const { return FAST == 1 ? true : (FAST == 2 ? false : c.fpu == 0); }
    SYN_DEV bool fpu_normal() const { return FAST == 2 ? true : (FAST == 1 ? false : c.fpu == 2); }
    SYN_DEV bool select_solved() const { return FAST ? true : c.select_solved != 0; }
    SYN_DEV bool solve() const { return FAST ? true : c.solve != 0; }
    SYN_DEV bool correct_values() const { return FAST ? true : c.correct_values != 0; }
    SYN_DEV bool auto_extend() const { return FAST ? true : c.auto_extend != 0; }
    SYN_DEV float cc() const { return c.c; }
    SYN_DEV float fpu_value() const { return c.fpu_value; }
};
inline bool cfg_is_fast(const DevMctsCfg& c) {
    return c.exploration == 1 && c.fpu == 0 && c.select_solved && c.solve && c.correct_values && c.auto_extend && c.noise == 0;
}
// the compile-time family of a configuration (CfgView): 1 parity, 2 the reference's self-play configuration, 0 anything else
inline int cfg_family(const DevMctsCfg& c) {
    const bool folded = c.exploration == 1 && c.select_solved && c.solve && c.correct_values && c.auto_extend && c.noise == 0;
    return folded ? (c.fpu == 0 ? 1 : (c.fpu == 2 ? 2 : 0)) : 0;
}

struct DevCounters {  // index order = syn_counters
    unsigned long long v[12];
};
enum { CTR_EXPLORES, CTR_SELECT_LEVELS, CTR_CHILDREN_SCANNED, CTR_EXPANSIONS, CTR_NEW_NODES, CTR_POLICY_EVALS,
       CTR_BACKPROP_LEVELS, CTR_SOLVER_CHILDREN, CTR_SOLVED_HITS, CTR_MOVES, CTR_GAMES, CTR_MAX_DEPTH, CTR_COUNT };

// ---------------------------------------------------------------------------------------------- meta / outcome
constexpr uint32_t META_NC_MASK = 0xFu;
SYN_DEV uint32_t meta_make(uint32_t nc, uint32_t action, bool some, uint32_t kind, uint32_t turns) {
    return nc | (action << 4) | ((some ? 1u : 0u) << 8) | (kind << 9) | (turns << 16);
}
SYN_DEV uint32_t meta_nc(uint32_t m) { return m & META_NC_MASK; }
SYN_DEV uint32_t meta_action(uint32_t m) { return (m >> 4) & 0xFu; }
SYN_DEV bool meta_some(uint32_t m) { return (m >> 8) & 1u; }
SYN_DEV uint32_t meta_kind(uint32_t m) { return (m >> 9) & 3u; }
SYN_DEV uint32_t meta_turns(uint32_t m) { return m >> 16; }

// Option<Outcome> -> integer key whose natural order is the reference's Ord (game.rs:46-60, None lowest):
//   None = 0 < Lose(t) (more turns greater) < Draw(t) (more turns greater) < Win(t) (FEWER turns greater)
SYN_DEV uint32_t outcome_key(bool some, uint32_t kind, uint32_t turns) {
    if (!some) return 0u;
    uint32_t rank = kind + 1u;  // Lose 1, Draw 2, Win 3
    uint32_t t = kind == 2u ? (0xFFFFu - turns) : turns;
    return (rank << 16) | (t & 0xFFFFu);
}
SYN_DEV void outcome_from_key(uint32_t key, bool& some, uint32_t& kind, uint32_t& turns) {
    some = key != 0u;
    kind = some ? (key >> 16) - 1u : 0u;
    uint32_t t = key & 0xFFFFu;
    turns = kind == 2u ? (0xFFFFu - t) : t;
}
// key of solution.map(reversed) (game.rs:29-35): Win<->Lose, Draw stays, turns + 1
SYN_DEV uint32_t outcome_key_reversed(uint32_t meta) {
    if (!meta_some(meta)) return 0u;
    uint32_t k = meta_kind(meta);
    uint32_t rk = k == 1u ? 1u : 2u - k;
    return outcome_key(true, rk, meta_turns(meta) + 1u);
}

// ---------------------------------------------------------------------------------------------- per-tree registers
// A node is one 32-byte record {stat (16 B), edge (16 B)}; stat[i] / edge[i] are views on record i. Interleaving the
// two halves keeps a sibling scan (<= 9 consecutive nodes) inside one contiguous 288-byte span instead of two 144-byte
// spans in different arrays: fewer cache lines and DRAM pages per level for the same two 16-byte loads per lane.
// The first `k` records of a tree (the nodes created first: the root, its children, the grandchildren — the top of the tree, read by
// every descent and written by every backprop) may live in a second place, `hot`: the latency-bound 16-trees-per-CU kernel keeps them
// in LDS (engine_kernels.cuh selfplay_kernel<WPS = 1>), where a level of the descent costs an LDS access instead of a memory round
// trip. k = 0 (every other kernel): the select folds away and the accesses stay plain global loads / stores. A tree never outlives
// its search, so the two places need no write-back.
struct StatView {
    float4* base;
    float4* hot;
    uint32_t k;
    SYN_DEV float4& operator[](uint32_t i) const { return (i < k ? hot : base)[2u * i]; }
};
struct EdgeView {
    uint4* base;
    uint4* hot;
    uint32_t k;
    SYN_DEV uint4& operator[](uint32_t i) const { return (i < k ? hot : base)[2u * i + 1u]; }
};

struct TreeCtx {
    // node pool slab of this tree (already offset by slot * cap)
    StatView stat;
    EdgeView edge;
    // root position of the current search
    uint64_t root_my, root_op;
    uint32_t next_node;   // nodes.len()
    uint32_t root_fc;     // first child / child count of the root, mirrored in registers so a descent never has
    uint32_t root_nc;     // to re-read the root record (its N is `iter`, its solution ends the search)
    int iter;             // explores done on this tree (root visit counts as 1) == root.num_visits
    bool root_solved;
};

struct ExploreCtx {  // what phase A hands to phase C
    uint32_t path0, path1, path2, path3;  // level L lives on lane L&15 of path<L>>4> (scalars: never indexed)
    int depth;            // level of the leaf (root = 0)
    uint32_t leaf;        // node the backprop starts from
    uint32_t fc;          // first child of the node that needs its priors (valid if needs_eval)
    uint32_t legal_mask;  // legal columns of that node
    bool needs_eval;
    bool solved;
    float p0, p1, p2;     // outcome distribution when no eval is needed
    uint64_t leaf_my, leaf_op;
};

SYN_DEV void path_set(ExploreCtx& X, int gl, int level, uint32_t node) {
    bool mine = gl == (level & 15);
    int m = level >> 4;
    X.path0 = (mine && m == 0) ? node : X.path0;
    X.path1 = (mine && m == 1) ? node : X.path1;
    X.path2 = (mine && m == 2) ? node : X.path2;
    X.path3 = (mine && m == 3) ? node : X.path3;
}
// (values, not a reference: a select between loads of struct fields gets rewritten into a load from a selected
//  address, which pins the whole struct in scratch)
SYN_DEV uint32_t path_get(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, int level) {
    int m = level >> 4;
    uint32_t v = p0;
    v = m == 1 ? p1 : v;
    v = m == 2 ? p2 : v;
    v = m == 3 ? p3 : v;
    return row_bcast_u32(v, level & 15);
}

// ---------------------------------------------------------------------------------------------- phase A
// explore() up to the point where the policy is needed: descend by PUCT, expand, auto-extend.  (mcts.rs:310-427)
template <bool COUNT, bool FAST>
SYN_DEV void tree_select_expand(const DevMctsCfg& cfg_, TreeCtx& T, ExploreCtx& X, int gl, uint32_t* ctr) {
    const CfgView<FAST> cfg{cfg_};
    uint32_t node = 0;
    int depth = 0;
    X.path0 = X.path1 = X.path2 = X.path3 = 0;
    uint64_t my = T.root_my, op = T.root_op;
    X.needs_eval = false;
    X.solved = false;
    X.p0 = X.p1 = X.p2 = 0.0f;
    X.fc = 0;
    X.legal_mask = 0;
    if (COUNT) ctr[CTR_EXPLORES]++;

    uint32_t fc, meta;
    float pN, pW0, pW2;
    if (T.next_node == 0) {
        // MCTS::with_capacity: push the root (mcts.rs:125) — unvisited, parent 0, action 0, prior 0
        if (gl == 0) {
            T.stat[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            T.edge[0] = make_uint4(0u, meta_make(0, 0, false, 0, 0), f32_bits(0.0f), 0u);
        }
        T.next_node = 1;
        fc = 0;
        meta = 0;
        pN = pW0 = pW2 = 0.0f;
    } else {
        // the root was expanded by this tree's first pass; it is never entered once solved (explore_n stops first)
        fc = T.root_fc;
        meta = T.root_nc;
        pN = (float)T.iter;
        pW0 = pW2 = 0.0f;
        if (!cfg.fpu_const()) {  // Fpu::ParentQ needs the root's W as well
            float4 s = T.stat[0];
            pW0 = s.y;
            pW2 = s.w;
        }
    }

    // ---- descent (mcts.rs:310-341). The loop body is ONLY the child scan: the four trees of a wave run it in
    // lockstep and a tree that has reached its leaf just idles (exec-masked) until the deepest one is done, so the
    // long expansion code below is executed once per wave instead of once per tree.
    bool hit_solved = false;
    for (;;) {
        if (meta_some(meta)) { hit_solved = true; break; }
        uint32_t nc = meta_nc(meta);
        if (nc == 0) break;

        // select_best_child: lane i scores child i (branch-free: every lane evaluates all three exploit forms)
        bool active = (uint32_t)gl < nc;
        uint32_t cid = fc + (active ? (uint32_t)gl : 0u);
        float4 cs = T.stat[cid];
        uint4 ce = T.edge[cid];
        // exploit_value (mcts.rs:343-359)
        float q_visited = -((cs.w - cs.y) / cs.x);
        float q_fpu = cfg.fpu_const() ? cfg.fpu_value() : (pW2 - pW0) / pN;
        uint32_t k = meta_kind(ce.y);
        // outcome.reversed().value(): child Win -> -1, Draw -> 0, Lose -> +1 (game.rs:29-43)
        float q_solved = cfg.select_solved() ? (k == 2u ? -1.0f : (k == 1u ? 0.0f : 1.0f)) : -__builtin_inff();
        float q = meta_nc(ce.y) == 0u ? q_fpu : q_visited;
        q = meta_some(ce.y) ? q_solved : q;
        // explore_value (mcts.rs:361-372)
        float u;
        if (cfg.puct()) {
            float visits = sqrtf(pN);
            u = cfg.cc() * bits_f32(ce.z) * visits / (1.0f + cs.x);
        } else {
            float visits = sqrtf(cfg.cc() * det_logf(pN));
            u = visits / sqrtf(cs.x);
        }
        float v = q + u;
        // Sequential scan semantics of `Some(v) > best` (strict, first wins): a NaN in child 0 is never replaced,
        // a NaN elsewhere never wins.
        if (v != v) v = gl == 0 ? __builtin_inff() : -__builtin_inff();
        v = active ? v : -__builtin_inff();
        int best = row_argmax_first(v, active ? gl : 255);
        if (COUNT) { ctr[CTR_SELECT_LEVELS]++; ctr[CTR_CHILDREN_SCANNED] += nc; }

        // descend: derive the child position in registers
        uint32_t bmeta = row_bcast_u32(ce.y, best);
        int a = (int)meta_action(bmeta);
        int ha = c4::col_height(my | op, a);
        uint64_t nmy = op, nop = my | (1ull << (ha + 7 * a));
        my = nmy;
        op = nop;
        node = fc + (uint32_t)best;
        depth++;
        path_set(X, gl, depth, node);
        fc = row_bcast_u32(ce.x, best);
        meta = bmeta;
        pN = row_bcast_f32(cs.x, best);
        if (!cfg.fpu_const()) {
            pW0 = row_bcast_f32(cs.y, best);
            pW2 = row_bcast_f32(cs.w, best);
        }
    }

    if (hit_solved) {
        // explore(): node already solved -> backprop its one-hot outcome (mcts.rs:314-316)
        uint32_t k = meta_kind(meta);
        X.p0 = k == 0u ? 1.0f : 0.0f;
        X.p1 = k == 1u ? 1.0f : 0.0f;
        X.p2 = k == 2u ? 1.0f : 0.0f;
        X.solved = true;
        if (COUNT) ctr[CTR_SOLVED_HITS]++;
    } else {
        // visit(): expansion, possibly repeated by auto-extend (mcts.rs:374-406)
        for (;;) {
            uint64_t occ = my | op;
            int h = c4::col_height(occ, gl < 9 ? gl : 0);
            bool legal = gl < 9 && h < c4::HEIGHT;
            uint32_t lmask = row_ballot(legal);
            uint32_t n_new = (uint32_t)__popc(lmask);
            uint32_t idx = (uint32_t)__popc(lmask & ((1u << gl) - 1u));
            uint32_t first = T.next_node;
            T.next_node += n_new;
            uint64_t bit = 1ull << (h + 7 * (gl < 9 ? gl : 0));
            uint64_t cop = my | bit;  // child.op_bb = the mover's stones (connect4.rs:224-229)
            bool w = c4::won(cop);
            bool full = (occ | bit) == c4::FULL;
            bool over = legal && (w || full);
            // Outcome::from(reward(child.player())): the mover won -> the child's side to move lost
            uint32_t cmeta = meta_make(0, (uint32_t)gl, over, w ? 0u : 1u, 0u);
            if (legal) {
                T.stat[first + idx] = make_float4(0.f, 0.f, 0.f, 0.f);
                T.edge[first + idx] = make_uint4(0u, cmeta, f32_bits(1.0f), node);
            }
            bool any_solved = row_ballot(over) != 0u;
            meta = (meta & ~META_NC_MASK) | n_new;
            if (gl == 0) *reinterpret_cast<uint2*>(&T.edge[node]) = make_uint2(first, meta);
            if (node == 0) { T.root_fc = first; T.root_nc = n_new; }
            if (COUNT) { ctr[CTR_EXPANSIONS]++; ctr[CTR_NEW_NODES] += n_new; }

            if (cfg.auto_extend() && n_new == 1u) {
                // recurse into the only child without calling the policy (mcts.rs:404-405)
                int a = __ffs((int)lmask) - 1;
                int ha = c4::col_height(occ, a);
                uint64_t abit = 1ull << (ha + 7 * a);
                uint64_t nmy = op, nop = my | abit;
                bool aw = c4::won(nop);
                bool afull = (occ | abit) == c4::FULL;
                node = first;
                depth++;
                path_set(X, gl, depth, node);
                my = nmy;
                op = nop;
                meta = meta_make(0, (uint32_t)a, aw || afull, aw ? 0u : 1u, 0u);
                if (aw || afull) {  // visit() of a solved node returns its one-hot outcome (mcts.rs:377-379)
                    X.p0 = aw ? 1.0f : 0.0f;
                    X.p1 = aw ? 0.0f : 1.0f;
                    X.p2 = 0.0f;
                    X.solved = true;
                    break;
                }
                continue;
            }
            X.needs_eval = true;
            X.solved = any_solved;
            X.fc = first;
            X.legal_mask = lmask;
            break;
        }
    }
    X.depth = depth;
    X.leaf = node;
    X.leaf_my = my;
    X.leaf_op = op;
}

// ---------------------------------------------------------------------------------------------- phase C
// Legal-move softmax of visit() (mcts.rs:409-423) for the node expanded in phase A. lane c = column c.
// `equal_noise_weight` >= 0: this is the root's first visit and PolicyNoise::Equal applies (mcts.rs:258-269):
// prior = prior * (1 - w) + w * (1 / num_children), for roots with at least two children.
SYN_DEV void tree_write_priors(TreeCtx& T, const ExploreCtx& X, int gl, float logit, float equal_noise_weight) {
    uint32_t lmask = X.legal_mask;
    bool legal = (lmask >> gl) & 1u;
    float mx = row_max_f32(legal ? logit : -__builtin_inff());
    float e = legal ? det_expf(logit - mx) : 0.0f;
    float total = 0.0f;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        float ec = row_bcast_f32(e, c);
        if ((lmask >> c) & 1u) total += ec;  // summed in child (= ascending column) order
    }
    float p = e / total;
    uint32_t nc = (uint32_t)__popc(lmask);
    if (equal_noise_weight >= 0.0f && nc >= 2u) {
        float noise = 1.0f / (float)nc;
        p = p * (1.0f - equal_noise_weight) + equal_noise_weight * noise;
    }
    uint32_t idx = (uint32_t)__popc(lmask & ((1u << gl) - 1u));
    if (legal) T.edge[X.fc + idx].z = f32_bits(p);
}

// backprop (mcts.rs:429-488). (d0,d1,d2) = outcome distribution from the leaf's point of view.
template <bool COUNT, bool FAST>
SYN_DEV void tree_backprop(const DevMctsCfg& cfg_, TreeCtx& T, const ExploreCtx& X, int gl, float d0, float d1,
                           float d2, bool solved, uint32_t* ctr) {
    const CfgView<FAST> cfg{cfg_};
    int level = X.depth;
    const uint32_t xp0 = X.path0, xp1 = X.path1, xp2 = X.path2, xp3 = X.path3;
    if (COUNT) {
        ctr[CTR_BACKPROP_LEVELS] += (uint32_t)(X.depth + 1);
        if ((uint32_t)(X.depth + 1) > ctr[CTR_MAX_DEPTH]) ctr[CTR_MAX_DEPTH] = (uint32_t)(X.depth + 1);
    }
    // --- solver walk: level by level while the subtree below is proven
    while (cfg.solve() && solved && level >= 0) {
        uint32_t nd = path_get(xp0, xp1, xp2, xp3, level);
        uint4 e = T.edge[nd];
        float4 s = T.stat[nd];
        uint32_t nc = meta_nc(e.y);
        bool active = (uint32_t)gl < nc;
        uint32_t cmeta = active ? T.edge[e.x + (uint32_t)gl].y : 0u;
        if (COUNT) ctr[CTR_SOLVER_CHILDREN] += nc;
        bool all_solved = row_ballot(active && !meta_some(cmeta)) == 0u;
        uint32_t key = active ? outcome_key_reversed(cmeta) : 0u;
        uint32_t own = outcome_key(meta_some(e.y), meta_kind(e.y), meta_turns(e.y));
        key = row_max_u32(key > own ? key : own);
        bool bsome;
        uint32_t bkind, bturns;
        outcome_from_key(key, bsome, bkind, bturns);
        if (bsome && bkind == 2u) {
            e.y = (e.y & 0xFFu) | (meta_make(0, 0, true, 2u, bturns) & ~0xFFu);
            if (cfg.correct_values()) {
                d0 = -s.y;
                d1 = -s.z;
                d2 = -s.w;
                d2 += s.x + 1.0f;
            }
        } else if (bsome && all_solved) {
            e.y = (e.y & 0xFFu) | (meta_make(0, 0, true, bkind, bturns) & ~0xFFu);
            if (cfg.correct_values()) {
                d0 = -s.y;
                d1 = -s.z;
                d2 = -s.w;
                if (bkind == 1u) d1 += s.x + 1.0f;
                else d0 += s.x + 1.0f;
            }
        } else {
            solved = false;
            break;  // this level (and everything above) is handled by the parallel sweep below
        }
        s.y += d0;
        s.z += d1;
        s.w += d2;
        s.x += 1.0f;
        if (gl == 0) {
            T.stat[nd] = s;
            T.edge[nd].y = e.y;
        }
        if (level == 0) T.root_solved = true;
        float t = d0;
        d0 = d2;
        d2 = t;
        level--;
    }
    // --- parallel sweep over the remaining levels 0..level: lane L&15 of chunk L>>4 owns level L
#pragma unroll
    for (int m = 0; m < 4; m++) {
        int L = m * 16 + gl;
        if (L <= level) {
            uint32_t nd = m == 0 ? xp0 : (m == 1 ? xp1 : (m == 2 ? xp2 : xp3));
            bool flip = ((level - L) & 1) != 0;  // delta[0] <-> delta[2] once per level climbed
            float4 s = T.stat[nd];
            s.y += flip ? d2 : d0;
            s.z += d1;
            s.w += flip ? d0 : d2;
            s.x += 1.0f;
            T.stat[nd] = s;
        }
    }
}

}  // namespace syn

// synthesis_amd — lane-per-tree self-play / search kernel (the high-concurrency launch shape).
//
// Same algorithm, node records and results as the row-per-tree kernels (engine_kernels.cuh, mcts.cuh) — the reference's
//   synthesis/src/mcts.rs:310-488       explore / select_best_child / visit / backprop
//   synthesis/src/alpha_zero.rs:229-338 run_game / sample_action / fill_state_info / store_rewards
// — but with ONE TREE PER LANE: a wave64 owns 64 independent trees, a 1024-thread workgroup 1024, the chip 262,144.
//
// Why a second shape: with one tree per 16-lane DPP row at most 9 of 16 lanes do useful work and a wave amortises every
// scalar/vector instruction of select/expand/backprop over only 4 trees; at 16 k concurrent games that kernel is
// SIMD-issue bound. Here every VALU instruction serves 64 trees (the per-tree code is the sequential
// algorithm, children scanned in a 9-step unrolled loop), each wave evaluates its own four 16-position MFMA tiles out of
// the workgroup's shared LDS weight image, and NOTHING synchronises across waves after the weights are staged: no
// barriers, no exchange buffers, every wave free-runs, so one wave's matrix-core phase overlaps its neighbours'
// pointer-chasing phases on the same SIMD. The price is concurrency (a CU wants 1024 games) and divergence (a wave
// descends until its deepest tree is done), which is why the row kernels stay for <= 16 k concurrent games.
//
//   phase A (per lane)   select + expand (global loads of 32-byte child records, 9 per level)
//   phase B (per wave)   up to four mlp_tile16 evaluations; leaf boards reach the tile layout by ds_bpermute, the 12
//                        outputs per position return through a 1 KB per-wave LDS patch
//   phase C (per lane)   legal softmax -> priors, backprop along parent links, move step when the search is over
#pragma once
#include "engine_kernels.cuh"

namespace syn {

// Node records of this launch shape (private to it: the pool's contents never outlive a launch). A tree slab of `cap`
// nodes is two arrays of 16-byte records:
//   sel[node] = { N, q, P, packed }              everything select_best_child needs about a child: ONE 16-byte load
//       q       = exploit_value as the parent will see it (mcts.rs:343-359): -((W_win - W_lose) / N), rewritten by every
//                 backprop (the float the reference recomputes at selection time); the constant of its outcome once the
//                 node is solved; the FPU constant while it is unvisited (Fpu::Const; Fpu::ParentQ is patched in by the
//                 descent). With the reference's config family the descent uses it as is — no decoding per child.
//       packed  = first_child[0:15] | num_children[16:19] | action[20:23] | solution[24:25] (0 none, 1 Lose, 2 Draw,
//                 3 Win) | turns[26:31]
//   aux[node] = { W_lose, W_draw, W_win, - }     touched only by backprop (never read while N == 0: no initialisation)
// Children of a node are contiguous, so a level of the descent is nine 16-byte loads from one 144-byte span (two cache
// lines) — half the requests and lines of the 32-byte records, which is what the per-CU vector-memory pipeline (the
// bottleneck of this shape) charges for. There are no parent links: the descent logs (node, N) per level into
// a per-wave path buffer [level][lane] (coalesced 512-byte rows) and backprop replays it from the leaf's level down to
// 0, all lanes of a wave on the same level, so backprop has no dependent pointer chase at all.
constexpr uint32_t PW_FC_MASK = 0xFFFFu, PW_NC_SHIFT = 16, PW_ACT_SHIFT = 20, PW_SOL_SHIFT = 24, PW_TURNS_SHIFT = 26;
constexpr uint32_t LANE_MAX_CAP = 1u << 16;  // first_child has 16 bits: trees of up to 65,536 nodes (7,280 explores)
SYN_DEV uint32_t pw_make(uint32_t fc, uint32_t nc, uint32_t action, bool some, uint32_t kind, uint32_t turns = 0) {
    return fc | (nc << PW_NC_SHIFT) | (action << PW_ACT_SHIFT) | ((some ? kind + 1u : 0u) << PW_SOL_SHIFT) |
           (turns << PW_TURNS_SHIFT);
}
SYN_DEV uint32_t pw_fc(uint32_t w) { return w & PW_FC_MASK; }
SYN_DEV uint32_t pw_nc(uint32_t w) { return (w >> PW_NC_SHIFT) & 0xFu; }
SYN_DEV uint32_t pw_action(uint32_t w) { return (w >> PW_ACT_SHIFT) & 0xFu; }
SYN_DEV bool pw_some(uint32_t w) { return ((w >> PW_SOL_SHIFT) & 3u) != 0u; }
SYN_DEV uint32_t pw_kind(uint32_t w) { return ((w >> PW_SOL_SHIFT) & 3u) - 1u; }  // 0 Lose, 1 Draw, 2 Win (if pw_some)
SYN_DEV uint32_t pw_turns(uint32_t w) { return w >> PW_TURNS_SHIFT; }
SYN_DEV uint32_t pw_with_solution(uint32_t w, uint32_t kind, uint32_t turns) {
    return (w & ((1u << PW_SOL_SHIFT) - 1u)) | ((kind + 1u) << PW_SOL_SHIFT) | (turns << PW_TURNS_SHIFT);
}
// exploit_value of a solved child (mcts.rs:343-350): outcome.reversed().value() — child Win -> -1, Draw -> 0,
// Lose -> +1 (game.rs:29-43) — or -inf when solved nodes are not to be selected
SYN_DEV float pw_q_solved(uint32_t kind, bool select_solved) {
    return select_solved ? (kind == 2u ? -1.0f : (kind == 1u ? 0.0f : 1.0f)) : -__builtin_inff();
}

struct LaneTree {
    unsigned char* slab;      // this lane's records: sel[cap] then aux[cap]
    uint64_t root_my, root_op;
    uint32_t next_node;       // nodes.len()
    uint32_t root_fc, root_nc;
    int iter;                 // passes done on this tree (root visit = 1) == root.num_visits
    bool root_solved;
    int job;                  // game / root index, -1 = idle
    int turn;
    uint32_t rng_index;
};

// Descent state of a lane. It persists across rounds: a round ends as soon as `thresh` lanes of the wave stand on a
// leaf (or nobody is descending any more), the lanes that are still on their way down simply continue in the next
// round. So a wave never idles 63 lanes while its deepest tree finishes, and the network runs on (nearly) full tiles.
struct LaneWalk {
    bool descending;
    bool pending;             // stands on an expanded leaf whose network call did not fit this round's full tiles
    uint32_t pend_fc, pend_lmask;
    uint32_t node, wcur;      // current node and its packed word
    float pN, pq;             // its N and stored q
    uint64_t my, op;          // its position
    int level;
};

struct LaneLeaf {             // phase A -> phase C (valid for lanes with at_leaf)
    bool at_leaf;             // this lane finished its descent in this round
    bool was_pending;         // ... in an earlier round (its position already missed the policy cache)
    uint32_t fc;              // first child of the node that needs its children created (valid if needs_eval)
    uint32_t legal_mask;
    bool needs_eval, solved;
    float p0, p1, p2;
};

SYN_DEV float4 ln_sel(const unsigned char* slab, uint32_t i) {
    return *reinterpret_cast<const float4*>(slab + (size_t)i * 16u);
}
SYN_DEV void st_sel(unsigned char* slab, uint32_t i, float N, float y, float P, uint32_t w) {
    *reinterpret_cast<float4*>(slab + (size_t)i * 16u) = make_float4(N, y, P, bits_f32(w));
}
SYN_DEV void st_sel_w(unsigned char* slab, uint32_t i, uint32_t w) {
    *reinterpret_cast<uint32_t*>(slab + (size_t)i * 16u + 12u) = w;
}
SYN_DEV float4 ln_aux(const unsigned char* slab, uint32_t cap, uint32_t i) {
    return *reinterpret_cast<const float4*>(slab + (size_t)(cap + i) * 16u);
}
SYN_DEV void st_aux(unsigned char* slab, uint32_t cap, uint32_t i, float4 v) {
    *reinterpret_cast<float4*>(slab + (size_t)(cap + i) * 16u) = v;
}

template <int MODE>
SYN_DEV void lane_start_job(const EngineParams& P, LaneTree& T) {
    int j = atomicAdd(P.job_next, 1);
    T.job = j < P.n_jobs ? j : -1;
    T.turn = 0;
    T.rng_index = 0;
    T.next_node = 0;
    T.root_fc = 0;
    T.root_nc = 0;
    T.iter = 0;
    T.root_solved = false;
    T.root_my = 0;
    T.root_op = 0;
    if (MODE == MODE_SEARCH && T.job >= 0) {
        T.root_my = P.in_my[T.job];
        T.root_op = P.in_op[T.job];
    }
}

// ---------------------------------------------------------------------------------------------- phase A
// pl = this lane's column of the wave's path buffer: level L lives at pl[L * 64]
template <bool COUNT, bool FAST>
SYN_DEV void lane_select_expand(const DevMctsCfg& cfg_, LaneTree& T, LaneWalk& Wk, LaneLeaf& X, bool active, uint4* pl,
                                uint32_t cap, int thresh, uint32_t* ctr) {
    const CfgView<FAST> cfg{cfg_};
    unsigned char* const slab = T.slab;
    const float y_unvisited = cfg.fpu_const() ? cfg.fpu_value() : 0.0f;
    const bool pending = active && Wk.pending;
    X.at_leaf = false;
    X.was_pending = pending;
    X.needs_eval = pending;
    X.solved = false;
    X.p0 = X.p1 = X.p2 = 0.0f;
    X.fc = Wk.pend_fc;
    X.legal_mask = Wk.pend_lmask;
    uint32_t node = Wk.node, wcur = Wk.wcur;
    float pN = Wk.pN, pq = Wk.pq;
    uint64_t my = Wk.my, op = Wk.op;
    int level = Wk.level;
    bool desc = active && Wk.descending;
    if (active && !desc && !pending) {
        // explore() starts at the root (mcts.rs:310-312)
        if (COUNT) ctr[CTR_EXPLORES]++;
        node = 0;
        level = 0;
        my = T.root_my;
        op = T.root_op;
        wcur = 0;  // the root has action 0 and no solution while it is searched
        pN = 0.0f;
        pq = 0.0f;
        if (T.next_node == 0) {
            T.next_node = 1;  // MCTS::with_capacity pushes the root (mcts.rs:125); its record is written by backprop
        } else {
            wcur = pw_make(T.root_fc, T.root_nc, 0, false, 0);
            pN = (float)T.iter;
            if (!cfg.fpu_const()) {  // Fpu::ParentQ at the first level needs the root's q
                const float4 a = ln_aux(slab, cap, 0);
                pq = -((a.z - a.x) / pN);
            }
        }
        pl[0] = make_uint4(0u, f32_bits(pN), wcur, 0u);
        desc = true;
    }

    // ---- descent (mcts.rs:310-341): every lane walks its own tree, one level per iteration
    // A round ends when `thresh` lanes (a whole number of 16-position tiles) stand on a leaf that needs the network, or
    // when nobody is descending any more.
    bool hit_solved = false, at_leaf = pending;
    for (;;) {
        if (desc) {
            if (pw_some(wcur)) { hit_solved = true; desc = false; at_leaf = true; }
            else if (pw_nc(wcur) == 0u) { desc = false; at_leaf = true; }
        }
        if (__ballot(desc) == 0ull || __popcll(__ballot(at_leaf && !hit_solved)) >= thresh) break;
        if (desc) {
            const uint32_t nc = pw_nc(wcur);
            const uint32_t fc = pw_fc(wcur);
            const float q_fpu = cfg.fpu_const() ? cfg.fpu_value() : -pq;  // parent.q() = -(stored q)
            const float visits = cfg.puct() ? sqrtf(pN) : sqrtf(cfg.cc() * det_logf(pN));
            // select_best_child: sequential scan, `Some(v) > best` (strict: first maximum wins, NaN never replaces)
            float best_v = 0.0f, bN = 0.0f, bq = 0.0f;
            uint32_t best_i = 0, bw = 0;
#pragma unroll
            for (uint32_t i = 0; i < 9; i++) {
                // indices past the last child re-read the last child (valid address, no predicate in front of the
                // loads, so the nine loads of a level are in flight together)
                const float4 s = ln_sel(slab, fc + (i < nc ? i : nc - 1u));
                const uint32_t w = f32_bits(s.w);
                // exploit_value: the record's q slot, except Fpu::ParentQ for an unvisited child
                const float q = (!cfg.fpu_const() && pw_nc(w) == 0u && !pw_some(w)) ? q_fpu : s.y;
                float u;
                if (cfg.puct()) u = cfg.cc() * s.z * visits / (1.0f + s.x);
                else u = visits / sqrtf(s.x);
                const float v = q + u;
                const bool take = i == 0u || (i < nc && v > best_v);
                best_v = take ? v : best_v;
                best_i = take ? i : best_i;
                bw = take ? w : bw;
                bN = take ? s.x : bN;
                bq = take ? s.y : bq;
            }
            if (COUNT) { ctr[CTR_SELECT_LEVELS]++; ctr[CTR_CHILDREN_SCANNED] += nc; }
            const int a = (int)pw_action(bw);
            const int ha = c4::col_height(my | op, a);
            const uint64_t nmy = op, nop = my | (1ull << (ha + 7 * a));
            my = nmy;
            op = nop;
            node = fc + best_i;
            wcur = bw;
            pN = bN;
            pq = bq;
            level++;
            pl[level * 64] = make_uint4(node, f32_bits(pN), wcur, f32_bits(pq));
        }
    }

    if (at_leaf && !pending) {
        if (hit_solved) {
            const uint32_t k = pw_kind(wcur);
            X.p0 = k == 0u ? 1.0f : 0.0f;
            X.p1 = k == 1u ? 1.0f : 0.0f;
            X.p2 = k == 2u ? 1.0f : 0.0f;
            X.solved = true;
            if (COUNT) ctr[CTR_SOLVED_HITS]++;
        } else {
            // visit() (mcts.rs:374-406): allocate the children; their records are written in phase C together with
            // their priors. Only an auto-extended single child is written here (prior 1.0, no policy call).
            for (;;) {
                const uint64_t occ = my | op;
                uint32_t lmask = 0;
#pragma unroll
                for (int c = 0; c < 9; c++)
                    if (c4::col_height(occ, c) < c4::HEIGHT) lmask |= 1u << c;
                const uint32_t n_new = (uint32_t)__popc(lmask);
                const uint32_t first = T.next_node;
                T.next_node = first + n_new;
                wcur = (wcur & ~(PW_FC_MASK | (0xFu << PW_NC_SHIFT))) | first | (n_new << PW_NC_SHIFT);
                st_sel_w(slab, node, wcur);
                pl[level * 64].z = wcur;  // the solver walk reads fc / nc of the path's nodes from the log
                if (node == 0) { T.root_fc = first; T.root_nc = n_new; }
                if (COUNT) { ctr[CTR_EXPANSIONS]++; ctr[CTR_NEW_NODES] += n_new; }

                if (cfg.auto_extend() && n_new == 1u) {
                    const int a = __ffs((int)lmask) - 1;
                    const int ha = c4::col_height(occ, a);
                    const uint64_t abit = 1ull << (ha + 7 * a);
                    const uint64_t nmy = op, nop = my | abit;
                    const bool aw = c4::won(nop);
                    const bool afull = (occ | abit) == c4::FULL;
                    wcur = pw_make(0, 0, (uint32_t)a, aw || afull, aw ? 0u : 1u);
                    st_sel(slab, first, 0.0f, (aw || afull) ? pw_q_solved(aw ? 0u : 1u, cfg.select_solved()) : y_unvisited, 1.0f, wcur);
                    node = first;
                    my = nmy;
                    op = nop;
                    level++;
                    pl[level * 64] = make_uint4(node, f32_bits(0.0f), wcur, 0u);
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
                X.fc = first;
                X.legal_mask = lmask;
                break;
            }
        }
    }
    X.at_leaf = at_leaf;
    Wk.descending = desc;
    Wk.node = node;
    Wk.wcur = wcur;
    Wk.pN = pN;
    Wk.pq = pq;
    Wk.my = my;
    Wk.op = op;
    Wk.level = level;
}

// ---------------------------------------------------------------------------------------------- phase C
// The rest of visit() for the node expanded in phase A (mcts.rs:389-423): creates its children (terminal ones already
// solved) with the legal-move softmax of the nine raw logits as priors. Returns any_solved.
SYN_DEV bool lane_create_children(unsigned char* slab, const LaneLeaf& X, uint64_t leaf_my, uint64_t leaf_op,
                                  const float (&lg)[9], float equal_noise_weight, float y_unvisited, bool select_solved) {
    const uint32_t lmask = X.legal_mask;
    float mx = -__builtin_inff();
#pragma unroll
    for (int c = 0; c < 9; c++)
        if ((lmask >> c) & 1u) mx = lg[c] > mx ? lg[c] : mx;
    float e[9];
    float total = 0.0f;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        e[c] = det_expf(lg[c] - mx);
        if ((lmask >> c) & 1u) total += e[c];  // summed in child (= ascending column) order
    }
    const uint32_t nc = (uint32_t)__popc(lmask);
    const float noise = 1.0f / (float)nc;
    const uint64_t my = leaf_my, occ = leaf_my | leaf_op;
    uint32_t idx = 0;
    bool any_solved = false;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        if ((lmask >> c) & 1u) {
            float p = e[c] / total;
            if (equal_noise_weight >= 0.0f && nc >= 2u) p = p * (1.0f - equal_noise_weight) + equal_noise_weight * noise;
            const int h = c4::col_height(occ, c);
            const uint64_t bit = 1ull << (h + 7 * c);
            const bool w = c4::won(my | bit);  // child.op_bb = the mover's stones (connect4.rs:224-229)
            const bool over = w || (occ | bit) == c4::FULL;
            // Outcome::from(reward(child.player())): the mover won -> the child's side to move lost
            st_sel(slab, X.fc + idx, 0.0f, over ? pw_q_solved(w ? 0u : 1u, select_solved) : y_unvisited, p,
                   pw_make(0, 0, (uint32_t)c, over, w ? 0u : 1u));
            any_solved = any_solved || over;
            idx++;
        }
    }
    return any_solved;
}

SYN_DEV int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// backprop (mcts.rs:429-488) replayed from the path buffer (entries {node, N, packed word, q|turns} as of the descent).
//   phase 1 (per lane): the MCTS-Solver walk, level by level while the subtree below stays proven. Everything a level
//            needs is addressed by its path entry, so its aux record, its children's records and the NEXT level's path
//            entry are fetched together: one memory round trip per level.
//   phase 2 (whole wave, four levels per step): every remaining level just adds the leaf's outcome distribution
//            (win/lose swapped once per level climbed) and one visit, so the levels are independent: the four path rows
//            and then the four aux records are fetched together — two round trips per four levels.
// `leaf_solved`: the walk starts at a node that already carries a solution (explore() hit a solved node, or an
// auto-extended terminal child); its q slot holds the outcome's constant and must stay that way even with the solver off.
template <bool COUNT, bool FAST>
SYN_DEV void lane_backprop(const DevMctsCfg& cfg_, LaneTree& T, int depth, float d0, float d1, float d2, bool solved,
                           bool leaf_solved, bool active, const uint4* pl, uint32_t cap, uint32_t* ctr,
                           unsigned long long* t_mid = nullptr) {
    const CfgView<FAST> cfg{cfg_};
    unsigned char* const slab = T.slab;
    if (COUNT && active) {
        ctr[CTR_BACKPROP_LEVELS] += (uint32_t)(depth + 1);
        if ((uint32_t)(depth + 1) > ctr[CTR_MAX_DEPTH]) ctr[CTR_MAX_DEPTH] = (uint32_t)(depth + 1);
    }
    int L = active ? depth : -1;
    bool keep_q = leaf_solved;  // only ever true for the first level handled
    // ---- phase 1
    if (cfg.solve() && solved && L >= 0) {
        uint4 pe = pl[L * 64];
        for (;;) {
            const uint32_t node = pe.x;
            float N = bits_f32(pe.y);
            uint32_t w = pe.z;
            const uint32_t nc = pw_nc(w), fc = pw_fc(w);
            // one batch: the node's sums, its children (indices past the last child re-read the last one), next entry
            const float4 a = ln_aux(slab, cap, node);
            float4 cs[9];
#pragma unroll
            for (uint32_t i = 0; i < 9; i++) cs[i] = ln_sel(slab, nc == 0u ? node : fc + (i < nc ? i : nc - 1u));
            const uint4 pe_next = pl[(L > 0 ? L - 1 : 0) * 64];
            // a node that was never backpropagated into has no aux record yet
            float W0 = N == 0.0f ? 0.0f : a.x, W1 = N == 0.0f ? 0.0f : a.y, W2 = N == 0.0f ? 0.0f : a.z;
            bool all_solved = true;
            uint32_t key = outcome_key(pw_some(w), pw_kind(w), pw_turns(w));
            if (COUNT) ctr[CTR_SOLVER_CHILDREN] += nc;
#pragma unroll
            for (uint32_t i = 0; i < 9; i++) {
                if (i < nc) {
                    const uint32_t cw = f32_bits(cs[i].w);
                    all_solved = all_solved && pw_some(cw);
                    // solution.map(reversed) (game.rs:29-35): Win<->Lose, Draw stays, turns + 1
                    const uint32_t ck = pw_kind(cw);
                    const uint32_t rk = pw_some(cw) ? outcome_key(true, ck == 1u ? 1u : 2u - ck, pw_turns(cw) + 1u) : 0u;
                    key = rk > key ? rk : key;
                }
            }
            bool bsome;
            uint32_t bkind, bturns;
            outcome_from_key(key, bsome, bkind, bturns);
            if (bsome && bkind == 2u) {
                if (cfg.correct_values()) {
                    d0 = -W0;
                    d1 = -W1;
                    d2 = -W2;
                    d2 += N + 1.0f;
                }
            } else if (bsome && all_solved) {
                if (cfg.correct_values()) {
                    d0 = -W0;
                    d1 = -W1;
                    d2 = -W2;
                    if (bkind == 1u) d1 += N + 1.0f;
                    else d0 += N + 1.0f;
                }
            } else {
                break;  // this level and everything above belongs to phase 2
            }
            w = pw_with_solution(w, bkind, bturns);
            st_sel_w(slab, node, w);
            if (L == 0) T.root_solved = true;
            W0 += d0;
            W1 += d1;
            W2 += d2;
            N += 1.0f;
            st_aux(slab, cap, node, make_float4(W0, W1, W2, 0.0f));
            *reinterpret_cast<float2*>(slab + (size_t)node * 16u) = make_float2(N, pw_q_solved(bkind, cfg.select_solved()));
            const float t = d0;
            d0 = d2;
            d2 = t;
            keep_q = false;
            L--;
            if (L < 0) break;
            pe = pe_next;
        }
    }
    if (t_mid) *t_mid = (unsigned long long)__builtin_readcyclecounter();
    // ---- phase 2: levels L..0 of this lane; (d0,d1,d2) is the delta for level L
    for (int base = wave_max_i32(L); base >= 0; base -= 4) {
        uint2 pe[4];
        float4 a[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int Lj = base - j;
            pe[j] = *reinterpret_cast<const uint2*>(pl + (Lj < 0 ? 0 : Lj) * 64);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int Lj = base - j;
            const bool ok = Lj >= 0 && Lj <= L;
            // a node that was never backpropagated into (N == 0: the fresh leaf of most explores) has no aux record to
            // read, and idle lanes have no node: both fetch their own path entry instead — a valid address that is hot in
            // L2 — so neither costs an HBM sector
            const unsigned char* src = (ok && pe[j].y != 0u) ? slab + (size_t)(cap + pe[j].x) * 16u
                                                             : reinterpret_cast<const unsigned char*>(pl + (Lj < 0 ? 0 : Lj) * 64);
            a[j] = *reinterpret_cast<const float4*>(src);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int Lj = base - j;
            if (Lj >= 0 && Lj <= L) {
                const uint32_t node = pe[j].x;
                float N = bits_f32(pe[j].y);
                const bool flip = ((L - Lj) & 1) != 0;  // delta[0] <-> delta[2] once per level climbed
                const float W0 = (N == 0.0f ? 0.0f : a[j].x) + (flip ? d2 : d0);
                const float W1 = (N == 0.0f ? 0.0f : a[j].y) + d1;
                const float W2 = (N == 0.0f ? 0.0f : a[j].z) + (flip ? d0 : d2);
                N += 1.0f;
                st_aux(slab, cap, node, make_float4(W0, W1, W2, 0.0f));
                const float q = -((W2 - W0) / N);
                if (keep_q && Lj == L) *reinterpret_cast<float*>(slab + (size_t)node * 16u) = N;
                else *reinterpret_cast<float2*>(slab + (size_t)node * 16u) = make_float2(N, q);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- end of a search
struct LaneRoot {
    uint32_t fc, nc, root_w, lmask;
    float rootN;
};
SYN_DEV LaneRoot lane_root(const LaneTree& T) {
    LaneRoot R;
    const float4 s = ln_sel(T.slab, 0);
    R.root_w = f32_bits(s.w);
    R.fc = pw_fc(R.root_w);
    R.nc = pw_nc(R.root_w);
    R.rootN = s.x;
    const uint64_t occ = T.root_my | T.root_op;
    uint32_t lm = 0;
#pragma unroll
    for (int c = 0; c < 9; c++)
        if (c4::col_height(occ, c) < c4::HEIGHT) lm |= 1u << c;
    R.lmask = R.nc == 0u ? 0u : lm;  // the children of the root are its legal columns in ascending order
    return R;
}

// MCTS::target_policy numerators (mcts.rs:174-211) per column (0 for non-children) and their sum in child order
SYN_DEV float lane_policy_weights(const LaneTree& T, const LaneRoot& R, float (&wts)[9]) {
    const bool first_visit = R.rootN == 1.0f;
    const bool root_win = pw_some(R.root_w) && pw_kind(R.root_w) == 2u;
    float total = 0.0f;
    uint32_t idx = 0;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        wts[c] = 0.0f;
        if ((R.lmask >> c) & 1u) {
            const float4 cs = ln_sel(T.slab, R.fc + idx);
            const uint32_t cw = f32_bits(cs.w);
            float v;
            if (first_visit) v = root_win ? ((pw_some(cw) && pw_kind(cw) == 0u) ? 1.0f : 0.0f) : 1.0f;
            else v = cs.x;
            wts[c] = v;
            total += v;
            idx++;
        }
    }
    return total;
}

SYN_DEV void lane_target_q(const LaneTree& T, const LaneRoot& R, uint32_t cap, float& q0, float& q1, float& q2) {
    if (pw_some(R.root_w)) {
        const uint32_t k = pw_kind(R.root_w);
        q0 = k == 0u ? 1.0f : 0.0f;
        q1 = k == 1u ? 1.0f : 0.0f;
        q2 = k == 2u ? 1.0f : 0.0f;
    } else {
        const float4 a = ln_aux(T.slab, cap, 0);
        q0 = a.x / R.rootN;
        q1 = a.y / R.rootN;
        q2 = a.z / R.rootN;
    }
}

// MCTS::best_action (mcts.rs:273-294); also returns the packed word of the chosen child
SYN_DEV int lane_best_action(const LaneTree& T, const LaneRoot& R, int action_selection, uint32_t& best_w) {
    int best = -1;
    float b0 = 0.0f, b1 = 0.0f;
    best_w = 0;
    uint32_t idx = 0;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        if ((R.lmask >> c) & 1u) {
            const float4 cs = ln_sel(T.slab, R.fc + idx);
            const uint32_t cw = f32_bits(cs.w);
            float k0, k1;
            if (pw_some(cw)) {
                const uint32_t kind = pw_kind(cw);
                const float t = (float)pw_turns(cw);
                if (kind == 2u) { k0 = 0.0f; k1 = t; }
                else if (kind == 1u) { k0 = 2.0f; k1 = -t; }
                else { k0 = 3.0f; k1 = -t; }
            } else {
                k0 = 1.0f;
                // -child.q(): the stored q, except for a never-visited child where the reference divides 0 by 0
                const float nq = cs.x == 0.0f ? -((0.0f - 0.0f) / cs.x) : cs.y;
                k1 = action_selection == 0 ? nq : cs.x;
            }
            const bool gt = best < 0 || (k0 > b0) || (k0 == b0 && k1 > b1);
            if (gt) { best = c; b0 = k0; b1 = k1; best_w = cw; }
            idx++;
        }
    }
    return best;
}

SYN_DEV uint32_t lane_child_w(const LaneTree& T, const LaneRoot& R, int action, bool& is_child) {
    is_child = ((R.lmask >> action) & 1u) != 0u;
    const uint32_t idx = (uint32_t)__popc(R.lmask & ((1u << action) - 1u));
    return is_child ? f32_bits(ln_sel(T.slab, R.fc + idx).w) : 0u;
}

// run_game's per-move tail (alpha_zero.rs:243-264) + game end (fill_state_info / store_rewards, 296-338)
template <bool COUNT>
SYN_DEV void lane_move_step(const EngineParams& P, LaneTree& T, uint32_t* ctr) {
    const DevRolloutCfg& rc = P.roll;
    const bool want_random = T.turn < rc.random_until;
    const bool maybe_sample = !want_random && T.turn < rc.sample_until;
    uint32_t rnd = 0;
    if (want_random || maybe_sample) {
        StdRng rng;
        rng.seed_from_u64(P.base_seed + P.first_game + (unsigned long long)T.job);
        rnd = rng.word(T.rng_index);
    }
    const LaneRoot R = lane_root(T);
    float pi[9];
    const float wtotal = lane_policy_weights(T, R, pi);
#pragma unroll
    for (int c = 0; c < 9; c++) pi[c] = pi[c] / wtotal;
    float q0, q1, q2;
    lane_target_q(T, R, P.cap, q0, q1, q2);
    const size_t pos = (size_t)T.job * 63 + (size_t)T.turn;
    P.states_bb[pos * 2 + 0] = T.root_my;
    P.states_bb[pos * 2 + 1] = T.root_op;
    P.root_nodes[pos] = T.next_node;
#pragma unroll
    for (int c = 0; c < 9; c++) P.pis[pos * 9 + c] = pi[c];
    P.vs[pos * 3 + 0] = q0;
    P.vs[pos * 3 + 1] = q1;
    P.vs[pos * 3 + 2] = q2;

    // sample_action (alpha_zero.rs:270-294)
    uint32_t best_w;
    const int best = lane_best_action(T, R, rc.action, best_w);
    int action;
    if (want_random) {
        const uint32_t n = (uint32_t)__popc(R.lmask);
        const uint32_t zone = 0xFFFFFFFFu - (0xFFFFFFFFu - n + 1u) % n;
        uint64_t mm = (uint64_t)rnd * (uint64_t)n;
        T.rng_index += 1;
        while ((uint32_t)mm > zone) {
            StdRng rng;
            rng.seed_from_u64(P.base_seed + P.first_game + (unsigned long long)T.job);
            mm = (uint64_t)rng.word(T.rng_index) * (uint64_t)n;
            T.rng_index += 1;
        }
        const uint32_t r = (uint32_t)(mm >> 32);
        uint32_t m = R.lmask;
        for (uint32_t i = 0; i < r; i++) m &= m - 1u;
        action = __ffs((int)m) - 1;
    } else if (maybe_sample && (!pw_some(best_w) || !rc.stop_when_solved)) {
        float total = pi[0];
        const float chosen_unit = bits_f32((rnd >> 9) | 0x3F800000u) - 1.0f;
        float cum[8];
#pragma unroll
        for (int c = 1; c < 9; c++) {
            cum[c - 1] = total;
            total += pi[c];
        }
        const float chosen = chosen_unit * total + 0.0f;
        T.rng_index += 1;
        int idx = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) idx = cum[c] <= chosen ? c + 1 : idx;
        action = idx;
    } else {
        action = best;
    }
    P.actions[pos] = (unsigned char)action;

    bool a_child;
    const uint32_t a_w = lane_child_w(T, R, action, a_child);
    bool sol_some = a_child && pw_some(a_w);
    uint32_t sol_kind = pw_kind(a_w);

    const uint64_t occ = T.root_my | T.root_op;
    const int h = c4::col_height(occ, action);
    const uint64_t bit = 1ull << (h + 7 * action);
    const uint64_t nmy = T.root_op, nop = T.root_my | bit;
    const bool w = c4::won(nop);
    const bool full = (occ | bit) == c4::FULL;
    if (w || full) {
        sol_some = true;
        sol_kind = w ? 0u : 1u;
    } else if (!rc.stop_when_solved) {
        sol_some = false;
    }
    T.turn += 1;
    if (COUNT) ctr[CTR_MOVES]++;

    if (!sol_some) {
        T.root_my = nmy;
        T.root_op = nop;
        T.next_node = 0;
        T.iter = 0;
        T.root_solved = false;
        return;
    }

    const int n = T.turn;
    const uint32_t last_kind = sol_kind == 1u ? 1u : 2u - sol_kind;
    for (int i = 0; i < n; i++) {
        const bool flip = ((n - 1 - i) & 1) != 0;
        const uint32_t zk = (flip && last_kind != 1u) ? 2u - last_kind : last_kind;
        const float z0 = zk == 0u ? 1.0f : 0.0f, z1 = zk == 1u ? 1.0f : 0.0f, z2 = zk == 2u ? 1.0f : 0.0f;
        const float t = (float)(i + 1) / (float)n;
        float* v = P.vs + ((size_t)T.job * 63 + (size_t)i) * 3;
        const float a0 = v[0], a1 = v[1], a2 = v[2];
        float o0, o1, o2;
        if (rc.value_target == 1) { o0 = a0; o1 = a1; o2 = a2; }
        else if (rc.value_target == 0) { o0 = z0; o1 = z1; o2 = z2; }
        else if (rc.value_target == 2) {
            const float p = rc.vt_p;
            o0 = a0 * p + z0 * (1.0f - p);
            o1 = a1 * p + z1 * (1.0f - p);
            o2 = a2 * p + z2 * (1.0f - p);
        } else {
            const float p = (1.0f - t) * rc.vt_from + t * rc.vt_to;
            o0 = a0 * (1.0f - p) + z0 * p;
            o1 = a1 * (1.0f - p) + z1 * p;
            o2 = a2 * (1.0f - p) + z2 * p;
        }
        v[0] = o0; v[1] = o1; v[2] = o2;
    }
    P.plies[T.job] = n;
    P.final_kind[T.job] = (unsigned char)sol_kind;
    if (COUNT) ctr[CTR_GAMES]++;
    lane_start_job<MODE_SELFPLAY>(P, T);
}

SYN_DEV void lane_search_finish(const EngineParams& P, LaneTree& T) {
    const LaneRoot R = lane_root(T);
    float pi[9];
    const float wtotal = lane_policy_weights(T, R, pi);
    float q0, q1, q2;
    lane_target_q(T, R, P.cap, q0, q1, q2);
    uint32_t bw;
    const int best = lane_best_action(T, R, P.action_selection, bw);
    DevSearchResult* out = P.results + T.job;
    uint32_t idx = 0;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const bool ch = ((R.lmask >> c) & 1u) != 0u;
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 ca = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch) {
            cs = ln_sel(T.slab, R.fc + idx);
            if (cs.x != 0.0f) ca = ln_aux(T.slab, P.cap, R.fc + idx);
            idx++;
        }
        const uint32_t cw = f32_bits(cs.w);
        out->child_N[c] = cs.x;
        out->child_W[c][0] = ca.x;
        out->child_W[c][1] = ca.y;
        out->child_W[c][2] = ca.z;
        out->child_P[c] = ch ? cs.z : 0.0f;
        const bool some = ch && pw_some(cw);
        out->child_sol[c][0] = some ? 1 : 0;
        out->child_sol[c][1] = some ? (int)pw_kind(cw) : 0;
        out->child_sol[c][2] = some ? (int)pw_turns(cw) : 0;
        out->target_pi[c] = pi[c] / wtotal;
    }
    const float4 ra = ln_aux(T.slab, P.cap, 0);
    out->root_N = R.rootN;
    out->root_W[0] = ra.x; out->root_W[1] = ra.y; out->root_W[2] = ra.z;
    const bool some = pw_some(R.root_w);
    out->root_sol[0] = some ? 1 : 0;
    out->root_sol[1] = some ? (int)pw_kind(R.root_w) : 0;
    out->root_sol[2] = some ? (int)pw_turns(R.root_w) : 0;
    out->num_nodes = T.next_node;
    out->best_action = best;
    out->target_q[0] = q0; out->target_q[1] = q1; out->target_q[2] = q2;
    lane_start_job<MODE_SEARCH>(P, T);
}

// cold path out of line, state by value (see engine_kernels.cuh: TreeGame)
template <bool COUNT>
__device__ __attribute__((noinline)) LaneTree lane_move_step_call(const EngineParams& P, LaneTree t, uint32_t* ctr) {
    lane_move_step<COUNT>(P, t, ctr);
    return t;
}
__device__ __attribute__((noinline)) LaneTree lane_search_finish_call(const EngineParams& P, LaneTree t) {
    lane_search_finish(P, t);
    return t;
}

// ---------------------------------------------------------------------------------------------- PolicyWithCache
// policies/cache.rs:19-32 on the device: memoises Policy::eval by position in one direct-mapped table shared by every game
// of the engine (the reference keeps one HashMap per worker thread). Entry = 64 bytes = one HBM sector:
//   { 12 output words (9 logits, 3 outcome probabilities) | my_bb ^ f | op_bb ^ rotl(f, 32) },  f = 64-bit fold of the outputs
// Lanes of different CUs overwrite slots without any lock; an entry torn by concurrent writers (or never written: all
// zero) fails the fold check and is simply a miss, so a hit always returns exactly what the network computed for that
// position — the cache cannot change a result (the reference's argument for a deterministic net).
SYN_DEV uint64_t cache_mix(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}
SYN_DEV uint64_t cache_slot(uint64_t my, uint64_t op, uint32_t shift) {
    return cache_mix(cache_mix(0x243F6A8885A308D3ull, my), op) >> shift;
}
SYN_DEV uint64_t cache_fold(const float (&lg)[9], float v0, float v1, float v2) {
    uint64_t h = 0x13198A2E03707344ull;
    h = cache_mix(h, (uint64_t)f32_bits(lg[0]) | ((uint64_t)f32_bits(lg[1]) << 32));
    h = cache_mix(h, (uint64_t)f32_bits(lg[2]) | ((uint64_t)f32_bits(lg[3]) << 32));
    h = cache_mix(h, (uint64_t)f32_bits(lg[4]) | ((uint64_t)f32_bits(lg[5]) << 32));
    h = cache_mix(h, (uint64_t)f32_bits(lg[6]) | ((uint64_t)f32_bits(lg[7]) << 32));
    h = cache_mix(h, (uint64_t)f32_bits(lg[8]) | ((uint64_t)f32_bits(v0) << 32));
    h = cache_mix(h, (uint64_t)f32_bits(v1) | ((uint64_t)f32_bits(v2) << 32));
    return h;
}
SYN_DEV bool cache_lookup(const uint4* table, uint32_t shift, uint64_t my, uint64_t op, float (&lg)[9], float& v0, float& v1,
                          float& v2) {
    const uint4* e = table + cache_slot(my, op, shift) * 4;
    const uint4 a = e[0], b = e[1], c = e[2], d = e[3];
    float t[9] = {bits_f32(a.x), bits_f32(a.y), bits_f32(a.z), bits_f32(a.w), bits_f32(b.x), bits_f32(b.y), bits_f32(b.z),
                  bits_f32(b.w), bits_f32(c.x)};
    const float u0 = bits_f32(c.y), u1 = bits_f32(c.z), u2 = bits_f32(c.w);
    const uint64_t f = cache_fold(t, u0, u1, u2);
    const uint64_t k0 = (uint64_t)d.x | ((uint64_t)d.y << 32), k1 = (uint64_t)d.z | ((uint64_t)d.w << 32);
    const bool hit = (k0 ^ f) == my && (k1 ^ ((f << 32) | (f >> 32))) == op;
    if (hit) {
#pragma unroll
        for (int i = 0; i < 9; i++) lg[i] = t[i];
        v0 = u0; v1 = u1; v2 = u2;
    }
    return hit;
}
SYN_DEV void cache_insert(uint4* table, uint32_t shift, uint64_t my, uint64_t op, const float (&lg)[9], float v0, float v1,
                          float v2) {
    uint4* e = table + cache_slot(my, op, shift) * 4;
    const uint64_t f = cache_fold(lg, v0, v1, v2);
    const uint64_t k0 = my ^ f, k1 = op ^ ((f << 32) | (f >> 32));
    e[0] = make_uint4(f32_bits(lg[0]), f32_bits(lg[1]), f32_bits(lg[2]), f32_bits(lg[3]));
    e[1] = make_uint4(f32_bits(lg[4]), f32_bits(lg[5]), f32_bits(lg[6]), f32_bits(lg[7]));
    e[2] = make_uint4(f32_bits(lg[8]), f32_bits(v0), f32_bits(v1), f32_bits(v2));
    e[3] = make_uint4((uint32_t)k0, (uint32_t)(k0 >> 32), (uint32_t)k1, (uint32_t)(k1 >> 32));
}

// ---------------------------------------------------------------------------------------------- RolloutPolicy
// policies/rollout.rs:8-31: uniformly random legal moves until the game ends; logits all zero, value = one-hot outcome for
// the player to move at the leaf. Random numbers: Rng::gen_range(0..n as u8) on the tree's own StdRng stream, word
// `rng_index` onwards (every rollout of a tree continues where the previous one stopped).
SYN_DEV void lane_rollout(uint64_t my, uint64_t op, unsigned long long seed, uint32_t& rng_index, float& v0, float& v1,
                          float& v2) {
    StdRng rng;
    rng.seed_from_u64(seed);
    rng.index = rng_index;
    bool leaf_player_moves = true;  // `my` is the side to move; the leaf itself is never terminal
    for (;;) {
        const uint64_t occ = my | op;
        uint32_t lmask = 0;
#pragma unroll
        for (int c = 0; c < 9; c++)
            if (c4::col_height(occ, c) < c4::HEIGHT) lmask |= 1u << c;
        const uint32_t n = (uint32_t)__popc(lmask);
        const uint32_t pick = rng.gen_range_u8(n);
        uint32_t m = lmask;
        for (uint32_t i = 0; i < pick; i++) m &= m - 1u;  // iter_actions().nth(pick)
        const int col = __ffs((int)m) - 1;
        const uint64_t bit = 1ull << (c4::col_height(occ, col) + 7 * col);
        const uint64_t mover = my | bit;
        if (c4::won(mover)) {  // reward(leaf player): +1 if it made the four, -1 otherwise
            v0 = leaf_player_moves ? 0.0f : 1.0f;
            v1 = 0.0f;
            v2 = leaf_player_moves ? 1.0f : 0.0f;
            break;
        }
        if ((occ | bit) == c4::FULL) {
            v0 = 0.0f; v1 = 1.0f; v2 = 0.0f;
            break;
        }
        my = op;
        op = mover;
        leaf_player_moves = !leaf_player_moves;
    }
    rng_index = rng.index;
}

// ---------------------------------------------------------------------------------------------- the kernel
template <int NW>
struct LaneLds {
    static constexpr size_t OUT_OFF = (size_t)MlpGeom::IMG_FLOATS * 4;  // 123,264 B weight + bias image
    static constexpr size_t IDX_OFF = OUT_OFF + (size_t)NW * 1024;     // + 1 KB result patch per wave
    static constexpr size_t BYTES = IDX_OFF + (size_t)NW * 64;         // + 64 B compaction index per wave
};

SYN_DEV uint64_t shfl_u64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// POLICY: 0 = Connect4Net on the matrix cores, 1 = RolloutPolicy (policies/rollout.rs; searches only)
template <int MODE, bool COUNT, bool FAST, int NW, bool PROF = false, int POLICY = 0>
__global__ __launch_bounds__(64 * NW, 1) void selfplay_kernel_lanes(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = 64 * NW;
    float* wimg = reinterpret_cast<float*>(smem_raw);
    const float* bimg = wimg + MlpGeom::W_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    float* outw = reinterpret_cast<float*>(smem_raw + LaneLds<NW>::OUT_OFF) + wave * 256;

    stage_weight_image(wimg, P.wimg, tid, NT);
    const FeatureTable FT = make_feature_table(lane >> 4);

    uint32_t ctr[COUNT ? CTR_COUNT : 1];
#pragma unroll
    for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) ctr[i] = 0;

    LaneTree T;
    const size_t slot = (size_t)blockIdx.x * NT + (size_t)tid;
    T.slab = reinterpret_cast<unsigned char*>(P.stat) + slot * (size_t)P.cap * 32u;
    // this lane's column of its wave's path buffer ([level 0..63][lane 0..63] entries of 16 bytes)
    uint4* const pl = P.path + ((size_t)blockIdx.x * NW + (size_t)wave) * 4096 + (size_t)lane;
    const uint32_t cap = P.cap;
    lane_start_job<MODE>(P, T);
    __syncthreads();  // the only workgroup barrier: weights staged. From here on every wave free-runs.

    const int n_explores = P.roll.num_explores;
    const int thresh = P.lane_thresh;
    unsigned char* const idxw = smem_raw + LaneLds<NW>::IDX_OFF + wave * 64;  // compaction: rank -> lane
    unsigned long long cache_hits = 0, cache_misses = 0;
    LaneWalk Wk;
    Wk.descending = false;
    Wk.pending = false;
    Wk.pend_fc = 0; Wk.pend_lmask = 0;
    Wk.node = 0; Wk.wcur = 0; Wk.pN = 0.0f; Wk.pq = 0.0f; Wk.my = 0; Wk.op = 0; Wk.level = 0;
    unsigned long long pA = 0, pB = 0, pC = 0, pC1 = 0, pC2 = 0, pM = 0, pT = 0, pTiles = 0, pRounds = 0, pLanes = 0, pEvals = 0;
#define SYN_STAMP() (PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
#define SYN_LAP(acc) if (PROF) { unsigned long long n_ = SYN_STAMP(); acc += n_ - pT; pT = n_; }
    for (;;) {
        const bool active = T.job >= 0;
        if (__ballot(active) == 0ull) break;
        pT = SYN_STAMP();
        LaneLeaf X;
        lane_select_expand<COUNT, FAST>(P.mcts, T, Wk, X, active, pl, cap, thresh, ctr);
        SYN_LAP(pA)
        // ---- phase B: the lanes that need the network, compacted into tiles of 16 positions. While other lanes are still
        // descending only whole tiles are evaluated: requests beyond `thresh` stay pending and go first next round.
        const bool want_nn = X.at_leaf && X.needs_eval;  // this lane's expanded leaf needs Policy::eval
        float lg[9];
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
#pragma unroll
        for (int c = 0; c < 9; c++) lg[c] = 0.0f;
        // PolicyWithCache: a position that some game already evaluated skips the network (and its tile slot)
        bool hit = false;
        if (POLICY == 1) {
            // RolloutPolicy: the "evaluation" is a random playout on this lane (no tiles, nothing deferred)
            if (want_nn) lane_rollout(Wk.my, Wk.op, P.base_seed + (unsigned long long)T.job, T.rng_index, v0, v1, v2);
            hit = want_nn;
        } else if (P.cache != nullptr && want_nn && !X.was_pending) {
            hit = cache_lookup(P.cache, P.cache_shift, Wk.my, Wk.op, lg, v0, v1, v2);
        }
        bool need = want_nn && !hit;
        const unsigned long long want_mask = __ballot(need);
        if (POLICY == 0 && P.cache != nullptr) {  // wave-uniform tallies (scalar registers), flushed once at the end of the kernel
            cache_hits += (unsigned long long)__popcll(__ballot(hit));
            cache_misses += (unsigned long long)__popcll(__ballot(want_nn && !hit && !X.was_pending));
        }
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(want_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want_mask, 0u));
        const int quota = __ballot(Wk.descending) != 0ull ? thresh : 64;
        Wk.pending = need && rank >= quota;
        if (Wk.pending) { Wk.pend_fc = X.fc; Wk.pend_lmask = X.legal_mask; }
        need = need && rank < quota;
        const bool fin = X.at_leaf && !Wk.pending;  // this lane's explore gets its network call / backprop in this round
        if (COUNT && (need || hit)) ctr[CTR_POLICY_EVALS]++;
        const unsigned long long need_mask = __ballot(need);
        const int n_need = __popcll(need_mask);
        idxw[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (need) idxw[rank] = (unsigned char)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint64_t hi, lo;
        feature_boards(Wk.my, Wk.op, hi, lo);
#pragma unroll 1
        for (int j = 0; j * 16 < n_need; j++) {
            if (PROF) pTiles++;
            const int src = (int)idxw[16 * j + (lane & 15)];  // (slots past the last request read lane 0: finite input)
            const uint64_t thi = shfl_u64(hi, src), tlo = shfl_u64(lo, src);
            f32x4 o = mlp_tile16(wimg, bimg, lane, FT, thi, tlo);
            const int q = lane >> 4;
            if (q == 2) {
                float a = o[1], b = o[2], c = o[3];
                value_softmax(a, b, c);
                o[1] = a; o[2] = b; o[3] = c;
            }
            if (q < 3) *reinterpret_cast<f32x4*>(outw + (lane & 15) * 16 + q * 4) = o;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (need && (rank >> 4) == j) {
                const float* mine = outw + (rank & 15) * 16;
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(mine);
                const f32x4 r1 = *reinterpret_cast<const f32x4*>(mine + 4);
                const f32x4 r2 = *reinterpret_cast<const f32x4*>(mine + 8);
                lg[0] = r0[0]; lg[1] = r0[1]; lg[2] = r0[2]; lg[3] = r0[3];
                lg[4] = r1[0]; lg[5] = r1[1]; lg[6] = r1[2]; lg[7] = r1[3];
                lg[8] = r2[0]; v0 = r2[1]; v1 = r2[2]; v2 = r2[3];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        SYN_LAP(pB)
        if (PROF) { pRounds++; pLanes += (unsigned long long)__popcll(__ballot(fin)); pEvals += (unsigned long long)n_need; }

        if (P.cache != nullptr && need) cache_insert(P.cache, P.cache_shift, Wk.my, Wk.op, lg, v0, v1, v2);

        // ---- phase C
        float d0 = X.p0, d1 = X.p1, d2 = X.p2;
        bool solved = X.solved;
        if (need || hit) {
            // PolicyNoise::Equal applies to the root's own expansion (mcts.rs:258-269): the first pass of a tree
            const CfgView<FAST> cv{P.mcts};
            solved = lane_create_children(T.slab, X, Wk.my, Wk.op, lg,
                                          (P.mcts.noise == 1 && T.iter == 0 && Wk.level == 0) ? P.mcts.noise_weight : -1.0f,
                                          cv.fpu_const() ? cv.fpu_value() : 0.0f, cv.select_solved());
            d0 = v0;
            d1 = v1;
            d2 = v2;
        }
        SYN_LAP(pC1)
        unsigned long long tmid = 0;
        lane_backprop<COUNT, FAST>(P.mcts, T, Wk.level, d0, d1, d2, solved, X.solved && !(need || hit), fin, pl, cap, ctr,
                                   PROF ? &tmid : nullptr);
        if (PROF) { pC2 += tmid - pT; pT = tmid; }
        SYN_LAP(pC)
        if (fin) {
            T.iter += 1;
            // explore_n (mcts.rs:139-147): the root visit, then up to n explores unless the root is solved
            if (T.iter > n_explores || T.root_solved) {
                // a private copy of the arguments goes to the callee and the slab pointer is re-derived afterwards, so
                // the hot loop's pointers never round-trip through memory (they would come back generic: flat_load)
                EngineParams Pc = P;
                if (MODE == MODE_SELFPLAY) T = lane_move_step_call<COUNT>(Pc, T, ctr);
                else T = lane_search_finish_call(Pc, T);
                T.slab = reinterpret_cast<unsigned char*>(P.stat) + slot * (size_t)P.cap * 32u;
            }
        }
        SYN_LAP(pM)
    }
#undef SYN_STAMP
#undef SYN_LAP
    if (PROF) {
        if (P.prof && lane == 0) {
            unsigned long long* o = P.prof + ((size_t)blockIdx.x * NW + wave) * 10;
            o[0] = pA; o[1] = pB; o[2] = pC; o[3] = pM; o[4] = pRounds; o[5] = pTiles; o[6] = pC1; o[7] = pC2; o[8] = pLanes; o[9] = pEvals;
        }
    }

    if (P.cache != nullptr && lane == 0 && (cache_hits | cache_misses) != 0ull) {
        atomicAdd(P.cache_stats + 0, cache_hits);
        atomicAdd(P.cache_stats + 1, cache_misses);
    }
    if (COUNT) {
        if (P.counters) {
#pragma unroll
            for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) {
                if (i == CTR_MAX_DEPTH) atomicMax(&P.counters[i], (unsigned long long)ctr[i]);
                else if (ctr[i]) atomicAdd(&P.counters[i], (unsigned long long)ctr[i]);
            }
        }
    }
}

}  // namespace syn

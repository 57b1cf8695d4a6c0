// synthesis_amd — the evaluator's baseline opponent on the device: FrozenMCTS over RolloutPolicy, one tree per lane.
//
// Reference: synthesis/src/evaluator.rs:230-534 (Node, exploit / with_capacity, best_action, explore, select_best_child,
// visit, backprop, explore_n) with policies/rollout.rs:8-31 as its policy — the only pairing the reference uses
// (evaluator.rs:170-228: eval_against_rollout_mcts, mcts_vs_mcts). It is NOT the self-play tree of mcts.rs: one scalar
// cum_value per node, the playout runs BEFORE the children are created, unvisited children score `fpu + prior` with no
// exploration term, Uct only, a lost child proves Win(0), explore_n never stops early.
//
// Shape: the baseline needs no network, so there is nothing to batch — every lane owns one root, its tree and its
// StdRng stream, and runs the sequential algorithm; a wave serves 64 matches, the chip 65,536 per pass. The random stream
// of a root starts at a caller-given word of StdRng::seed_from_u64(seed) and the position after the search is returned,
// because the reference draws all playouts of one match from ONE generator (evaluator.rs:171-172, 207-208).
//
// Node record (16 bytes; a node's children are contiguous, in ascending legal-column order, so the child's action and
// position are re-derived from the parent's board during the descent instead of being stored):
//   .x cum_value   .y num_visits   .z action_prob
//   .w first_child[0:20] | visited[21] | solution: some[22] kind[23:24] turns[25:30]   (kind 0 Lose 1 Draw 2 Win)
// num_children is not stored either: it is the number of free columns of the node's position. 21 bits of node index hold the
// reference's largest baseline, VanillaMCTS204800 (1 + 9 * 204,801 = 1,843,210 nodes at most).
// Path: node id | num_children << 21 of every level of the current descent, [level][lane] per wave (the reference's parent
// pointers).
#pragma once
#include "device_common.cuh"

namespace syn {

struct FrozenParams {
    uint4* pool;                 // node records
    size_t nodes_per_tree;       // record capacity of one lane's slab
    uint32_t* path;              // [wave][level 0..63][lane]
    const unsigned long long* in_my;
    const unsigned long long* in_op;
    const unsigned long long* seeds;
    unsigned long long* rng_words;  // in: first stream word of each root; out: first unused word
    const int* explores;         // per root
    int n_roots;
    int n_lanes;                 // trees the pool holds at this record capacity (<= n_roots); roots beyond take a later turn
    float c, fpu_value;
    int solve, action_selection;
    struct FrozenResult* results;
    int* error;
};

struct FrozenResult {
    float child_N[9], child_cum[9], child_P[9];
    int child_sol[9][3];
    float root_N, root_cum;
    int root_sol[3];
    uint32_t num_nodes;
    int best_action;
};

namespace fz {
constexpr uint32_t LOSE = 0, DRAW = 1, WIN = 2;
constexpr uint32_t ID_MASK = 0x1FFFFFu, VISITED = 1u << 21, SOL_MASK = 0x7FC00000u;
SYN_DEV uint32_t first_child(uint32_t w) { return w & ID_MASK; }
SYN_DEV bool visited(uint32_t w) { return (w >> 21) & 1u; }
SYN_DEV bool some(uint32_t w) { return (w >> 22) & 1u; }
SYN_DEV uint32_t kind(uint32_t w) { return (w >> 23) & 3u; }
SYN_DEV uint32_t turns(uint32_t w) { return (w >> 25) & 63u; }
SYN_DEV uint32_t sol_bits(uint32_t k, uint32_t t) { return (1u << 22) | (k << 23) | (t << 25); }
SYN_DEV bool unvisited(uint32_t w) { return !visited(w) && !some(w); }
// game.rs:46-60 as one integer key: Lose(t) ascending < Draw(t) ascending < Win(t) descending
SYN_DEV uint32_t order_key(uint32_t w) { return (kind(w) << 8) | (kind(w) == WIN ? 255u - turns(w) : turns(w)); }
SYN_DEV float value_of_kind(uint32_t k) { return k == WIN ? 1.0f : (k == DRAW ? 0.0f : -1.0f); }  // game.rs:37-43
// free columns: stones stack without holes, so a column is free exactly when its top cell (row 6) is empty
SYN_DEV uint32_t legal_mask(uint64_t occ) {
    const uint64_t free_cells = ~occ;
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < 9; c++) m |= ((uint32_t)(free_cells >> (6 + 7 * c)) & 1u) << c;
    return m;
}
SYN_DEV int nth_set(uint32_t m, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) m &= m - 1u;
    return __ffs((int)m) - 1;
}
}  // namespace fz

// The lane's view of its root's StdRng: the key, the stream position, and a ring of four 16-word ChaCha12 output blocks in
// LDS ([slot][word][lane], conflict-free). Blocks are generated AHEAD, at the start of a playout, for all lanes of the wave
// together (prefetch): a block function is ~800 instructions, and generated on demand inside the playout loop it would run
// in almost every iteration for the few lanes that happen to cross a block boundary there.
struct FrozenRng {
    StdRng base;
    uint32_t index;      // next output word
    uint32_t hi;         // blocks [hi - 4, hi) that were generated since `start` are in the ring (slot = block & 3)
    uint32_t* lds;       // this lane's column: word k of slot s at lds[(s * 16 + k) * 64]

    SYN_DEV void start(uint32_t first_word) { index = first_word; hi = first_word >> 4; }
    SYN_DEV void generate_next() {
        uint32_t out[16];
        base.block16(hi, out);
        uint32_t* slot = lds + (hi & 3u) * 16u * 64u;
#pragma unroll
        for (int k = 0; k < 16; k++) slot[k * 64] = out[k];
        hi++;
    }
    // at a point where the wave is converged: make the next 33+ words available (three blocks from the current one)
    SYN_DEV void prefetch() {
        const uint32_t want = (index >> 4) + 3u;
        while (hi < want) generate_next();
    }
    SYN_DEV uint32_t next_u32() {
        const uint32_t blk = index >> 4;
        while (blk >= hi) generate_next();  // only a playout that outruns the prefetched window gets here
        return lds[((blk & 3u) * 16u + (index++ & 15u)) * 64u];
    }
    // Rng::gen_range(0..n) for u8 (rand 0.8.3 UniformInt<u8>::sample_single), as StdRng::gen_range_u8
    SYN_DEV uint32_t gen_range_u8(uint32_t n) {
        const uint32_t ints_to_reject = (0xFFFFFFFFu - n + 1u) % n;
        const uint32_t zone = 0xFFFFFFFFu - ints_to_reject;
        for (;;) {
            const uint32_t v = next_u32();
            const uint64_t m = (uint64_t)v * (uint64_t)n;
            if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
        }
    }
};

// policies/rollout.rs:8-31 on the root's stream: uniformly random legal moves to the end; returns dist[2] - dist[0] of the
// one-hot outcome for the player to move at the leaf
SYN_DEV float frozen_playout(uint64_t my, uint64_t op, FrozenRng& rng) {
    rng.prefetch();
    bool leaf_player_moves = true;
    for (;;) {
        const uint64_t occ = my | op;
        const uint32_t lmask = fz::legal_mask(occ);
        const uint32_t pick = rng.gen_range_u8((uint32_t)__popc(lmask));
        const int col = fz::nth_set(lmask, pick);
        const uint64_t bit = 1ull << (c4::col_height(occ, col) + 7 * col);
        const uint64_t mover = my | bit;
        if (c4::won(mover)) return leaf_player_moves ? 1.0f : -1.0f;
        if ((occ | bit) == c4::FULL) return 0.0f;
        my = op;
        op = mover;
        leaf_player_moves = !leaf_player_moves;
    }
}

// One wave per workgroup: the waves of a small batch spread over all CUs instead of filling a quarter of them.
__global__ __launch_bounds__(64) void frozen_rollout_kernel(FrozenParams P) {
    __shared__ uint32_t rng_blocks[4 * 16 * 64];
    const int lane = threadIdx.x;
    const size_t gwave = blockIdx.x;
    const size_t glane = (size_t)blockIdx.x * 64 + threadIdx.x;
    uint4* const pool = P.pool + glane * P.nodes_per_tree;
    uint32_t* const path = P.path + gwave * 4096 + lane;  // level l at path[l * 64]
    const uint32_t node_cap = (uint32_t)(P.nodes_per_tree < fz::ID_MASK ? P.nodes_per_tree : fz::ID_MASK);

    if (glane >= (size_t)P.n_lanes) return;
    for (size_t root = glane; root < (size_t)P.n_roots; root += (size_t)P.n_lanes) {
        const uint64_t root_my = P.in_my[root], root_op = P.in_op[root];
        FrozenRng rng;
        rng.base.seed_from_u64(P.seeds[root]);
        rng.start((uint32_t)P.rng_words[root]);
        rng.lds = rng_blocks + threadIdx.x;
        uint32_t next_node = 1;
        bool overflow = false;
        pool[0] = make_uint4(0u, 0u, 0u, 0u);  // Node::unvisited(0, game, None, 0, 0.0)
        const int total = P.explores[root] + 1;  // with_capacity's own visit of the root, then explore_n (evaluator.rs:325-338)

        for (int it = 0; it < total && !overflow; it++) {
            // ---- explore (evaluator.rs:391-406): descend to a solved or unvisited node
            // The descent only FINDS the node; its visit (playout + children) runs after the loop, once for the whole wave —
            // inside the loop the playout would be replayed for every distinct leaf depth among the 64 lanes.
            uint32_t id = 0, level = 0, lmask, nc;
            uint64_t my = root_my, op = root_op, occ;
            uint4 nd = pool[0];  // below the root a node's record arrives with its parent's child scan: one round trip per level
            float value = 0.0f;
            bool solved = false, visit;
            for (;;) {
                occ = my | op;
                lmask = fz::legal_mask(occ);
                nc = (uint32_t)__popc(lmask);
                path[level * 64] = id | (nc << 21);
                if (fz::some(nd.w)) { value = fz::value_of_kind(fz::kind(nd.w)); solved = true; visit = false; break; }
                if (!fz::visited(nd.w)) { visit = true; break; }
                // ---- select_best_child (evaluator.rs:408-437)
                const uint32_t fc = fz::first_child(nd.w);
                const float visits = sqrtf(P.c * det_logf(bits_f32(nd.y)));
                uint32_t best = 0;
                float best_value = -INFINITY;
                uint4 best_rec = make_uint4(0u, 0u, 0u, 0u);
                // all nine child records in one batch of independent loads (unconditional: indices past the last child are
                // clamped onto it), then the scan — loaded inside the scan loop they are nine dependent round trips
                uint4 chs[9];
#pragma unroll
                for (int j = 0; j < 9; j++) chs[j] = pool[fc + ((uint32_t)j < nc ? (uint32_t)j : nc - 1u)];
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    const uint4 ch = chs[j];
                    float v;
                    if ((uint32_t)j >= nc) {
                        v = -INFINITY;   // not a child: never selected (j == 0 is always one)
                    } else if (fz::unvisited(ch.w)) {
                        v = P.fpu_value + bits_f32(ch.z);
                    } else {
                        // a solved child counts from the parent's side: outcome.reversed().value()
                        const float q = fz::some(ch.w) ? -fz::value_of_kind(fz::kind(ch.w)) : -bits_f32(ch.x) / bits_f32(ch.y);
                        v = q + visits / sqrtf(bits_f32(ch.y));
                    }
                    if (j == 0 || ((uint32_t)j < nc && v > best_value)) { best = (uint32_t)j; best_value = v; best_rec = ch; }
                }
                const int col = fz::nth_set(lmask, best);
                const uint64_t mover = my | (1ull << (c4::col_height(occ, col) + 7 * col));
                my = op;
                op = mover;
                id = fc + best;
                nd = best_rec;
                level++;
            }
            if (visit) {
                // ---- visit (evaluator.rs:439-483): playout first, then one child per legal column
                value = frozen_playout(my, op, rng);
                if (next_node + nc > node_cap) { overflow = true; break; }
                // stable softmax over the legal actions (evaluator.rs:468-477); RolloutPolicy's logits are all zero, so
                // their maximum is zero too — the exp / sum / divide sequence is kept as the reference performs it
                const float max_logit = 0.0f;
                float e[9], tot = 0.0f;
#pragma unroll
                for (int j = 0; j < 9; j++) e[j] = 0.0f;
#pragma unroll
                for (int j = 0; j < 9; j++)
                    if ((uint32_t)j < nc) { e[j] = det_expf(0.0f - max_logit); tot += e[j]; }
                uint32_t m = lmask;
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    if ((uint32_t)j < nc) {
                        const int col = __ffs((int)m) - 1;
                        m &= m - 1u;
                        const uint64_t bit = 1ull << (c4::col_height(occ, col) + 7 * col);
                        uint32_t w = 0;
                        if (c4::won(my | bit)) { w = fz::sol_bits(fz::LOSE, 0); solved = true; }       // reward(child.player()) = -1
                        else if ((occ | bit) == c4::FULL) { w = fz::sol_bits(fz::DRAW, 0); solved = true; }
                        pool[next_node + j] = make_uint4(0u, 0u, f32_bits(e[j] / tot), w);
                    }
                }
                pool[id] = make_uint4(nd.x, nd.y, nd.z, (nd.w & fz::SOL_MASK) | next_node | fz::VISITED);
                next_node += nc;
            }
            if (overflow) break;

            // ---- backprop (evaluator.rs:485-527) along the recorded path
            // Loads run ahead of the level being processed (a level never modifies its ancestors' records): the path entry two
            // levels up and the node record one level up are requested before this level's work, so a level costs its own
            // arithmetic and store instead of two dependent round trips. The leaf's record is still in `nd` from the descent.
            uint32_t pe_cur = path[level * 64];
            uint32_t pe_up = path[(level >= 1u ? level - 1u : 0u) * 64];
            uint4 nd_cur = visit ? make_uint4(nd.x, nd.y, nd.z, (nd.w & fz::SOL_MASK) | (next_node - nc) | fz::VISITED) : nd;
            for (int l = (int)level; l >= 0; l--) {
                const uint32_t pe_up2 = path[(l >= 2 ? l - 2 : 0) * 64];
                const uint4 nd_up = pool[l >= 1 ? (pe_up & fz::ID_MASK) : 0u];
                const uint32_t pe = pe_cur;
                const uint32_t nid = pe & fz::ID_MASK, nc = pe >> 21;
                uint4 nd = nd_cur;
                pe_cur = pe_up;
                pe_up = pe_up2;
                nd_cur = nd_up;
                if (P.solve && solved && !fz::some(nd.w)) {
                    bool all_solved = true, have_worst = false;
                    uint32_t worst = 0;
                    const uint32_t fc = fz::first_child(nd.w);
                    const uint32_t n_scan = fz::visited(nd.w) ? nc : 0u;
                    for (uint32_t j = 0; j < n_scan; j++) {
                        const uint32_t cw = pool[fc + j].w;
                        if (!fz::some(cw)) all_solved = false;  // is_unvisited() || is_unsolved()
                        else if (!have_worst || fz::order_key(cw) < fz::order_key(worst)) { worst = cw; have_worst = true; }
                    }
                    const float cum = bits_f32(nd.x), nv = bits_f32(nd.y);
                    if (have_worst && fz::kind(worst) == fz::LOSE) {
                        nd.w = (nd.w & ~fz::SOL_MASK) | fz::sol_bits(fz::WIN, 0);
                        value = -cum + (nv + 1.0f);
                    } else if (fz::visited(nd.w) && all_solved) {
                        // worst.reversed(): Win -> Lose, Draw -> Draw, one turn later (never Lose here)
                        const uint32_t k = fz::kind(worst) == fz::WIN ? fz::LOSE : fz::DRAW;
                        nd.w = (nd.w & ~fz::SOL_MASK) | fz::sol_bits(k, fz::turns(worst) + 1u);
                        value = k == fz::DRAW ? -cum : -cum - (nv + 1.0f);
                    } else {
                        solved = false;
                    }
                }
                nd.x = f32_bits(bits_f32(nd.x) + value);
                nd.y = f32_bits(bits_f32(nd.y) + 1.0f);
                pool[nid] = nd;
                value = -value;
            }
        }
        if (overflow) *P.error = 2;

        // ---- results + best_action (evaluator.rs:364-389)
        FrozenResult& R = P.results[root];
        const uint4 rt = pool[0];
        const uint32_t lmask = fz::legal_mask(root_my | root_op);
        int best_action = -1;
        float best_value = -INFINITY;
        uint32_t m = lmask;
        const uint32_t fc = fz::first_child(rt.w);
        uint32_t j = 0;
        for (int a = 0; a < 9; a++) {
            uint4 ch = make_uint4(0u, 0u, 0u, 0u);
            const bool legal = (m >> a) & 1u;
            if (legal && !overflow) ch = pool[fc + j];
            if (legal) j++;
            R.child_N[a] = bits_f32(ch.y);
            R.child_cum[a] = bits_f32(ch.x);
            R.child_P[a] = bits_f32(ch.z);
            R.child_sol[a][0] = fz::some(ch.w) ? 1 : 0;
            R.child_sol[a][1] = fz::some(ch.w) ? (int)fz::kind(ch.w) : 0;
            R.child_sol[a][2] = fz::some(ch.w) ? (int)fz::turns(ch.w) : 0;
            if (!legal || overflow || fz::unvisited(ch.w)) continue;
            float v;
            if (fz::some(ch.w)) v = fz::kind(ch.w) == fz::WIN ? -INFINITY : (fz::kind(ch.w) == fz::DRAW ? 1e6f : INFINITY);
            else v = P.action_selection == 0 ? -bits_f32(ch.x) / bits_f32(ch.y) : bits_f32(ch.y);
            if (best_action < 0 || v > best_value) { best_value = v; best_action = a; }
        }
        R.root_N = bits_f32(rt.y);
        R.root_cum = bits_f32(rt.x);
        R.root_sol[0] = fz::some(rt.w) ? 1 : 0;
        R.root_sol[1] = fz::some(rt.w) ? (int)fz::kind(rt.w) : 0;
        R.root_sol[2] = fz::some(rt.w) ? (int)fz::turns(rt.w) : 0;
        R.num_nodes = next_node;
        R.best_action = best_action;
        P.rng_words[root] = (unsigned long long)rng.index;
    }
}

}  // namespace syn

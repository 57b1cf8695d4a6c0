// synthesis_amd — the fused self-play / search kernel and the stand-alone batched kernels.
//
// selfplay_kernel: ONE launch plays whole games. A workgroup (4 waves) owns 16 trees (one per DPP row) for its entire
// life and keeps its share of the network weights in registers; per explore it runs
//     phase A  (all rows)   select + expand                       [node pool: HBM/L2, latency-bound gathers]
//     phase B  (all waves)  Connect4Net on the 16-position tile   [f32 MFMA, layers split over the 4 waves]
//     phase C  (all rows)   legal softmax -> priors, backprop, and — when a search finishes — the whole move step
//                           of run_game (targets, action sampling, game step, next root / next game)
// separated by two workgroup barriers. Trees never communicate across workgroups, so there is no grid barrier, no
// inter-workgroup visibility protocol and no host round trip until every game of the call has finished. Finished games
// pull the next game index from one global counter (a result depends only on the game index, never on the slot).
#pragma once
#include "mcts.cuh"
#include "mlp.cuh"
#include "f16x2_tile.cuh"

namespace syn {

enum { MODE_SELFPLAY = 0, MODE_SEARCH = 1 };

// mirrors syn_search_result (include/synthesis_amd.h)
struct DevSearchResult {
    float child_N[9];
    float child_W[9][3];
    float child_P[9];
    int child_sol[9][3];
    float root_N;
    float root_W[3];
    int root_sol[3];
    uint32_t num_nodes;
    int best_action;
    float target_pi[9];
    float target_q[3];
};

struct EngineParams {
    DevMctsCfg mcts;
    DevRolloutCfg roll;
    const float* wimg;       // weight image, fragment order (MlpGeom::IMG_FLOATS floats)
    float4* stat;            // node pool: 32-byte records; stat and edge are the same base address (two typed views)
    uint4* edge;
    uint32_t cap;            // nodes per tree slab
    int n_jobs;              // games (self-play) or roots (search)
    int* job_next;           // [0] global job counter (syn_cancel raises it past n_jobs), [1] games finished, [8] error word
    unsigned long long* counters;  // DevCounters (may be null)
    unsigned long long* prof;      // PROF builds: per-wave cycle sums [A, wait1, B, wait2, C, iterations]
    int* error;                    // set to 1 by a quad whose bounded spin gave up (quad-async kernel)
    int lane_thresh;               // lane kernel: a round ends once this many lanes of a wave stand on a leaf
    uint4* cache;                  // lane kernel: PolicyWithCache table (4 x 16 B per entry), null = off
    uint32_t cache_shift;          // 64 - log2(entries)
    unsigned long long* cache_stats;  // [hits, misses]
    uint4* path;                   // lane kernel: per-wave descent log [wave][level 0..63][lane 0..63]
    unsigned char* vw_buf;         // producer/consumer kernel: parked state + network outputs per virtual wave (pc_kernel.cuh)
    int nv;                        // producer/consumer kernel: virtual waves per tree wave
    int debug_prio;                // producer/consumer kernel: bit 0 = matrix waves at priority 3, bit 1 = tree waves at priority 3
    int debug_stub;                // PROF builds only: replace the network by a cheap stand-in (tree-side ceiling)
    // self-play
    unsigned long long base_seed;  // game g uses StdRng::seed_from_u64(base_seed + first_game + g)
    unsigned long long first_game;
    int* plies;
    unsigned long long* states_bb;
    float* pis;
    float* vs;
    unsigned char* actions;
    uint32_t* root_nodes;
    unsigned char* final_kind;
    // search
    const unsigned long long* in_my;
    const unsigned long long* in_op;
    DevSearchResult* results;
    int action_selection;
};

// ---------------------------------------------------------------------------------------------- end-of-search helpers
struct RootView {      // lane c = column c of the root
    bool is_child;     // column c is a child of the root
    float N, W0, W1, W2, P;
    uint32_t meta;
    uint32_t nc;
    float rootN, rW0, rW1, rW2;
    uint32_t root_meta;
    uint32_t lmask;
};

SYN_DEV RootView load_root_view(const TreeCtx& T, int gl) {
    RootView R;
    uint4 e = T.edge[0];
    float4 s = T.stat[0];
    R.root_meta = e.y;
    R.nc = meta_nc(e.y);
    R.rootN = s.x; R.rW0 = s.y; R.rW1 = s.z; R.rW2 = s.w;
    uint64_t occ = T.root_my | T.root_op;
    bool legal = gl < 9 && c4::col_height(occ, gl < 9 ? gl : 0) < c4::HEIGHT;
    R.lmask = row_ballot(legal);
    // the root's children are its legal columns in ascending order (expansion order)
    uint32_t idx = (uint32_t)__popc(R.lmask & ((1u << gl) - 1u));
    R.is_child = legal && idx < R.nc;
    uint32_t cid = e.x + (R.is_child ? idx : 0u);
    float4 cs = T.stat[cid];
    uint4 ce = T.edge[cid];
    R.N = cs.x; R.W0 = cs.y; R.W1 = cs.z; R.W2 = cs.w;
    R.P = bits_f32(ce.z);
    R.meta = ce.y;
    return R;
}

// MCTS::target_policy (mcts.rs:174-211): returns pi for column gl (0 for non-children)
SYN_DEV float target_policy(const RootView& R, int gl) {
    float v;
    if (R.rootN == 1.0f) {
        bool root_win = meta_some(R.root_meta) && meta_kind(R.root_meta) == 2u;
        if (root_win) v = (meta_some(R.meta) && meta_kind(R.meta) == 0u) ? 1.0f : 0.0f;
        else v = 1.0f;
    } else {
        v = R.N;
    }
    v = R.is_child ? v : 0.0f;
    float total = 0.0f;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        float vc = row_bcast_f32(v, c);
        if ((R.lmask >> c) & 1u) total += vc;
    }
    return v / total;
}

// MCTS::target_q (mcts.rs:213-225)
SYN_DEV void target_q(const RootView& R, float& q0, float& q1, float& q2) {
    if (meta_some(R.root_meta)) {
        uint32_t k = meta_kind(R.root_meta);
        q0 = k == 0u ? 1.0f : 0.0f;
        q1 = k == 1u ? 1.0f : 0.0f;
        q2 = k == 2u ? 1.0f : 0.0f;
    } else {
        q0 = R.rW0 / R.rootN;
        q1 = R.rW1 / R.rootN;
        q2 = R.rW2 / R.rootN;
    }
}

// MCTS::best_action (mcts.rs:273-294): sequential scan in child order with Option<(f32,f32)> `>` semantics
SYN_DEV int best_action(const RootView& R, int action_selection) {
    float k0, k1;
    if (meta_some(R.meta)) {
        uint32_t kind = meta_kind(R.meta);
        float t = (float)meta_turns(R.meta);
        if (kind == 2u) { k0 = 0.0f; k1 = t; }
        else if (kind == 1u) { k0 = 2.0f; k1 = -t; }
        else { k0 = 3.0f; k1 = -t; }
    } else {
        k0 = 1.0f;
        k1 = action_selection == 0 ? -((R.W2 - R.W0) / R.N) : R.N;
    }
    int best = -1;
    float b0 = 0.0f, b1 = 0.0f;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        float c0 = row_bcast_f32(k0, c), c1 = row_bcast_f32(k1, c);
        bool child = row_bcast_u32(R.is_child ? 1u : 0u, c) != 0u;
        bool gt = best < 0 || (c0 > b0) || (c0 == b0 && c1 > b1);
        if (child && gt) { best = c; b0 = c0; b1 = c1; }
    }
    return best;
}

// ---------------------------------------------------------------------------------------------- the fused kernel
struct GameCtx {          // self-play state of the game a row is playing
    int job;              // game / root index, -1 = idle
    int turn;             // plies played so far
    uint32_t rng_index;   // StdRng words consumed by this game
};

template <int MODE>
SYN_DEV void start_job(const EngineParams& P, TreeCtx& T, GameCtx& G, int gl) {
    int j = 0;
    if (gl == 0) j = atomicAdd(P.job_next, 1);
    j = (int)row_bcast_u32((uint32_t)j, 0);
    G.job = j < P.n_jobs ? j : -1;
    G.turn = 0;
    G.rng_index = 0;
    T.next_node = 0;
    T.root_fc = 0;
    T.root_nc = 0;
    T.iter = 0;
    T.root_solved = false;
    if (MODE == MODE_SELFPLAY) {
        T.root_my = 0;  // G::new() (connect4.rs:181-188)
        T.root_op = 0;
    } else if (G.job >= 0) {
        T.root_my = P.in_my[G.job];
        T.root_op = P.in_op[G.job];
    }
}

// run_game's per-move tail (alpha_zero.rs:243-264) + game end (fill_state_info / store_rewards, 296-338)
template <bool COUNT>
SYN_DEV void selfplay_move_step(const EngineParams& P, TreeCtx& T, GameCtx& G, int gl, uint32_t* ctr) {
    const DevRolloutCfg& rc = P.roll;
    // sample_action consumes exactly one StdRng word in the common case (gen_range's rejection zone is 4 / 2^32):
    // generate it FIRST, while nothing but the tree handle is live (ChaCha12 needs ~32 registers of its own).
    const bool want_random = G.turn < rc.random_until;
    const bool maybe_sample = !want_random && G.turn < rc.sample_until;
    uint32_t rnd = 0;
    if (want_random || maybe_sample) {
        StdRng rng;
        rng.seed_from_u64(P.base_seed + P.first_game + (unsigned long long)G.job);
        rnd = rng.word(G.rng_index);
    }
    RootView R = load_root_view(T, gl);
    float pi = target_policy(R, gl);
    float q0, q1, q2;
    target_q(R, q0, q1, q2);
    const size_t pos = (size_t)G.job * 63 + (size_t)G.turn;
    // buffer.add(&game, &search_policy, ..) + StateInfo::q (alpha_zero.rs:248-250)
    if (gl == 0) {
        P.states_bb[pos * 2 + 0] = T.root_my;
        P.states_bb[pos * 2 + 1] = T.root_op;
        P.root_nodes[pos] = T.next_node;
    }
    if (gl < 9) P.pis[pos * 9 + gl] = pi;
    if (gl < 3) P.vs[pos * 3 + gl] = gl == 0 ? q0 : (gl == 1 ? q1 : q2);

    // sample_action (alpha_zero.rs:270-294)
    int best = best_action(R, rc.action);
    uint32_t best_meta = row_bcast_u32(R.meta, best);
    int action;
    if (want_random) {
        // Rng::gen_range(0..n) for u8 (device_common.cuh StdRng::gen_range_u8), first candidate = rnd
        uint32_t n = (uint32_t)__popc(R.lmask);
        uint32_t zone = 0xFFFFFFFFu - (0xFFFFFFFFu - n + 1u) % n;
        uint64_t mm = (uint64_t)rnd * (uint64_t)n;
        G.rng_index += 1;
        while ((uint32_t)mm > zone) {  // rejected (probability ~1e-9): draw again
            StdRng rng;
            rng.seed_from_u64(P.base_seed + P.first_game + (unsigned long long)G.job);
            mm = (uint64_t)rng.word(G.rng_index) * (uint64_t)n;
            G.rng_index += 1;
        }
        uint32_t r = (uint32_t)(mm >> 32);
        uint32_t m = R.lmask;
        for (uint32_t i = 0; i < r; i++) m &= m - 1u;  // iter_actions().nth(r)
        action = __ffs((int)m) - 1;
    } else if (maybe_sample && (!meta_some(best_meta) || !rc.stop_when_solved)) {
        // WeightedIndex::new(search_policy).sample(rng): 9 weights by column
        float cum[8];
        float total = row_bcast_f32(pi, 0);
#pragma unroll
        for (int c = 1; c < 9; c++) {
            cum[c - 1] = total;
            total += row_bcast_f32(pi, c);
        }
        // Uniform<f32>::new(0, total).sample: [1,2) from the top 23 bits, minus 1, times total
        float chosen = (bits_f32((rnd >> 9) | 0x3F800000u) - 1.0f) * total + 0.0f;
        G.rng_index += 1;
        int idx = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) idx = cum[c] <= chosen ? c + 1 : idx;
        action = idx;
    } else {
        action = best;
    }
    if (gl == 0) P.actions[pos] = (unsigned char)action;

    // solution = mcts.solution(&action) (alpha_zero.rs:254, mcts.rs:296-306)
    bool a_child = row_bcast_u32(R.is_child ? 1u : 0u, action) != 0u;
    uint32_t a_meta = row_bcast_u32(R.meta, action);
    bool sol_some = a_child && meta_some(a_meta);
    uint32_t sol_kind = meta_kind(a_meta);

    // game.step(&action) (connect4.rs:221-233)
    uint64_t occ = T.root_my | T.root_op;
    int h = c4::col_height(occ, action);
    uint64_t bit = 1ull << (h + 7 * action);
    uint64_t nmy = T.root_op, nop = T.root_my | bit;
    bool w = c4::won(nop);
    bool full = (occ | bit) == c4::FULL;
    if (w || full) {
        sol_some = true;
        sol_kind = w ? 0u : 1u;  // reward(player to move).into(): the mover won -> Lose(0); else Draw(0)
    } else if (!rc.stop_when_solved) {
        sol_some = false;
    }
    G.turn += 1;
    if (COUNT) ctr[CTR_MOVES]++;

    if (!sol_some) {
        // next move: fresh tree on the new position (alpha_zero.rs:241-242)
        T.root_my = nmy;
        T.root_op = nop;
        T.next_node = 0;
        T.iter = 0;
        T.root_solved = false;
        return;
    }

    // ---- game over: fill_state_info(solution.reversed()) + store_rewards
    const int n = G.turn;
    const uint32_t last_kind = sol_kind == 1u ? 1u : 2u - sol_kind;  // kind of solution.reversed()
    for (int i = gl; i < n; i += 16) {
        // outcome seen from position i: reversed once per step back from the last position
        bool flip = ((n - 1 - i) & 1) != 0;
        uint32_t zk = (flip && last_kind != 1u) ? 2u - last_kind : last_kind;
        float z0 = zk == 0u ? 1.0f : 0.0f, z1 = zk == 1u ? 1.0f : 0.0f, z2 = zk == 2u ? 1.0f : 0.0f;
        float t = (float)(i + 1) / (float)n;
        float* v = P.vs + ((size_t)G.job * 63 + (size_t)i) * 3;
        float a0 = v[0], a1 = v[1], a2 = v[2];
        float o0, o1, o2;
        if (rc.value_target == 1) { o0 = a0; o1 = a1; o2 = a2; }
        else if (rc.value_target == 0) { o0 = z0; o1 = z1; o2 = z2; }
        else if (rc.value_target == 2) {
            float p = rc.vt_p;
            o0 = a0 * p + z0 * (1.0f - p);
            o1 = a1 * p + z1 * (1.0f - p);
            o2 = a2 * p + z2 * (1.0f - p);
        } else {
            float p = (1.0f - t) * rc.vt_from + t * rc.vt_to;
            o0 = a0 * (1.0f - p) + z0 * p;
            o1 = a1 * (1.0f - p) + z1 * p;
            o2 = a2 * (1.0f - p) + z2 * p;
        }
        v[0] = o0; v[1] = o1; v[2] = o2;
    }
    if (gl == 0) {
        P.plies[G.job] = n;
        P.final_kind[G.job] = (unsigned char)sol_kind;
        atomicAdd(P.job_next + 1, 1);  // games finished so far (syn_progress)
    }
    if (COUNT) ctr[CTR_GAMES]++;
    start_job<MODE_SELFPLAY>(P, T, G, gl);
}

SYN_DEV void search_finish(const EngineParams& P, TreeCtx& T, GameCtx& G, int gl) {
    RootView R = load_root_view(T, gl);
    float pi = target_policy(R, gl);
    float q0, q1, q2;
    target_q(R, q0, q1, q2);
    int best = best_action(R, P.action_selection);
    DevSearchResult* out = P.results + G.job;
    if (gl < 9) {
        bool ch = R.is_child;
        out->child_N[gl] = ch ? R.N : 0.0f;
        out->child_W[gl][0] = ch ? R.W0 : 0.0f;
        out->child_W[gl][1] = ch ? R.W1 : 0.0f;
        out->child_W[gl][2] = ch ? R.W2 : 0.0f;
        out->child_P[gl] = ch ? R.P : 0.0f;
        bool some = ch && meta_some(R.meta);
        out->child_sol[gl][0] = some ? 1 : 0;
        out->child_sol[gl][1] = some ? (int)meta_kind(R.meta) : 0;
        out->child_sol[gl][2] = some ? (int)meta_turns(R.meta) : 0;
        out->target_pi[gl] = pi;
    }
    if (gl == 0) {
        out->root_N = R.rootN;
        out->root_W[0] = R.rW0; out->root_W[1] = R.rW1; out->root_W[2] = R.rW2;
        bool some = meta_some(R.root_meta);
        out->root_sol[0] = some ? 1 : 0;
        out->root_sol[1] = some ? (int)meta_kind(R.root_meta) : 0;
        out->root_sol[2] = some ? (int)meta_turns(R.root_meta) : 0;
        out->num_nodes = T.next_node;
        out->best_action = best;
        out->target_q[0] = q0; out->target_q[1] = q1; out->target_q[2] = q2;
    }
    start_job<MODE_SEARCH>(P, T, G, gl);
}

// Out-of-line forms of the cold end-of-search code (once per ~570 explores): used by the 4-quad kernel, whose 128-VGPR
// budget must be set by the hot loop, not by the root view / targets / ChaCha12; the caller-saved spills then happen
// only around the rare call. State goes in and out BY VALUE: passing TreeCtx by reference would park its node-pool
// pointers in memory, and the hot loop's loads would degrade from global_load to flat_load (which also tick the LDS
// counter). Inlined everywhere else: with registers to spare the call only costs.
struct TreeGame {
    TreeCtx T;
    GameCtx G;
};
template <bool COUNT>
__device__ __attribute__((noinline)) TreeGame selfplay_move_step_call(const EngineParams& P, TreeGame s, int gl,
                                                                      uint32_t* ctr) {
    selfplay_move_step<COUNT>(P, s.T, s.G, gl, ctr);
    return s;
}
__device__ __attribute__((noinline)) TreeGame search_finish_call(const EngineParams& P, TreeGame s, int gl) {
    search_finish(P, s.T, s.G, gl);
    return s;
}

// LDS per workgroup (16 trees): [bias image 1,408 B][exA 8 KB][exB 6 KB][leaf boards 16 x 16 B][net outputs 16 x 64 B]
// = 17,024 B, so several workgroups fit one CU; the weights themselves live in registers (mlp.cuh, split variant).
struct EngineLds {
    static constexpr int TPW = 16;
    static constexpr size_t BIAS_OFF = 0;
    static constexpr size_t EXA_OFF = (size_t)MlpGeom::B_FLOATS * 4;
    static constexpr size_t EXB_OFF = EXA_OFF + 8 * 64 * 16;
    static constexpr size_t LEAF_OFF = EXB_OFF + 6 * 64 * 16;
    static constexpr size_t OUT_OFF = LEAF_OFF + TPW * 16;
    static constexpr size_t FLAG_OFF = OUT_OFF + TPW * 64;
    static constexpr size_t W345_OFF = FLAG_OFF + 16;
    static constexpr size_t W345_FLOATS = MlpGeom::W_FLOATS - MlpGeom::W_OFF[2];  // layers 3-5: 9,984 floats
    static constexpr int HOT_NODES = 256;                                         // records per tree kept in LDS (WPS = 1 only)
    static constexpr size_t HOT_OFF = W345_OFF;                                   // 16 trees x 256 records x 32 B = 128 KB
    static constexpr size_t BYTES_WPS1 = HOT_OFF + (size_t)TPW * HOT_NODES * 32;  // all weights in registers
    static constexpr size_t BYTES_WPS2 = W345_OFF + W345_FLOATS * 4;              // + 39,936 B
};

// One workgroup = 256 threads = 4 waves = 16 trees (one per DPP row) = one 16-position MFMA tile.
// WPS = waves per SIMD the register allocation must allow: 1 -> up to 512 VGPRs, one workgroup (16 trees) per CU;
// 2 -> at most 256 VGPRs so two workgroups (32 trees) share a CU.
template <int MODE, bool COUNT, int WPS, bool FAST, bool PROF = false>
__global__ __launch_bounds__(256, WPS) void selfplay_kernel(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = 256;
    float* bimg = reinterpret_cast<float*>(smem_raw + EngineLds::BIAS_OFF);
    f32x4* exA = reinterpret_cast<f32x4*>(smem_raw + EngineLds::EXA_OFF);
    f32x4* exB = reinterpret_cast<f32x4*>(smem_raw + EngineLds::EXB_OFF);
    uint4* leafbuf = reinterpret_cast<uint4*>(smem_raw + EngineLds::LEAF_OFF);
    float* outbuf = reinterpret_cast<float*>(smem_raw + EngineLds::OUT_OFF);
    int* evalflag = reinterpret_cast<int*>(smem_raw + EngineLds::FLAG_OFF);  // [2]: double-buffered "tile needs eval"

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int gl = tid & 15;
    const int t = tid >> 4;  // tree (row) index inside the workgroup

    // this wave's share of the network -> registers, for the lifetime of the kernel
    constexpr bool W345_LDS = WPS >= 2;
    const float* w345 = reinterpret_cast<const float*>(smem_raw + EngineLds::W345_OFF);
    MlpSplitWeights W;
    mlp_split_load_weights<W345_LDS>(P.wimg, wave, lane, W);
    if (W345_LDS) {
        f32x4* dst = reinterpret_cast<f32x4*>(smem_raw + EngineLds::W345_OFF);
        const f32x4* src = reinterpret_cast<const f32x4*>(P.wimg + MlpGeom::W_OFF[2]);
        for (int i = tid; i < (int)(EngineLds::W345_FLOATS / 4); i += NT) dst[i] = src[i];
    }
    const FeatureTable FT = make_feature_table(lane >> 4);
    for (int i = tid; i < MlpGeom::B_FLOATS; i += NT) bimg[i] = P.wimg[MlpGeom::W_FLOATS + i];
    if (tid < 2) evalflag[tid] = 0;

    uint32_t ctr[COUNT ? CTR_COUNT : 1];
#pragma unroll
    for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) ctr[i] = 0;

    TreeCtx T;
    const size_t slot = (size_t)blockIdx.x * EngineLds::TPW + (size_t)t;
    T.stat.base = P.stat + 2 * slot * P.cap;
    T.edge.base = P.edge + 2 * slot * P.cap;
    // WPS = 1 (<= 16 trees per CU: every explore is a dependent chain A -> B -> C on four waves): the first 256 records of each tree
    // in LDS — the root's children and grandchildren are always among them — so the top two levels of a descent and of a backprop
    // cost LDS accesses instead of memory round trips
    T.stat.hot = WPS == 1 ? reinterpret_cast<float4*>(smem_raw + EngineLds::HOT_OFF) + (size_t)t * EngineLds::HOT_NODES * 2 : nullptr;
    T.edge.hot = reinterpret_cast<uint4*>(T.stat.hot);
    T.stat.k = T.edge.k = WPS == 1 ? (uint32_t)EngineLds::HOT_NODES : 0u;
    GameCtx G;
    start_job<MODE>(P, T, G, gl);
    __syncthreads();

    const int n_explores = P.roll.num_explores;
    int it = 0;
    unsigned long long pA = 0, pW1 = 0, pB = 0, pW2 = 0, pC = 0, pT = 0;
#define SYN_STAMP() (PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
    for (;;) {
        pT = SYN_STAMP();
        const bool active = G.job >= 0;
        ExploreCtx X = {};
        if (active) {
            tree_select_expand<COUNT, FAST>(P.mcts, T, X, gl, ctr);
            if (X.needs_eval) {
                if (gl == 0) {
                    uint64_t hi, lo;  // the two derived boards layer 1 reads (mlp.cuh: feature_boards)
                    feature_boards(X.leaf_my, X.leaf_op, hi, lo);
                    leafbuf[t] = make_uint4((uint32_t)hi, (uint32_t)(hi >> 32), (uint32_t)lo, (uint32_t)(lo >> 32));
                    evalflag[it & 1] = 1;
                }
                if (COUNT) ctr[CTR_POLICY_EVALS]++;
            }
        }
        if (PROF) { unsigned long long n = SYN_STAMP(); pA += n - pT; pT = n; }
        if (!__syncthreads_or(active ? 1 : 0)) break;  // barrier 1: leaf boards visible; exit when every row is idle
        if (PROF) { unsigned long long n = SYN_STAMP(); pW1 += n - pT; pT = n; }

        // ---- phase B: the four waves evaluate the tile together (4 internal barriers); skipped (uniformly) when no
        //      tree of the workgroup needs the network this round
        if (evalflag[it & 1]) {
            const int j = lane & 15, q = lane >> 4;
            uint4 b = leafbuf[j];
            uint64_t hi = (uint64_t)b.x | ((uint64_t)b.y << 32), lo = (uint64_t)b.z | ((uint64_t)b.w << 32);
            f32x4 o = mlp_split_tile16<W345_LDS>(W, bimg, w345, exA, exB, wave, lane, FT, hi, lo);
            if (wave == 0) {
                if (q == 2) {
                    float v0 = o[1], v1 = o[2], v2 = o[3];
                    value_softmax(v0, v1, v2);
                    o[1] = v0; o[2] = v1; o[3] = v2;
                }
                if (q < 3) *reinterpret_cast<f32x4*>(outbuf + j * 16 + q * 4) = o;
            }
        }
        if (tid == 0) evalflag[(it + 1) & 1] = 0;
        if (PROF) { unsigned long long n = SYN_STAMP(); pB += n - pT; pT = n; }
        lds_barrier();  // barrier 2: network outputs (LDS) visible; node-pool stores keep draining in the background
        if (PROF) { unsigned long long n = SYN_STAMP(); pW2 += n - pT; pT = n; }

        // ---- phase C
        if (active) {
            float d0 = X.p0, d1 = X.p1, d2 = X.p2;
            if (X.needs_eval) {
                const float* o = outbuf + t * 16;
                float logit = o[gl < 9 ? gl : 0];
                tree_write_priors(T, X, gl, logit, (P.mcts.noise == 1 && T.iter == 0 && X.leaf == 0u) ? P.mcts.noise_weight : -1.0f);
                // one 16-byte LDS read (also keeps these loads from being merged with the X.p* loads above into a
                // pointer phi, which would pin X in scratch)
                f32x4 ov = *reinterpret_cast<const f32x4*>(o + 8);
                d0 = ov[1];
                d1 = ov[2];
                d2 = ov[3];
            }
            tree_backprop<COUNT, FAST>(P.mcts, T, X, gl, d0, d1, d2, X.solved, ctr);
            T.iter += 1;
            // explore_n (mcts.rs:139-147): the root visit, then up to n explores unless the root is solved
            if (T.iter > n_explores || T.root_solved) {
                if (MODE == MODE_SELFPLAY) selfplay_move_step<COUNT>(P, T, G, gl, ctr);
                else search_finish(P, T, G, gl);
            }
        }
        if (PROF) { unsigned long long n = SYN_STAMP(); pC += n - pT; pT = n; }
        it++;
    }
#undef SYN_STAMP
    if (PROF) {
        if (P.prof && lane == 0) {
            unsigned long long* o = P.prof + ((size_t)blockIdx.x * (NT / 64) + wave) * 6;
            o[0] = pA; o[1] = pW1; o[2] = pB; o[3] = pW2; o[4] = pC; o[5] = (unsigned long long)it;
        }
    }

    if (COUNT) {
        if (P.counters && gl == 0) {
#pragma unroll
            for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) {
                if (i == CTR_MAX_DEPTH) atomicMax(&P.counters[i], (unsigned long long)ctr[i]);
                else if (ctr[i]) atomicAdd(&P.counters[i], (unsigned long long)ctr[i]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- quad-async kernel
// Many trees per CU. One workgroup = NQ quads; a quad = 4 waves = 16 trees = one MFMA tile, exactly the unit of the
// kernel above — but here the quads of a workgroup share ONE LDS copy of the weight image (123 KB, which is what limits
// a CU to one such workgroup) and are otherwise independent: each quad runs its own A -> B -> C loop and synchronises
// only with itself through an LDS spin barrier, so while one quad's waves issue MFMAs another quad's waves are
// chasing pointers or back-propagating on the same SIMDs. The 14 KB activation-exchange buffers are a pool of NEX sets
// handed out with an LDS compare-and-swap token (a quad holds one only during phase B).
// Spins are bounded: if a barrier or token wait exceeds SPIN_LIMIT polls the quad raises P.error and leaves.
template <int NQ>
struct QuadLds {
    static constexpr int NEX = NQ >= 3 ? 2 : NQ;                     // exchange sets (2 x 14 KB fit beside the weights)
    static constexpr size_t WIMG_OFF = 0;
    static constexpr size_t EX_OFF = (size_t)MlpGeom::IMG_FLOATS * 4;            // [NEX][exA 8 KB | exB 6 KB]
    static constexpr size_t EX_BYTES = 14 * 1024;
    static constexpr size_t LEAF_OFF = EX_OFF + NEX * EX_BYTES;                  // [NQ][16] uint4
    static constexpr size_t OUT_OFF = LEAF_OFF + (size_t)NQ * 256;               // [NQ][16][16] float
    static constexpr size_t SYNC_OFF = OUT_OFF + (size_t)NQ * 1024;              // [NQ][8] int + [NEX] int
    static constexpr size_t BYTES = SYNC_OFF + (size_t)NQ * 32 + 16;
};
enum { QS_BAR = 0, QS_ACT0 = 1, QS_ACT1 = 2, QS_EVAL0 = 3, QS_EVAL1 = 4, QS_SET = 5 };

template <int MODE, bool COUNT, bool FAST, int NQ, bool PROF = false>
__global__ __launch_bounds__(256 * NQ) void selfplay_kernel_quads(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using L = QuadLds<NQ>;
    constexpr int NT = 256 * NQ;
    constexpr int SPIN_LIMIT = 1 << 22;
    const float* wimg = reinterpret_cast<const float*>(smem_raw + L::WIMG_OFF);
    const float* bimg = wimg + MlpGeom::W_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int quad = wave >> 2;
    const int mw = wave & 3;
    const int gl = tid & 15;
    const int t = (tid >> 4) & 15;  // tree (row) index inside the quad

    uint4* leafbuf = reinterpret_cast<uint4*>(smem_raw + L::LEAF_OFF) + quad * 16;
    float* outbuf = reinterpret_cast<float*>(smem_raw + L::OUT_OFF) + quad * 256;
    int* qs = reinterpret_cast<int*>(smem_raw + L::SYNC_OFF) + quad * 8;
    int* ex_owner = reinterpret_cast<int*>(smem_raw + L::SYNC_OFF) + NQ * 8;

    stage_weight_image(reinterpret_cast<float*>(smem_raw + L::WIMG_OFF), P.wimg, tid, NT);
    if (tid < NQ * 8 + L::NEX) reinterpret_cast<int*>(smem_raw + L::SYNC_OFF)[tid] = 0;
    uint32_t ft_mw;  // this wave's quarter of the layer-1 shift table (features 16*mw + 4*r + q)
    {
        const FeatureTable FT = make_feature_table(lane >> 4);
        ft_mw = mw == 0 ? FT.t[0] : (mw == 1 ? FT.t[1] : (mw == 2 ? FT.t[2] : FT.t[3]));
    }

    uint32_t ctr[COUNT ? CTR_COUNT : 1];
#pragma unroll
    for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) ctr[i] = 0;

    TreeCtx T;
    const size_t slot = ((size_t)blockIdx.x * NQ + quad) * 16 + (size_t)t;
    T.stat.base = P.stat + 2 * slot * P.cap;
    T.edge.base = P.edge + 2 * slot * P.cap;
    T.stat.hot = nullptr; T.edge.hot = nullptr; T.stat.k = T.edge.k = 0u;
    GameCtx G;
    start_job<MODE>(P, T, G, gl);
    __syncthreads();  // weights staged; the only workgroup-wide barrier of the kernel

    int gen = 0;
    bool failed = false;
    // LDS spin barrier of this quad: monotonic arrival counter, one arrival per wave
    // Only LDS traffic is exchanged between the waves of a quad (leaf boards, activations, network outputs): wait for
    // this wave's LDS operations only. A full workgroup release fence would also drain vmcnt, i.e. expose the latency
    // of the node-pool stores (expansion, backprop) that nobody else ever reads, at every barrier.
    auto quad_sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        gen += 4;
        if (lane == 0) __hip_atomic_fetch_add(&qs[QS_BAR], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // every lane polls the same LDS word (a broadcast read); readfirstlane makes the exit test a scalar branch, so
        // the spin is ds_read / s_waitcnt / v_readfirstlane / s_cmp / s_cbranch / s_sleep with no exec-mask bookkeeping
        const int target = __builtin_amdgcn_readfirstlane(gen);
        int spins = 0;
        for (;;) {
            int v = __hip_atomic_load(&qs[QS_BAR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_readfirstlane(v) >= target) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { failed = true; break; }
        }
        asm volatile("" ::: "memory");  // LDS is coherent inside the CU: only stop the compiler from hoisting reads
    };

    const int n_explores = P.roll.num_explores;
    int it = 0;
    unsigned long long pA = 0, pW1 = 0, pB = 0, pW2 = 0, pC = 0, pT = 0;
#define SYN_STAMP() (PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
#define SYN_LAP(acc) if (PROF) { unsigned long long n_ = SYN_STAMP(); acc += n_ - pT; pT = n_; }
    for (;;) {
        pT = SYN_STAMP();
        const bool active = G.job >= 0;
        ExploreCtx X = {};
        if (active) {
            tree_select_expand<COUNT, FAST>(P.mcts, T, X, gl, ctr);
            if (X.needs_eval) {
                if (gl == 0) {
                    uint64_t hi, lo;
                    feature_boards(X.leaf_my, X.leaf_op, hi, lo);
                    leafbuf[t] = make_uint4((uint32_t)hi, (uint32_t)(hi >> 32), (uint32_t)lo, (uint32_t)(lo >> 32));
                    qs[QS_EVAL0 + (it & 1)] = 1;
                }
                if (COUNT) ctr[CTR_POLICY_EVALS]++;
            }
        }
        // wave-level "any tree active" -> quad flag (double-buffered by iteration parity)
        if (__ballot(active) != 0ull && lane == 0) qs[QS_ACT0 + (it & 1)] = 1;
        SYN_LAP(pA)
        quad_sync();  // barrier 1: leaf boards + flags visible to the quad
        SYN_LAP(pW1)
        const bool alive = __hip_atomic_load(&qs[QS_ACT0 + (it & 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
        const bool need_eval = __hip_atomic_load(&qs[QS_EVAL0 + (it & 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
        if (!alive || __any(failed)) break;

        // ---- phase B
        if (need_eval) {
            // wave 0 takes an exchange set from the pool, the quad learns which through barrier 1b
            if (mw == 0 && lane == 0) {
                int got = -1, spins = 0;
                while (got < 0) {
#pragma unroll
                    for (int s = 0; s < L::NEX; s++) {
                        int expected = 0;
                        if (got < 0 && __hip_atomic_compare_exchange_strong(&ex_owner[s], &expected, quad + 1,
                                                                            __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                            __HIP_MEMORY_SCOPE_WORKGROUP))
                            got = s;
                    }
                    if (got < 0) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > SPIN_LIMIT) { failed = true; got = 0; }
                    }
                }
                qs[QS_SET] = got;
            }
            quad_sync();
            // the quad now holds one of the scarce exchange sets: let its MFMA chains win issue arbitration against
            // the other quads' pointer-chasing on the same SIMDs until the set is handed back
            __builtin_amdgcn_s_setprio(3);
            SYN_LAP(pW1)  // token wait is booked with barrier 1
            const int set = __hip_atomic_load(&qs[QS_SET], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            f32x4* exA = reinterpret_cast<f32x4*>(smem_raw + L::EX_OFF + (size_t)set * L::EX_BYTES);
            f32x4* exB = exA + 8 * 64;
            const int j = lane & 15, q = lane >> 4;
            uint4 b = leafbuf[j];
            uint64_t hi = (uint64_t)b.x | ((uint64_t)b.y << 32), lo = (uint64_t)b.z | ((uint64_t)b.w << 32);
            f32x4 o = mlp_quad_tile16(wimg, bimg, exA, exB, mw, lane, ft_mw, hi, lo, quad_sync);
            if (mw == 0) {
                // L5 has consumed exB: nobody touches the exchange set any more -> hand it back
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(&ex_owner[set], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (q == 2) {
                    float v0 = o[1], v1 = o[2], v2 = o[3];
                    value_softmax(v0, v1, v2);
                    o[1] = v0; o[2] = v1; o[3] = v2;
                }
                if (q < 3) *reinterpret_cast<f32x4*>(outbuf + j * 16 + q * 4) = o;
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (mw == 0 && lane == 0) {  // reset next iteration's flags (their writers run after barrier 2)
            qs[QS_ACT0 + ((it + 1) & 1)] = 0;
            qs[QS_EVAL0 + ((it + 1) & 1)] = 0;
        }
        SYN_LAP(pB)
        quad_sync();  // barrier 2: network outputs visible
        SYN_LAP(pW2)

        // ---- phase C
        if (active) {
            float d0 = X.p0, d1 = X.p1, d2 = X.p2;
            if (X.needs_eval) {
                const float* o = outbuf + t * 16;
                float logit = o[gl < 9 ? gl : 0];
                tree_write_priors(T, X, gl, logit, (P.mcts.noise == 1 && T.iter == 0 && X.leaf == 0u) ? P.mcts.noise_weight : -1.0f);
                f32x4 ov = *reinterpret_cast<const f32x4*>(o + 8);
                d0 = ov[1];
                d1 = ov[2];
                d2 = ov[3];
            }
            tree_backprop<COUNT, FAST>(P.mcts, T, X, gl, d0, d1, d2, X.solved, ctr);
            T.iter += 1;
            if (T.iter > n_explores || T.root_solved) {
                if (NQ >= 4) {
                    TreeGame s{T, G};
                    // a private copy goes to the callee: taking the address of the kernel-argument struct itself
                    // would demote every pointer loaded from it to the generic address space
                    EngineParams Pc = P;
                    if (MODE == MODE_SELFPLAY) s = selfplay_move_step_call<COUNT>(Pc, s, gl, ctr);
                    else s = search_finish_call(Pc, s, gl);
                    T = s.T;
                    G = s.G;
                    // re-derive the node-pool pointers from the kernel arguments so they never come back from
                    // memory (keeps the hot loop on global_load / global_store)
                    T.stat.base = P.stat + 2 * slot * P.cap;
                    T.edge.base = P.edge + 2 * slot * P.cap;
                } else {
                    if (MODE == MODE_SELFPLAY) selfplay_move_step<COUNT>(P, T, G, gl, ctr);
                    else search_finish(P, T, G, gl);
                }
            }
        }
        SYN_LAP(pC)
        it++;
    }
#undef SYN_STAMP
#undef SYN_LAP
    if (PROF) {
        if (P.prof && lane == 0) {
            unsigned long long* o = P.prof + ((size_t)blockIdx.x * (NT / 64) + wave) * 6;
            o[0] = pA; o[1] = pW1; o[2] = pB; o[3] = pW2; o[4] = pC; o[5] = (unsigned long long)it;
        }
    }
    if (__any(failed) && lane == 0 && P.error) atomicExch(P.error, 1);

    if (COUNT) {
        if (P.counters && gl == 0) {
#pragma unroll
            for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) {
                if (i == CTR_MAX_DEPTH) atomicMax(&P.counters[i], (unsigned long long)ctr[i]);
                else if (ctr[i]) atomicAdd(&P.counters[i], (unsigned long long)ctr[i]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- stand-alone kernels
// Batched Policy::eval (policies.rs:47-59): n positions -> logits[n][9], value[n][3]. Each wave walks 16-position
// tiles grid-stride with the weight image resident in LDS.
template <int NT>
__global__ __launch_bounds__(NT) void policy_eval_kernel(const float* __restrict__ g_wimg,
                                                         const unsigned long long* __restrict__ my_bb,
                                                         const unsigned long long* __restrict__ op_bb, int n,
                                                         float* __restrict__ logits, float* __restrict__ value) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    stage_weight_image(smem, g_wimg, tid, NT);
    __syncthreads();
    const int ntiles = (n + 15) >> 4;
    const int j = lane & 15, q = lane >> 4;
    const FeatureTable FT = make_feature_table(q);
    for (int tile = blockIdx.x * (NT / 64) + wave; tile < ntiles; tile += gridDim.x * (NT / 64)) {
        // the weight fragments do not depend on the tile: make their LDS OFFSET opaque per iteration so the ~120 reads stay
        // next to the MFMAs that consume them instead of being hoisted out of the loop into spilled registers. (An opaque
        // POINTER would also lose its address space: the reads then become flat_load and wait on vmcnt and lgkmcnt.)
        uint32_t img_off = 0;
        asm volatile("" : "+v"(img_off));
        const float* wimg = smem + img_off;
        const float* bimg = wimg + MlpGeom::W_FLOATS;
        int pos = tile * 16 + j;
        bool valid = pos < n;
        uint64_t my = valid ? my_bb[pos] : 0ull, op = valid ? op_bb[pos] : 0ull;
        uint64_t hi, lo;
        feature_boards(my, op, hi, lo);
        f32x4 o = mlp_tile16(wimg, bimg, lane, FT, hi, lo);
        if (valid) {
            if (q < 2) {
#pragma unroll
                for (int r = 0; r < 4; r++) logits[(size_t)pos * 9 + q * 4 + r] = o[r];
            } else if (q == 2) {
                logits[(size_t)pos * 9 + 8] = o[0];
                float v0 = o[1], v1 = o[2], v2 = o[3];
                value_softmax(v0, v1, v2);
                value[(size_t)pos * 3 + 0] = v0;
                value[(size_t)pos * 3 + 1] = v1;
                value[(size_t)pos * 3 + 2] = v2;
            }
        }
    }
}

// The same in the f16x2 arithmetic (f16x2_tile.cuh): g_img = the F16Geom image.
template <int NT>
__global__ __launch_bounds__(NT) void policy_eval_f16x2_kernel(const uint32_t* __restrict__ g_img,
                                                               const unsigned long long* __restrict__ my_bb,
                                                               const unsigned long long* __restrict__ op_bb, int n,
                                                               float* __restrict__ logits, float* __restrict__ value) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < F16Geom::IMG_WORDS / 4; i += NT) reinterpret_cast<uint4*>(smem16)[i] = reinterpret_cast<const uint4*>(g_img)[i];
    __syncthreads();
    const int ntiles = (n + 15) >> 4;
    const int j = lane & 15, q = lane >> 4;
    for (int tile = blockIdx.x * (NT / 64) + wave; tile < ntiles; tile += gridDim.x * (NT / 64)) {
        uint32_t img_off = 0;   // opaque per iteration: the image reads stay LDS reads next to their MFMAs
        asm volatile("" : "+v"(img_off));
        const uint32_t* img = smem16 + img_off;
        int pos = tile * 16 + j;
        bool valid = pos < n;
        uint64_t my = valid ? my_bb[pos] : 0ull, op = valid ? op_bb[pos] : 0ull;
        uint64_t hi, lo;
        feature_boards(my, op, hi, lo);
        f32x4 o = f16x2_tile16<3>(img, lane, hi, lo);
        const float os = reinterpret_cast<const float*>(img + F16Geom::SCALE_WORD0)[4];
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] *= os;
        if (valid) {
            if (q < 2) {
#pragma unroll
                for (int r = 0; r < 4; r++) logits[(size_t)pos * 9 + q * 4 + r] = o[r];
            } else if (q == 2) {
                logits[(size_t)pos * 9 + 8] = o[0];
                float v0 = o[1], v1 = o[2], v2 = o[3];
                value_softmax(v0, v1, v2);
                value[(size_t)pos * 3 + 0] = v0;
                value[(size_t)pos * 3 + 1] = v1;
                value[(size_t)pos * 3 + 2] = v2;
            }
        }
    }
}

// Game::features (connect4.rs:235-258): out[n][63]
__global__ void features_kernel(const unsigned long long* __restrict__ my_bb,
                                const unsigned long long* __restrict__ op_bb, int n, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)n * 63;
    for (; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t pos = i / 63;
        int f = (int)(i - pos * 63);
        uint64_t my = my_bb[pos], op = op_bb[pos];
        out[i] = c4::feature(my, op, c4::next_free_cells(my | op), f);
    }
}

// slimnn::Linear::forward (linear.rs:17-25): separate multiply and add, ascending input index; one thread per output
__global__ void linear_kernel(int I, int O, const float* __restrict__ W, const float* __restrict__ b,
                              const float* __restrict__ x, int batch, float* __restrict__ y, int relu) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)batch * O;
    for (; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t nb = i / O;
        int o = (int)(i - nb * O);
        float acc = b[o];
        const float* xr = x + nb * I;
        const float* wr = W + (size_t)o * I;
        for (int k = 0; k < I; k++) acc += xr[k] * wr[k];
        y[i] = relu ? __builtin_fmaxf(acc, 0.0f) : acc;
    }
}

// The same layer with the weights transposed into LDS once per workgroup ([input][output]: the lanes of a wave read consecutive
// outputs, conflict-free) and the input row read as a wave-uniform broadcast: per multiply-add one LDS read instead of 64
// divergent global look-ups. Same arithmetic: separate multiply and add, ascending input index. Needs I * O * 4 bytes of LDS.
__global__ void linear_kernel_lds(int I, int O, const float* __restrict__ W, const float* __restrict__ b,
                                  const float* __restrict__ x, int batch, float* __restrict__ y, int relu) {
    extern __shared__ float wt[];
    for (int idx = threadIdx.x; idx < I * O; idx += blockDim.x) {
        const int o = idx / I, k = idx - o * I;
        wt[k * O + o] = W[idx];
    }
    __syncthreads();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)batch * O;
    for (; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t nb = i / O;
        const int o = (int)(i - nb * O);
        float acc = b[o];
        const float* xr = x + nb * I;
        for (int k = 0; k < I; k++) acc += xr[k] * wt[k * O + o];
        y[i] = relu ? __builtin_fmaxf(acc, 0.0f) : acc;
    }
}

// slimnn::Conv2d::forward (conv.rs:45-85): accumulation order ci -> k1 -> k2 per output element
__global__ void conv2d_kernel(int CIN, int COUT, int K, int RP, int CP, int S, int H_IN, int W_IN, int H_OUT,
                              int W_OUT, const float* __restrict__ W, const float* __restrict__ b,
                              const float* __restrict__ x, int batch, float* __restrict__ y, int relu) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t per = (size_t)COUT * H_OUT * W_OUT;
    size_t total = (size_t)batch * per;
    for (; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t nb = i / per;
        size_t rem = i - nb * per;
        int co = (int)(rem / ((size_t)H_OUT * W_OUT));
        int rr = (int)(rem - (size_t)co * H_OUT * W_OUT);
        int r = rr / W_OUT, c = rr - r * W_OUT;
        float acc = b[co];
        const float* xb = x + nb * (size_t)CIN * H_IN * W_IN;
        for (int ci = 0; ci < CIN; ci++)
            for (int k1 = 0; k1 < K; k1++) {
                int in_row = r * S + k1;
                if (RP <= in_row && in_row < H_IN + RP)
                    for (int k2 = 0; k2 < K; k2++) {
                        int in_col = c * S + k2;
                        if (CP <= in_col && in_col < W_IN + CP) {
                            float w = W[(((size_t)co * CIN + ci) * K + k1) * K + k2];
                            float v = xb[((size_t)ci * H_IN + (in_row - RP)) * W_IN + (in_col - CP)];
                            acc += w * v;
                        }
                    }
            }
        y[i] = relu ? __builtin_fmaxf(acc, 0.0f) : acc;
    }
}

// PMC calibration probe (MI355X guide §HBM: FETCH_SIZE / WRITE_SIZE are only calibrated for wide streaming accesses):
// reproduces the node pool's access shape — every 16-lane row gathers one 288-byte sibling span (9 lanes x two 16-byte
// loads) at span_off[i] and optionally rewrites 16 bytes per lane — over a buffer far larger than the caches, so the
// counters can be compared with exactly known byte / cache-line counts (tools/calibrate_pmc.py).
__global__ void calib_gather_kernel(const float4* __restrict__ base, const unsigned* __restrict__ span_off, int n_spans,
                                    float4* __restrict__ sink, int do_write) {
    int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    int gl = threadIdx.x & 15;
    int nrows = (gridDim.x * blockDim.x) >> 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = row; i < n_spans; i += nrows) {
        if (gl < 9) {
            size_t rec = (size_t)span_off[i] + (size_t)gl;  // 32-byte records
            float4 a = base[2 * rec], b = base[2 * rec + 1];
            acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
            if (do_write) const_cast<float4*>(base)[2 * rec] = make_float4(acc.x, a.y, a.z, a.w);
        }
    }
    if (acc.x == 123.456f) sink[row] = acc;  // keep the loads alive
}

// parity probes for the device primitives
__global__ void debug_rng_kernel(unsigned long long seed, int n, uint32_t* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        StdRng r;
        r.seed_from_u64(seed);
        out[i] = r.word((uint32_t)i);
    }
}
// slimnn activations as layers (slimnn/src/activations.rs:31-63) on [batch][n] rows: 0 = ReLU (x.max(0.0)), 1 = Tanh
// (det_tanhf), 2 = Softmax::apply_1d — exp of every element (det_expf, NO max subtraction, as the reference), summed in index
// order, each element divided by the total. One thread per row for the softmax (the sum is sequential by definition).
__global__ void activation_kernel(int kind, const float* x, int batch, int n, float* y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (kind == 2) {
        if (i >= batch) return;
        const float* xr = x + (size_t)i * n;
        float* yr = y + (size_t)i * n;
        float total = 0.0f;
        for (int k = 0; k < n; k++) {
            const float e = det_expf(xr[k]);
            yr[k] = e;
            total += e;
        }
        for (int k = 0; k < n; k++) yr[k] = yr[k] / total;
        return;
    }
    if ((size_t)i >= (size_t)batch * n) return;
    const float v = x[i];
    y[i] = kind == 0 ? __builtin_fmaxf(v, 0.0f) : det_tanhf(v);
}

__global__ void debug_fast_div_kernel(const float* a, const float* b, int n, float* fast, float* full) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int j = i ^ 1;  // partner element: both halves of the packed instructions carry live data
        const f32x2 q = div2_by_small_int(f32x2{a[i], a[j < n ? j : i]}, f32x2{b[i], b[j < n ? j : i]});   // (the descent's form)
        fast[i] = q[0];
        full[i] = a[i] / b[i];
    }
}
// Enumeration behind the descent's two shortened sequences (device_common.cuh): thread t takes the significand t (2^23 of them).
// mism[0] += quotients a / b, a = 1.m x {2^-60, 1, 2^59}, b = b_lo .. b_hi (integers), where div2_by_small_int differs from the IEEE
// division; mism[1] += the same for div2_safe_range; mism[2] += square roots of 1.m x {1, 2, 2^-60, 2^59} where sqrt_normal_range
// differs from sqrtf.
__global__ void debug_small_int_math_kernel(int b_lo, int b_hi, unsigned long long* mism) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= (1u << 23)) return;
    const float scale[3] = {0x1p-60f, 1.0f, 0x1p59f};
    uint32_t bad_small = 0, bad_safe = 0, bad_sqrt = 0;
    for (int e = 0; e < 3; e++) {
        const float a = bits_f32(0x3F800000u | m) * scale[e];   // (exact: a power of two)
        for (int b = b_lo; b <= b_hi; b += 2) {
            const float b0 = (float)b, b1 = (float)(b + 1 <= b_hi ? b + 1 : b);
            const f32x2 q = div2_by_small_int(f32x2{a, a}, f32x2{b0, b1});
            const f32x2 q2 = div2_safe_range(f32x2{a, a}, f32x2{b0, b1});
            const float r0 = a / b0, r1 = a / b1;
            bad_small += (f32_bits(q[0]) != f32_bits(r0)) + (f32_bits(q[1]) != f32_bits(r1));
            bad_safe += (f32_bits(q2[0]) != f32_bits(r0)) + (f32_bits(q2[1]) != f32_bits(r1));
        }
    }
    const float sq[4] = {1.0f, 2.0f, 0x1p-60f, 0x1p59f};
    for (int e = 0; e < 4; e++) {
        const float x = bits_f32(0x3F800000u | m) * sq[e];
        bad_sqrt += f32_bits(sqrt_normal_range(x)) != f32_bits(sqrtf(x));
    }
    if (bad_small) atomicAdd(&mism[0], (unsigned long long)bad_small);
    if (bad_safe) atomicAdd(&mism[1], (unsigned long long)bad_safe);
    if (bad_sqrt) atomicAdd(&mism[2], (unsigned long long)bad_sqrt);
}
__global__ void debug_math_kernel(const float* a, const float* b, int n, float* e, float* d, float* s) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        e[i] = det_expf(a[i]);
        d[i] = a[i] / b[i];
        s[i] = b[i] < 0.0f ? det_logf(a[i]) : sqrtf(a[i]);  // b < 0 selects the log probe
    }
}

}  // namespace syn

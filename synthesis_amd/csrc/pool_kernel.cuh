// synthesis_amd — lane-per-tree kernel with the trees UNBOUND from the lanes for the descent (launch shape 8).
//
// Same algorithm, node blocks, path log, device functions and results as lane_kernel.cuh — the reference's
//   synthesis/src/mcts.rs:310-488       explore / select_best_child / visit / backprop
//   synthesis/src/alpha_zero.rs:229-338 run_game / sample_action / fill_state_info / store_rewards
// What changes is which lane works on which tree, and when.
//
// In lane_kernel.cuh a lane owns one tree. A round is A (descend) -> B (network tiles) -> C (children, backprop), and a round's
// descent runs until 64 (f16x2) or 48 (f32) lanes stand on a leaf: a lane whose tree reaches its leaf at level 2 idles while the
// deepest tree of the wave walks to level 12. The stamps of round 5 (profiles/r05_phase_stamps.txt) put the descent at 46-52 % of
// a round with 23-28 of 64 lanes active; an iteration costs the same ~190 vector instructions and the same memory round trip
// whether 25 or 64 lanes take a level.
//
// Here a wave owns a POOL of M trees (64 < M <= 128) and its 64 lanes are workers:
//   * a tree is DEAD (no game left), READY (between two explores), BOUND (a lane is walking it down) or LEAF (its descent has
//     arrived; it waits for the network and / or its backprop). One state byte per tree in LDS; lane l keeps the books of trees
//     l and l + 64 (two ballots give the wave the 128-bit sets it needs).
//   * descent: a lane whose tree arrives writes the cursor (8 dwords) to the tree's record in global memory, marks it LEAF and
//     binds the next READY tree in the SAME iteration (ballot-compacted: the r-th free lane takes the r-th READY tree) — so as
//     long as the pool holds READY trees every iteration takes 64 levels. The cursor of a tree that keeps descending never
//     leaves its lane's registers: only arrivals and new explores touch the records.
//   * a round fires when 64 trees are LEAF (or nothing else can move): the 64 leaves are compacted onto the lanes (the two halves
//     of the pool take turns to be ranked first: no leaf waits more than two rounds), every lane loads its leaf's tree state,
//     phases B and C run exactly as in lane_kernel.cuh (full tiles without a `pending` queue), the tree states go back and the
//     trees are READY again. The lanes' own descents are parked in LDS meanwhile (8 dwords per lane; 11 with Fpu::Func).
// With M = 128 the bound trees number 128 - LEAF - READY >= 64 whenever READY is empty and fewer than 64 are LEAF: the
// descent runs at 64 of 64 lanes in steady state. Trees share nothing and a game's result depends on nothing but its index,
// so a schedule cannot change a result: every parity test of the lane kernel holds this kernel to the oracle unchanged.
//
// Per-tree record (global memory, [field][tree] per wave so that lanes touching neighbouring trees coalesce; 9 KB per wave):
//   root position (4), next_block | num_nodes << 16, iter | root_solved << 14 | root_sol << 15, job, turn | rng_index << 8,
//   Fpu::Func draws; the parked leaf cursor: record, block | kind << 14 | solved << 16 | level << 17, q | turns, N, position (4).
// Written and read by different lanes of the SAME wave only (wavefront scope: program order, no cache action needed).
#pragma once
#include "lane_kernel.cuh"

namespace syn {

struct PoolGeom {
    static constexpr int M_MAX = 128;                                   // trees per wave: two per book-keeping lane
    static constexpr int FIELDS = 18;                                   // dwords per tree record (17 used)
    static constexpr size_t PARK_OFF = (size_t)FIELDS * M_MAX * 4;      // behind the records: the lanes' own descents during phases B / C,
    static constexpr int PARK_DW = 11;                                  // [dword][lane] (8 dwords; 11 with Fpu::Func)
    static constexpr size_t WAVE_BYTES = PARK_OFF + (size_t)PARK_DW * 64 * 4 + 256;   // 12,288 B
};
enum : uint32_t { PT_DEAD = 0, PT_READY = 1, PT_BOUND = 2, PT_LEAF = 3 };
enum { PF_RMY0 = 0, PF_RMY1, PF_ROP0, PF_ROP1, PF_ALLOC, PF_ITER, PF_JOB, PF_TURN, PF_DRAWS, PF_REC, PF_META, PF_QT, PF_PN,
       PF_MY0, PF_MY1, PF_OP0, PF_OP1 };

template <int NW, int FAST>
struct PoolLds {
    static constexpr size_t IMG = (size_t)MlpGeom::IMG_FLOATS * 4 > (size_t)F16Geom::IMG_WORDS * 4 ? (size_t)MlpGeom::IMG_FLOATS * 4
                                                                                                  : (size_t)F16Geom::IMG_WORDS * 4;
    static constexpr size_t IDX_OFF = IMG;                                   // 64 B compaction index per wave (rank -> tree)
    static constexpr size_t FT_OFF = IDX_OFF + (size_t)NW * 64;              // the four feature shift tables (16 B each)
    static constexpr size_t STATE_OFF = FT_OFF + 64;                         // 128 state bytes per wave
    // what the start of an explore needs of a tree — root position (4 dwords) and pass counter | "root not expanded yet" << 31 —
    // lives in LDS ([dword][tree] per wave): binding a READY tree costs no memory round trip in front of its first level
    static constexpr size_t BEGIN_OFF = STATE_OFF + (size_t)NW * 128;
    static constexpr int BEGIN_DW = 5;
    static constexpr size_t BYTES = BEGIN_OFF + (size_t)BEGIN_DW * 4 * PoolGeom::M_MAX * NW;
    static_assert(BYTES <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");
};

SYN_DEV void pool_lds_sync() {   // LDS hand-off between lanes of one wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
SYN_DEV int pool_rank(unsigned long long m) {   // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// POLICY: 0 = Connect4Net on the f32 matrix cores, 3 = Connect4Net in the f16x2 arithmetic. FAST: 1 = the parity configuration,
// 2 = the reference's own self-play configuration (Fpu::Func(|| Normal(mean, std))): the two compile-time-folded families.
template <int MODE, bool COUNT, int FAST, int NW, int POLICY>
__global__ __launch_bounds__(64 * NW, 1) void selfplay_kernel_pool(EngineParams P) {
    static_assert(FAST == 1 || FAST == 2, "compile-time-folded configuration families only");
    static_assert(POLICY == 0 || POLICY == 3, "Connect4Net only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using L = PoolLds<NW, FAST>;
    constexpr int NT = 64 * NW;
    float* wimg = reinterpret_cast<float*>(smem_raw);
    const float* bimg = wimg + MlpGeom::W_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    if (POLICY == 0) stage_weight_image(wimg, P.wimg, tid, NT);
    if (POLICY == 3) {
        const uint4* src = reinterpret_cast<const uint4*>(P.wimg);
        uint4* dst = reinterpret_cast<uint4*>(smem_raw);
        for (int i = tid; i < F16Geom::IMG_WORDS / 4; i += NT) dst[i] = src[i];
    }
    if (tid < 4) {
        const FeatureTable f = make_feature_table(tid);
        *reinterpret_cast<uint4*>(smem_raw + L::FT_OFF + tid * 16) = make_uint4(f.t[0], f.t[1], f.t[2], f.t[3]);
    }

    uint32_t ctr[COUNT ? CTR_COUNT : 1];
#pragma unroll
    for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) ctr[i] = 0;

    const int M = P.nv;   // trees of this wave's pool
    const size_t wave_g = (size_t)blockIdx.x * NW + (size_t)wave;
    uint32_t* const trec = reinterpret_cast<uint32_t*>(P.vw_buf + wave_g * PoolGeom::WAVE_BYTES);   // [field][M_MAX]
    unsigned char* const st8 = smem_raw + L::STATE_OFF + wave * 128;
    unsigned char* const idxw = smem_raw + L::IDX_OFF + wave * 64;
    uint32_t* const pk = reinterpret_cast<uint32_t*>(P.vw_buf + wave_g * PoolGeom::WAVE_BYTES + PoolGeom::PARK_OFF) + lane;   // dword k at pk[k * 64]
    uint32_t* const bg = reinterpret_cast<uint32_t*>(smem_raw + L::BEGIN_OFF) + wave * (L::BEGIN_DW * PoolGeom::M_MAX);   // dword k of tree t at bg[k * 128 + t]
#define SYN_BG(t, k) bg[(k) * PoolGeom::M_MAX + (t)]
    const uint32_t bcap = P.cap / 4u;
    const size_t slab_bytes = (size_t)P.cap * 32u;
    unsigned char* const slab0 = reinterpret_cast<unsigned char*>(P.stat) + wave_g * (size_t)M * slab_bytes;
    uint4* const path0 = P.path + wave_g * 2 * PATH_ENTRIES;   // two path buffers per wave: tree t -> buffer t >> 6, column t & 63
#define SYN_SLAB(t) (slab0 + (size_t)(t) * slab_bytes)
#define SYN_PATH(t) (path0 + (size_t)((t) >> 6) * PATH_ENTRIES + (size_t)((t) & 63))
#define SYN_REC(t, f) trec[(f) * PoolGeom::M_MAX + (t)]

    // ---- every tree of the pool takes its first job
    int n_ready = 0, n_leaf = 0;
    {
        uint32_t s[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int t = lane + 64 * h;
            s[h] = PT_DEAD;
            if (t < M) {
                LaneTree T;
                lane_start_job<MODE>(P, T);
                SYN_REC(t, PF_RMY0) = (uint32_t)T.root_my; SYN_REC(t, PF_RMY1) = (uint32_t)(T.root_my >> 32);
                SYN_REC(t, PF_ROP0) = (uint32_t)T.root_op; SYN_REC(t, PF_ROP1) = (uint32_t)(T.root_op >> 32);
                SYN_REC(t, PF_ALLOC) = 0u; SYN_REC(t, PF_ITER) = 0u; SYN_REC(t, PF_JOB) = (uint32_t)T.job;
                SYN_REC(t, PF_TURN) = 0u; SYN_REC(t, PF_DRAWS) = 0u;
                SYN_BG(t, 0) = (uint32_t)T.root_my; SYN_BG(t, 1) = (uint32_t)(T.root_my >> 32);
                SYN_BG(t, 2) = (uint32_t)T.root_op; SYN_BG(t, 3) = (uint32_t)(T.root_op >> 32);
                SYN_BG(t, 4) = 0x80000000u;
                s[h] = T.job >= 0 ? PT_READY : PT_DEAD;
            }
            st8[t] = (unsigned char)s[h];
        }
        n_ready = __popcll(__ballot(s[0] == PT_READY)) + __popcll(__ballot(s[1] == PT_READY));
    }
    __syncthreads();  // the only workgroup barrier: weights staged. From here on every wave free-runs.

    const int n_explores = P.roll.num_explores;
    const int fire = P.debug_prio;        // a round fires once this many trees are LEAF (whole tiles: 64 by default)
    const int scan_min = P.lane_thresh;   // Fpu::Func: a scan iteration is taken once this many bound lanes wait for draws
    unsigned long long cache_hits = 0, cache_misses = 0;
    // the descent this lane is walking (tree < 0: none)
    int tree = -1;
    LaneCursor C;
    C.rec = REC_ROOT; C.blk = 0; C.kind = 0; C.qt = 0; C.nsolved = false; C.pN = 0.0f; C.my = 0; C.op = 0; C.level = 0;
    uint32_t lm = 0;
    LaneTree Td;   // what lane_descend_level reads of a tree: its slab and (Fpu::Func) its scan counter
    Td.slab = slab0; Td.fpu_draws = 0;
    Td.root_my = Td.root_op = 0; Td.next_block = Td.num_nodes = 0; Td.iter = 0; Td.root_solved = false; Td.root_sol = 0; Td.job = -1;
    Td.turn = 0; Td.rng_index = 0;
    uint4* pl = path0 + lane;
    uint64_t noise_seed = 0;
    bool fwait = false;
    int sel_flip = 0;

    for (;;) {
        // =============================================================================================== phase A
        for (;;) {
            // (1) arrivals: explore() stops on a solved node or on one that visit() has to expand (mcts.rs:314-320)
            const bool arrived = tree >= 0 && (C.nsolved || C.blk == 0u);
            if (arrived) {
                SYN_REC(tree, PF_REC) = C.rec;
                SYN_REC(tree, PF_META) = C.blk | (C.kind << 14) | ((C.nsolved ? 1u : 0u) << 16) | ((uint32_t)C.level << 17);
                SYN_REC(tree, PF_QT) = C.qt;
                SYN_REC(tree, PF_PN) = f32_bits(C.pN);
                SYN_REC(tree, PF_MY0) = (uint32_t)C.my; SYN_REC(tree, PF_MY1) = (uint32_t)(C.my >> 32);
                SYN_REC(tree, PF_OP0) = (uint32_t)C.op; SYN_REC(tree, PF_OP1) = (uint32_t)(C.op >> 32);
                if (FAST == 2) SYN_REC(tree, PF_DRAWS) = Td.fpu_draws;
                st8[tree] = (unsigned char)PT_LEAF;
                tree = -1;
            }
            n_leaf += __popcll(__ballot(arrived));
            // (2) a round fires on 64 leaves, or when nothing can move any more
            const unsigned long long freem = __ballot(tree < 0);
            if (n_leaf >= fire || (freem == ~0ull && n_ready == 0)) break;
            // (3) free lanes bind READY trees: the r-th free lane takes the r-th READY tree and starts its explore at the root
            if (freem != 0ull && n_ready > 0) {
                pool_lds_sync();
                const uint32_t s0 = st8[lane], s1 = st8[lane + 64];
                const unsigned long long r0 = __ballot(s0 == PT_READY), r1 = __ballot(s1 == PT_READY);
                const int nfree = __popcll(freem);
                const int k0 = pool_rank(r0), k1 = __popcll(r0) + pool_rank(r1);
                if (s0 == PT_READY && k0 < nfree) { idxw[k0] = (unsigned char)lane; st8[lane] = (unsigned char)PT_BOUND; }
                if (s1 == PT_READY && k1 < nfree) { idxw[k1] = (unsigned char)(lane + 64); st8[lane + 64] = (unsigned char)PT_BOUND; }
                pool_lds_sync();
                const int nr = __popcll(r0) + __popcll(r1);
                const int take = nfree < nr ? nfree : nr;
                const int fr = pool_rank(freem);
                if (tree < 0 && fr < take) {
                    tree = (int)idxw[fr];
                    Td.slab = SYN_SLAB(tree);
                    pl = SYN_PATH(tree);
                    if (FAST == 2) {   // (global loads, in flight while the root's line is fetched: used by the level's draws only)
                        const uint32_t job = SYN_REC(tree, PF_JOB), turn = SYN_REC(tree, PF_TURN) & 0xFFu;
                        Td.fpu_draws = SYN_REC(tree, PF_DRAWS);
                        noise_seed = noise_tree_seed(P.base_seed + (MODE == MODE_SELFPLAY ? P.first_game : 0ull) + (uint64_t)job,
                                                     MODE == MODE_SELFPLAY ? turn : 0u);
                        fwait = false;
                    }
                    const uint32_t it = SYN_BG(tree, 4);
                    C.my = (uint64_t)SYN_BG(tree, 0) | ((uint64_t)SYN_BG(tree, 1) << 32);
                    C.op = (uint64_t)SYN_BG(tree, 2) | ((uint64_t)SYN_BG(tree, 3) << 32);
                    C.rec = REC_ROOT;
                    C.level = 0;
                    C.nsolved = false;
                    C.kind = 0;
                    C.qt = 0;
                    // MCTS::with_capacity pushes the root unexpanded (mcts.rs:125): a fresh tree arrives at level 0 (its first
                    // pass is the root's own visit); afterwards the root's block is block 1 and its N the pass counter
                    const bool fresh = (it >> 31) != 0u;
                    C.blk = fresh ? 0u : 1u;
                    C.pN = fresh ? 0.0f : (float)(it & 0x3FFFu);
                    lm = legal_mask_of(C.my | C.op);
                    pl[0] = make_uint4(REC_ROOT, f32_bits(C.pN), pm_make(C.blk, (uint32_t)__popc(lm), false, 0) | (C.blk != 0u ? PM_HAS_W : 0u), 0u);
                    if (COUNT) ctr[CTR_EXPLORES]++;
                }
                n_ready = nr - take;
            }
            // (4) one level (= one cache line) for every bound lane whose node is to be descended through (mcts.rs:322)
            const bool can = tree >= 0 && !C.nsolved && C.blk != 0u;
            if (FAST == 2) {
                // Fpu::Func: lanes whose node needs draws wait; a scan iteration (one noise_fpu_scan for everybody who needs draws,
                // then the level) is taken once `scan_min` lanes wait or nobody can take a level without draws
                const unsigned long long wm = __ballot(can && fwait), dm = __ballot(can && !fwait);
                const bool scan_now = dm == 0ull || __popcll(wm) >= scan_min;
                if (can && (scan_now || !fwait)) {
                    FpuHold H;
                    H.wait = false;
                    lane_descend_level<COUNT, FAST, true>(P.mcts, Td, C, lm, pl, ctr, noise_seed, &H, scan_now, nullptr);
                    fwait = H.wait;
                }
            } else {
                if (can) lane_descend_level<COUNT, FAST>(P.mcts, Td, C, lm, pl, ctr, 0ull, nullptr, true, nullptr);
            }
        }
        if (n_leaf == 0) break;   // nothing bound, nothing READY, nothing LEAF: every job of the launch is done

        // =============================================================================================== the round's trees
        // up to 64 LEAF trees onto the lanes; the half of the pool that is ranked first alternates
        pool_lds_sync();
        int bt;
        int nsel;
        {
            const uint32_t s0 = st8[lane], s1 = st8[lane + 64];
            const bool m0 = s0 == PT_LEAF, m1 = s1 == PT_LEAF;
            const bool ma = sel_flip ? m1 : m0, mb = sel_flip ? m0 : m1;
            const int ta = sel_flip ? lane + 64 : lane, tb = sel_flip ? lane : lane + 64;
            const unsigned long long a = __ballot(ma), b = __ballot(mb);
            const int ka = pool_rank(a), kb = __popcll(a) + pool_rank(b);
            if (ma && ka < 64) idxw[ka] = (unsigned char)ta;
            if (mb && kb < 64) idxw[kb] = (unsigned char)tb;
            const int total = __popcll(a) + __popcll(b);   // == n_leaf
            nsel = total < 64 ? total : 64;
            sel_flip ^= 1;
            pool_lds_sync();
            bt = lane < nsel ? (int)idxw[lane] : -1;
        }
        const bool active = bt >= 0;
        // this lane's own descent waits in LDS
        pk[0] = C.rec;
        pk[64] = C.blk | (C.kind << 14) | ((C.nsolved ? 1u : 0u) << 16) | ((uint32_t)C.level << 17) | ((uint32_t)(tree & 127) << 24) | (tree >= 0 ? 0x80000000u : 0u);
        pk[2 * 64] = C.qt;
        pk[3 * 64] = f32_bits(C.pN);
        pk[4 * 64] = (uint32_t)C.my; pk[5 * 64] = (uint32_t)(C.my >> 32);
        pk[6 * 64] = (uint32_t)C.op; pk[7 * 64] = (uint32_t)(C.op >> 32);
        if (FAST == 2) {
            pk[8 * 64] = Td.fpu_draws | (fwait ? 0x80000000u : 0u);
            pk[9 * 64] = (uint32_t)noise_seed; pk[10 * 64] = (uint32_t)(noise_seed >> 32);
        }
        // the leaf's tree and cursor
        LaneTree T;
        LaneCursor Cx;
        const int btz = active ? bt : 0;
        T.slab = SYN_SLAB(btz);
        uint4* const plx = SYN_PATH(btz);
        T.root_my = T.root_op = 0; T.turn = 0; T.rng_index = 0; T.fpu_draws = 0;
        T.next_block = 0; T.num_nodes = 0; T.iter = 0; T.root_solved = false; T.root_sol = 0; T.job = -1;
        Cx.rec = REC_ROOT; Cx.blk = 0; Cx.kind = 0; Cx.qt = 0; Cx.nsolved = false; Cx.pN = 0.0f; Cx.my = 0; Cx.op = 0; Cx.level = 0;
        LaneLeaf X;
        X.at_leaf = active; X.was_pending = false; X.needs_eval = false; X.solved = false; X.legal_mask = 0;
        X.p0 = X.p1 = X.p2 = 0.0f;
        if (active) {
            const uint32_t alloc = SYN_REC(bt, PF_ALLOC), it = SYN_REC(bt, PF_ITER), meta = SYN_REC(bt, PF_META);
            T.job = (int)SYN_REC(bt, PF_JOB);
            Cx.rec = SYN_REC(bt, PF_REC);
            Cx.qt = SYN_REC(bt, PF_QT);
            Cx.pN = bits_f32(SYN_REC(bt, PF_PN));
            Cx.my = (uint64_t)SYN_REC(bt, PF_MY0) | ((uint64_t)SYN_REC(bt, PF_MY1) << 32);
            Cx.op = (uint64_t)SYN_REC(bt, PF_OP0) | ((uint64_t)SYN_REC(bt, PF_OP1) << 32);
            if (FAST == 2) T.fpu_draws = SYN_REC(bt, PF_DRAWS);
            T.next_block = alloc & 0xFFFFu;
            T.num_nodes = alloc >> 16;
            T.iter = (int)(it & 0x3FFFu);
            T.root_solved = ((it >> 14) & 1u) != 0u;
            T.root_sol = it >> 15;
            Cx.blk = meta & 0x3FFFu;
            Cx.kind = (meta >> 14) & 3u;
            Cx.nsolved = ((meta >> 16) & 1u) != 0u;
            Cx.level = (int)((meta >> 17) & 0x7Fu);
            if (T.next_block == 0u) { T.next_block = 1u; T.num_nodes = 1u; }   // (the root was pushed when the explore began)
            lane_arrive<COUNT, FAST>(P.mcts, T, Cx, X, Cx.nsolved, plx, bcap, ctr, P.error);
        }

        // =============================================================================================== phase B
        const bool want_nn = active && X.needs_eval;
        float lg[9];
        float v0 = X.p0, v1 = X.p1, v2 = X.p2;
#pragma unroll
        for (int c = 0; c < 9; c++) lg[c] = 0.0f;
        bool hit = false;
        if (P.cache != nullptr && want_nn) hit = cache_lookup(P.cache, P.cache_shift, Cx.my, Cx.op, lg, v0, v1, v2);
        const bool need = want_nn && !hit;
        const unsigned long long need_mask = __ballot(need);
        if (P.cache != nullptr) {
            cache_hits += (unsigned long long)__popcll(__ballot(hit));
            cache_misses += (unsigned long long)__popcll(need_mask);
        }
        const int rank = pool_rank(need_mask);
        if (COUNT && (need || hit)) ctr[CTR_POLICY_EVALS]++;
        const int n_need = __popcll(need_mask);
        idxw[lane] = 0;
        pool_lds_sync();
        if (need) idxw[rank] = (unsigned char)lane;
        pool_lds_sync();
#pragma unroll 1
        for (int j = 0; j * 16 < n_need; j++) {
            const int src = (int)idxw[16 * j + (lane & 15)];  // (slots past the last request read lane 0: finite input)
            f32x4 o;
            uint64_t hi, lo;
            feature_boards(Cx.my, Cx.op, hi, lo);
            const uint64_t thi = shfl_u64(hi, src), tlo = shfl_u64(lo, src);
            if (POLICY == 3) {
                uint32_t img_off = 0;  // opaque per tile: the image reads stay LDS reads next to their MFMAs
                asm volatile("" : "+v"(img_off));
                const uint32_t* img16 = reinterpret_cast<const uint32_t*>(smem_raw) + img_off;
                o = f16x2_tile16<3>(img16, lane, thi, tlo);
                const float os = reinterpret_cast<const float*>(img16 + F16Geom::SCALE_WORD0)[4];   // exact power of two
#pragma unroll
                for (int r = 0; r < 4; r++) o[r] *= os;
            } else {
                const uint4 ftw = *reinterpret_cast<const uint4*>(smem_raw + L::FT_OFF + (lane >> 4) * 16);
                FeatureTable FT;
                FT.t[0] = ftw.x; FT.t[1] = ftw.y; FT.t[2] = ftw.z; FT.t[3] = ftw.w;
                o = mlp_tile16_pipe(wimg, bimg, lane, FT, thi, tlo);
            }
            // every lane fetches "its" twelve outputs through the LDS crossbar (lane_kernel.cuh, phase B)
            const int pos4 = (rank & 15) << 2;
            float t12[12];
#pragma unroll
            for (int qq = 0; qq < 3; qq++)
#pragma unroll
                for (int k = 0; k < 4; k++)
                    t12[4 * qq + k] = bits_f32((uint32_t)__builtin_amdgcn_ds_bpermute(pos4 + 64 * qq, (int)f32_bits(o[k])));
            const bool mine = need && (rank >> 4) == j;
#pragma unroll
            for (int c = 0; c < 9; c++) lg[c] = mine ? t12[c] : lg[c];
            v0 = mine ? t12[9] : v0;
            v1 = mine ? t12[10] : v1;
            v2 = mine ? t12[11] : v2;
        }

        // =============================================================================================== phase C
        bool solved = X.solved;
        uint32_t leaf_flag = 0, leaf_code = 0;
        if (need || hit) {
            float pr[9];
            lane_softmaxes(X.legal_mask, lg, pr, need, v0, v1, v2);
            if (P.cache != nullptr && need) cache_insert(P.cache, P.cache_shift, Cx.my, Cx.op, lg, v0, v1, v2);
            const CfgView<FAST> cv{P.mcts};
            leaf_code = lane_write_children(T.slab, Cx.blk, X.legal_mask, Cx.my, Cx.op, pr, 0, P.mcts.noise_weight, P.mcts.noise_alpha, 0ull,
                                            cv.fpu_const() ? cv.fpu_value() : 0.0f, leaf_flag, false);
            solved = (leaf_code & LEAF_ANY_SOLVED) != 0u;
        }
        lane_backprop<COUNT, FAST, true>(P.mcts, T, Cx.level, v0, v1, v2, solved, active, plx, ctr, leaf_flag, nullptr, nullptr, leaf_code);
        bool alive = false;
        if (active) {
            T.iter += 1;
            // explore_n (mcts.rs:139-147): the root visit, then up to n explores unless the root is solved
            if (T.iter > n_explores || T.root_solved) {
                const KernargPtr Pc = lane_kernarg();
                T.root_my = (uint64_t)SYN_REC(bt, PF_RMY0) | ((uint64_t)SYN_REC(bt, PF_RMY1) << 32);
                T.root_op = (uint64_t)SYN_REC(bt, PF_ROP0) | ((uint64_t)SYN_REC(bt, PF_ROP1) << 32);
                const uint32_t tr = SYN_REC(bt, PF_TURN);
                T.turn = (int)(tr & 0xFFu);
                T.rng_index = tr >> 8;
                if (MODE == MODE_SELFPLAY) T = lane_move_step_call<COUNT>(Pc, T, ctr);
                else T = lane_search_finish_call(Pc, T);
                SYN_REC(bt, PF_RMY0) = (uint32_t)T.root_my; SYN_REC(bt, PF_RMY1) = (uint32_t)(T.root_my >> 32);
                SYN_REC(bt, PF_ROP0) = (uint32_t)T.root_op; SYN_REC(bt, PF_ROP1) = (uint32_t)(T.root_op >> 32);
                SYN_REC(bt, PF_TURN) = (uint32_t)T.turn | (T.rng_index << 8);
                SYN_REC(bt, PF_JOB) = (uint32_t)T.job;
                SYN_BG(bt, 0) = (uint32_t)T.root_my; SYN_BG(bt, 1) = (uint32_t)(T.root_my >> 32);
                SYN_BG(bt, 2) = (uint32_t)T.root_op; SYN_BG(bt, 3) = (uint32_t)(T.root_op >> 32);
            }
            SYN_BG(bt, 4) = (uint32_t)T.iter | (T.next_block == 0u ? 0x80000000u : 0u);
            SYN_REC(bt, PF_ALLOC) = T.next_block | (T.num_nodes << 16);
            SYN_REC(bt, PF_ITER) = (uint32_t)T.iter | ((T.root_solved ? 1u : 0u) << 14) | (T.root_sol << 15);
            if (FAST == 2) SYN_REC(bt, PF_DRAWS) = T.fpu_draws;
            alive = T.job >= 0;
            st8[bt] = (unsigned char)(alive ? PT_READY : PT_DEAD);
        }
        n_leaf -= nsel;
        n_ready += __popcll(__ballot(alive));
        // back to this lane's own descent
        {
            const uint32_t meta = pk[64];
            C.rec = pk[0];
            C.blk = meta & 0x3FFFu;
            C.kind = (meta >> 14) & 3u;
            C.nsolved = ((meta >> 16) & 1u) != 0u;
            C.level = (int)((meta >> 17) & 0x7Fu);
            tree = (meta >> 31) != 0u ? (int)((meta >> 24) & 127u) : -1;
            C.qt = pk[2 * 64];
            C.pN = bits_f32(pk[3 * 64]);
            C.my = (uint64_t)pk[4 * 64] | ((uint64_t)pk[5 * 64] << 32);
            C.op = (uint64_t)pk[6 * 64] | ((uint64_t)pk[7 * 64] << 32);
            lm = legal_mask_of(C.my | C.op);
            const int tz = tree >= 0 ? tree : 0;
            Td.slab = SYN_SLAB(tz);
            pl = SYN_PATH(tz);
            if (FAST == 2) {
                const uint32_t d = pk[8 * 64];
                Td.fpu_draws = d & 0x7FFFFFFFu;
                fwait = (d >> 31) != 0u;
                noise_seed = (uint64_t)pk[9 * 64] | ((uint64_t)pk[10 * 64] << 32);
            }
        }
    }
#undef SYN_SLAB
#undef SYN_PATH
#undef SYN_REC
#undef SYN_BG

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

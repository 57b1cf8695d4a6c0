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
//   phase A (per lane)   select + expand (one 128-byte block = one cache line per level)
//   phase B (per wave)   up to four mlp_tile16 evaluations; leaf boards reach the tile layout by ds_bpermute, the 12
//                        outputs per position return through a 1 KB per-wave LDS patch
//   phase C (per lane)   legal softmax -> children records, backprop replayed from the path log, move step when the
//                        search is over
#pragma once
#include "engine_kernels.cuh"
#include "noise.cuh"
#include "convnet.cuh"
#include "f16x2_tile.cuh"

namespace syn {

// Node storage of this launch shape (private to it: the pool's contents never outlive a launch): 128-byte BLOCKS, each
// exactly one aligned cache line — the unit the memory system charges for (tools/ubench/gather_blocks.hip: 49 G random
// lines/s whether a visit needs 16 or 128 bytes of the line; a 144-byte children block costs two). A block belongs to one
// expanded node (its owner) and holds everything a visit of that node touches:
//   bytes   0..11   W_lose, W_draw, W_win of the owner         (read-modify-written by backprop)
//   bytes  12..15   spare
//   bytes 16..123   nine 12-byte child records { q | turns, P, N[0:14] solved[15] | block[16:29] kind[30:31] }
//       q      = exploit value as select_best_child will see it (mcts.rs:343-359): -((W_win - W_lose) / N) rewritten by
//                every backprop; the FPU constant while unvisited (Fpu::ParentQ is patched in by the descent); once the
//                child is solved the slot holds the solution's turn count and the value comes from `kind`
//       block  = the child's own block (0 = not expanded); N counts visits (<= 32,767)
// So one level of the descent is ONE line (seven 16-byte loads of the record area), and one level of backprop is one line
// read (the owner's sums, skipped for a fresh node) plus two dirtied lines (own header, own record in the parent's block).
// Children are the legal columns of the owner's position in ascending order, so their count and their actions are derived
// from the board the descent carries anyway. A node is named by its record (parent block * 16 + slot); the root has no
// record (its N is the pass counter). There are no parent links: the descent logs {record, N, block|count|solution,
// q|turns} per level into a per-wave path buffer [level][lane] (coalesced 1 KB rows) and backprop replays it.
constexpr uint32_t LANE_MAX_CAP = 1u << 16;   // blocks per tree = cap / 4 must fit the 14-bit block field
constexpr uint32_t REC_ROOT = 0xFFFFFFFFu;
// Path buffer of one wave: two planes of [level 0..63][lane 0..63] 16-byte entries. Plane 0 = {record, N, block|count|
// solution (pm_*), q|turns}; plane 1 = the node's own sums {W_lose, W_draw, W_win} as the descent saw them in the line it
// read anyway (flag PM_HAS_W in plane 0's .z). Backprop then needs NO random line read for a level it only adds to: a
// dirtied line costs the memory system one more line transaction, a re-read line another (tools/ubench/rw_lines.hip:
// ~50 G random line reads/s, ~23 G read+dirty visits/s) — the re-reads were 3.4 of 15 line transactions per explore.
constexpr uint32_t PATH_PLANE = 4096;         // entries per plane
constexpr uint32_t PATH_ENTRIES = 2 * PATH_PLANE;
constexpr uint32_t PM_HAS_W = 1u << 21;

SYN_DEV uint32_t nf_make(uint32_t n, bool solved, uint32_t blk, uint32_t kind) {
    return n | ((solved ? 1u : 0u) << 15) | (blk << 16) | (kind << 30);
}
SYN_DEV float nf_N(uint32_t nf) { return (float)(nf & 0x7FFFu); }
SYN_DEV bool nf_solved(uint32_t nf) { return ((nf >> 15) & 1u) != 0u; }
SYN_DEV uint32_t nf_blk(uint32_t nf) { return (nf >> 16) & 0x3FFFu; }
SYN_DEV uint32_t nf_kind(uint32_t nf) { return nf >> 30; }
// path entry .z: block[0:13] | num_children[14:17] | solved[18] | kind[19:20]
SYN_DEV uint32_t pm_make(uint32_t blk, uint32_t nc, bool solved, uint32_t kind) {
    return blk | (nc << 14) | ((solved ? 1u : 0u) << 18) | (kind << 19);
}
SYN_DEV uint32_t pm_blk(uint32_t m) { return m & 0x3FFFu; }
SYN_DEV uint32_t pm_nc(uint32_t m) { return (m >> 14) & 0xFu; }
SYN_DEV bool pm_solved(uint32_t m) { return ((m >> 18) & 1u) != 0u; }
SYN_DEV uint32_t pm_kind(uint32_t m) { return (m >> 19) & 3u; }
// exploit_value of a solved child (mcts.rs:343-350): outcome.reversed().value() — child Win -> -1, Draw -> 0,
// Lose -> +1 (game.rs:29-43) — or -inf when solved nodes are not to be selected
SYN_DEV float pw_q_solved(uint32_t kind, bool select_solved) {
    return select_solved ? 1.0f - (float)kind : -__builtin_inff();  // kind 0 Lose -> +1, 1 Draw -> 0, 2 Win -> -1 (exact)
}
SYN_DEV unsigned char* blk_ptr(unsigned char* slab, uint32_t b) { return slab + (size_t)b * 128u; }
SYN_DEV unsigned char* rec_ptr(unsigned char* slab, uint32_t rec) {
    return slab + (size_t)(rec >> 4) * 128u + 16u + (rec & 15u) * 12u;
}
SYN_DEV uint32_t legal_mask_of(uint64_t occ) { return c4::legal_columns(occ); }
typedef uint32_t lu3 __attribute__((ext_vector_type(3)));
SYN_DEV void st_rec(unsigned char* slab, uint32_t rec, uint32_t qt, float P, uint32_t nf) {
    *reinterpret_cast<lu3*>(rec_ptr(slab, rec)) = lu3{qt, f32_bits(P), nf};
}
// Records of a block's unused slots (a node with fewer than nine legal columns): exploit value -inf, prior 0, no visits — the
// descent scans all nine slots without a count check and such a slot can never win the strict `>` of select_best_child.
SYN_DEV void st_rec_none(unsigned char* slab, uint32_t rec) { st_rec(slab, rec, 0xFF800000u, 0.0f, 0u); }
SYN_DEV lu3 ld_rec(unsigned char* slab, uint32_t rec) { return *reinterpret_cast<const lu3*>(rec_ptr(slab, rec)); }

struct LaneTree {
    unsigned char* slab;      // this lane's blocks
    uint64_t root_my, root_op;
    uint32_t next_block;      // 0 = fresh tree; block ids start at 1, the root's block is always 1
    uint32_t num_nodes;       // nodes.len() of the reference tree (the root + every child record created)
    int iter;                 // passes done on this tree (root visit = 1) == root.num_visits
    bool root_solved;
    uint32_t root_sol;        // kind | turns << 2 of the root's solution
    int job;                  // game / root index, -1 = idle
    int turn;
    uint32_t rng_index;
    uint32_t fpu_draws;       // Fpu::Func Normal draws this tree has taken (noise.cuh); 0 at every new root
};

// Descent state of a lane. It persists across rounds: a round ends as soon as `thresh` lanes of the wave stand on a
// leaf (or nobody is descending any more), the lanes that are still on their way down simply continue in the next
// round. So a wave never idles 63 lanes while its deepest tree finishes, and the network runs on (nearly) full tiles.
struct LaneWalk {
    bool descending;
    bool pending;             // stands on an expanded leaf whose network call did not fit this round's full tiles
    uint32_t pend_lmask;
    uint32_t rec, blk;        // current node: its record and its own block (0 = none)
    bool solved;
    uint32_t kind;
    uint32_t qt;              // its q slot (q bits, or turns when solved)
    float pN;                 // its N
    uint64_t my, op;          // its position
    int level;
};

struct LaneLeaf {             // phase A -> phase C (valid for lanes with at_leaf)
    bool at_leaf;             // this lane finished its descent in this round
    bool was_pending;         // ... in an earlier round (its position already missed the policy cache)
    uint32_t legal_mask;      // legal columns of the node whose children are to be created (valid if needs_eval)
    bool needs_eval, solved;
    float p0, p1, p2;
};

// Diagnostic build only (SYN_DEBUG=1 SYN_PROFILE=1, template parameter PROF): cycle stamps INSIDE the phases. "wait" = from the
// issue of a step's loads to their arrival (an explicit s_waitcnt vmcnt(0) that the production build does not have), "alu" = the
// rest of the step (arithmetic, stores, control flow). The stamps sit inside divergent code, so the first active lane adds the
// (wave-uniform) interval to this wave's row of P.prof with one no-return atomic. nullptr everywhere else: it all compiles away.
enum { LP_A_WAIT = 10, LP_A_ALU, LP_A_ITERS, LP_A_LANES, LP_A_ARRIVE, LP_W_WAIT, LP_W_ALU, LP_W_ITERS, LP_W_LANES, LP_S_WAIT, LP_S_ALU,
       LP_S_STEPS, LP_S_LANES, LP_C_SOFT, LP_C_WRITE, LP_C_LANES, LP_B_GATHER, LP_B_TILE, LP_B_SCATTER, LP_M_CALLS, LP_M_LANES, LP_M_TOTAL, LP_M_T1, LP_M_T2, LP_M_T3, LP_M_T4, LP_M_T5, LP_FIELDS };
// `ablate` (SYN_DEBUG=1 SYN_PROFILE=1 SYN_ABLATE=<mask>): additive sensitivity runs — a component is executed TWICE (same addresses,
// same values: results unchanged) and the launch's slowdown is that component's marginal cost in the real, contended kernel; the
// stamps are off in such a run. 1 child-record stores, 2 backprop sweep stores, 4 path-log stores of the descent, 8 the descent's
// line loads, 16 every network tile, 32 the leaf softmaxes, 64 the solver walk's line loads.
enum { ABL_CHILD_ST = 1, ABL_SWEEP_ST = 2, ABL_LOG_ST = 4, ABL_LINE_LD = 8, ABL_TILE = 16, ABL_SOFTMAX = 32, ABL_WALK_LD = 64 };
struct LaneProf { unsigned long long* row; int ablate; };
SYN_DEV bool lp_abl(const LaneProf* lp, int bit) { return lp != nullptr && (lp->ablate & bit) != 0; }
SYN_DEV void lp_fence() { asm volatile("" ::: "memory"); }
constexpr size_t PROF_TIMELINE_OFF = (size_t)4096 * LP_FIELDS;  // behind the rows of up to 4096 waves
SYN_DEV unsigned long long lp_now() { return (unsigned long long)__builtin_readcyclecounter(); }
SYN_DEV void lp_wait_vm(const LaneProf* lp) { if (lp->ablate == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
SYN_DEV void lp_add(const LaneProf* lp, int field, unsigned long long v) {
    if (lp->ablate != 0) return;
    const unsigned long long ex = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)ex) - 1) atomicAdd(lp->row + field, v);
}
// one interval + how many lanes shared it
SYN_DEV void lp_step(const LaneProf* lp, int f_time, int f_count, int f_lanes, unsigned long long dt) {
    if (lp->ablate != 0) return;
    const unsigned long long ex = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)ex) - 1) {
        atomicAdd(lp->row + f_time, dt);
        if (f_count >= 0) atomicAdd(lp->row + f_count, 1ull);
        if (f_lanes >= 0) atomicAdd(lp->row + f_lanes, (unsigned long long)__popcll(ex));
    }
}

template <int MODE>
SYN_DEV void lane_start_job(const EngineParams& P, LaneTree& T) {
    int j = atomicAdd(P.job_next, 1);
    T.job = j < P.n_jobs ? j : -1;
    T.turn = 0;
    T.rng_index = 0;
    T.next_block = 0;
    T.num_nodes = 0;
    T.iter = 0;
    T.root_solved = false;
    T.root_sol = 0;
    T.root_my = 0;
    T.root_op = 0;
    T.fpu_draws = 0;
    if (MODE == MODE_SEARCH && T.job >= 0) {
        T.root_my = P.in_my[T.job];
        T.root_op = P.in_op[T.job];
    }
}

SYN_DEV uint32_t lane_alloc_block(LaneTree& T, uint32_t bcap, int* error) {
    uint32_t b = T.next_block;
    T.next_block = b + 1u;
    if (b >= bcap) {  // cannot happen for explores <= max_explores (engine.hip sizes the slab); never write outside it
        *error = 2;
        b = bcap - 1u;
    }
    return b;
}

// Fpu::Func in the one-tree-per-lane descent loop: a lane whose node needs draws waits (`wait`, with the slots in `need`) until no
// lane of the wave can take a level without them; then ONE scan serves all waiting lanes (lane_select_expand). A draw is a function
// of (tree, scan index, slot) only, so when it is computed does not matter.
struct FpuHold {
    bool wait;
};

// ---------------------------------------------------------------------------------------------- phase A
// The node a descent stands on: its record and own block (0 = none), solution, q slot, N and position.
struct LaneCursor {
    uint32_t rec, blk, kind, qt;
    bool nsolved;
    float pN;
    uint64_t my, op;
    int level;
};

// explore() starts at the root (mcts.rs:310-312). pl = this lane's column of the wave's path buffer: level L lives at
// pl[L * 64]. ROOT_IN_T: the root position is live in T.root_my / T.root_op (pc_kernel.cuh) instead of parked in LDS at pk[].
template <bool COUNT, int FAST, bool ROOT_IN_T>
SYN_DEV void lane_begin_explore(const DevMctsCfg& cfg_, LaneTree& T, LaneCursor& C, uint4* pl, uint32_t* ctr, const uint32_t* pk,
                                int pk_stride) {
    const CfgView<FAST> cfg{cfg_};
    if (COUNT) ctr[CTR_EXPLORES]++;
    C.rec = REC_ROOT;
    C.level = 0;
    // the root position is parked in LDS between searches (it is only needed here and at the end of a search)
    if (ROOT_IN_T) {
        C.my = T.root_my;
        C.op = T.root_op;
    } else {
        C.my = (uint64_t)pk[0] | ((uint64_t)pk[pk_stride] << 32);
        C.op = (uint64_t)pk[2 * pk_stride] | ((uint64_t)pk[3 * pk_stride] << 32);
    }
    C.nsolved = false;
    C.kind = 0;
    C.qt = 0;
    if (T.next_block == 0) {
        T.next_block = 1;  // MCTS::with_capacity pushes the root (mcts.rs:125): not expanded yet, no block
        T.num_nodes = 1;
        C.blk = 0;
        C.pN = 0.0f;
    } else {
        C.blk = 1;           // the root's block
        C.pN = (float)T.iter;
        if (!cfg.fpu_const() && !cfg.fpu_normal()) {  // Fpu::ParentQ at the first level needs the root's q
            const float4 a = *reinterpret_cast<const float4*>(blk_ptr(T.slab, 1));
            C.qt = f32_bits(-((a.z - a.x) / C.pN));
        }
    }
    pl[0] = make_uint4(REC_ROOT, f32_bits(C.pN),
                       pm_make(C.blk, (uint32_t)__popc(legal_mask_of(C.my | C.op)), false, 0) | (C.blk != 0u ? PM_HAS_W : 0u), 0u);
}

// One level of the descent (mcts.rs:310-341: select_best_child + the step into the chosen child) = one cache line.
// lm = legal columns of the cursor's position (updated as the descent drops stones: a column leaves the mask when its seventh
// stone lands); the children of a node are its legal columns in ascending order.
// DEFER (lane_select_expand): with `scan_now` false a lane that needs draws only records that in H and returns; the caller's
// next call with scan_now true (wave-uniform) takes the draws and the level.
template <bool COUNT, int FAST, bool DEFER = false>
SYN_DEV void lane_descend_level(const DevMctsCfg& cfg_, LaneTree& T, LaneCursor& C, uint32_t& lm, uint4* pl, uint32_t* ctr,
                                uint64_t noise_seed, FpuHold* H = nullptr, bool scan_now = true, LaneProf* lp = nullptr,
                                const float* fpu_tbl = FPU_NORMAL_TABLE) {
    const CfgView<FAST> cfg{cfg_};
    unsigned char* const slab = T.slab;
    const uint32_t nc = (uint32_t)__popc(lm);
    const uint4* line = reinterpret_cast<const uint4*>(blk_ptr(slab, C.blk));
    const unsigned long long lp_t0 = lp ? lp_now() : 0ull;
    const uint4 hdr = line[0];  // the node's own sums: logged for backprop (same line, no extra transaction)
    uint32_t d[28];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        const uint4 t = line[1 + j];
        d[4 * j] = t.x; d[4 * j + 1] = t.y; d[4 * j + 2] = t.z; d[4 * j + 3] = t.w;
    }
    if (lp_abl(lp, ABL_LINE_LD)) {
        lp_fence();
        uint32_t sink = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint4 t = line[j];
            sink ^= t.x ^ t.y ^ t.z ^ t.w;
        }
        asm volatile("" ::"v"(sink));
    }
    unsigned long long lp_t1 = 0ull;
    if (lp) { lp_wait_vm(lp); lp_t1 = lp_now(); lp_add(lp, LP_A_WAIT, lp_t1 - lp_t0); }
    const float q_fpu = cfg.fpu_const() ? cfg.fpu_value() : -bits_f32(C.qt);  // parent.q() = -(stored q)
    const float visits = cfg.puct() ? (cfg_.fast_div != 0 ? sqrt_normal_range(C.pN) : sqrtf(C.pN)) : sqrtf(cfg.cc() * det_logf(C.pN));
    // select_best_child: sequential scan, `Some(v) > best` (strict: first maximum wins, NaN never replaces). Slots past
    // the last child hold -inf / prior 0 (st_rec_none), so all nine are scored without a count check.
    // what an unexpanded, unsolved child scores under a non-constant FPU: the parent's q (Fpu::ParentQ), or one
    // Normal(mean, std) draw per such child per scan, in child order (Fpu::Func, mcts.rs:351-356)
    float qf[9];
#pragma unroll
    for (uint32_t i = 0; i < 9; i++) qf[i] = q_fpu;
    if (cfg.fpu_normal()) {
        // T.fpu_draws counts this tree's scans that took a draw (noise.cuh)
        uint32_t need = 0;
#pragma unroll
        for (uint32_t i = 0; i < 9; i++) {
            const uint32_t nf = d[3 * i + 2];
            need |= (i < nc && !nf_solved(nf) && nf_blk(nf) == 0u) ? (1u << i) : 0u;
        }
        if (DEFER && !scan_now) {
            if (need != 0u) {
                H->wait = true;
                return;
            }
        } else if (__ballot(need != 0u) != 0ull) {
            noise_fpu_scan(noise_seed, T.fpu_draws, need, cfg.fpu_value(), cfg_.fpu_std, qf, fpu_tbl);
            T.fpu_draws += need != 0u ? 1u : 0u;
        }
    }
    pl[PATH_PLANE + C.level * 64] = hdr;  // (the entry's PM_HAS_W flag was set when the descent arrived here)
    if (lp_abl(lp, ABL_LOG_ST)) { lp_fence(); pl[PATH_PLANE + C.level * 64] = hdr; lp_fence(); }
    float vv[9];
    if (cfg.puct()) {
        // explore_value = ((c * P) * sqrt(N_parent)) / (1 + n) (mcts.rs:361-372), two children per instruction.
        // hdr.w != 0: this node has a prior outside the packed division's exact range (lane_create_children).
        const bool exact_fast = cfg_.fast_div != 0 && __ballot(hdr.w != 0u) == 0ull;
        f32x2 uu[5];
        if (exact_fast) {
#pragma unroll
            for (uint32_t i = 0; i < 9; i += 2) {
                const uint32_t j = i + 1 < 9 ? i + 1 : i;
                f32x2 a = f32x2{cfg.cc(), cfg.cc()} * f32x2{bits_f32(d[3 * i + 1]), bits_f32(d[3 * j + 1])};
                a = a * f32x2{visits, visits};
                uu[i >> 1] = div2_by_small_int(a, f32x2{1.0f, 1.0f} + f32x2{nf_N(d[3 * i + 2]), nf_N(d[3 * j + 2])});
            }
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 9; i += 2) {
                const uint32_t j = i + 1 < 9 ? i + 1 : i;
                const float a0 = cfg.cc() * bits_f32(d[3 * i + 1]) * visits, a1 = cfg.cc() * bits_f32(d[3 * j + 1]) * visits;
                uu[i >> 1] = f32x2{a0 / (1.0f + nf_N(d[3 * i + 2])), a1 / (1.0f + nf_N(d[3 * j + 2]))};
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < 9; i++) {
            const uint32_t nf = d[3 * i + 2];
            // exploit_value: solved -> its outcome; unvisited under Fpu::ParentQ -> parent's q; else the q slot
            float q = (!cfg.fpu_const() && nf_blk(nf) == 0u) ? qf[i] : bits_f32(d[3 * i]);
            q = nf_solved(nf) ? pw_q_solved(nf_kind(nf), cfg.select_solved()) : q;
            vv[i] = q + uu[i >> 1][i & 1];
        }
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 9; i++) {
            const uint32_t nf = d[3 * i + 2];
            float q = (!cfg.fpu_const() && nf_blk(nf) == 0u) ? qf[i] : bits_f32(d[3 * i]);
            q = nf_solved(nf) ? pw_q_solved(nf_kind(nf), cfg.select_solved()) : q;
            vv[i] = q + visits / sqrtf(nf_N(nf));
        }
    }
    if (!cfg.fpu_const()) {  // Fpu::ParentQ gives an unused slot (no block) the parent's q: mask by the child count instead
#pragma unroll
        for (uint32_t i = 1; i < 9; i++) vv[i] = i < nc ? vv[i] : -__builtin_inff();
    }
    float best_v = vv[0];
    uint32_t best_i = 0, b_qt = d[0], b_nf = d[2], b_m = lm;
    uint32_t m = lm & (lm - 1u);  // child i = the i-th legal column = the lowest set bit of m
#pragma unroll
    for (uint32_t i = 1; i < 9; i++) {
        const bool take = vv[i] > best_v;
        best_v = take ? vv[i] : best_v;
        best_i = take ? i : best_i;
        b_qt = take ? d[3 * i] : b_qt;
        b_nf = take ? d[3 * i + 2] : b_nf;
        b_m = take ? m : b_m;
        m &= m - 1u;
    }
    if (COUNT) { ctr[CTR_SELECT_LEVELS]++; ctr[CTR_CHILDREN_SCANNED] += nc; }
    const int a = __ffs((int)b_m) - 1;
    const int ha = c4::col_height(C.my | C.op, a);
    const uint64_t nmy = C.op, nop = C.my | (1ull << (ha + 7 * a));
    C.my = nmy;
    C.op = nop;
    lm = ha == c4::HEIGHT - 1 ? lm & ~(1u << a) : lm;
    C.rec = C.blk * 16u + best_i;
    C.blk = nf_blk(b_nf);
    C.nsolved = nf_solved(b_nf);
    C.kind = nf_kind(b_nf);
    C.qt = b_qt;
    C.pN = nf_N(b_nf);
    C.level++;
    // an expanded, unsolved node is descended through: its line (and with it its own sums) gets read and logged
    pl[C.level * 64] = make_uint4(C.rec, f32_bits(C.pN),
                                  pm_make(C.blk, (uint32_t)__popc(lm), C.nsolved, C.kind) |
                                      ((!C.nsolved && C.blk != 0u) ? PM_HAS_W : 0u), C.qt);
    // Fpu::Func, deferred scans: a node with N visits has at most N - 1 expanded children, so a child entered with N <= its number
    // of children is certain to score an unexpanded child — it waits for the wave's next scan right away instead of finding that
    // out with a wasted pass over its line (a child whose unexpanded children are all terminal waits for nothing: harmless)
    if (DEFER && cfg.fpu_normal() && H != nullptr && !C.nsolved && C.blk != 0u && C.pN <= (float)__popc(lm)) H->wait = true;
    if (lp_abl(lp, ABL_LOG_ST)) {
        lp_fence();
        pl[C.level * 64] = make_uint4(C.rec, f32_bits(C.pN),
                                      pm_make(C.blk, (uint32_t)__popc(lm), C.nsolved, C.kind) |
                                          ((!C.nsolved && C.blk != 0u) ? PM_HAS_W : 0u), C.qt);
        lp_fence();
    }
    if (lp) lp_step(lp, LP_A_ALU, LP_A_ITERS, LP_A_LANES, lp_now() - lp_t1);
}

// The descent stands on a leaf: a solved node (explore() returns its outcome) or an unexpanded one, which visit()
// (mcts.rs:374-406) gives its block; the children's records are written in phase C together with their priors. Only an
// auto-extended single child is written here (prior 1.0, no policy call). Sets X.{p0,p1,p2,solved,needs_eval,legal_mask}.
template <bool COUNT, int FAST>
SYN_DEV void lane_arrive(const DevMctsCfg& cfg_, LaneTree& T, LaneCursor& C, LaneLeaf& X, bool hit_solved, uint4* pl,
                         uint32_t bcap, uint32_t* ctr, int* error) {
    const CfgView<FAST> cfg{cfg_};
    unsigned char* const slab = T.slab;
    const float y_unvisited = cfg.fpu_const() ? cfg.fpu_value() : 0.0f;
    if (hit_solved) {
        X.p0 = C.kind == 0u ? 1.0f : 0.0f;
        X.p1 = C.kind == 1u ? 1.0f : 0.0f;
        X.p2 = C.kind == 2u ? 1.0f : 0.0f;
        X.solved = true;
        if (COUNT) ctr[CTR_SOLVED_HITS]++;
        if (C.blk == 0u) {
            // first visit of a terminal node: it gets a block for its outcome sums
            C.blk = lane_alloc_block(T, bcap, error);
            *reinterpret_cast<unsigned short*>(rec_ptr(slab, C.rec) + 10) = (unsigned short)(C.blk | (C.kind << 14));
            pl[C.level * 64].z = pm_make(C.blk, 0u, true, C.kind);
        }
    } else {
        for (;;) {
            const uint64_t occ = C.my | C.op;
            const uint32_t lmask = legal_mask_of(occ);
            const uint32_t n_new = (uint32_t)__popc(lmask);
            const uint32_t nb = lane_alloc_block(T, bcap, error);
            T.num_nodes += n_new;
            if (C.rec != REC_ROOT) *reinterpret_cast<unsigned short*>(rec_ptr(slab, C.rec) + 10) = (unsigned short)nb;
            C.blk = nb;
            pl[C.level * 64].z = pm_make(nb, n_new, false, 0u);
            if (COUNT) { ctr[CTR_EXPANSIONS]++; ctr[CTR_NEW_NODES] += n_new; }

            if (cfg.auto_extend() && n_new == 1u) {
                const int a = __ffs((int)lmask) - 1;
                const int ha = c4::col_height(occ, a);
                const uint64_t abit = 1ull << (ha + 7 * a);
                const uint64_t nmy = C.op, nop = C.my | abit;
                const bool aw = c4::won(nop);
                const bool over = aw || (occ | abit) == c4::FULL;
                // the only child: slot 0 of the new block (the block's own sums are written by backprop)
                C.rec = nb * 16u;
                C.kind = aw ? 0u : 1u;
                C.nsolved = over;
                C.qt = over ? 0u : f32_bits(y_unvisited);
                st_rec(slab, C.rec, C.qt, 1.0f, nf_make(0u, over, 0u, over ? C.kind : 0u));
#pragma unroll
                for (uint32_t k = 1; k < 9; k++) st_rec_none(slab, C.rec + k);
                C.blk = 0;
                C.my = nmy;
                C.op = nop;
                C.pN = 0.0f;
                C.level++;
                pl[C.level * 64] = make_uint4(C.rec, f32_bits(0.0f), pm_make(0u, (uint32_t)__popc(legal_mask_of(C.my | C.op)), over, over ? C.kind : 0u), C.qt);
                if (over) {  // visit() of a solved node returns its one-hot outcome (mcts.rs:377-379)
                    X.p0 = aw ? 1.0f : 0.0f;
                    X.p1 = aw ? 0.0f : 1.0f;
                    X.p2 = 0.0f;
                    X.solved = true;
                    C.blk = lane_alloc_block(T, bcap, error);
                    *reinterpret_cast<unsigned short*>(rec_ptr(slab, C.rec) + 10) = (unsigned short)(C.blk | (C.kind << 14));
                    pl[C.level * 64].z = pm_make(C.blk, 0u, true, C.kind);
                    break;
                }
                continue;
            }
            X.needs_eval = true;
            X.legal_mask = lmask;
            break;
        }
    }
}

// Phase A of a round of the one-tree-per-lane kernels: start an explore, descend, stop on a leaf.
template <bool COUNT, int FAST, bool ROOT_IN_T = false>
SYN_DEV void lane_select_expand(const DevMctsCfg& cfg_, LaneTree& T, LaneWalk& Wk, LaneLeaf& X, bool active, uint4* pl,
                                uint32_t bcap, int thresh, uint32_t* ctr, int* error, const uint32_t* pk, int pk_stride,
                                uint64_t noise_seed = 0, LaneProf* lp = nullptr, int scan_min = 64,
                                const float* fpu_tbl = FPU_NORMAL_TABLE) {
    const bool pending = active && Wk.pending;
    X.at_leaf = false;
    X.was_pending = pending;
    X.needs_eval = pending;
    X.solved = false;
    X.p0 = X.p1 = X.p2 = 0.0f;
    X.legal_mask = Wk.pend_lmask;
    LaneCursor C;
    C.rec = Wk.rec; C.blk = Wk.blk; C.kind = Wk.kind; C.qt = Wk.qt;
    C.nsolved = Wk.solved;
    C.pN = Wk.pN;
    C.my = Wk.my; C.op = Wk.op;
    C.level = Wk.level;
    bool desc = active && Wk.descending;
    if (active && !desc && !pending) {
        lane_begin_explore<COUNT, FAST, ROOT_IN_T>(cfg_, T, C, pl, ctr, pk, pk_stride);
        desc = true;
    }

    // ---- descent (mcts.rs:310-341): every lane walks its own tree, one level (= one cache line) per iteration.
    // A round ends when `thresh` lanes (a whole number of 16-position tiles) stand on a leaf that needs the network, or
    // when nobody is descending any more.
    bool hit_solved = false, at_leaf = pending;
    // legal columns of the current position: computed once per round and updated as the descent drops stones
    uint32_t lm = legal_mask_of(C.my | C.op);
    if (CfgView<FAST>{cfg_}.fpu_normal() && scan_min > 0) {
        // Fpu::Func with deferred scans (scan_min = 0: no deferral — the plain loop below, whose levels take their draws on the spot):
        // levels that need no draws first, then one scan for every lane that waits for draws (FpuHold)
        FpuHold H;
        H.wait = false;
        for (;;) {
            if (desc && !H.wait) {
                if (C.nsolved) { hit_solved = true; desc = false; at_leaf = true; }
                else if (C.blk == 0u) { desc = false; at_leaf = true; }
            }
            if (__popcll(__ballot(at_leaf && !hit_solved)) >= thresh) break;
            // a scan iteration — one noise_fpu_scan for every lane that needs draws, then the level for EVERY descending lane — is
            // taken when no lane can take a level without draws, or once `scan_min` lanes wait for one (64 = the first rule alone)
            const unsigned long long waiting = __ballot(desc && H.wait);
            const bool scan_now = __ballot(desc && !H.wait) == 0ull || __popcll(waiting) >= scan_min;
            if (scan_now && __ballot(desc) == 0ull) break;
            if (desc && (scan_now || !H.wait)) {
                H.wait = false;
                lane_descend_level<COUNT, FAST, true>(cfg_, T, C, lm, pl, ctr, noise_seed, &H, scan_now, lp, fpu_tbl);
            }
        }
    } else
    for (;;) {
        if (desc) {
            if (C.nsolved) { hit_solved = true; desc = false; at_leaf = true; }
            else if (C.blk == 0u) { desc = false; at_leaf = true; }
        }
        if (__ballot(desc) == 0ull || __popcll(__ballot(at_leaf && !hit_solved)) >= thresh) break;
        if (desc) lane_descend_level<COUNT, FAST>(cfg_, T, C, lm, pl, ctr, noise_seed, nullptr, true, lp, fpu_tbl);
    }

    const unsigned long long lp_ta = lp ? lp_now() : 0ull;
    if (at_leaf && !pending) lane_arrive<COUNT, FAST>(cfg_, T, C, X, hit_solved, pl, bcap, ctr, error);
    if (lp) lp_add(lp, LP_A_ARRIVE, lp_now() - lp_ta);
    X.at_leaf = at_leaf;
    Wk.descending = desc;
    Wk.rec = C.rec;
    Wk.blk = C.blk;
    Wk.solved = C.nsolved;
    Wk.kind = C.kind;
    Wk.qt = C.qt;
    Wk.pN = C.pN;
    Wk.my = C.my;
    Wk.op = C.op;
    Wk.level = C.level;
}

// ---------------------------------------------------------------------------------------------- phase C
// The two softmaxes of a fresh leaf, per lane: the legal-move softmax of the nine raw logits (mcts.rs:407-423: max over the
// children, exp(l - max), sum in child order, divide) -> pr[c] for every legal column c (other entries unspecified), and — for a
// lane whose outputs came from the network in this round (`do_value`) — policies.rs:54-57's softmax over the three outcome logits
// in (v0, v1, v2). Twelve exponentials and twelve divisions per lane: they run two at a time (det_expf2_in_range, one refined
// reciprocal per sum + div2_by_shared) whenever every argument of the wave is inside those functions' exact ranges — logit gaps
// below 41, i.e. always for a sane network — and through det_expf / IEEE division otherwise. Same bits either way.
SYN_DEV void lane_softmaxes(uint32_t lmask, const float (&lg)[9], float (&pr)[9], bool do_value, float& v0, float& v1, float& v2) {
    float mx = -__builtin_inff();
#pragma unroll
    for (int c = 0; c < 9; c++)
        if ((lmask >> c) & 1u) mx = lg[c] > mx ? lg[c] : mx;
    float vm = v0;
    vm = v1 > vm ? v1 : vm;
    vm = v2 > vm ? v2 : vm;
    float x[12];
    bool in_range = true;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const bool legal = ((lmask >> c) & 1u) != 0u;
        const float d = lg[c] - mx;
        in_range = in_range && (!legal || d >= -41.0f);  // (a legal d is <= 0, or NaN: then the comparison fails)
        x[c] = legal ? d : 0.0f;
    }
    {
        const float d0 = v0 - vm, d1 = v1 - vm, d2 = v2 - vm;
        in_range = in_range && (!do_value || (d0 >= -41.0f && d1 >= -41.0f && d2 >= -41.0f));
        x[9] = do_value ? d0 : 0.0f;
        x[10] = do_value ? d1 : 0.0f;
        x[11] = do_value ? d2 : 0.0f;
    }
    if (__ballot(!in_range) == 0ull) {
        float e[12];
#pragma unroll
        for (int i = 0; i < 12; i += 2) {
            const f32x2 ee = det_expf2_in_range(f32x2{x[i], x[i + 1]});
            e[i] = ee[0];
            e[i + 1] = ee[1];
        }
        float total = 0.0f;
#pragma unroll
        for (int c = 0; c < 9; c++) total += ((lmask >> c) & 1u) ? e[c] : 0.0f;  // child (= ascending column) order; + 0.0 is exact
        float vt = 0.0f;
        vt += e[9];
        vt += e[10];
        vt += e[11];
        // 1 <= total <= 9, 1 <= vt <= 3, every e >= exp(-41) > 2^-60: inside the packed division's exact range
        const float ry = rcp_refined_safe_range(total), rv = rcp_refined_safe_range(vt);
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const f32x2 q = div2_by_shared(f32x2{e[i], e[i + 1]}, total, ry);
            pr[i] = q[0];
            pr[i + 1] = q[1];
        }
        pr[8] = div2_by_shared(f32x2{e[8], e[8]}, total, ry)[0];
        const f32x2 qa = div2_by_shared(f32x2{e[9], e[10]}, vt, rv), qb = div2_by_shared(f32x2{e[11], e[11]}, vt, rv);
        if (do_value) { v0 = qa[0]; v1 = qa[1]; v2 = qb[0]; }
    } else {
        float e[9];
        float total = 0.0f;
#pragma unroll
        for (int c = 0; c < 9; c++) {
            e[c] = det_expf(lg[c] - mx);
            if ((lmask >> c) & 1u) total += e[c];
        }
#pragma unroll
        for (int c = 0; c < 9; c++) pr[c] = e[c] / total;
        if (do_value) value_softmax(v0, v1, v2);
    }
}

// The rest of visit() for the node expanded in phase A (mcts.rs:389-423): writes its block — one record per legal column
// (terminal children already solved) with the priors `pr` of lane_softmaxes. Returns what backprop's solver walk needs to know
// about the new node without reading its line back: LEAF_ANY_SOLVED (some child is terminal), LEAF_ANY_WIN (some child's mover made
// four: the child is Lose(0), the new node Win(1)), LEAF_ALL_OVER (every child is terminal).
// `hdr_flag` (out): the value for word 3 of the block's header — non-zero iff some prior lies outside the range the descent's
// packed division is exact on (tiny but non-zero, or not finite); backprop writes the header (sums + this word).
// Root noise (mcts.rs:229-269) applies when `noise_kind` != 0 (the caller passes it for the root's own expansion only):
// 1 = PolicyNoise::Equal{weight}, 2 = PolicyNoise::Dirichlet{alpha, weight} sampled from the tree's stream (noise.cuh).
enum : uint32_t { LEAF_ANY_SOLVED = 1u, LEAF_ANY_WIN = 2u, LEAF_ALL_OVER = 4u };
SYN_DEV uint32_t lane_write_children(unsigned char* slab, uint32_t blk, uint32_t lmask, uint64_t leaf_my, uint64_t leaf_op,
                                 const float (&pr)[9], int noise_kind, float noise_weight, float noise_alpha, uint64_t noise_seed,
                                 float y_unvisited, uint32_t& hdr_flag, bool store_twice = false) {
    const uint32_t nc = (uint32_t)__popc(lmask);
    const float noise = 1.0f / (float)nc;
    float dir[9];
#pragma unroll
    for (int c = 0; c < 9; c++) dir[c] = 0.0f;
    if (__ballot(noise_kind == 2 && nc >= 2u) != 0ull) {
        if (noise_kind == 2 && nc >= 2u) noise_dirichlet(noise_seed, noise_alpha, nc, dir);  // dir[i] = noise of child i
    }
    const uint64_t my = leaf_my, occ = leaf_my | leaf_op;
    // (the block's header — its own sums — is written by the backprop that follows every expansion)
    // every cell that would give the mover four in a row, computed once for the whole expansion instead of one won() per
    // child; the cell a child's stone lands on is the column's lowest free cell. A child's game is over if its stone makes the
    // four or fills the board (the last of the 63 cells).
    const uint64_t win_drop = c4::winning_cells(my) & c4::next_free_cells(occ);
    const bool last_cell = __popcll(occ) == 62;
    unsigned char* const rec0 = rec_ptr(slab, blk * 16u);
    uint32_t idx = 0;
    bool any_solved = false, flag = false, any_win = false, all_over = true;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const bool legal = ((lmask >> c) & 1u) != 0u;
        float p = pr[c];
        if (noise_kind == 1 && nc >= 2u) p = p * (1.0f - noise_weight) + noise_weight * noise;
        if (noise_kind == 2 && nc >= 2u) {
            float dn = 0.0f;
#pragma unroll
            for (uint32_t k = 0; k < 9; k++) dn = (k == idx) ? dir[k] : dn;
            p = p * (1.0f - noise_weight) + noise_weight * dn;
        }
        const bool w = (win_drop & (0x7Full << (7 * c))) != 0ull;  // won(child.op_bb): the mover's stones plus this one (connect4.rs:224-229)
        const bool over = w || last_cell;
        // Outcome::from(reward(child.player())): the mover won -> the child's side to move lost; turns 0.
        // Every COLUMN stores exactly one record, without a branch: a legal column the record of its child (slot idx, in ascending
        // column order), a full column one of the "no child" records {-inf, 0, 0} that fill the slots behind the last child (slot
        // nc + the number of full columns before it) — the nine stores cover the nine slots exactly once.
        const uint32_t slot = legal ? idx : nc + ((uint32_t)c - idx);
        const uint32_t r_qt = legal ? (over ? 0u : f32_bits(y_unvisited)) : 0xFF800000u;
        const uint32_t r_p = legal ? f32_bits(p) : 0u;
        const uint32_t r_nf = legal ? nf_make(0u, over, 0u, over ? (w ? 0u : 1u) : 0u) : 0u;
        *reinterpret_cast<lu3*>(rec0 + slot * 12u) = lu3{r_qt, r_p, r_nf};
        if (store_twice) {
            lp_fence();
            *reinterpret_cast<lu3*>(rec0 + slot * 12u) = lu3{r_qt, r_p, r_nf};
            lp_fence();
        }
        flag = flag || (legal && !(p == 0.0f || (p >= PRIOR_SAFE_MIN && p <= 2.0f)));
        any_solved = any_solved || (legal && over);
        any_win = any_win || (legal && w);
        all_over = all_over && (!legal || over);
        idx += legal ? 1u : 0u;
    }
    hdr_flag = flag ? 1u : 0u;
    return (any_solved ? LEAF_ANY_SOLVED : 0u) | (any_win ? LEAF_ANY_WIN : 0u) | (all_over ? LEAF_ALL_OVER : 0u);
}

// Both steps for callers that hold the value head's probabilities already (pc_kernel.cuh).
SYN_DEV bool lane_create_children(unsigned char* slab, uint32_t blk, const LaneLeaf& X, uint64_t leaf_my, uint64_t leaf_op,
                                  const float (&lg)[9], int noise_kind, float noise_weight, float noise_alpha, uint64_t noise_seed,
                                  float y_unvisited, uint32_t& hdr_flag) {
    float pr[9];
    float u0 = 0.0f, u1 = 0.0f, u2 = 0.0f;
    lane_softmaxes(X.legal_mask, lg, pr, false, u0, u1, u2);
    return (lane_write_children(slab, blk, X.legal_mask, leaf_my, leaf_op, pr, noise_kind, noise_weight, noise_alpha, noise_seed,
                                y_unvisited, hdr_flag) & LEAF_ANY_SOLVED) != 0u;
}

SYN_DEV int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// backprop (mcts.rs:429-488) replayed from the path buffer.
//   phase 1 (per lane): the MCTS-Solver walk, level by level while the subtree below stays proven. A level needs ONE
//            line — the node's block holds its sums and its children's solutions — fetched together with the next
//            level's path entry: one memory round trip per level.
//   phase 2 (whole wave, four levels per step): every remaining level just adds the leaf's outcome distribution
//            (win/lose swapped once per level climbed) and one visit, so the levels are independent; the node's sums
//            come from the path log's second plane (the descent read them with the line it needed anyway), so four
//            levels cost ONE round trip of coalesced rows and no random line read — only the dirtied lines remain.
// `leaf_flag`: header word 3 of the node expanded in this pass (lane_create_children), 0 otherwise; every other visited node
// keeps the word it has (a node that was never backpropagated into has no header yet: 0).
// LEAF_KNOWN (the one-tree-per-lane kernel): the solver walk's FIRST level — the leaf itself — is decided without reading the
// leaf's line. `leaf_code` != 0: the node was expanded in this pass and lane_write_children reported its children (it has no
// visits, no sums and no solution of its own yet): a winning child makes it Win(1), else it is solved only if every child is
// terminal (then a Draw(1)). `leaf_code` == 0: explore() stopped on a node that already was solved (mcts.rs:314-316, or a terminal
// node's first visit): nothing below a solved node changes once it is solved — no descent passes through it — so the maximum over
// its children that backprop recomputes (mcts.rs:441-449) is its own solution and, unless that is a Win, all its children are still
// solved; only its sums are needed, and only if it was visited before (one 16-byte load).
template <bool COUNT, int FAST, bool LEAF_KNOWN = false>
SYN_DEV void lane_backprop(const DevMctsCfg& cfg_, LaneTree& T, int depth, float d0, float d1, float d2, bool solved,
                           bool active, const uint4* pl, uint32_t* ctr, uint32_t leaf_flag, unsigned long long* t_mid = nullptr,
                           LaneProf* lp = nullptr, uint32_t leaf_code = 0u) {
    const CfgView<FAST> cfg{cfg_};
    unsigned char* const slab = T.slab;
    if (COUNT && active) {
        ctr[CTR_BACKPROP_LEVELS] += (uint32_t)(depth + 1);
        if ((uint32_t)(depth + 1) > ctr[CTR_MAX_DEPTH]) ctr[CTR_MAX_DEPTH] = (uint32_t)(depth + 1);
    }
    int L = active ? depth : -1;
    // ---- phase 1
    if (cfg.solve() && solved && L >= 0) {
        uint4 pe = pl[L * 64];
        bool first = LEAF_KNOWN;
        for (;;) {
            const uint32_t rec = pe.x, meta = pe.z;
            float N = bits_f32(pe.y);
            const uint32_t blk = pm_blk(meta), nc = pm_nc(meta);
            const uint4* line = reinterpret_cast<const uint4*>(blk_ptr(slab, blk));
            float W0 = 0.0f, W1 = 0.0f, W2 = 0.0f;
            uint32_t w_old = 0u;
            bool bsome, all_solved = true;
            uint32_t bkind, bturns;
            uint4 pe_next;
            if (LEAF_KNOWN && first) {
                // the leaf's own level (see the comment above the function)
                pe_next = pl[(L > 0 ? L - 1 : 0) * 64];
                if (leaf_code == 0u && N != 0.0f) {
                    const uint4 hdr = line[0];
                    W0 = bits_f32(hdr.x); W1 = bits_f32(hdr.y); W2 = bits_f32(hdr.z);
                    w_old = hdr.w;
                }
                if (leaf_code != 0u) {
                    bsome = true;
                    bkind = (leaf_code & LEAF_ANY_WIN) ? 2u : 1u;
                    bturns = 1u;
                    all_solved = (leaf_code & LEAF_ALL_OVER) != 0u;
                    if (COUNT) ctr[CTR_SOLVER_CHILDREN] += nc;
                } else {
                    bsome = pm_solved(meta);
                    bkind = pm_kind(meta);
                    bturns = pe.w;
                    if (COUNT) ctr[CTR_SOLVER_CHILDREN] += nc;
                }
                if (lp) lp_step(lp, LP_W_WAIT, LP_W_ITERS, LP_W_LANES, 0ull);
            } else {
                // one line: the node's sums and its children's records; plus the next level's path entry
                const uint4 hdr = line[0];
                uint32_t d[28];
#pragma unroll
                for (int j = 0; j < 7; j++) {
                    const uint4 t = line[1 + j];
                    d[4 * j] = t.x; d[4 * j + 1] = t.y; d[4 * j + 2] = t.z; d[4 * j + 3] = t.w;
                }
                pe_next = pl[(L > 0 ? L - 1 : 0) * 64];
                if (lp_abl(lp, ABL_WALK_LD)) {
                    lp_fence();
                    uint32_t sink = 0;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint4 t = line[j];
                        sink ^= t.x ^ t.y ^ t.z ^ t.w;
                    }
                    asm volatile("" ::"v"(sink));
                }
                if (lp) {
                    const unsigned long long t0_ = lp_now();
                    lp_wait_vm(lp);
                    lp_step(lp, LP_W_WAIT, LP_W_ITERS, LP_W_LANES, lp_now() - t0_);
                }
                // a node that was never backpropagated into has nothing in its header yet
                W0 = N == 0.0f ? 0.0f : bits_f32(hdr.x);
                W1 = N == 0.0f ? 0.0f : bits_f32(hdr.y);
                W2 = N == 0.0f ? 0.0f : bits_f32(hdr.z);
                w_old = hdr.w;
                uint32_t key = outcome_key(pm_solved(meta), pm_kind(meta), pe.w);
                if (COUNT) ctr[CTR_SOLVER_CHILDREN] += nc;
                // (selects, not nine branches: a slot past the last child contributes nothing — a terminal node's block, read here as
                //  the leaf by the kernels without LEAF_KNOWN, has no initialised slots at all)
                const uint32_t live = (1u << nc) - 1u;
                uint32_t unsolved = 0u;
#pragma unroll
                for (uint32_t i = 0; i < 9; i++) {
                    const uint32_t nf = d[3 * i + 2];
                    const bool sol = nf_solved(nf) && ((live >> i) & 1u) != 0u;
                    unsolved |= sol ? 0u : (1u << i);
                    // solution.map(reversed) (game.rs:29-35): Win<->Lose, Draw stays, turns + 1
                    const uint32_t ck = nf_kind(nf);
                    const uint32_t rk = sol ? outcome_key(true, ck == 1u ? 1u : 2u - ck, d[3 * i] + 1u) : 0u;
                    key = rk > key ? rk : key;
                }
                all_solved = (unsolved & live) == 0u;
                outcome_from_key(key, bsome, bkind, bturns);
            }
            first = false;
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
            W0 += d0;
            W1 += d1;
            W2 += d2;
            N += 1.0f;
            const uint32_t w3 = N != 1.0f ? w_old : (L == depth ? leaf_flag : 0u);  // (N was incremented above)
            *reinterpret_cast<float4*>(blk_ptr(slab, blk)) = make_float4(W0, W1, W2, bits_f32(w3));
            if (rec != REC_ROOT) {
                // the node is (still) solved: its q slot carries the turn count, its record the outcome
                unsigned char* r = rec_ptr(slab, rec);
                *reinterpret_cast<uint32_t*>(r) = bturns;
                *reinterpret_cast<uint32_t*>(r + 8) = nf_make((uint32_t)N, true, blk, bkind);
            } else {
                T.root_solved = true;
                T.root_sol = bkind | (bturns << 2);
            }
            const float t = d0;
            d0 = d2;
            d2 = t;
            L--;
            if (L < 0) break;
            pe = pe_next;
        }
    }
    if (t_mid) *t_mid = (unsigned long long)__builtin_readcyclecounter();
    // ---- phase 2: levels L..0 of this lane; (d0,d1,d2) is the delta for level L
    for (int base = wave_max_i32(L); base >= 0; base -= 4) {
        uint4 pe[4];
        float4 a[4];
        const unsigned long long lp_t0 = lp ? lp_now() : 0ull;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int Lj = base - j;
            pe[j] = pl[(Lj < 0 ? 0 : Lj) * 64];
            // the node's sums as the descent logged them (coalesced row of the second plane)
            a[j] = *reinterpret_cast<const float4*>(pl + PATH_PLANE + (Lj < 0 ? 0 : Lj) * 64);
        }
        unsigned long long lp_t1 = 0ull;
        if (lp) {
            lp_wait_vm(lp);
            lp_t1 = lp_now();
            unsigned long long nl = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) nl += (unsigned long long)__popcll(__ballot(base - j >= 0 && base - j <= L));
            lp_add(lp, LP_S_WAIT, lp_t1 - lp_t0);
            lp_add(lp, LP_S_LANES, nl);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int Lj = base - j;
            const bool ok = Lj >= 0 && Lj <= L;
            // Only a visited node the descent did not read through lacks logged sums: a solved leaf that was visited before,
            // reached with the solver switched off (with it on, phase 1 has handled that level). It reads its header.
            if (ok && pe[j].y != 0u && (pe[j].z & PM_HAS_W) == 0u)
                a[j] = *reinterpret_cast<const float4*>(blk_ptr(slab, pm_blk(pe[j].z)));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int Lj = base - j;
            if (Lj >= 0 && Lj <= L) {
                const uint32_t rec = pe[j].x, meta = pe[j].z;
                float N = bits_f32(pe[j].y);
                const bool flip = ((L - Lj) & 1) != 0;  // delta[0] <-> delta[2] once per level climbed
                const float W0 = (N == 0.0f ? 0.0f : a[j].x) + (flip ? d2 : d0);
                const float W1 = (N == 0.0f ? 0.0f : a[j].y) + d1;
                const float W2 = (N == 0.0f ? 0.0f : a[j].z) + (flip ? d0 : d2);
                const uint32_t w3 = N != 0.0f ? f32_bits(a[j].w) : (Lj == depth ? leaf_flag : 0u);
                N += 1.0f;
                *reinterpret_cast<float4*>(blk_ptr(slab, pm_blk(meta))) = make_float4(W0, W1, W2, bits_f32(w3));
                if (rec != REC_ROOT) {
                    unsigned char* r = rec_ptr(slab, rec);
                    // a solved node (explore() hit it, or the solver is off) keeps its turn count in the q slot
                    if (!pm_solved(meta)) *reinterpret_cast<float*>(r) = -((W2 - W0) / N);
                    *reinterpret_cast<unsigned short*>(r + 8) = (unsigned short)((uint32_t)N | (pm_solved(meta) ? 0x8000u : 0u));
                }
                if (lp_abl(lp, ABL_SWEEP_ST)) {
                    lp_fence();
                    *reinterpret_cast<float4*>(blk_ptr(slab, pm_blk(meta))) = make_float4(W0, W1, W2, bits_f32(w3));
                    if (rec != REC_ROOT) {
                        unsigned char* r = rec_ptr(slab, rec);
                        if (!pm_solved(meta)) *reinterpret_cast<float*>(r) = -((W2 - W0) / N);
                        *reinterpret_cast<unsigned short*>(r + 8) = (unsigned short)((uint32_t)N | (pm_solved(meta) ? 0x8000u : 0u));
                    }
                    lp_fence();
                }
            }
        }
        if (lp) lp_step(lp, LP_S_ALU, LP_S_STEPS, -1, lp_now() - lp_t1);
    }
}

// ---------------------------------------------------------------------------------------------- end of a search
// The root's block read ONCE — one memory round trip for everything the end of a search needs (the code below used to fetch every
// child record two or three times, each behind its own wait: ~25 dependent round trips per move, executed by one or two lanes of
// the wave while the other 62 stand by): the root's own sums and the nine child records in slot order (slot i = the i-th legal
// column in ascending order; entries past `nc` are unspecified).
struct LaneRoot {
    uint32_t nc, lmask;
    float rootN;
    float4 hdr;
    uint32_t d[28];
};
SYN_DEV LaneRoot lane_root(const LaneTree& T) {
    LaneRoot R;
    R.rootN = (float)T.iter;
    R.lmask = T.next_block > 1u ? legal_mask_of(T.root_my | T.root_op) : 0u;  // the root's children: its legal columns
    R.nc = (uint32_t)__popc(R.lmask);
    const uint4* line = reinterpret_cast<const uint4*>(blk_ptr(T.slab, 1));     // (block 1 exists in every slab)
    R.hdr = *reinterpret_cast<const float4*>(line);
#pragma unroll
    for (int j = 0; j < 7; j++) {
        const uint4 t = line[1 + j];
        R.d[4 * j] = t.x; R.d[4 * j + 1] = t.y; R.d[4 * j + 2] = t.z; R.d[4 * j + 3] = t.w;
    }
    return R;
}
// column of slot k
SYN_DEV int lane_slot_column(uint32_t lmask, uint32_t k) {
    uint32_t m = lmask;
    for (uint32_t i = 0; i < k; i++) m &= m - 1u;
    return __ffs((int)m) - 1;
}

// MCTS::target_policy numerators (mcts.rs:174-211) per child slot (0 past the last child) and their sum in child order
SYN_DEV float lane_policy_weights(const LaneTree& T, const LaneRoot& R, float (&wts)[9]) {
    const bool first_visit = R.rootN == 1.0f;
    const bool root_win = T.root_solved && (T.root_sol & 3u) == 2u;
    float total = 0.0f;
#pragma unroll
    for (uint32_t i = 0; i < 9; i++) {
        wts[i] = 0.0f;
        if (i < R.nc) {
            const uint32_t nf = R.d[3 * i + 2];
            float v;
            if (first_visit) v = root_win ? ((nf_solved(nf) && nf_kind(nf) == 0u) ? 1.0f : 0.0f) : 1.0f;
            else v = nf_N(nf);
            wts[i] = v;
            total += v;
        }
    }
    return total;
}

SYN_DEV void lane_target_q(const LaneTree& T, const LaneRoot& R, float& q0, float& q1, float& q2) {
    if (T.root_solved) {
        const uint32_t k = T.root_sol & 3u;
        q0 = k == 0u ? 1.0f : 0.0f;
        q1 = k == 1u ? 1.0f : 0.0f;
        q2 = k == 2u ? 1.0f : 0.0f;
    } else {
        q0 = R.hdr.x / R.rootN;
        q1 = R.hdr.y / R.rootN;
        q2 = R.hdr.z / R.rootN;
    }
}

// MCTS::best_action (mcts.rs:273-294); also returns the record word (N | solved | block | kind) of the chosen child
SYN_DEV int lane_best_action(const LaneRoot& R, int action_selection, uint32_t& best_nf) {
    int best = -1;
    float b0 = 0.0f, b1 = 0.0f;
    best_nf = 0;
    uint32_t m = R.lmask;
#pragma unroll
    for (uint32_t i = 0; i < 9; i++) {
        if (i < R.nc) {
            const int c = __ffs((int)m) - 1;
            m &= m - 1u;
            const uint32_t nf = R.d[3 * i + 2];
            float k0, k1;
            if (nf_solved(nf)) {
                const uint32_t kind = nf_kind(nf);
                const float t = (float)R.d[3 * i];
                if (kind == 2u) { k0 = 0.0f; k1 = t; }
                else if (kind == 1u) { k0 = 2.0f; k1 = -t; }
                else { k0 = 3.0f; k1 = -t; }
            } else {
                k0 = 1.0f;
                // -child.q(): the stored q, except for a never-visited child where the reference divides 0 by 0
                const float cN = nf_N(nf);
                const float nq = cN == 0.0f ? -((0.0f - 0.0f) / cN) : bits_f32(R.d[3 * i]);
                k1 = action_selection == 0 ? nq : cN;
            }
            const bool gt = best < 0 || (k0 > b0) || (k0 == b0 && k1 > b1);
            if (gt) { best = c; b0 = k0; b1 = k1; best_nf = nf; }
        }
    }
    return best;
}

SYN_DEV uint32_t lane_child_nf(const LaneRoot& R, int action, bool& is_child) {
    is_child = ((R.lmask >> action) & 1u) != 0u;
    const uint32_t slot = (uint32_t)__popc(R.lmask & ((1u << action) - 1u));
    uint32_t nf = 0;
#pragma unroll
    for (uint32_t i = 0; i < 9; i++) nf = slot == i ? R.d[3 * i + 2] : nf;
    return is_child ? nf : 0u;
}

// run_game's per-move tail (alpha_zero.rs:243-264) + game end (fill_state_info / store_rewards, 296-338)
template <bool COUNT>
SYN_DEV void lane_move_step(const EngineParams& P, LaneTree& T, uint32_t* ctr, const LaneProf* lp = nullptr) {
    const DevRolloutCfg& rc = P.roll;
    unsigned long long lp_t = lp ? lp_now() : 0ull;
#define SYN_MLAP(f) if (lp) { lp_wait_vm(lp); const unsigned long long n_ = lp_now(); lp_add(lp, f, n_ - lp_t); lp_t = n_; }
    const bool want_random = T.turn < rc.random_until;
    const bool maybe_sample = !want_random && T.turn < rc.sample_until;
    const LaneRoot R = lane_root(T);   // (its loads are in flight while the generator below runs)
    uint32_t rnd = 0;
    if (want_random || maybe_sample) {
        StdRng rng;
        rng.seed_from_u64(P.base_seed + P.first_game + (unsigned long long)T.job);
        rnd = rng.word(T.rng_index);
    }
    SYN_MLAP(LP_M_T1)
    float ps[9];   // target policy per child slot
    const float wtotal = lane_policy_weights(T, R, ps);
#pragma unroll
    for (int i = 0; i < 9; i++) ps[i] = ps[i] / wtotal;
    const float pz = 0.0f / wtotal;   // what the reference's division leaves in a non-child column
    float q0, q1, q2;
    lane_target_q(T, R, q0, q1, q2);
    const size_t pos = (size_t)T.job * 63 + (size_t)T.turn;
    P.states_bb[pos * 2 + 0] = T.root_my;
    P.states_bb[pos * 2 + 1] = T.root_op;
    P.root_nodes[pos] = T.num_nodes;
    // pi by column: every column first, then the children over it (two stores of one lane to one address stay in order)
#pragma unroll
    for (int c = 0; c < 9; c++) P.pis[pos * 9 + c] = pz;
    {
        uint32_t m = R.lmask;
#pragma unroll
        for (uint32_t i = 0; i < 9; i++) {
            if (i < R.nc) {
                const int c = __ffs((int)m) - 1;
                m &= m - 1u;
                P.pis[pos * 9 + c] = ps[i];
            }
        }
    }
    P.vs[pos * 3 + 0] = q0;
    P.vs[pos * 3 + 1] = q1;
    P.vs[pos * 3 + 2] = q2;

    SYN_MLAP(LP_M_T2)
    // sample_action (alpha_zero.rs:270-294)
    uint32_t best_nf;
    const int best = lane_best_action(R, rc.action, best_nf);
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
        action = lane_slot_column(R.lmask, (uint32_t)(mm >> 32));
    } else if (maybe_sample && (!nf_solved(best_nf) || !rc.stop_when_solved)) {
        // WeightedIndex over the nine columns: a column that is no child weighs 0 and never changes a cumulative sum, so the
        // partition point among the columns is the column of the partition point among the child slots
        const float chosen_unit = bits_f32((rnd >> 9) | 0x3F800000u) - 1.0f;
        float total = ps[0];
        float cum[8];
#pragma unroll
        for (int i = 1; i < 9; i++) {
            cum[i - 1] = total;
            total += (uint32_t)i < R.nc ? ps[i] : 0.0f;
        }
        const float chosen = chosen_unit * total + 0.0f;
        T.rng_index += 1;
        uint32_t slot = 0;
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) slot = (i + 1u < R.nc && cum[i] <= chosen) ? i + 1u : slot;
        // (with the last column no child its cumulative weight is the total, which `chosen` stays below)
        action = lane_slot_column(R.lmask, slot);
    } else {
        action = best;
    }
    P.actions[pos] = (unsigned char)action;
    SYN_MLAP(LP_M_T3)

    bool a_child;
    const uint32_t a_nf = lane_child_nf(R, action, a_child);
    bool sol_some = a_child && nf_solved(a_nf);
    uint32_t sol_kind = nf_kind(a_nf);

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
    SYN_MLAP(LP_M_T4)

    if (!sol_some) {
        T.root_my = nmy;
        T.root_op = nop;
        T.next_block = 0;
        T.num_nodes = 0;
        T.iter = 0;
        T.root_solved = false;
        T.root_sol = 0;
        T.fpu_draws = 0;
        return;
    }

    const int n = T.turn;
    const uint32_t last_kind = sol_kind == 1u ? 1u : 2u - sol_kind;
    // store_rewards: ValueTarget::Q keeps the q already stored move by move (alpha_zero.rs:319): nothing to rewrite
    if (rc.value_target != 1)
    for (int i = 0; i < n; i++) {
        const bool flip = ((n - 1 - i) & 1) != 0;
        const uint32_t zk = (flip && last_kind != 1u) ? 2u - last_kind : last_kind;
        const float z0 = zk == 0u ? 1.0f : 0.0f, z1 = zk == 1u ? 1.0f : 0.0f, z2 = zk == 2u ? 1.0f : 0.0f;
        const float t = (float)(i + 1) / (float)n;
        float* v = P.vs + ((size_t)T.job * 63 + (size_t)i) * 3;
        const float a0 = v[0], a1 = v[1], a2 = v[2];
        float o0, o1, o2;
        if (rc.value_target == 0) { o0 = z0; o1 = z1; o2 = z2; }
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
    atomicAdd(P.job_next + 1, 1);  // games finished so far (syn_progress)
    if (COUNT) ctr[CTR_GAMES]++;
    lane_start_job<MODE_SELFPLAY>(P, T);
    SYN_MLAP(LP_M_T5)
#undef SYN_MLAP
}

SYN_DEV void lane_search_finish(const EngineParams& P, LaneTree& T) {
    const LaneRoot R = lane_root(T);
    float ps[9];
    const float wtotal = lane_policy_weights(T, R, ps);
    float q0, q1, q2;
    lane_target_q(T, R, q0, q1, q2);
    uint32_t bnf;
    const int best = lane_best_action(R, P.action_selection, bnf);
    DevSearchResult* out = P.results + T.job;
    // every column as "no child" first, then the children over their columns
    const float pz = 0.0f / wtotal;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        out->child_N[c] = 0.0f;
        out->child_W[c][0] = 0.0f; out->child_W[c][1] = 0.0f; out->child_W[c][2] = 0.0f;
        out->child_P[c] = 0.0f;
        out->child_sol[c][0] = 0; out->child_sol[c][1] = 0; out->child_sol[c][2] = 0;
        out->target_pi[c] = pz;
    }
    uint32_t m = R.lmask;
#pragma unroll
    for (uint32_t i = 0; i < 9; i++) {
        if (i < R.nc) {
            const int c = __ffs((int)m) - 1;
            m &= m - 1u;
            const uint32_t r0 = R.d[3 * i], r1 = R.d[3 * i + 1], nf = R.d[3 * i + 2];
            float4 ca = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nf_N(nf) != 0.0f) ca = *reinterpret_cast<const float4*>(blk_ptr(T.slab, nf_blk(nf)));
            out->child_N[c] = nf_N(nf);
            out->child_W[c][0] = ca.x;
            out->child_W[c][1] = ca.y;
            out->child_W[c][2] = ca.z;
            out->child_P[c] = bits_f32(r1);
            const bool some = nf_solved(nf);
            out->child_sol[c][0] = some ? 1 : 0;
            out->child_sol[c][1] = some ? (int)nf_kind(nf) : 0;
            out->child_sol[c][2] = some ? (int)r0 : 0;
            out->target_pi[c] = ps[i] / wtotal;
        }
    }
    out->root_N = R.rootN;
    out->root_W[0] = R.hdr.x; out->root_W[1] = R.hdr.y; out->root_W[2] = R.hdr.z;
    out->root_sol[0] = T.root_solved ? 1 : 0;
    out->root_sol[1] = T.root_solved ? (int)(T.root_sol & 3u) : 0;
    out->root_sol[2] = T.root_solved ? (int)(T.root_sol >> 2) : 0;
    out->num_nodes = T.num_nodes;
    out->best_action = best;
    out->target_q[0] = q0; out->target_q[1] = q1; out->target_q[2] = q2;
    lane_start_job<MODE_SEARCH>(P, T);
}

// Cold path out of line, the tree's state by value. The callee reads the launch parameters where they already are — the kernel
// argument segment (every lane-per-tree kernel takes EngineParams as its only argument, at offset 0) — through scalar loads. (It
// used to receive a reference to a private copy of the 300-byte struct: twenty dependent scratch load -> wait -> store pairs in
// front of every call, on top of the caller-saved registers; a reference to the kernel's own parameter would have demoted the hot
// loop's pointers to the generic address space.)
typedef const __attribute__((address_space(4))) void* KernargPtr;
SYN_DEV KernargPtr lane_kernarg() { return (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr(); }
SYN_DEV EngineParams lane_params_from_kernarg(KernargPtr k) {
    EngineParams P;
    __builtin_memcpy(&P, k, sizeof(EngineParams));   // (only the fields the callee uses survive: scalar loads)
    return P;
}
template <bool COUNT>
__device__ __attribute__((noinline)) LaneTree lane_move_step_call(KernargPtr k, LaneTree t, uint32_t* ctr) {
    const EngineParams P = lane_params_from_kernarg(k);
    lane_move_step<COUNT>(P, t, ctr);
    return t;
}
// (diagnostic build: the same with stamps between its parts)
__device__ __attribute__((noinline)) LaneTree lane_move_step_call_prof(KernargPtr k, LaneTree t, const LaneProf* lp) {
    const EngineParams P = lane_params_from_kernarg(k);
    lane_move_step<false>(P, t, nullptr, lp);
    return t;
}
__device__ __attribute__((noinline)) LaneTree lane_search_finish_call(KernargPtr k, LaneTree t) {
    const EngineParams P = lane_params_from_kernarg(k);
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
// A ring of three 16-word ChaCha12 output blocks per lane in LDS ([slot][word][lane]); the blocks a playout is going to use
// are generated before its loop, when the wave is converged (see frozen_kernel.cuh: generated on demand inside the loop, the
// block function would run in almost every iteration for the few lanes crossing a block boundary there).
struct RolloutRing {
    uint32_t* lds;   // this lane's column: word k of slot s at lds[(s * 16 + k) * 64]
    uint32_t hi;     // blocks [hi - 3, hi) generated for `job` are in the ring (slot = block % 3)
    int job;
};

SYN_DEV void lane_rollout(uint64_t my, uint64_t op, unsigned long long seed, int job, uint32_t& rng_index, RolloutRing& ring,
                          float& v0, float& v1, float& v2) {
    StdRng rng;
    rng.seed_from_u64(seed);
    uint32_t index = rng_index;
    if (ring.job != job) { ring.job = job; ring.hi = index >> 4; }  // another tree's stream: nothing of it is cached
    auto generate_next = [&]() {
        uint32_t out[16];
        rng.block16(ring.hi, out);
        uint32_t* slot = ring.lds + (ring.hi % 3u) * 16u * 64u;
#pragma unroll
        for (int k = 0; k < 16; k++) slot[k * 64] = out[k];
        ring.hi++;
    };
    {
        const uint32_t want = (index >> 4) + 3u;
        while (ring.hi < want) generate_next();
    }
    bool leaf_player_moves = true;  // `my` is the side to move; the leaf itself is never terminal
    for (;;) {
        const uint64_t occ = my | op;
        uint32_t lmask = 0;
#pragma unroll
        for (int c = 0; c < 9; c++)
            if (c4::col_height(occ, c) < c4::HEIGHT) lmask |= 1u << c;
        const uint32_t n = (uint32_t)__popc(lmask);
        // Rng::gen_range(0..n as u8), rand 0.8.3 UniformInt<u8>::sample_single (as StdRng::gen_range_u8)
        const uint32_t zone = 0xFFFFFFFFu - (0xFFFFFFFFu - n + 1u) % n;
        uint32_t pick;
        for (;;) {
            const uint32_t blk = index >> 4;
            while (blk >= ring.hi) generate_next();  // only a playout that outruns the prefetched window
            const uint32_t v = ring.lds[((blk % 3u) * 16u + (index++ & 15u)) * 64u];
            const uint64_t m = (uint64_t)v * (uint64_t)n;
            if ((uint32_t)m <= zone) { pick = (uint32_t)(m >> 32); break; }
        }
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
    rng_index = index;
}

// ---------------------------------------------------------------------------------------------- the kernel
template <int NW>
struct LaneLds {
    // weight + bias image: 123,264 B (Connect4Net f32, mlp.cuh) or 124,320 B (Connect4Net as f16 pairs, f16x2_tile.cuh)
    static constexpr size_t IDX_OFF = (size_t)MlpGeom::IMG_FLOATS * 4 > (size_t)F16Geom::IMG_WORDS * 4 ? (size_t)MlpGeom::IMG_FLOATS * 4 : (size_t)F16Geom::IMG_WORDS * 4;
    static constexpr size_t FT_OFF = IDX_OFF + (size_t)NW * 64;        // + 64 B compaction index per wave
    static constexpr size_t PARK_OFF = FT_OFF + 64;                    // + the four feature shift tables (16 B each)
    static constexpr size_t FPU_OFF = PARK_OFF + (size_t)NW * 64 * 20;  // + 5 parked dwords per lane (root boards, turn|rng)
    // + the Fpu::Func draw's table of normal quantiles (noise.cuh: 2,945 floats; staged by the configuration families that draw)
    static constexpr size_t BYTES = FPU_OFF + (((size_t)FPU_NORMAL_CELLS + 1) * 4 + 15) / 16 * 16;
    static_assert(BYTES <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");
};

SYN_DEV uint64_t shfl_u64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// POLICY: 0 = Connect4Net on the matrix cores, 1 = RolloutPolicy (policies/rollout.rs; searches only),
//         2 = Connect4ConvNet on the matrix cores (convnet.cuh: its image takes the first 66 KB of the Connect4Net image's LDS)
//         3 = Connect4Net in the f16x2 arithmetic (f16x2_tile.cuh: two-term f16 split on v_mfma_f32_16x16x32_f16; P.wimg is that image)
template <int MODE, bool COUNT, int FAST, int NW, bool PROF = false, int POLICY = 0>
__global__ __launch_bounds__(64 * NW, 1) void selfplay_kernel_lanes(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = 64 * NW;
    float* wimg = reinterpret_cast<float*>(smem_raw);
    const float* bimg = wimg + MlpGeom::W_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    if (POLICY == 0) stage_weight_image(wimg, P.wimg, tid, NT);
    if (POLICY == 2) stage_conv_image(wimg, P.wimg, tid, NT);
    if (POLICY == 3) {
        const uint4* src = reinterpret_cast<const uint4*>(P.wimg);
        uint4* dst = reinterpret_cast<uint4*>(smem_raw);
        for (int i = tid; i < F16Geom::IMG_WORDS / 4; i += NT) dst[i] = src[i];
    }
    // RolloutPolicy needs no weights: the image's LDS holds the waves' ChaCha12 block rings instead (12 KB per wave)
    RolloutRing ring;
    ring.lds = reinterpret_cast<uint32_t*>(smem_raw) + (size_t)wave * (3 * 16 * 64) + lane;
    ring.hi = 0;
    ring.job = -1;
    static_assert(POLICY != 1 || (size_t)NW * 3 * 16 * 64 * 4 <= (size_t)MlpGeom::IMG_FLOATS * 4, "rings must fit the image region");
    if (tid < 4) {
        const FeatureTable f = make_feature_table(tid);
        *reinterpret_cast<uint4*>(smem_raw + LaneLds<NW>::FT_OFF + tid * 16) = make_uint4(f.t[0], f.t[1], f.t[2], f.t[3]);
    }
    // Fpu::Func: the draw's table of normal quantiles next to the weights (the parity family never draws)
    float* const fpu_tbl = reinterpret_cast<float*>(smem_raw + LaneLds<NW>::FPU_OFF);
    if (FAST != 1)
        for (int i = tid; i <= (int)FPU_NORMAL_CELLS; i += NT) fpu_tbl[i] = FPU_NORMAL_TABLE[i];

    uint32_t ctr[COUNT ? CTR_COUNT : 1];
#pragma unroll
    for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) ctr[i] = 0;

    LaneTree T;
    const size_t slot = (size_t)blockIdx.x * NT + (size_t)tid;
    T.slab = reinterpret_cast<unsigned char*>(P.stat) + slot * (size_t)P.cap * 32u;
    // this lane's column of its wave's path buffer ([level 0..63][lane 0..63] entries of 16 bytes)
    uint4* const pl = P.path + ((size_t)blockIdx.x * NW + (size_t)wave) * PATH_ENTRIES + (size_t)lane;
    const uint32_t bcap = P.cap / 4u;  // 128-byte blocks in this lane's slab
    lane_start_job<MODE>(P, T);
    // Cold per-lane state lives in LDS between the ends of searches (5 dwords per lane, lane-linear): the root position and
    // the game's turn / RNG position. That takes them out of the register budget of the hot loop.
    uint32_t* const pk = reinterpret_cast<uint32_t*>(smem_raw + LaneLds<NW>::PARK_OFF) + tid;
#define SYN_PARK()                                                                                           \
    do {                                                                                                     \
        pk[0] = (uint32_t)T.root_my; pk[NT] = (uint32_t)(T.root_my >> 32);                                   \
        pk[2 * NT] = (uint32_t)T.root_op; pk[3 * NT] = (uint32_t)(T.root_op >> 32);                          \
        pk[4 * NT] = (uint32_t)T.turn | (T.rng_index << 8);                                                  \
        T.root_my = 0; T.root_op = 0; T.turn = 0;                                                            \
        if (POLICY != 1) T.rng_index = 0;                                                                    \
    } while (0)
#define SYN_UNPARK()                                                                                         \
    do {                                                                                                     \
        T.root_my = (uint64_t)pk[0] | ((uint64_t)pk[NT] << 32);                                              \
        T.root_op = (uint64_t)pk[2 * NT] | ((uint64_t)pk[3 * NT] << 32);                                     \
        T.turn = (int)(pk[4 * NT] & 0xFFu);                                                                  \
        if (POLICY != 1) T.rng_index = pk[4 * NT] >> 8;                                                      \
    } while (0)
    SYN_PARK();
    __syncthreads();  // the only workgroup barrier: weights staged. From here on every wave free-runs.

    const int n_explores = P.roll.num_explores;
    const int thresh = P.lane_thresh;
    unsigned char* const idxw = smem_raw + LaneLds<NW>::IDX_OFF + wave * 64;  // compaction: rank -> lane
    unsigned long long cache_hits = 0, cache_misses = 0;
    LaneWalk Wk;
    Wk.descending = false;
    Wk.pending = false;
    Wk.pend_lmask = 0;
    Wk.rec = REC_ROOT; Wk.blk = 0; Wk.solved = false; Wk.kind = 0; Wk.qt = 0; Wk.pN = 0.0f; Wk.my = 0; Wk.op = 0; Wk.level = 0;
    unsigned long long pA = 0, pB = 0, pC = 0, pC1 = 0, pC2 = 0, pM = 0, pT = 0, pTiles = 0, pRounds = 0, pLanes = 0, pEvals = 0;
#define SYN_STAMP() (PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
#define SYN_LAP(acc) if (PROF) { unsigned long long n_ = SYN_STAMP(); acc += n_ - pT; pT = n_; }
    // seed of this lane's current tree for Fpu::Func / Dirichlet draws (noise.cuh): stream = seed + game (or root) index,
    // turn from the parked game state; only the runtime-switched configurations evaluate it
    auto lane_noise_seed = [&]() -> uint64_t {
        if (FAST == 1 || (FAST == 0 && P.mcts.fpu != 2 && P.mcts.noise != 2)) return 0ull;
        const uint64_t stream = P.base_seed + (MODE == MODE_SELFPLAY ? P.first_game : 0ull) + (uint64_t)(uint32_t)T.job;
        return noise_tree_seed(stream, MODE == MODE_SELFPLAY ? (pk[4 * NT] & 0xFFu) : 0u);
    };
    LaneProf lp_store;
    lp_store.row = (PROF && P.prof) ? P.prof + ((size_t)blockIdx.x * NW + wave) * LP_FIELDS : nullptr;
    lp_store.ablate = PROF ? P.debug_stub : 0;
    LaneProf* const lp = (PROF && P.prof) ? &lp_store : nullptr;
    for (;;) {
        const bool active = T.job >= 0;
        if (__ballot(active) == 0ull) break;
        pT = SYN_STAMP();
        LaneLeaf X;
        lane_select_expand<COUNT, FAST>(P.mcts, T, Wk, X, active, pl, bcap, thresh, ctr, P.error, pk, NT, lane_noise_seed(), lp, P.nv, fpu_tbl);
        SYN_LAP(pA)
        // PROF: timeline of the three waves of SIMD 0 of workgroup 0 (rounds 2000..2015): [A end = B start, B end, C end]
        const bool tl = PROF && P.prof && blockIdx.x == 0 && (wave & 3) == 0 && pRounds >= 2000 && pRounds < 2016;
        if (tl && lane == 0) P.prof[PROF_TIMELINE_OFF + ((wave >> 2) * 16 + (pRounds - 2000)) * 3 + 0] = SYN_STAMP();
        // ---- phase B: the lanes that need the network, compacted into tiles of 16 positions. While other lanes are still
        // descending only whole tiles are evaluated: requests beyond `thresh` stay pending and go first next round.
        const bool want_nn = X.at_leaf && X.needs_eval;  // this lane's expanded leaf needs Policy::eval
        float lg[9];
        float v0 = X.p0, v1 = X.p1, v2 = X.p2;  // outcome distribution to back up: a solved leaf's, else the network's
#pragma unroll
        for (int c = 0; c < 9; c++) lg[c] = 0.0f;
        // PolicyWithCache: a position that some game already evaluated skips the network (and its tile slot)
        bool hit = false;
        if (POLICY == 1) {
            // RolloutPolicy: the "evaluation" is a random playout on this lane (no tiles, nothing deferred)
            if (want_nn) lane_rollout(Wk.my, Wk.op, P.base_seed + (unsigned long long)T.job, T.job, T.rng_index, ring, v0, v1, v2);
            hit = want_nn;
        } else if (P.cache != nullptr && want_nn && !X.was_pending) {
            hit = cache_lookup(P.cache, P.cache_shift, Wk.my, Wk.op, lg, v0, v1, v2);
        }
        bool need = want_nn && !hit;
        const unsigned long long want_mask = __ballot(need);
        if (POLICY != 1 && P.cache != nullptr) {  // wave-uniform tallies (scalar registers), flushed once at the end of the kernel
            cache_hits += (unsigned long long)__popcll(__ballot(hit));
            cache_misses += (unsigned long long)__popcll(__ballot(want_nn && !hit && !X.was_pending));
        }
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(want_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want_mask, 0u));
        const int quota = __ballot(Wk.descending) != 0ull ? thresh : 64;
        Wk.pending = need && rank >= quota;
        if (Wk.pending) Wk.pend_lmask = X.legal_mask;
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
#pragma unroll 1
        for (int j = 0; j * 16 < n_need; j++) {
            if (PROF) pTiles++;
            const unsigned long long lp_b0 = lp ? lp_now() : 0ull;
            // everything a tile needs is re-derived here instead of living in registers across the whole matrix phase:
            // the two derived boards (10 VALU) and the lane's feature shift table (one 16-byte LDS read)
            const int src = (int)idxw[16 * j + (lane & 15)];  // (slots past the last request read lane 0: finite input)
            f32x4 o;
            if (POLICY == 2) {
                // Connect4ConvNet reads the two bitplanes themselves
                const uint64_t tmy = shfl_u64(Wk.my, src), top = shfl_u64(Wk.op, src);
                uint32_t img_off = 0;  // opaque per tile: the image reads stay LDS reads next to their MFMAs
                asm volatile("" : "+v"(img_off));
                o = conv_tile16(wimg + img_off, lane, tmy, top);
            } else if (POLICY == 3) {
                uint64_t hi, lo;
                feature_boards(Wk.my, Wk.op, hi, lo);
                const uint64_t thi = shfl_u64(hi, src), tlo = shfl_u64(lo, src);
                uint32_t img_off = 0;  // opaque per tile: the image reads stay LDS reads next to their MFMAs
                asm volatile("" : "+v"(img_off));
                const uint32_t* img16 = reinterpret_cast<const uint32_t*>(smem_raw) + img_off;
                o = f16x2_tile16<3>(img16, lane, thi, tlo);
                const float os = reinterpret_cast<const float*>(img16 + F16Geom::SCALE_WORD0)[4];   // exact power of two
#pragma unroll
                for (int r = 0; r < 4; r++) o[r] *= os;
            } else {
                uint64_t hi, lo;
                feature_boards(Wk.my, Wk.op, hi, lo);
                const uint4 ftw = *reinterpret_cast<const uint4*>(smem_raw + LaneLds<NW>::FT_OFF + (lane >> 4) * 16);
                FeatureTable FT;
                FT.t[0] = ftw.x; FT.t[1] = ftw.y; FT.t[2] = ftw.z; FT.t[3] = ftw.w;
                const uint64_t thi = shfl_u64(hi, src), tlo = shfl_u64(lo, src);
                if (lp_abl(lp, ABL_TILE)) {
                    uint64_t thi2 = thi, tlo2 = tlo;
                    asm volatile("" : "+v"(thi2), "+v"(tlo2));
                    const f32x4 o2 = mlp_tile16_pipe(wimg, bimg, lane, FT, thi2, tlo2);
                    asm volatile("" ::"v"(o2));
                }
                o = mlp_tile16_pipe(wimg, bimg, lane, FT, thi, tlo);   // (the hand-pipelined tile at every wave count: the plain one spills around the crossbar fetches below)
            }
            // (raw outputs: the softmax over the three outcome logits runs per tree lane in phase C, lane_softmaxes)
            const unsigned long long lp_b1 = lp ? lp_now() : 0ull;
            if (lp) lp_add(lp, LP_B_TILE, lp_b1 - lp_b0);
            // The 12 outputs of position (rank & 15) sit in register r of lanes (rank & 15) + 16 q, q = 0..2: every lane fetches
            // "its" twelve values through the LDS crossbar (ds_bpermute: no LDS memory, no barrier) and the lanes whose request was
            // in this tile keep them. (Rounds 1-3 passed them through a 1 KB LDS patch per wave behind two wave barriers per tile.)
            {
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
            if (lp) lp_add(lp, LP_B_SCATTER, lp_now() - lp_b1);
        }

        if (tl && lane == 0) P.prof[PROF_TIMELINE_OFF + ((wave >> 2) * 16 + (pRounds - 2000)) * 3 + 1] = SYN_STAMP();
        SYN_LAP(pB)
        if (PROF) { pRounds++; pLanes += (unsigned long long)__popcll(__ballot(fin)); pEvals += (unsigned long long)n_need; }

        // ---- phase C
        bool solved = X.solved;
        uint32_t leaf_flag = 0, leaf_code = 0;
        if (need || hit) {
            // the leaf's two softmaxes (RolloutPolicy delivers outcome probabilities already), then — with the probabilities —
            // the PolicyWithCache entry of a position the network has just evaluated
            float pr[9];
            const unsigned long long lp_c0 = lp ? lp_now() : 0ull;
            if (lp_abl(lp, ABL_SOFTMAX)) {
                float pr2[9], w0 = v0, w1 = v1, w2 = v2, lg2[9];
#pragma unroll
                for (int c = 0; c < 9; c++) { lg2[c] = lg[c]; asm volatile("" : "+v"(lg2[c])); }
                lane_softmaxes(X.legal_mask, lg2, pr2, POLICY != 1 && need, w0, w1, w2);
#pragma unroll
                for (int c = 0; c < 9; c++) asm volatile("" ::"v"(pr2[c]));
                asm volatile("" ::"v"(w0), "v"(w1), "v"(w2));
            }
            lane_softmaxes(X.legal_mask, lg, pr, POLICY != 1 && need, v0, v1, v2);
            if (lp) lp_step(lp, LP_C_SOFT, -1, LP_C_LANES, lp_now() - lp_c0);
            if (POLICY != 1 && P.cache != nullptr && need) cache_insert(P.cache, P.cache_shift, Wk.my, Wk.op, lg, v0, v1, v2);
            // PolicyNoise::Equal applies to the root's own expansion (mcts.rs:258-269): the first pass of a tree
            const CfgView<FAST> cv{P.mcts};
            leaf_code = lane_write_children(T.slab, Wk.blk, X.legal_mask, Wk.my, Wk.op, pr,
                                         (FAST == 0 && T.iter == 0 && Wk.level == 0) ? P.mcts.noise : 0, P.mcts.noise_weight,
                                         P.mcts.noise_alpha, lane_noise_seed(),
                                         cv.fpu_const() ? cv.fpu_value() : 0.0f, leaf_flag, lp_abl(lp, ABL_CHILD_ST));
            solved = (leaf_code & LEAF_ANY_SOLVED) != 0u;
            if (lp) lp_add(lp, LP_C_WRITE, lp_now() - lp_c0);
        }
        SYN_LAP(pC1)
        unsigned long long tmid = 0;
        lane_backprop<COUNT, FAST, true>(P.mcts, T, Wk.level, v0, v1, v2, solved, fin, pl, ctr, leaf_flag, PROF ? &tmid : nullptr, lp, leaf_code);
        if (PROF) { pC2 += tmid - pT; pT = tmid; }
        SYN_LAP(pC)
        if (fin) {
            T.iter += 1;
            // explore_n (mcts.rs:139-147): the root visit, then up to n explores unless the root is solved
            if (T.iter > n_explores || T.root_solved) {
                // the callee reads the launch parameters from the kernel argument segment; the slab pointer is re-derived
                // afterwards, so the hot loop's pointers never round-trip through memory (they would come back generic: flat_load)
                const KernargPtr Pc = lane_kernarg();
                if (lp) lp_step(lp, LP_M_CALLS, -1, LP_M_LANES, 1ull);
                const unsigned long long lp_m0 = lp ? lp_now() : 0ull;
                SYN_UNPARK();
                if (MODE == MODE_SELFPLAY) {
                    if (PROF && lp) T = lane_move_step_call_prof(Pc, T, lp);
                    else T = lane_move_step_call<COUNT>(Pc, T, ctr);
                }
                else T = lane_search_finish_call(Pc, T);
                if (lp) lp_add(lp, LP_M_TOTAL, lp_now() - lp_m0);
                T.slab = reinterpret_cast<unsigned char*>(P.stat) + slot * (size_t)P.cap * 32u;
                SYN_PARK();
            }
        }
        if (tl && lane == 0) P.prof[PROF_TIMELINE_OFF + ((wave >> 2) * 16 + (pRounds - 1 - 2000)) * 3 + 2] = SYN_STAMP();
        SYN_LAP(pM)
    }
#undef SYN_STAMP
#undef SYN_LAP
#undef SYN_PARK
#undef SYN_UNPARK
    if (PROF) {
        if (P.prof && lane == 0) {
            unsigned long long* o = P.prof + ((size_t)blockIdx.x * NW + wave) * LP_FIELDS;
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

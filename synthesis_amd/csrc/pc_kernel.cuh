// synthesis_amd — producer/consumer self-play / search kernel (the headline launch shape from 393,216 concurrent games).
//
// Same algorithm, node blocks, path log and per-tree code as the lane-per-tree kernel (lane_kernel.cuh):
//   synthesis/src/mcts.rs:310-488       explore / select_best_child / visit / backprop
//   synthesis/src/alpha_zero.rs:229-338 run_game / sample_action / fill_state_info / store_rewards
// What changes is WHO runs the network. In the lane kernel every wave alternates tree phases (dependent memory round
// trips, ~107k cycles per round) with its own matrix phase (~41k cycles of MFMA), and the matrix pipe of a SIMD idles
// whenever all of its waves are in their tree phases at once (60 % of the f32 MFMA roof with three waves per SIMD). Here
// a 16-wave workgroup (one per CU) splits into
//   * 4 MATRIX waves, one per SIMD (chosen by HW_ID.simd_id): nothing but mlp_tile16 on 16-position tiles popped from an
//     LDS ring, at raised priority. A matrix wave issues MFMAs back to back as long as the ring is not empty.
//   * 12 TREE waves, each time-slicing NV "virtual waves" of 64 trees (one tree per lane). One visit of a virtual wave =
//         unpark its per-lane state (19 dwords, coalesced rows in global memory, L2-resident)
//         harvest the network outputs of the leaves it submitted on its previous visit
//         phase C  children + priors, backprop from the path log, move step when a search ends      (lane_kernel.cuh)
//         phase A  select + expand of the next explore                                              (lane_kernel.cuh)
//         submit   feature boards of the new leaves -> LDS, one ring entry per 16-position tile
//         park
//     so a tree wave never waits for its own tiles: while virtual wave k's leaves are with the matrix waves it runs
//     virtual wave k+1. Trees per CU = 768 * NV; the node pool is 45 GB per NV (this is what 288 GB of HBM3E is for).
//
// Hand-offs (all inside one workgroup = one CU, so LDS words order everything; no other workgroup is ever involved):
//   tree -> matrix : boards[vw][rank] (LDS, 16 B per position), then ONE 64-bit LDS store {ticket + 1, descriptor} into
//                    ring[ticket % 256]; tickets come from an LDS fetch-add; LDS operations of a wave execute in order.
//   matrix -> tree : outputs (12 floats per position, raw; the value softmax runs on the tree side) to global memory
//                    outs[vw][rank], s_waitcnt vmcnt(0) one tile LATER (the store was issued ~15k cycles earlier, so the
//                    wait is free), then an LDS fetch-add on done[vw]; the tree wave polls done[vw] == submitted[vw]
//                    before it loads. Same CU = same vector L1: plain loads see the stores (workgroup scope).
// Results depend only on the game index: nothing here reads a clock or a slot id into an arithmetic result.
#pragma once
#include "lane_kernel.cuh"

namespace syn {

struct PcGeom {
    static constexpr int TREE_WAVES = 12;
    static constexpr int MAT_WAVES = 4;
    static constexpr int NT = 64 * (TREE_WAVES + MAT_WAVES);
    static constexpr int NV_MAX = 3;                                 // LDS board slots: 36 KB
    static constexpr int STATE_WORDS = 19;
    static constexpr size_t VW_OUTS_OFF = (size_t)STATE_WORDS * 64 * 4;   // outs[64][12] f32, rank-indexed
    static constexpr size_t VW_HITS_OFF = VW_OUTS_OFF + 64 * 48;          // hits[64][12] f32, lane-indexed (policy cache)
    static constexpr size_t VW_BYTES = VW_HITS_OFF + 64 * 48;             // 10,752 B per virtual wave
    static constexpr int RING = 256;                                  // >= 36 virtual waves x 4 tiles in flight
    static constexpr uint32_t POISON = 0xFFFFFFFFu;
};

struct PcLds {
    static constexpr size_t FT_OFF = (size_t)MlpGeom::IMG_FLOATS * 4;                       // 123,264 B weight + bias image
    static constexpr size_t BOARD_OFF = FT_OFF + 64;                                        // [vw local][rank] 16 B
    static constexpr size_t RING_OFF = BOARD_OFF + (size_t)PcGeom::TREE_WAVES * PcGeom::NV_MAX * 1024;
    static constexpr size_t CTRL_OFF = RING_OFF + (size_t)PcGeom::RING * 8;
    static constexpr size_t BYTES = CTRL_OFF + 512;                                         // 162,752 B of 163,840
    // control words (uint32 index)
    enum { C_HEAD = 0, C_TAIL = 1, C_EXITED = 2, C_TREE_TK = 3, C_ABORT = 4, C_ROLE = 8, C_DONE = 16, C_SUBM = 56, C_WORDS = 96 };
};
static_assert(PcLds::BYTES <= 163840, "LDS budget of one CU");

SYN_DEV uint32_t pc_lds_add(uint32_t* p, uint32_t v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
SYN_DEV uint32_t pc_lds_ld(const uint32_t* p) { return *reinterpret_cast<const volatile uint32_t*>(p); }

// per-lane state of one tree between two visits of its virtual wave
struct PcFlags {
    bool fin, need, hit, xsolved;
    uint32_t legal_mask;
};

template <int MODE, bool COUNT, bool FAST, bool PROF = false>
__global__ __launch_bounds__(PcGeom::NT, 1) void selfplay_kernel_pc(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* wimg = reinterpret_cast<float*>(smem_raw);
    const float* bimg = wimg + MlpGeom::W_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    uint32_t* const ctrl = reinterpret_cast<uint32_t*>(smem_raw + PcLds::CTRL_OFF);
    unsigned long long* const ring = reinterpret_cast<unsigned long long*>(smem_raw + PcLds::RING_OFF);
    uint4* const boards = reinterpret_cast<uint4*>(smem_raw + PcLds::BOARD_OFF);

    stage_weight_image(wimg, P.wimg, tid, PcGeom::NT);
    if (tid < 4) {
        const FeatureTable f = make_feature_table(tid);
        *reinterpret_cast<uint4*>(smem_raw + PcLds::FT_OFF + tid * 16) = make_uint4(f.t[0], f.t[1], f.t[2], f.t[3]);
    }
    if (tid < PcLds::C_WORDS) ctrl[tid] = 0;
    if (tid < PcGeom::RING) ring[tid] = 0ull;
    __syncthreads();
    // ---- roles: the first wave to arrive on each SIMD becomes that SIMD's matrix wave (HW_ID.simd_id = bits 5:4)
    const uint32_t simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
    uint32_t role_tk = 0;
    if (lane == 0) role_tk = pc_lds_add(&ctrl[PcLds::C_ROLE + (simd & 3u)], 1u);
    role_tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)role_tk);
    int ti = PcGeom::TREE_WAVES;
    if (role_tk != 0u) {
        uint32_t t = 0;
        if (lane == 0) t = pc_lds_add(&ctrl[PcLds::C_TREE_TK], 1u);
        ti = __builtin_amdgcn_readfirstlane((int)t);
    }
    // (a workgroup that was not spread 4/4/4/4 over the SIMDs still ends up with 12 tree waves and 4 matrix waves)
    const bool is_matrix = ti >= PcGeom::TREE_WAVES;
    const int NV = P.nv;
    const size_t vw_block0 = (size_t)blockIdx.x * PcGeom::TREE_WAVES * (size_t)NV;  // first virtual wave of this workgroup
    unsigned long long pWait = 0, pBusy = 0, pTiles = 0, pA = 0, pC = 0, pRounds = 0, pFin = 0;
#define PC_STAMP() (PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull)

    if (is_matrix) {
        // =========================================================================================== matrix wave
        // The matrix wave is throughput work that always has an instruction ready; the tree waves are latency-bound chains
        // of short VALU bursts between memory round trips, and an f32 MFMA in flight blocks the SIMD's VALU port for more
        // than half of its 32 cycles (tools/ubench/mfma_valu_overlap.hip: a VALU wave runs 2.4x slower beside a saturated
        // matrix pipe). So the TREE waves get the higher priority and the matrix wave takes the issue slots they leave.
        if (P.debug_prio & 1) __builtin_amdgcn_s_setprio(3);
        const int q = lane >> 4;
        auto claim = [&]() {
            uint32_t t = 0;
            if (lane == 0) t = pc_lds_add(&ctrl[PcLds::C_HEAD], 1u);
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        };
        int prev_vwl = -1;  // the tile whose output store is still in flight (not yet signalled)
        auto signal_prev = [&]() {
            if (prev_vwl >= 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // issued a whole tile ago: already drained
                if (lane == 0) pc_lds_add(&ctrl[PcLds::C_DONE + prev_vwl], 1u);
                prev_vwl = -1;
            }
        };
        uint32_t tk = claim();
        for (;;) {
            unsigned long long t0 = PC_STAMP();
            unsigned long long e;
            uint32_t spins = 0;
            for (;;) {
                e = *reinterpret_cast<const volatile unsigned long long*>(&ring[tk & (PcGeom::RING - 1)]);
                if ((uint32_t)(e >> 32) == tk + 1u) break;
                signal_prev();  // the ring is empty: nobody should wait for a tile that is already computed
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 26) || pc_lds_ld(&ctrl[PcLds::C_ABORT]) != 0u) {
                    if (lane == 0) { *P.error = 3; ctrl[PcLds::C_ABORT] = 1u; }
                    e = (unsigned long long)PcGeom::POISON;
                    break;
                }
            }
            const uint32_t desc = (uint32_t)e;
            if (desc == PcGeom::POISON) break;
            if (PROF) { unsigned long long t1 = PC_STAMP(); pWait += t1 - t0; t0 = t1; }
            const int vwl = (int)(desc & 63u), j = (int)((desc >> 6) & 3u);
            const uint32_t tk_next = claim();
            // the lane's feature shift table is re-read per tile (one 16-byte LDS read) so that nothing derived from it
            // is hoisted out of the loop into long-lived registers
            int ft_off = q * 16;
            asm volatile("" : "+v"(ft_off));  // opaque OFFSET (an opaque or volatile pointer would degrade to flat_load)
            const uint4 ftw = *reinterpret_cast<const uint4*>(smem_raw + PcLds::FT_OFF + ft_off);
            FeatureTable FT;
            FT.t[0] = ftw.x; FT.t[1] = ftw.y; FT.t[2] = ftw.z; FT.t[3] = ftw.w;
            const uint4 b = boards[vwl * 64 + 16 * j + (lane & 15)];
            const uint64_t hi = (uint64_t)b.x | ((uint64_t)b.y << 32), lo = (uint64_t)b.z | ((uint64_t)b.w << 32);
            f32x4 o;
            if (PROF && P.debug_stub) {
                // diagnostic only (SYN_DEBUG=1 SYN_PROFILE=1 SYN_PC_STUB=1): no network, cheap position-dependent outputs —
                // measures what the tree waves alone sustain
                const uint32_t hsh = (uint32_t)(hi * 0x9E3779B97F4A7C15ull >> 40) + (uint32_t)q * 977u;
                o = f32x4{(float)(hsh & 7u) * 0.25f, (float)((hsh >> 3) & 7u) * 0.25f, (float)((hsh >> 6) & 7u) * 0.25f,
                          (float)((hsh >> 9) & 7u) * 0.25f};
            } else {
                o = mlp_tile16_pipe(wimg, bimg, lane, FT, hi, lo);
            }
            signal_prev();
            unsigned char* vwb = P.vw_buf + (vw_block0 + (size_t)vwl) * PcGeom::VW_BYTES;
            if (q < 3) *reinterpret_cast<f32x4*>(vwb + PcGeom::VW_OUTS_OFF + (size_t)(16 * j + (lane & 15)) * 48 + q * 16) = o;
            prev_vwl = vwl;
            tk = tk_next;
            if (PROF) { pBusy += PC_STAMP() - t0; pTiles++; }
        }
        signal_prev();
        if (PROF && P.prof && lane == 0) {
            unsigned long long* o = P.prof + ((size_t)blockIdx.x * 16 + (tid >> 6)) * 8;
            o[0] = 1; o[1] = pWait; o[2] = pBusy; o[3] = pTiles;
        }
        return;
    }

    // =============================================================================================== tree wave
    if (P.debug_prio & 2) __builtin_amdgcn_s_setprio(3);
    uint32_t ctr[COUNT ? CTR_COUNT : 1];
#pragma unroll
    for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) ctr[i] = 0;
    unsigned long long cache_hits = 0, cache_misses = 0;
    const uint32_t bcap = P.cap / 4u;
    const int n_explores = P.roll.num_explores;
    const int thresh = P.lane_thresh;

    LaneTree T;
    LaneWalk Wk;
    PcFlags F;
#define PC_PARK(st)                                                                                                   \
    do {                                                                                                              \
        (st)[0 * 64] = (uint32_t)T.root_my; (st)[1 * 64] = (uint32_t)(T.root_my >> 32);                               \
        (st)[2 * 64] = (uint32_t)T.root_op; (st)[3 * 64] = (uint32_t)(T.root_op >> 32);                               \
        (st)[4 * 64] = (uint32_t)T.turn | (T.rng_index << 8);                                                         \
        (st)[5 * 64] = (uint32_t)T.job;                                                                               \
        (st)[6 * 64] = T.next_block;                                                                                  \
        (st)[7 * 64] = T.num_nodes;                                                                                   \
        (st)[8 * 64] = (uint32_t)T.iter | ((T.root_solved ? 1u : 0u) << 16) | (T.root_sol << 17);                     \
        (st)[9 * 64] = Wk.rec;                                                                                        \
        (st)[10 * 64] = Wk.blk | ((uint32_t)Wk.level << 14) | ((Wk.descending ? 1u : 0u) << 21) |                     \
                        ((Wk.pending ? 1u : 0u) << 22) | ((Wk.solved ? 1u : 0u) << 23) | (Wk.kind << 24);             \
        (st)[11 * 64] = Wk.qt;                                                                                        \
        (st)[12 * 64] = f32_bits(Wk.pN);                                                                              \
        (st)[13 * 64] = (uint32_t)Wk.my; (st)[14 * 64] = (uint32_t)(Wk.my >> 32);                                     \
        (st)[15 * 64] = (uint32_t)Wk.op; (st)[16 * 64] = (uint32_t)(Wk.op >> 32);                                     \
        (st)[17 * 64] = (F.fin ? 1u : 0u) | ((F.need ? 1u : 0u) << 1) | ((F.hit ? 1u : 0u) << 2) |                    \
                        ((F.xsolved ? 1u : 0u) << 3) | (F.legal_mask << 4) | (Wk.pend_lmask << 13);                   \
        (st)[18 * 64] = T.fpu_draws;                                                                                  \
    } while (0)
#define PC_UNPARK(st)                                                                                                 \
    do {                                                                                                              \
        uint32_t w_[PcGeom::STATE_WORDS];                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < PcGeom::STATE_WORDS; i_++) w_[i_] = (st)[i_ * 64];                     \
        T.root_my = (uint64_t)w_[0] | ((uint64_t)w_[1] << 32);                                                        \
        T.root_op = (uint64_t)w_[2] | ((uint64_t)w_[3] << 32);                                                        \
        T.turn = (int)(w_[4] & 0xFFu); T.rng_index = w_[4] >> 8;                                                      \
        T.job = (int)w_[5]; T.next_block = w_[6]; T.num_nodes = w_[7];                                                \
        T.iter = (int)(w_[8] & 0xFFFFu); T.root_solved = ((w_[8] >> 16) & 1u) != 0u; T.root_sol = w_[8] >> 17;        \
        Wk.rec = w_[9];                                                                                               \
        Wk.blk = w_[10] & 0x3FFFu; Wk.level = (int)((w_[10] >> 14) & 0x7Fu);                                          \
        Wk.descending = ((w_[10] >> 21) & 1u) != 0u; Wk.pending = ((w_[10] >> 22) & 1u) != 0u;                        \
        Wk.solved = ((w_[10] >> 23) & 1u) != 0u; Wk.kind = (w_[10] >> 24) & 3u;                                       \
        Wk.qt = w_[11]; Wk.pN = bits_f32(w_[12]);                                                                     \
        Wk.my = (uint64_t)w_[13] | ((uint64_t)w_[14] << 32);                                                          \
        Wk.op = (uint64_t)w_[15] | ((uint64_t)w_[16] << 32);                                                          \
        F.fin = (w_[17] & 1u) != 0u; F.need = ((w_[17] >> 1) & 1u) != 0u; F.hit = ((w_[17] >> 2) & 1u) != 0u;         \
        F.xsolved = ((w_[17] >> 3) & 1u) != 0u; F.legal_mask = (w_[17] >> 4) & 0x1FFu;                                \
        Wk.pend_lmask = (w_[17] >> 13) & 0x1FFu;                                                                      \
        T.fpu_draws = w_[18];                                                                                         \
    } while (0)

    // every virtual wave starts with 64 fresh trees (first visit: nothing to harvest)
    for (int k = 0; k < NV; k++) {
        lane_start_job<MODE>(P, T);
        Wk.descending = false; Wk.pending = false; Wk.pend_lmask = 0;
        Wk.rec = REC_ROOT; Wk.blk = 0; Wk.solved = false; Wk.kind = 0; Wk.qt = 0; Wk.pN = 0.0f; Wk.my = 0; Wk.op = 0; Wk.level = 0;
        F.fin = false; F.need = false; F.hit = false; F.xsolved = false; F.legal_mask = 0;
        uint32_t* st = reinterpret_cast<uint32_t*>(P.vw_buf + (vw_block0 + (size_t)(ti * NV + k)) * PcGeom::VW_BYTES) + lane;
        PC_PARK(st);
    }

    auto pc_noise_seed = [&]() -> uint64_t {  // this lane's current tree (noise.cuh); runtime-switched configurations only
        if (FAST || (P.mcts.fpu != 2 && P.mcts.noise != 2)) return 0ull;
        const uint64_t stream = P.base_seed + (MODE == MODE_SELFPLAY ? P.first_game : 0ull) + (uint64_t)(uint32_t)T.job;
        return noise_tree_seed(stream, MODE == MODE_SELFPLAY ? (uint32_t)T.turn : 0u);
    };
    uint32_t dead = 0;
    int alive = NV;
    int k = 0;
    bool aborted = false;
    while (alive > 0 && !aborted) {
        if ((dead >> k) & 1u) { k = k + 1 == NV ? 0 : k + 1; continue; }
        const int vwl = ti * NV + k;
        const size_t gvw = vw_block0 + (size_t)vwl;
        unsigned char* const vwb = P.vw_buf + gvw * PcGeom::VW_BYTES;
        uint32_t* const st = reinterpret_cast<uint32_t*>(vwb) + lane;
        PC_UNPARK(st);
        const size_t slot = gvw * 64 + (size_t)lane;
        T.slab = reinterpret_cast<unsigned char*>(P.stat) + slot * (size_t)P.cap * 32u;
        uint4* const pl = P.path + gvw * PATH_ENTRIES + (size_t)lane;
        unsigned long long tA = PC_STAMP();

        // ---- harvest + phase C for the explores this virtual wave finished on its previous visit
        if (__ballot(F.fin) != 0ull) {
            float lg[9];
#pragma unroll
            for (int c = 0; c < 9; c++) lg[c] = 0.0f;
            // a solved leaf backs up its one-hot outcome (mcts.rs:377-379), everything else the network's distribution
            float v0 = (F.xsolved && Wk.kind == 0u) ? 1.0f : 0.0f, v1 = (F.xsolved && Wk.kind == 1u) ? 1.0f : 0.0f,
                  v2 = (F.xsolved && Wk.kind == 2u) ? 1.0f : 0.0f;
            const unsigned long long need_mask = __ballot(F.need);
            if (need_mask != 0ull) {
                const uint32_t want = pc_lds_ld(&ctrl[PcLds::C_SUBM + vwl]);
                uint32_t spins = 0;
                while (pc_lds_ld(&ctrl[PcLds::C_DONE + vwl]) != want) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1u << 25) || pc_lds_ld(&ctrl[PcLds::C_ABORT]) != 0u) {
                        if (lane == 0) { *P.error = 3; ctrl[PcLds::C_ABORT] = 1u; }
                        aborted = true;
                        break;
                    }
                }
                if (aborted) break;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            if (PROF) { unsigned long long t1 = PC_STAMP(); pWait += t1 - tA; tA = t1; }
            if (F.need || F.hit) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(need_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need_mask, 0u));
                const float* src = reinterpret_cast<const float*>(vwb + (F.need ? PcGeom::VW_OUTS_OFF + (size_t)rank * 48
                                                                                : PcGeom::VW_HITS_OFF + (size_t)lane * 48));
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(src);
                const f32x4 r1 = *reinterpret_cast<const f32x4*>(src + 4);
                const f32x4 r2 = *reinterpret_cast<const f32x4*>(src + 8);
                lg[0] = r0[0]; lg[1] = r0[1]; lg[2] = r0[2]; lg[3] = r0[3];
                lg[4] = r1[0]; lg[5] = r1[1]; lg[6] = r1[2]; lg[7] = r1[3];
                lg[8] = r2[0]; v0 = r2[1]; v1 = r2[2]; v2 = r2[3];
                if (F.need) {
                    value_softmax(v0, v1, v2);  // policies.rs:54-57 (the matrix waves deliver the raw outcome logits)
                    if (P.cache != nullptr) cache_insert(P.cache, P.cache_shift, Wk.my, Wk.op, lg, v0, v1, v2);
                }
            }
            bool solved = F.xsolved;
            uint32_t leaf_flag = 0;
            if (F.need || F.hit) {
                const CfgView<FAST> cv{P.mcts};
                LaneLeaf Xc;
                Xc.legal_mask = F.legal_mask;
                solved = lane_create_children(T.slab, Wk.blk, Xc, Wk.my, Wk.op, lg,
                                              (!FAST && T.iter == 0 && Wk.level == 0) ? P.mcts.noise : 0, P.mcts.noise_weight,
                                              P.mcts.noise_alpha, pc_noise_seed(),
                                              cv.fpu_const() ? cv.fpu_value() : 0.0f, leaf_flag);
            }
            lane_backprop<COUNT, FAST>(P.mcts, T, Wk.level, v0, v1, v2, solved, F.fin, pl, ctr, leaf_flag);
            if (F.fin) {
                T.iter += 1;
                if (PROF) pFin++;
                // explore_n (mcts.rs:139-147): the root visit, then up to n explores unless the root is solved
                if (T.iter > n_explores || T.root_solved) {
                    const KernargPtr Pc = lane_kernarg();
                    if (MODE == MODE_SELFPLAY) T = lane_move_step_call<COUNT>(Pc, T, ctr);
                    else T = lane_search_finish_call(Pc, T);
                    T.slab = reinterpret_cast<unsigned char*>(P.stat) + slot * (size_t)P.cap * 32u;
                }
            }
            if (PROF) { unsigned long long t1 = PC_STAMP(); pC += t1 - tA; tA = t1; }
        }

        const bool active = T.job >= 0;
        if (__ballot(active) == 0ull) {  // every tree of this virtual wave has run out of jobs
            dead |= 1u << k;
            alive--;
            k = k + 1 == NV ? 0 : k + 1;
            continue;
        }

        // ---- phase A: the next explore of every tree (lanes still on their way down simply continue)
        LaneLeaf X;
        lane_select_expand<COUNT, FAST, true>(P.mcts, T, Wk, X, active, pl, bcap, thresh, ctr, P.error, nullptr, 0, pc_noise_seed());

        // ---- submit: the leaves that need Policy::eval, compacted by rank into 16-position tiles
        const bool want_nn = X.at_leaf && X.needs_eval;
        bool hit = false;
        if (P.cache != nullptr && want_nn && !X.was_pending) {
            float lg[9], c0, c1, c2;
            hit = cache_lookup(P.cache, P.cache_shift, Wk.my, Wk.op, lg, c0, c1, c2);
            if (hit) {
                float* dst = reinterpret_cast<float*>(vwb + PcGeom::VW_HITS_OFF + (size_t)lane * 48);
                *reinterpret_cast<f32x4*>(dst) = f32x4{lg[0], lg[1], lg[2], lg[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{lg[4], lg[5], lg[6], lg[7]};
                *reinterpret_cast<f32x4*>(dst + 8) = f32x4{lg[8], c0, c1, c2};
            }
        }
        bool need = want_nn && !hit;
        const unsigned long long want_mask = __ballot(need);
        if (P.cache != nullptr) {
            cache_hits += (unsigned long long)__popcll(__ballot(hit));
            cache_misses += (unsigned long long)__popcll(__ballot(want_nn && !hit && !X.was_pending));
        }
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(want_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want_mask, 0u));
        const int quota = __ballot(Wk.descending) != 0ull ? thresh : 64;
        Wk.pending = need && rank >= quota;
        if (Wk.pending) Wk.pend_lmask = X.legal_mask;
        need = need && rank < quota;
        if (COUNT && (need || hit)) ctr[CTR_POLICY_EVALS]++;
        const int n_need = __popcll(__ballot(need));
        const int ntiles = (n_need + 15) >> 4;
        if (need) {
            uint64_t hi, lo;
            feature_boards(Wk.my, Wk.op, hi, lo);
            boards[vwl * 64 + rank] = make_uint4((uint32_t)hi, (uint32_t)(hi >> 32), (uint32_t)lo, (uint32_t)(lo >> 32));
        }
        asm volatile("" ::: "memory");  // LDS operations of a wave execute in issue order: boards before ring entries
        if (lane == 0 && ntiles > 0) ctrl[PcLds::C_SUBM + vwl] = pc_lds_ld(&ctrl[PcLds::C_SUBM + vwl]) + (uint32_t)ntiles;
        if (lane < ntiles) {
            const uint32_t t = pc_lds_add(&ctrl[PcLds::C_TAIL], 1u);
            *reinterpret_cast<volatile unsigned long long*>(&ring[t & (PcGeom::RING - 1)]) =
                ((unsigned long long)(t + 1u) << 32) | (unsigned long long)((uint32_t)vwl | ((uint32_t)lane << 6));
        }
        F.fin = X.at_leaf && !Wk.pending;
        F.need = need;
        F.hit = hit;
        F.xsolved = X.solved;
        F.legal_mask = X.legal_mask;
        PC_PARK(st);
        if (PROF) { pA += PC_STAMP() - tA; pRounds++; }
        k = k + 1 == NV ? 0 : k + 1;
    }
#undef PC_PARK
#undef PC_UNPARK

    // ---- exit: the last tree wave to leave sends one poison entry per matrix wave
    if (lane == 0) {
        const uint32_t n_out = pc_lds_add(&ctrl[PcLds::C_EXITED], 1u);
        if (n_out == (uint32_t)PcGeom::TREE_WAVES - 1u || aborted) {
            for (int i = 0; i < PcGeom::MAT_WAVES; i++) {
                const uint32_t t = pc_lds_add(&ctrl[PcLds::C_TAIL], 1u);
                *reinterpret_cast<volatile unsigned long long*>(&ring[t & (PcGeom::RING - 1)]) =
                    ((unsigned long long)(t + 1u) << 32) | (unsigned long long)PcGeom::POISON;
            }
        }
    }
    if (PROF && P.prof && lane == 0) {
        unsigned long long* o = P.prof + ((size_t)blockIdx.x * 16 + (tid >> 6)) * 8;
        o[0] = 2; o[1] = pWait; o[2] = pA; o[3] = pC; o[4] = pRounds; o[5] = pFin;
    }
#undef PC_STAMP
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

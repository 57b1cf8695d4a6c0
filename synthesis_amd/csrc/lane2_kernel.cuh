// synthesis_amd — lane-per-tree kernel with TWO trees per lane (the launch shape of the headline benchmark).
//
// Same algorithm, node blocks, path log and results as lane_kernel.cuh — the reference's
//   synthesis/src/mcts.rs:310-488       explore / select_best_child / visit / backprop
//   synthesis/src/alpha_zero.rs:229-338 run_game / sample_action / fill_state_info / store_rewards
// — and the same device functions for every step. What changes is who waits for whom.
//
// With one tree per lane a round of a wave is A (descend until 48 of the 64 lanes stand on a leaf) -> B (network tiles) ->
// C (children, backprop). A lane that reaches its leaf in the first iteration of A idles until the 48th lane arrives: the
// descent loop — more than 40 % of the kernel's vector instructions and all of its dependent memory round trips per explore —
// ran at roughly 40 % lane occupancy. Here every lane owns two trees ("contexts"). One of them is *current* and is what the
// descent loop works on; as soon as it stands on a leaf (expanded, waiting for the network or for backprop) the lane switches
// to its other tree, if that one can descend. So the loop's iterations are spent on lanes that all have a level to do, a round
// needs fewer of them (and fewer memory round trips) per explore, and phases B / C still see one finished tree per lane and
// round — the round ends once `thresh` lanes hold a tree that waits for the network. A tree that waits is served oldest first.
//
// The other tree's registers (18 dwords) are the price: the shape is built for 8 or 12 waves per workgroup (2 or 3 per SIMD: 256 /
// 168 VGPRs), i.e. 1,024 or 1,536 trees per CU. Results never depend on which slot or context
// plays a game (trees share nothing), so every parity test of the lane kernel holds this kernel to the oracle unchanged.
#pragma once
#include "lane_kernel.cuh"

namespace syn {

template <int NW>
struct Lane2Lds {
    static constexpr size_t OUT_OFF = (size_t)MlpGeom::IMG_FLOATS * 4;     // 123,264 B weight + bias image
    // + 768 B result patch per wave (16 positions x 12 outputs); its first 64 bytes double as the wave's compaction index (rank ->
    //   lane), which every lane reads into a register before the first tile's results overwrite it
    static constexpr size_t FT_OFF = OUT_OFF + (size_t)NW * 768;
    static constexpr size_t PARK_OFF = FT_OFF + 64;                       // + the four feature shift tables (16 B each)
    static constexpr size_t BYTES = PARK_OFF + (size_t)NW * 64 * 2 * 20;  // + 5 parked dwords per lane and context
    static_assert(BYTES <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");
};

// what a context is doing
enum : uint32_t {
    CS_IDLE = 0,    // no game / root left for it
    CS_READY = 1,   // between two explores of its tree: the next one starts at the root
    CS_DESC = 2,    // on its way down (the cursor is valid)
    CS_EVAL = 3,    // stands on an expanded leaf that needs Policy::eval
    CS_MISSED = 4,  // ... and already missed the policy cache / did not fit an earlier round's whole tiles
    CS_SOLVED = 5,  // stands on a solved leaf: backprop only
};

// The registers of the context that is not current.
struct Lane2Other {
    uint32_t st;
    uint32_t next_block, num_nodes, root_sol /* | root_solved << 31 */, fpu_draws;
    int iter, job;
    uint32_t rec, blk, kind /* | nsolved << 2 */, qt;
    float pN;
    uint64_t my, op;
    int level;
};

template <class X>
SYN_DEV void swap_if(bool c, X& a, X& b) {
    const X t = a;
    a = c ? b : a;
    b = c ? t : b;
}

// POLICY: 0 = Connect4Net, 2 = Connect4ConvNet (RolloutPolicy searches stay with the one-tree-per-lane kernel)
template <int MODE, bool COUNT, bool FAST, int NW, int POLICY = 0, int TILE = 0>
__global__ __launch_bounds__(64 * NW, 1) void selfplay_kernel_lanes2(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = 64 * NW;
    static_assert(POLICY == 0 || POLICY == 2, "network policies only");
    float* wimg = reinterpret_cast<float*>(smem_raw);
    const float* bimg = wimg + MlpGeom::W_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    float* outw = reinterpret_cast<float*>(smem_raw + Lane2Lds<NW>::OUT_OFF) + wave * 192;

    if (POLICY == 0) stage_weight_image(wimg, P.wimg, tid, NT);
    if (POLICY == 2) stage_conv_image(wimg, P.wimg, tid, NT);
    if (tid < 4) {
        const FeatureTable f = make_feature_table(tid);
        *reinterpret_cast<uint4*>(smem_raw + Lane2Lds<NW>::FT_OFF + tid * 16) = make_uint4(f.t[0], f.t[1], f.t[2], f.t[3]);
    }

    uint32_t ctr[COUNT ? CTR_COUNT : 1];
#pragma unroll
    for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) ctr[i] = 0;

    // context c of this lane: tree slab 2 * slot + c, path column c of its wave's pair of path buffers, parked dwords at pk0 + c * 5 NT
    const size_t slot = (size_t)blockIdx.x * NT + (size_t)tid;
    const size_t slab_bytes = (size_t)P.cap * 32u;
    unsigned char* const slab0 = reinterpret_cast<unsigned char*>(P.stat) + slot * 2u * slab_bytes;
    uint4* const pl0 = P.path + ((size_t)blockIdx.x * NW + (size_t)wave) * (2u * PATH_ENTRIES) + (size_t)lane;
    uint32_t* const pk0 = reinterpret_cast<uint32_t*>(smem_raw + Lane2Lds<NW>::PARK_OFF) + tid;
    const uint32_t bcap = P.cap / 4u;  // 128-byte blocks in a slab

    // Cold per-context state lives in LDS between the ends of searches (5 dwords: the root position and the game's turn / RNG
    // position), as in lane_kernel.cuh.
#define SYN_PARK(pk_)                                                                                        \
    do {                                                                                                     \
        (pk_)[0] = (uint32_t)T.root_my; (pk_)[NT] = (uint32_t)(T.root_my >> 32);                             \
        (pk_)[2 * NT] = (uint32_t)T.root_op; (pk_)[3 * NT] = (uint32_t)(T.root_op >> 32);                    \
        (pk_)[4 * NT] = (uint32_t)T.turn | (T.rng_index << 8);                                               \
        T.root_my = 0; T.root_op = 0; T.turn = 0; T.rng_index = 0;                                           \
    } while (0)
#define SYN_UNPARK(pk_)                                                                                      \
    do {                                                                                                     \
        T.root_my = (uint64_t)(pk_)[0] | ((uint64_t)(pk_)[NT] << 32);                                        \
        T.root_op = (uint64_t)(pk_)[2 * NT] | ((uint64_t)(pk_)[3 * NT] << 32);                               \
        T.turn = (int)((pk_)[4 * NT] & 0xFFu);                                                               \
        T.rng_index = (pk_)[4 * NT] >> 8;                                                                    \
    } while (0)

    LaneTree T;      // the current context's tree ...
    LaneCursor C;    // ... and its descent
    uint32_t st;
    Lane2Other O;
    uint32_t cx = 0;  // which context is current
    {
        // context 1 first (it becomes the other one), then context 0
        T.slab = slab0 + slab_bytes;
        lane_start_job<MODE>(P, T);
        SYN_PARK(pk0 + 5 * NT);
        O.st = T.job >= 0 ? CS_READY : CS_IDLE;
        O.next_block = 0; O.num_nodes = 0; O.root_sol = 0; O.fpu_draws = 0; O.iter = 0; O.job = T.job;
        O.rec = REC_ROOT; O.blk = 0; O.kind = 0; O.qt = 0; O.pN = 0.0f; O.my = 0; O.op = 0; O.level = 0;
        T.slab = slab0;
        lane_start_job<MODE>(P, T);
        SYN_PARK(pk0);
        st = T.job >= 0 ? CS_READY : CS_IDLE;
        C.rec = REC_ROOT; C.blk = 0; C.kind = 0; C.qt = 0; C.nsolved = false; C.pN = 0.0f; C.my = 0; C.op = 0; C.level = 0;
    }
    uint4* pl = pl0;
    uint32_t* pk = pk0;
    __syncthreads();  // the only workgroup barrier: weights staged. From here on every wave free-runs.

    const int n_explores = P.roll.num_explores;
    const int thresh = P.lane_thresh;
    unsigned char* const idxw = reinterpret_cast<unsigned char*>(outw);  // compaction: byte (rank & 15) * 4 + (rank >> 4) = lane of that rank
    unsigned long long cache_hits = 0, cache_misses = 0;

    // the lanes with `c` exchange their two contexts
    auto swap_contexts = [&](bool c) {
        swap_if(c, st, O.st);
        swap_if(c, T.next_block, O.next_block);
        swap_if(c, T.num_nodes, O.num_nodes);
        uint32_t rs = T.root_sol | ((T.root_solved ? 1u : 0u) << 31);
        swap_if(c, rs, O.root_sol);
        T.root_sol = rs & 0x7FFFFFFFu;
        T.root_solved = (rs >> 31) != 0u;
        if (!FAST) swap_if(c, T.fpu_draws, O.fpu_draws);
        swap_if(c, T.iter, O.iter);
        swap_if(c, T.job, O.job);
        swap_if(c, C.rec, O.rec);
        swap_if(c, C.blk, O.blk);
        uint32_t kq = C.kind | ((C.nsolved ? 1u : 0u) << 2);
        swap_if(c, kq, O.kind);
        C.kind = kq & 3u;
        C.nsolved = (kq >> 2) != 0u;
        swap_if(c, C.qt, O.qt);
        swap_if(c, C.pN, O.pN);
        swap_if(c, C.my, O.my);
        swap_if(c, C.op, O.op);
        swap_if(c, C.level, O.level);
        cx ^= c ? 1u : 0u;
        T.slab = slab0 + (cx != 0u ? slab_bytes : (size_t)0);
        pl = pl0 + (cx != 0u ? PATH_ENTRIES : 0u);
        pk = pk0 + (cx != 0u ? 5 * NT : 0);
    };
    // seed of the current tree for Fpu::Func / Dirichlet draws (noise.cuh): stream = seed + game (or root) index, turn from
    // the parked game state; only the runtime-switched configurations evaluate it
    auto lane_noise_seed = [&]() -> uint64_t {
        if (FAST || (P.mcts.fpu != 2 && P.mcts.noise != 2)) return 0ull;
        const uint64_t stream = P.base_seed + (MODE == MODE_SELFPLAY ? P.first_game : 0ull) + (uint64_t)(uint32_t)T.job;
        return noise_tree_seed(stream, MODE == MODE_SELFPLAY ? (pk[4 * NT] & 0xFFu) : 0u);
    };

    for (;;) {
        if (__ballot(st != CS_IDLE || O.st != CS_IDLE) == 0ull) break;

        // ---- phase A: the descent loop over the current contexts, switching a lane to its other tree whenever the current one
        // stands on a leaf (or has nothing to do) and the other one can move
        uint32_t lm = legal_mask_of(C.my | C.op);
        for (;;) {
            // a descent that stands on a leaf: solved node -> backprop only; unexpanded node -> visit() gives it its block
            const bool arrived = st == CS_DESC && (C.nsolved || C.blk == 0u);
            if (__ballot(arrived) != 0ull) {
                if (arrived) {
                    LaneLeaf X;
                    X.solved = false;
                    X.needs_eval = false;
                    lane_arrive<COUNT, FAST>(P.mcts, T, C, X, C.nsolved, pl, bcap, ctr, P.error);
                    st = X.solved ? CS_SOLVED : CS_EVAL;
                }
            }
            const bool cur_runs = st == CS_DESC || st == CS_READY;
            const bool sw = !cur_runs && (O.st == CS_DESC || O.st == CS_READY);
            if (__ballot(sw) != 0ull) {
                swap_contexts(sw);
                lm = legal_mask_of(C.my | C.op);
            }
            const bool starts = st == CS_READY;
            if (__ballot(starts) != 0ull) {
                if (starts) {
                    lane_begin_explore<COUNT, FAST, false>(P.mcts, T, C, pl, ctr, pk, NT);
                    st = CS_DESC;
                    lm = legal_mask_of(C.my | C.op);
                }
            }
            const bool go = st == CS_DESC;
            const bool waits = st == CS_EVAL || st == CS_MISSED || O.st == CS_EVAL || O.st == CS_MISSED;
            if (__ballot(go) == 0ull || __popcll(__ballot(waits)) >= thresh) break;
            if (go && C.blk != 0u && !C.nsolved) lane_descend_level<COUNT, FAST>(P.mcts, T, C, lm, pl, ctr, lane_noise_seed());
        }
        // serve the tree that has waited longer: the other context, if it stands on a leaf
        {
            const bool serve_other = O.st >= CS_EVAL;
            if (__ballot(serve_other) != 0ull) swap_contexts(serve_other);
        }

        // ---- phase B: the current contexts that need the network, compacted into tiles of 16 positions. While some tree of
        // the wave can still descend only whole tiles are evaluated: the remainder stays on its leaf and goes first next round.
        const bool at_leaf = st >= CS_EVAL;
        const bool want_nn = st == CS_EVAL || st == CS_MISSED;
        float lg[9];
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;  // outcome distribution to back up: a solved leaf's, else the network's
        if (st == CS_SOLVED) {
            v0 = C.kind == 0u ? 1.0f : 0.0f;
            v1 = C.kind == 1u ? 1.0f : 0.0f;
            v2 = C.kind == 2u ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < 9; c++) lg[c] = 0.0f;
        // PolicyWithCache: a position that some game already evaluated skips the network (and its tile slot)
        bool hit = false;
        if (P.cache != nullptr && st == CS_EVAL) hit = cache_lookup(P.cache, P.cache_shift, C.my, C.op, lg, v0, v1, v2);
        bool need = want_nn && !hit;
        const unsigned long long want_mask = __ballot(need);
        if (P.cache != nullptr) {  // wave-uniform tallies (scalar registers), flushed once at the end of the kernel
            cache_hits += (unsigned long long)__popcll(__ballot(hit));
            cache_misses += (unsigned long long)__popcll(__ballot(st == CS_EVAL && !hit));
        }
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(want_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want_mask, 0u));
        const bool can_move = __ballot(st == CS_DESC || st == CS_READY || O.st == CS_DESC || O.st == CS_READY) != 0ull;
        const int quota = can_move ? (__popcll(want_mask) & ~15) : 64;
        const bool deferred = need && rank >= quota;
        need = need && rank < quota;
        if (deferred) st = CS_MISSED;
        const bool fin = at_leaf && !deferred;  // this context's explore gets its network call / backprop in this round
        if (COUNT && (need || hit)) ctr[CTR_POLICY_EVALS]++;
        const unsigned long long need_mask = __ballot(need);
        const int n_need = __popcll(need_mask);
        idxw[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (need) idxw[(rank & 15) * 4 + (rank >> 4)] = (unsigned char)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // tile slot (lane & 15) of tile j evaluates the position of lane byte j of src4 (slots past the last request: lane 0, finite input)
        const uint32_t src4 = *reinterpret_cast<const uint32_t*>(idxw + (lane & 15) * 4);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int j = 0; j * 16 < n_need; j++) {
            // everything a tile needs is re-derived here instead of living in registers across the whole matrix phase
            const int src = (int)((src4 >> (8 * j)) & 0xFFu);
            f32x4 o;
            if (POLICY == 2) {
                const uint64_t tmy = shfl_u64(C.my, src), top = shfl_u64(C.op, src);
                uint32_t img_off = 0;  // opaque per tile: the image reads stay LDS reads next to their MFMAs
                asm volatile("" : "+v"(img_off));
                o = conv_tile16(wimg + img_off, lane, tmy, top);
            } else {
                uint64_t hi, lo;
                feature_boards(C.my, C.op, hi, lo);
                const uint4 ftw = *reinterpret_cast<const uint4*>(smem_raw + Lane2Lds<NW>::FT_OFF + (lane >> 4) * 16);
                FeatureTable FT;
                FT.t[0] = ftw.x; FT.t[1] = ftw.y; FT.t[2] = ftw.z; FT.t[3] = ftw.w;
                const uint64_t thi = shfl_u64(hi, src), tlo = shfl_u64(lo, src);
                o = TILE == 0 ? mlp_tile16_pipe(wimg, bimg, lane, FT, thi, tlo) : mlp_tile16(wimg, bimg, lane, FT, thi, tlo);
            }
            // (raw outputs: the softmax over the three outcome logits runs per tree lane in phase C, lane_softmaxes)
            const int q = lane >> 4;
            if (q < 3) *reinterpret_cast<f32x4*>(outw + (lane & 15) * 12 + q * 4) = o;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (need && (rank >> 4) == j) {
                const float* mine = outw + (rank & 15) * 12;
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

        // ---- phase C
        bool solved = st == CS_SOLVED;
        uint32_t leaf_flag = 0;
        if (need || hit) {
            // the leaf's two softmaxes, then — with the probabilities — the PolicyWithCache entry of a position the network has
            // just evaluated
            const uint32_t lmask = legal_mask_of(C.my | C.op);
            float pr[9];
            lane_softmaxes(lmask, lg, pr, need, v0, v1, v2);
            if (P.cache != nullptr && need) cache_insert(P.cache, P.cache_shift, C.my, C.op, lg, v0, v1, v2);
            // root noise applies to the root's own expansion (mcts.rs:229-269): the first pass of a tree
            const CfgView<FAST> cv{P.mcts};
            solved = (lane_write_children(T.slab, C.blk, lmask, C.my, C.op, pr,
                                          (!FAST && T.iter == 0 && C.level == 0) ? P.mcts.noise : 0, P.mcts.noise_weight,
                                          P.mcts.noise_alpha, lane_noise_seed(),
                                          cv.fpu_const() ? cv.fpu_value() : 0.0f, leaf_flag) & LEAF_ANY_SOLVED) != 0u;
        }
        lane_backprop<COUNT, FAST>(P.mcts, T, C.level, v0, v1, v2, solved, fin, pl, ctr, leaf_flag, nullptr);
        if (fin) {
            T.iter += 1;
            st = CS_READY;
            // explore_n (mcts.rs:139-147): the root visit, then up to n explores unless the root is solved
            if (T.iter > n_explores || T.root_solved) {
                // a private copy of the arguments goes to the callee and the slab pointer is re-derived afterwards, so
                // the hot loop's pointers never round-trip through memory (they would come back generic: flat_load)
                const KernargPtr Pc = lane_kernarg();
                SYN_UNPARK(pk);
                if (MODE == MODE_SELFPLAY) T = lane_move_step_call<COUNT>(Pc, T, ctr);
                else T = lane_search_finish_call(Pc, T);
                T.slab = slab0 + (cx != 0u ? slab_bytes : (size_t)0);
                SYN_PARK(pk);
                st = T.job >= 0 ? CS_READY : CS_IDLE;
            }
        }
    }
#undef SYN_PARK
#undef SYN_UNPARK

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

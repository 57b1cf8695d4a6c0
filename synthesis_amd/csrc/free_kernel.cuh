// synthesis_amd — the fused engine kernel for AT MOST 16 TREES PER CU (BASELINE configs[1]'s 4,096 concurrent games) in the f16x2
// network arithmetic: four free-running waves of four trees, every wave evaluating its own leaves.
//
// At 16 trees per CU nothing is throughput: an explore (synthesis/src/mcts.rs:310-325) is a dependent chain select/expand -> network
// -> priors/backprop, and the chip waits for it. The row-per-tree kernel (engine_kernels.cuh selfplay_kernel<WPS = 1>) runs the 16 trees
// of a CU in lock step — four waves of four trees, two workgroup barriers per explore, one f32 tile split over the four waves because
// a single wave would need 15k matrix-pipe cycles for it: every explore lasts as long as the DEEPEST of 16 descents (stamps,
// profiles/r04_phase_stamps.txt: A 5.8k + wait 3.8k + B 7.4k + wait 1.3k + C 2.8k cycles).
// The f16x2 tile (f16x2_tile.cuh) is short enough for ONE wave (2.9k matrix-pipe cycles), so the lock step can go:
//   * the same four waves of four trees (one tree per DPP row, one lane per Connect4 column: tree code unchanged; a wave instruction
//     costs its four issue cycles whether one row or four are active, so fewer trees per wave only multiply the vector work —
//     measured: one tree per wave on 16 waves, leaves through an LDS mailbox, ran A 14.2k / C 11.2k cycles instead of 5.8k / 2.8k,
//     8.7k games/s against 14.0k; profiles/NOTES.md round 5);
//   * no barrier after start-up: a wave selects / expands on its four trees, evaluates ITS OWN leaves in a tile of its own (four of
//     the sixteen positions used — nobody else wants the SIMD's matrix pipe), writes priors and backs up, at its own pace;
//   * the 124 KB f16x2 image and the first 64 records of every tree (StatView::hot) share the CU's LDS.
// Results depend on the game / root index only, exactly as in every other launch shape (tests/test_gpu_f16x2.py).
#pragma once
#include "engine_kernels.cuh"
#include "f16x2_tile.cuh"

namespace syn {

struct FreeLds {
    static constexpr int TREES = 16;
    static constexpr int HOT_NODES = 64;
    static constexpr size_t IMG_OFF = 0;
    static constexpr size_t OUT_OFF = (size_t)F16Geom::IMG_WORDS * 4;              // 16 x 16 floats (9 logits, pad, 3 probabilities)
    static constexpr size_t HOT_OFF = OUT_OFF + TREES * 64;
    static constexpr size_t BYTES = HOT_OFF + (size_t)TREES * HOT_NODES * 32;      // 158,112 B
};
enum { FP_A = 0, FP_B, FP_C, FP_ITERS, FP_TILES, FP_LEAVES, FP_FIELDS = 8 };

template <int MODE, bool COUNT, bool FAST, bool PROF = false>
__global__ __launch_bounds__(256, 1) void selfplay_kernel_free(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = 256;
    const uint32_t* img = reinterpret_cast<const uint32_t*>(smem_raw + FreeLds::IMG_OFF);
    float* outbuf = reinterpret_cast<float*>(smem_raw + FreeLds::OUT_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gl = tid & 15;
    const int t = tid >> 4;   // tree (row) index inside the workgroup

    {
        const uint4* src = reinterpret_cast<const uint4*>(P.wimg);
        uint4* dst = reinterpret_cast<uint4*>(smem_raw + FreeLds::IMG_OFF);
        for (int i = tid; i < F16Geom::IMG_WORDS / 4; i += NT) dst[i] = src[i];
    }

    uint32_t ctr[COUNT ? CTR_COUNT : 1];
#pragma unroll
    for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) ctr[i] = 0;

    TreeCtx T;
    const size_t slot = (size_t)blockIdx.x * FreeLds::TREES + (size_t)t;
    T.stat.base = P.stat + 2 * slot * P.cap;
    T.edge.base = P.edge + 2 * slot * P.cap;
    T.stat.hot = reinterpret_cast<float4*>(smem_raw + FreeLds::HOT_OFF) + (size_t)t * FreeLds::HOT_NODES * 2;
    T.edge.hot = reinterpret_cast<uint4*>(T.stat.hot);
    T.stat.k = T.edge.k = (uint32_t)FreeLds::HOT_NODES;
    GameCtx G;
    start_job<MODE>(P, T, G, gl);
    __syncthreads();   // image staged: the only workgroup barrier, from here on every wave free-runs

    const int n_explores = P.roll.num_explores;
    unsigned long long pr[FP_FIELDS];
#pragma unroll
    for (int i = 0; i < FP_FIELDS; i++) pr[i] = 0;
    unsigned long long pT = 0;
#define SYN_STAMP() (PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
#define SYN_LAP(f) if (PROF) { unsigned long long n_ = SYN_STAMP(); pr[f] += n_ - pT; pT = n_; }
    for (;;) {
        const bool active = G.job >= 0;
        if (__ballot(active) == 0ull) break;
        pT = SYN_STAMP();
        ExploreCtx X = {};
        if (active) {
            tree_select_expand<COUNT, FAST>(P.mcts, T, X, gl, ctr);
            if (COUNT && X.needs_eval) ctr[CTR_POLICY_EVALS]++;
        }
        const unsigned long long need = __ballot(active && X.needs_eval);   // (every lane of a row agrees)
        SYN_LAP(FP_A)
        if (need != 0ull) {
            // this wave's own tile: position j = the leaf of row (j & 3) — four of the sixteen columns are used
            uint64_t hi = 0, lo = 0;
            if (active && X.needs_eval) feature_boards(X.leaf_my, X.leaf_op, hi, lo);
            const int src = 16 * (lane & 3);
            const uint64_t thi = (uint64_t)(uint32_t)__shfl((int)(uint32_t)hi, src, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(hi >> 32), src, 64) << 32);
            const uint64_t tlo = (uint64_t)(uint32_t)__shfl((int)(uint32_t)lo, src, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(lo >> 32), src, 64) << 32);
            uint32_t img_off = 0;   // opaque per tile: the image reads stay LDS reads next to their MFMAs
            asm volatile("" : "+v"(img_off));
            f32x4 o = f16x2_tile16<3>(img + img_off, lane, thi, tlo);
            const float os = reinterpret_cast<const float*>(img + img_off + F16Geom::SCALE_WORD0)[4];
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] *= os;
            const int j = lane & 15, q = lane >> 4;
            {   // (whole wave: the packed form tests its range with a ballot; only the q == 2 lanes keep the result)
                float v0 = o[1], v1 = o[2], v2 = o[3];
                if (q != 2) { v0 = 0.0f; v1 = 0.0f; v2 = 0.0f; }
                value_softmax_packed(v0, v1, v2);
                if (q == 2) { o[1] = v0; o[2] = v1; o[3] = v2; }
            }
            if (q < 3 && j < 4) *reinterpret_cast<f32x4*>(outbuf + (wave * 4 + j) * 16 + q * 4) = o;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its LDS operations complete in order)
            if (PROF) { pr[FP_TILES]++; pr[FP_LEAVES] += (unsigned long long)(__popcll(need) >> 4); }
        }
        SYN_LAP(FP_B)

        // ---- priors + backprop (+ the end of the search)
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
                if (MODE == MODE_SELFPLAY) selfplay_move_step<COUNT>(P, T, G, gl, ctr);
                else search_finish(P, T, G, gl);
            }
        }
        if (PROF) pr[FP_ITERS]++;
        SYN_LAP(FP_C)
    }
#undef SYN_STAMP
#undef SYN_LAP
    if (PROF) {
        if (P.prof && lane == 0) {
            unsigned long long* o = P.prof + ((size_t)blockIdx.x * (NT / 64) + wave) * FP_FIELDS;
#pragma unroll
            for (int i = 0; i < FP_FIELDS; i++) o[i] = pr[i];
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

}  // namespace syn

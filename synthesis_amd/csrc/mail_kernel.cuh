// synthesis_amd — the fused engine kernel for AT MOST 16 TREES PER CU (BASELINE configs[1]'s 4,096 concurrent games) in the f16x2
// network arithmetic: one tree per WAVE, leaves through an LDS mailbox, network tiles by whichever wave is waiting.
//
// At 16 trees per CU nothing is throughput: an explore (synthesis/src/mcts.rs:310-325) is a dependent chain select/expand -> network
// -> priors/backprop, and the chip waits for it. The row-per-tree kernel (engine_kernels.cuh selfplay_kernel<WPS = 1>) runs the 16 trees
// of a CU in lock step — four waves of four trees, two workgroup barriers per explore, one tile split over the four waves: every
// explore lasts as long as the DEEPEST of 16 descents (stamps, profiles/r04_phase_stamps.txt: A 5.8k + wait 3.8k + B 7.4k + wait 1.3k
// + C 2.8k cycles). Here:
//   * 16 waves per workgroup, one workgroup per CU, ONE TREE PER WAVE (its row 0: lanes 0-15, one lane per Connect4 column as in the
//     row kernel; the other 48 lanes idle through the tree phases) — no barrier after start-up, a tree's explore takes its own time;
//   * a wave whose leaf needs Policy::eval posts the two feature boards in the CU's LDS mailbox and then serves the mailbox itself:
//     it claims every posted, unclaimed leaf (its own included — or somebody else already has), evaluates ONE f16x2 tile
//     (f16x2_tile.cuh: a single wave runs the whole network for 16 positions in ~6k cycles, where the f32 tile needs four cooperating
//     waves to get under 7k) and publishes the outputs. Requests that arrive together share a tile; nobody waits for a quorum
//     (P.lane_thresh > 1 asks for one, bounded by a timeout: measured slower, profiles/NOTES.md);
//   * the 124 KB f16x2 image, the mailbox and the first 64 records of every tree (StatView::hot) share the CU's LDS.
// Results depend on the game / root index only, exactly as in every other launch shape (tests/test_gpu_f16x2.py).
#pragma once
#include "engine_kernels.cuh"
#include "f16x2_tile.cuh"

namespace syn {

struct MailLds {
    static constexpr int TREES = 16;
    static constexpr int HOT_NODES = 64;
    static constexpr size_t IMG_OFF = 0;
    static constexpr size_t STATE_OFF = (size_t)F16Geom::IMG_WORDS * 4;             // 16 request states + [16] = busy waves
    static constexpr size_t LEAF_OFF = STATE_OFF + 128;                            // 16 x (hi, lo) feature boards
    static constexpr size_t OUT_OFF = LEAF_OFF + TREES * 16;                       // 16 x 16 floats (9 logits, pad, 3 probabilities)
    static constexpr size_t HOT_OFF = OUT_OFF + TREES * 64;
    static constexpr size_t BYTES = HOT_OFF + (size_t)TREES * HOT_NODES * 32;      // 158,496 B
};
enum { MAIL_IDLE = 0, MAIL_POSTED = 1, MAIL_CLAIMED = 2, MAIL_DONE = 3 };
enum { MP_A = 0, MP_WAIT, MP_SERVE, MP_C, MP_ITERS, MP_TILES, MP_LEAVES, MP_FIELDS = 8 };

SYN_DEV void lds_release() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int MODE, bool COUNT, bool FAST, bool PROF = false>
__global__ __launch_bounds__(1024, 1) void selfplay_kernel_mail(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = 1024;
    const uint32_t* img = reinterpret_cast<const uint32_t*>(smem_raw + MailLds::IMG_OFF);
    volatile int* state = reinterpret_cast<volatile int*>(smem_raw + MailLds::STATE_OFF);
    int* state_nv = reinterpret_cast<int*>(smem_raw + MailLds::STATE_OFF);
    uint4* leafbuf = reinterpret_cast<uint4*>(smem_raw + MailLds::LEAF_OFF);
    float* outbuf = reinterpret_cast<float*>(smem_raw + MailLds::OUT_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gl = tid & 15;
    const bool row0 = lane < 16;

    {
        const uint4* src = reinterpret_cast<const uint4*>(P.wimg);
        uint4* dst = reinterpret_cast<uint4*>(smem_raw + MailLds::IMG_OFF);
        for (int i = tid; i < F16Geom::IMG_WORDS / 4; i += NT) dst[i] = src[i];
    }
    if (tid < 32) state_nv[tid] = 0;

    uint32_t ctr[COUNT ? CTR_COUNT : 1];
#pragma unroll
    for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) ctr[i] = 0;

    TreeCtx T;
    const size_t slot = (size_t)blockIdx.x * MailLds::TREES + (size_t)wave;
    T.stat.base = P.stat + 2 * slot * P.cap;
    T.edge.base = P.edge + 2 * slot * P.cap;
    T.stat.hot = reinterpret_cast<float4*>(smem_raw + MailLds::HOT_OFF) + (size_t)wave * MailLds::HOT_NODES * 2;
    T.edge.hot = reinterpret_cast<uint4*>(T.stat.hot);
    T.stat.k = T.edge.k = (uint32_t)MailLds::HOT_NODES;
    GameCtx G;
    G.job = -1; G.turn = 0; G.rng_index = 0;
    T.next_node = 0; T.root_fc = 0; T.root_nc = 0; T.iter = 0; T.root_solved = false; T.root_my = 0; T.root_op = 0;
    __syncthreads();   // mailbox zeroed before anybody counts itself busy
    if (row0) start_job<MODE>(P, T, G, gl);
    bool counted = __ballot(row0 && G.job >= 0) != 0ull;   // this wave is counted in the mailbox's busy-wave count
    if (counted && lane == 0) atomicAdd(&state_nv[16], 1);
    __syncthreads();   // image staged, every wave counted. The last workgroup barrier: from here on every wave free-runs.

    const int n_explores = P.roll.num_explores;
    const int thresh = P.lane_thresh < 1 ? 1 : (P.lane_thresh > 16 ? 16 : P.lane_thresh);
    const long long timeout = 3000;
    unsigned long long pr[MP_FIELDS];
#pragma unroll
    for (int i = 0; i < MP_FIELDS; i++) pr[i] = 0;
    unsigned long long pT = 0;
#define SYN_STAMP() (PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
#define SYN_LAP(f) if (PROF) { unsigned long long n_ = SYN_STAMP(); pr[f] += n_ - pT; pT = n_; }
    for (;;) {
        const bool active = row0 && G.job >= 0;
        if (__ballot(active) == 0ull) break;
        pT = SYN_STAMP();
        ExploreCtx X = {};
        if (active) {
            tree_select_expand<COUNT, FAST>(P.mcts, T, X, gl, ctr);
            if (COUNT && X.needs_eval) ctr[CTR_POLICY_EVALS]++;
        }
        const bool need = __ballot(active && X.needs_eval) != 0ull;   // wave-uniform: this wave's tree waits for the network
        SYN_LAP(MP_A)
        if (need) {
            if (lane == 0) {
                uint64_t hi, lo;
                feature_boards(X.leaf_my, X.leaf_op, hi, lo);
                leafbuf[wave] = make_uint4((uint32_t)hi, (uint32_t)(hi >> 32), (uint32_t)lo, (uint32_t)(lo >> 32));
                lds_release();
                atomicSub(&state_nv[16], 1);
                state[wave] = MAIL_POSTED;
            }
            const long long t_post = (long long)__builtin_readcyclecounter();
            for (;;) {
                const int mine = state[wave];
                if (mine == MAIL_DONE) break;
                bool served = false;
                if (mine == MAIL_POSTED) {
                    const int s_l = row0 ? state[lane] : MAIL_IDLE;
                    const int nposted = __popcll(__ballot(s_l == MAIL_POSTED));
                    const bool go = nposted >= thresh || state[16] <= 0 || ((long long)__builtin_readcyclecounter() - t_post) > timeout;
                    if (go) {
                        bool got = false;
                        if (row0 && s_l == MAIL_POSTED) got = atomicCAS(&state_nv[lane], MAIL_POSTED, MAIL_CLAIMED) == MAIL_POSTED;
                        const unsigned long long claimed = __ballot(got);
                        asm volatile("" ::: "memory");   // the claimed leaves are read after the claim
                        if (claimed != 0ull) {
                            SYN_LAP(MP_WAIT)
                            const int j = lane & 15, q = lane >> 4;
                            const bool valid = (claimed >> j) & 1ull;
                            const uint4 b = valid ? leafbuf[j] : make_uint4(0u, 0u, 0u, 0u);
                            const uint64_t hi = (uint64_t)b.x | ((uint64_t)b.y << 32), lo = (uint64_t)b.z | ((uint64_t)b.w << 32);
                            uint32_t img_off = 0;   // opaque per tile: the image reads stay LDS reads next to their MFMAs
                            asm volatile("" : "+v"(img_off));
                            f32x4 o = f16x2_tile16<3>(img + img_off, lane, hi, lo);
                            const float os = reinterpret_cast<const float*>(img + img_off + F16Geom::SCALE_WORD0)[4];
#pragma unroll
                            for (int r = 0; r < 4; r++) o[r] *= os;
                            if (q == 2) {
                                float v0 = o[1], v1 = o[2], v2 = o[3];
                                value_softmax(v0, v1, v2);
                                o[1] = v0; o[2] = v1; o[3] = v2;
                            }
                            if (q < 3 && valid) *reinterpret_cast<f32x4*>(outbuf + j * 16 + q * 4) = o;
                            lds_release();
                            if (got) state[lane] = MAIL_DONE;
                            served = true;
                            if (PROF) { pr[MP_TILES]++; pr[MP_LEAVES] += (unsigned long long)__popcll(claimed); }
                            SYN_LAP(MP_SERVE)
                        }
                    }
                }
                if (!served) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");   // the outputs are read after MAIL_DONE was seen
            if (lane == 0) atomicAdd(&state_nv[16], 1);
            SYN_LAP(MP_WAIT)
        }

        // ---- priors + backprop (+ the end of the search)
        if (active) {
            float d0 = X.p0, d1 = X.p1, d2 = X.p2;
            if (X.needs_eval) {
                const float* o = outbuf + wave * 16;
                float logit = o[gl < 9 ? gl : 0];
                tree_write_priors(T, X, gl, logit, (P.mcts.noise == 1 && T.iter == 0 && X.leaf == 0u) ? P.mcts.noise_weight : -1.0f);
                f32x4 ov = *reinterpret_cast<const f32x4*>(o + 8);
                d0 = ov[1];
                d1 = ov[2];
                d2 = ov[3];
                lds_release();
                if (gl == 0) state[wave] = MAIL_IDLE;   // (only this wave posts into this slot: nobody can overwrite the outputs before)
            }
            tree_backprop<COUNT, FAST>(P.mcts, T, X, gl, d0, d1, d2, X.solved, ctr);
            T.iter += 1;
            if (T.iter > n_explores || T.root_solved) {
                if (MODE == MODE_SELFPLAY) selfplay_move_step<COUNT>(P, T, G, gl, ctr);
                else search_finish(P, T, G, gl);
            }
        }
        if (PROF) pr[MP_ITERS]++;
        SYN_LAP(MP_C)
    }
#undef SYN_STAMP
#undef SYN_LAP
    if (counted && lane == 0) atomicSub(&state_nv[16], 1);   // out of jobs: the others need not wait for this wave's leaves any more
    if (PROF) {
        if (P.prof && lane == 0) {
            unsigned long long* o = P.prof + ((size_t)blockIdx.x * (NT / 64) + wave) * MP_FIELDS;
#pragma unroll
            for (int i = 0; i < MP_FIELDS; i++) o[i] = pr[i];
        }
    }
    if (COUNT) {
        if (P.counters && lane == 0) {
#pragma unroll
            for (int i = 0; i < (COUNT ? CTR_COUNT : 1); i++) {
                if (i == CTR_MAX_DEPTH) atomicMax(&P.counters[i], (unsigned long long)ctr[i]);
                else if (ctr[i]) atomicAdd(&P.counters[i], (unsigned long long)ctr[i]);
            }
        }
    }
}

}  // namespace syn

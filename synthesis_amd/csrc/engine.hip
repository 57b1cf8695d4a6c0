// synthesis_amd — host side of the engine and the C ABI (include/synthesis_amd.h).
//
// Host role (what the reference's Rust driver does around the hot path, synthesis/src/alpha_zero.rs:181-209):
// own the device node pool and the weight image, turn the plain-C configs into kernel parameters, launch ONE fused
// kernel per call and hand results back. No torch, no CPU fallback: every entry point either runs the HIP kernels on
// the handle's GPU or returns an error code.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/synthesis_amd.h"
#include "engine_kernels.cuh"
#include "eval_small.cuh"
#include "lane_kernel.cuh"
#ifdef SYN_DEBUG_SHAPES   // measured dead ends kept as parity-tested debug shapes: `make DEBUG_SHAPES=1` (default: not in the library)
#include "lane2_kernel.cuh"
#include "pc_kernel.cuh"
#endif
#include "frozen_kernel.cuh"
#include "train_kernels.cuh"
#include "train_mfma.cuh"
#include "train_epoch.cuh"
#include "convnet.cuh"
#include "layer_kernels.cuh"
#include "train_conv.cuh"
#include "train_conv_mfma.cuh"
#include "lane_instances.h"
#include "free_kernel.cuh"
#ifdef SYN_DEBUG_SHAPES
#include "pool_kernel.cuh"   // round 6's measured dead end (trees unbound from the lanes): a debug shape like the two above
#endif

#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cmath>
#include <thread>

using namespace syn;

// The Connect4ConvNet instantiations of the lane-per-tree kernel are compiled in engine_conv.hip (a second translation unit, built in
// parallel); here they are only declared.
namespace syn {
#define SYN_CONV_LANES(MODE, COUNT)                                                                        \
    extern template __global__ void selfplay_kernel_lanes<MODE, COUNT, true, 4, false, 2>(EngineParams);   \
    extern template __global__ void selfplay_kernel_lanes<MODE, COUNT, false, 4, false, 2>(EngineParams);  \
    extern template __global__ void selfplay_kernel_lanes<MODE, COUNT, true, 8, false, 2>(EngineParams);   \
    extern template __global__ void selfplay_kernel_lanes<MODE, COUNT, false, 8, false, 2>(EngineParams);  \
    extern template __global__ void selfplay_kernel_lanes<MODE, COUNT, true, 16, false, 2>(EngineParams);  \
    extern template __global__ void selfplay_kernel_lanes<MODE, COUNT, false, 16, false, 2>(EngineParams);
SYN_CONV_LANES(MODE_SEARCH, false)
SYN_CONV_LANES(MODE_SELFPLAY, false)
SYN_CONV_LANES(MODE_SELFPLAY, true)
#undef SYN_CONV_LANES
// ... and (DEBUG_SHAPES=1 builds only) the two-trees-per-lane kernels in engine_lanes2.hip
#ifdef SYN_DEBUG_SHAPES
#define SYN_LANES2(MODE, COUNT)                                                                      \
    extern template __global__ void selfplay_kernel_lanes2<MODE, COUNT, true, 8, 0>(EngineParams);   \
    extern template __global__ void selfplay_kernel_lanes2<MODE, COUNT, false, 8, 0>(EngineParams);  \
    extern template __global__ void selfplay_kernel_lanes2<MODE, COUNT, true, 8, 2>(EngineParams);   \
    extern template __global__ void selfplay_kernel_lanes2<MODE, COUNT, false, 8, 2>(EngineParams);  \
    extern template __global__ void selfplay_kernel_lanes2<MODE, COUNT, true, 12, 0>(EngineParams);  \
    extern template __global__ void selfplay_kernel_lanes2<MODE, COUNT, false, 12, 0>(EngineParams); \
    extern template __global__ void selfplay_kernel_lanes2<MODE, COUNT, true, 12, 2>(EngineParams);  \
    extern template __global__ void selfplay_kernel_lanes2<MODE, COUNT, false, 12, 2>(EngineParams);
extern template __global__ void selfplay_kernel_lanes2<MODE_SELFPLAY, false, true, 8, 0, 1>(EngineParams);
SYN_LANES2(MODE_SEARCH, false)
SYN_LANES2(MODE_SELFPLAY, false)
SYN_LANES2(MODE_SELFPLAY, true)
#undef SYN_LANES2
#endif
// ... and the Connect4Net / RolloutPolicy lane-per-tree kernels in engine_lanes_fast.hip and engine_lanes_gen.hip
#define SYN_X(MODE, COUNT, FAST, NW, PROF, POLICY) \
    extern template __global__ void selfplay_kernel_lanes<MODE, COUNT, FAST, NW, PROF, POLICY>(EngineParams);
SYN_LANES_FAST_LIST(SYN_X)
SYN_LANES_GEN_LIST(SYN_X)
SYN_LANES_REF_LIST(SYN_X)
SYN_LANES_F16_LIST(SYN_X)
SYN_LANES_F16_GEN_LIST(SYN_X)
#undef SYN_X
// ... and (DEBUG_SHAPES=1 builds only) the pool kernels (pool_kernel.cuh) in engine_pool.hip / engine_pool_f16.hip
#ifdef SYN_DEBUG_SHAPES
#define SYN_X(MODE, COUNT, FAST, NW, POLICY) extern template __global__ void selfplay_kernel_pool<MODE, COUNT, FAST, NW, POLICY>(EngineParams);
SYN_POOL_F32_LIST(SYN_X)
SYN_POOL_F16_LIST(SYN_X)
#undef SYN_X
#endif
// ... and the free-running four-trees-per-wave kernels (free_kernel.cuh) in engine_free.hip
#define SYN_FREE(MODE, COUNT)                                                                      \
    extern template __global__ void selfplay_kernel_free<MODE, COUNT, true, false>(EngineParams);  \
    extern template __global__ void selfplay_kernel_free<MODE, COUNT, false, false>(EngineParams);
SYN_FREE(MODE_SEARCH, false)
SYN_FREE(MODE_SELFPLAY, false)
SYN_FREE(MODE_SELFPLAY, true)
#undef SYN_FREE
extern template __global__ void selfplay_kernel_free<MODE_SELFPLAY, false, true, true>(EngineParams);
}  // namespace syn

static_assert(sizeof(DevSearchResult) == sizeof(syn_search_result), "search result layout must match the C ABI");
static_assert(sizeof(syn_counters) == sizeof(unsigned long long) * CTR_COUNT, "counter layout must match the C ABI");

static thread_local std::string g_create_error;

// Developer knobs (launch-shape overrides, in-kernel phase stamps) are read from the environment ONLY when SYN_DEBUG=1 is
// set as well: a stray variable must never change the launch shape of a production call.
static const char* debug_env(const char* name) {
    const char* on = std::getenv("SYN_DEBUG");
    if (!on || on[0] != '1') return nullptr;
    return std::getenv(name);
}

struct syn_engine {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;  // syn_progress / syn_cancel: independent of the launch stream
    int* h_pin = nullptr;              // 64 pinned bytes for their transfers: [0..1] syn_progress, [8] syn_cancel's word, [9] its pre-read
    // The call in flight (syn_selfplay_run / syn_mcts_search* / syn_frozen_search_rollout) as syn_progress / syn_cancel see it.
    // call_mu orders their state changes and serialises the users of aux_stream and h_pin.
    std::mutex call_mu;
    int call_state = 0;                // 0 idle, 1 launching (the job counter's reset is not enqueued yet), 2 running
    int cancel_pending = 0;            // a syn_cancel that arrived while launching: applied right behind the counter's reset
    int cancel_applied = 0;            // the running call's job counter was raised past its job count
    int started_at_cancel = 0;         // jobs handed out when that happened (read just before the write)
    int running_jobs = 0;              // job count of the call in flight (0 = none)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int slots = 0;
    int last_shape = 0, last_grid = 0, last_threads = 0;
    int pool_trees = 0;  // (DEBUG_SHAPES builds) > 64: the pool kernel with that many trees per wave; never set: the kernel is forced by SYN_POOL only
    int last_pool_trees = 0;
    int pool_slots = 0;  // tree slabs actually allocated (slots rounded up to the largest workgroup + slack)
    int max_explores = 0;
    uint32_t cap = 0;
    float4* d_stat = nullptr;
    uint4* d_edge = nullptr;
    float* d_wimg = nullptr;
    int net_kind = 0;  // which network d_wimg holds: 0 = Connect4Net (mlp.cuh), 1 = Connect4ConvNet (convnet.cuh)
    // Connect4Net in the f16x2 arithmetic (f16x2_tile.cuh; syn_set_network_arithmetic): its image, the parameters it is built from
    // (host copy, refreshed by syn_load_weights / syn_trainer_publish_weights) and whether the image is current
    int net_arith = SYN_NET_ARITH_F32;
    uint32_t* d_wimg16 = nullptr;
    std::vector<float> host_blob;
    bool img16_current = false;
    bool has_weights = false;
    int* d_job_next = nullptr;
    uint4* d_cache = nullptr;      // PolicyWithCache table (policy_cache_log2 > 0)
    int cache_log2 = 0;
    unsigned long long* d_cache_stats = nullptr;
    unsigned long long last_cache_hits = 0, last_cache_misses = 0;
    void* d_train_data = nullptr;  // syn_train_set_data: [my u64 n][op u64 n][pi 9 f32 n][v 3 f32 n]
    size_t train_data_cap = 0, train_data_n = 0;
    uint4* d_path = nullptr;   // lane kernel's per-wave descent logs
    size_t path_bytes = 0;
    unsigned char* d_vw = nullptr;  // producer/consumer kernel: per-virtual-wave parked state + network outputs
    size_t vw_bytes = 0;
    unsigned long long* d_counters = nullptr;
    // self-play output buffers (device), grown on demand
    int out_games = 0;
    int* d_plies = nullptr;
    unsigned long long* d_states = nullptr;
    float* d_pis = nullptr;
    float* d_vs = nullptr;
    unsigned char* d_actions = nullptr;
    uint32_t* d_root_nodes = nullptr;
    unsigned char* d_final = nullptr;
    // scratch for host-pointer entry points
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    std::atomic<bool> eval_attr_set[5] = {{false}, {false}, {false}, {false}, {false}};   // launch_policy_eval: the kernels' LDS attribute is set
    size_t eval_poll_max = 1024;       // contexts: batches up to this size signal completion through pinned memory (SYN_DEBUG=1 SYN_EVAL_POLL_MAX)
    size_t eval_zero_copy_out = 4096;  // contexts: results of up to this many positions are written into the pinned buffer by the kernel itself
    struct syn_eval_ctx* eval_ctx = nullptr;   // syn_policy_eval_batch's own evaluation context (created on first use)
    std::string err;
    float last_kernel_ms = 0.0f;
    int last_launches = 0;
    // learner state (syn_trainer_init): parameters, Adam moments, gradient + loss scratch
    float* d_tw = nullptr;
    float* d_tm = nullptr;
    float* d_tv = nullptr;
    float* d_tgrad = nullptr;
    float* d_tloss = nullptr;
    float* d_twimg = nullptr;  // the trainer's weights in the inference fragment order (forward A operands; = the published image)
    float* d_ttimg = nullptr;  // ... and transposed fragments for the activation gradients (train_mfma.cuh)
    float* d_timg2 = nullptr;  // the second buffer of both images for the persistent epoch kernel (train_epoch.cuh): [fwd][transposed]
    unsigned* d_tsync = nullptr;  // its arrival counter and status word
    float* d_tsnap = nullptr;     // snapshot of [w][m][v][fwd image][transposed image] taken before an epoch launch (restored if it aborts)
    long long train_step = 0;
    bool epoch_barrier_checked = false;  // syn_trainer_init's self-check of the epoch kernel's one-XCD barrier has run on this engine
    bool epoch_device_scope = false;     // ... and it failed (or is running its second half): the epoch kernel uses the device-scope barrier
    long long epoch_fallbacks = 0;  // syn_train_epoch calls whose persistent kernel gave up and ran through the queued launches
    float* d_cxbuf = nullptr;       // Connect4ConvNet learner on four workgroups: the exchange buffer (train_conv_mfma.cuh ConvMwGeom)
    bool conv_mw_checked = false;   // syn_trainer_init_conv's self-check of that kernel against the one-workgroup kernel has run
    bool conv_mw_disabled = false;  // ... and it failed: this engine keeps the one-workgroup epoch kernel
    int conv_mw_force = -1;         // self-check only: 0 = one workgroup, 1 = four
    DevTrainHyper train_hp{};
    bool has_trainer = false;
    int trainer_kind = 0;  // 0 = Connect4Net (train_mfma.cuh / train_epoch.cuh), 1 = Connect4ConvNet (train_conv_mfma.cuh)
    int train_bf16 = 0;    // Connect4ConvNet learner: 1 = the bf16 matrix-core variant of the gradient step (syn_trainer_set_precision)
};

static int fail(syn_engine* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_error = buf;
    return code;
}

// for the library's host-only translation units (lockstep_capi.cpp)
extern "C" int syn_internal_fail(syn_engine* h, int code, const char* msg) { return fail(h, code, "%s", msg); }
extern "C" int syn_internal_concurrent_games(const syn_engine* h) { return h ? h->slots : 0; }
#ifdef SYN_DEBUG_SHAPES
// present only in `make DEBUG_SHAPES=1` builds: the tests of the two debug launch shapes skip when it is absent
extern "C" int syn_internal_debug_shapes(void) { return 1; }
#endif

#define HIP_TRY(h, call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(h, SYN_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static int ensure_scratch(syn_engine* h, size_t bytes) {
    if (bytes <= h->scratch_bytes) return SYN_OK;
    if (h->d_scratch) HIP_TRY(h, hipFree(h->d_scratch));
    h->d_scratch = nullptr;
    h->scratch_bytes = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(h, hipMalloc(&h->d_scratch, want));
    h->scratch_bytes = want;
    return SYN_OK;
}

// ------------------------------------------------------------------------------------------------ weight image
// blob order: l_k.weight[O][I] then l_k.bias[O] for k = 1..5 (study-connect4/src/policies.rs:20-24)
static void build_weight_image(const float* blob, std::vector<float>& img) {
    img.assign(MlpGeom::IMG_FLOATS, 0.0f);
    size_t off = 0;
    for (int l = 0; l < MlpGeom::NL; l++) {
        const int K = MlpGeom::K[l], O = MlpGeom::O[l], S4 = MlpGeom::S4[l], NOB = MlpGeom::NOB[l];
        const float* W = blob + off;
        const float* b = W + (size_t)K * O;
        off += (size_t)K * O + O;
        for (int s4 = 0; s4 < S4; s4++)
            for (int ob = 0; ob < NOB; ob++)
                for (int lane = 0; lane < 64; lane++)
                    for (int r = 0; r < 4; r++) {
                        int i = lane & 15, q = lane >> 4;
                        int unit = mlp_unit_of_row(l, ob, i);
                        int k = 16 * s4 + 4 * r + q;
                        float v = (unit < O && k < K) ? W[(size_t)unit * K + k] : 0.0f;
                        img[MlpGeom::W_OFF[l] + ((s4 * NOB + ob) * 64 + lane) * 4 + r] = v;
                    }
        for (int ob = 0; ob < NOB; ob++)
            for (int q = 0; q < 4; q++)
                for (int r = 0; r < 4; r++) {
                    int unit = mlp_unit_of_row(l, ob, 4 * q + r);
                    img[MlpGeom::W_FLOATS + MlpGeom::B_OFF[l] + (ob * 4 + q) * 4 + r] = unit < O ? b[unit] : 0.0f;
                }
    }
}


// ------------------------------------------------------------------------------------------------ progress / cancel bracket
// Every entry point that plays jobs from the device-side job counter (d_job_next[0]) brackets its launch with a CallScope:
//   CallScope scope(h, n_jobs);          state = launching: a syn_cancel from another thread is remembered, nothing touches the device
//   ... enqueue the counter's reset ...
//   scope.armed();                       state = running; a remembered cancel is enqueued right behind the reset (same stream)
//   ... launch, copies, synchronise ...
//   scope.cancelled()                    did a cancel reach this call?  (the caller then decides from what actually finished)
// and the destructor returns the engine to idle. syn_cancel on an idle engine is an error: there is nothing to cancel.
constexpr int CANCEL_WORD = 0x40000000;
struct CallScope {
    syn_engine* e;
    CallScope(syn_engine* h, int n_jobs) : e(h) {
        std::lock_guard<std::mutex> g(e->call_mu);
        e->call_state = 1;
        e->cancel_pending = 0;
        e->cancel_applied = 0;
        e->started_at_cancel = 0;
        e->running_jobs = n_jobs;
    }
    hipError_t armed() {
        std::lock_guard<std::mutex> g(e->call_mu);
        e->call_state = 2;
        if (!e->cancel_pending) return hipSuccess;
        e->cancel_pending = 0;
        e->cancel_applied = 1;
        e->started_at_cancel = 0;
        e->h_pin[8] = CANCEL_WORD;
        return hipMemcpyAsync(e->d_job_next, e->h_pin + 8, 4, hipMemcpyHostToDevice, e->stream);
    }
    bool cancelled() {
        std::lock_guard<std::mutex> g(e->call_mu);
        return e->cancel_applied != 0;
    }
    ~CallScope() {
        std::lock_guard<std::mutex> g(e->call_mu);
        e->call_state = 0;
        e->cancel_pending = 0;
        e->running_jobs = 0;
    }
};

// ------------------------------------------------------------------------------------------------ config checks
static int convert_mcts(syn_engine* h, const syn_mcts_config* c, DevMctsCfg& d) {
    if (!c) return fail(h, SYN_ERR_INVALID_ARGUMENT, "mcts config is NULL");
    if (c->exploration != SYN_EXPLORATION_UCT && c->exploration != SYN_EXPLORATION_POLYNOMIAL_UCT)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown exploration %d", c->exploration);
    if (c->fpu == SYN_FPU_FUNC)
        return fail(h, SYN_ERR_UNSUPPORTED, "SYN_FPU_FUNC (Fpu::Func(fn() -> f32), config.rs:25) is a host function: a device kernel cannot call it. "
                                            "It runs on the host trees (syn_mcts_search_lockstep / syn_selfplay_run_lockstep); on the device, "
                                            "SYN_FPU_NORMAL is the reference's own Normal(mean, std) closure");
    if (c->fpu != SYN_FPU_CONST && c->fpu != SYN_FPU_PARENT_Q && c->fpu != SYN_FPU_NORMAL)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown fpu %d", c->fpu);
    if (c->fpu == SYN_FPU_NORMAL && !(c->fpu_std >= 0.0f && c->fpu_std < 1e30f && c->fpu_value == c->fpu_value))
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "SYN_FPU_NORMAL needs a finite mean and a standard deviation >= 0 (Normal::new)");
    if (c->root_policy_noise != SYN_NOISE_NONE && c->root_policy_noise != SYN_NOISE_EQUAL && c->root_policy_noise != SYN_NOISE_DIRICHLET)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown root policy noise %d", c->root_policy_noise);
    if (c->root_policy_noise != SYN_NOISE_NONE && !(c->noise_weight >= 0.0f))
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "PolicyNoise weight must be >= 0");
    if (c->root_policy_noise == SYN_NOISE_DIRICHLET && !(c->noise_alpha > 0.0f && c->noise_alpha < 1e30f))
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "PolicyNoise::Dirichlet alpha must be > 0 (Dirichlet::new_with_size)");
    d.exploration = c->exploration;
    d.c = c->c;
    d.solve = c->solve != 0;
    d.correct_values = c->correct_values_on_solve != 0;
    d.select_solved = c->select_solved_nodes != 0;
    d.auto_extend = c->auto_extend != 0;
    d.fpu = c->fpu;
    d.fpu_value = c->fpu_value;
    d.noise = c->root_policy_noise;
    d.noise_weight = c->noise_weight;
    d.fpu_std = c->fpu_std;
    d.noise_alpha = c->noise_alpha;
    // (c * prior * sqrt(N)) / (1 + n) goes through div2_safe_range only when the numerator stays inside its exact range
    d.fast_div = (c->c >= 0x1p-10f && c->c <= 0x1p10f) ? 1 : 0;
    return SYN_OK;
}

// ------------------------------------------------------------------------------------------------ launches
template <int MODE, bool COUNT, bool PROF = false>
static hipError_t launch_engine(syn_engine* h, const EngineParams& P, int jobs, int* out_grid = nullptr,
                                int* out_nt = nullptr) {
    // one 256-thread workgroup per 16 tree slots; the kernel needs ~17 KB of LDS and <= 256 VGPRs, so up to two
    // workgroups are resident per CU (concurrency beyond 2 x 16 x CUs queues behind resident workgroups)
    int want_slots = h->slots;
    if (jobs < want_slots) want_slots = ((jobs + 15) / 16) * 16;
    int grid = want_slots / 16;
    if (grid < 1) grid = 1;
    const bool fast = cfg_is_fast(P.mcts);  // compile-time-folded config family (mcts.cuh CfgView)
#define SYN_LAUNCH(WPS, FAST)                                                                                      \
    {                                                                                                              \
        auto k = selfplay_kernel<MODE, COUNT, WPS, FAST, PROF>;                                                    \
        size_t lds = WPS == 1 ? EngineLds::BYTES_WPS1 : EngineLds::BYTES_WPS2;                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, h->stream, P);                                           \
    }
    // Kernel choice by trees per CU: <= 16 -> one 16-tree workgroup per CU, weights in registers (latency-optimal);
    // <= 32 -> two such workgroups per CU (hybrid register/LDS weights); more -> the quad-async kernel (NQ quads of 16
    // trees per workgroup sharing one LDS weight image). SYN_DEBUG=1 SYN_QUADS=0..4 overrides (0 = never use the quad kernel).
    // Lane-per-tree kernel (lane_kernel.cuh): one tree per lane, NW waves per workgroup, one workgroup per CU.
    // SYN_DEBUG=1 SYN_LANES=<waves per workgroup: 4, 8, 12 or 16> forces it (0 = never); by default it takes over once every CU can
    // be given 256 trees (4 waves; measured 41.9k games/s at 65,536 concurrent games against 31.8k for the queued
    // row-per-tree workgroups), 8 waves up to 512 trees per CU, 12 up to 768, 16 beyond (selection below).
    // Producer/consumer kernel (pc_kernel.cuh): 12 tree waves x NV virtual waves of 64 trees + 4 matrix waves per CU. Measured
    // slower than the symmetric lane kernel (the f32 MFMA shares the SIMD's FP32 datapath with the VALU, so
    // dedicating waves to the matrix pipe frees nothing), so it is never chosen automatically:
    // SYN_DEBUG=1 SYN_PC=<NV 1..3> selects it (parity tests, profiling).
#ifdef SYN_DEBUG_SHAPES
    {
        int nv = 0;
        if (const char* ev = debug_env("SYN_PC")) nv = std::atoi(ev);
        if (nv >= 1 && nv <= PcGeom::NV_MAX && h->cap <= LANE_MAX_CAP && h->net_kind == 0) {
            const int per_wg = 64 * PcGeom::TREE_WAVES * nv;
            const int pgrid = (want_slots + per_wg - 1) / per_wg;
            const size_t nvw = (size_t)pgrid * PcGeom::TREE_WAVES * nv;
            const size_t need_path = nvw * PATH_ENTRIES * sizeof(uint4);
            if (need_path > h->path_bytes) {
                if (h->d_path) (void)hipFree(h->d_path);
                h->d_path = nullptr;
                h->path_bytes = 0;
                hipError_t pe = hipMalloc(&h->d_path, need_path);
                if (pe != hipSuccess) return pe;
                h->path_bytes = need_path;
            }
            const size_t need_vw = nvw * PcGeom::VW_BYTES;
            if (need_vw > h->vw_bytes) {
                if (h->d_vw) (void)hipFree(h->d_vw);
                h->d_vw = nullptr;
                h->vw_bytes = 0;
                hipError_t pe = hipMalloc(&h->d_vw, need_vw);
                if (pe != hipSuccess) return pe;
                h->vw_bytes = need_vw;
            }
            EngineParams PL = P;
            PL.path = h->d_path;
            PL.vw_buf = h->d_vw;
            PL.nv = nv;
            PL.debug_prio = 2;
            if (const char* ev = debug_env("SYN_PC_PRIO")) PL.debug_prio = std::atoi(ev);
            PL.debug_stub = (PROF && debug_env("SYN_PC_STUB") != nullptr) ? 1 : 0;
            PL.lane_thresh = 48;
            if (const char* ev = debug_env("SYN_LANE_THRESH")) PL.lane_thresh = std::atoi(ev);
            if (PL.lane_thresh < 16 || PL.lane_thresh > 64) PL.lane_thresh = 48;
            PL.lane_thresh &= ~15;  // whole tiles
#define SYN_LAUNCH_PC(FAST)                                                                                        \
    {                                                                                                              \
        auto k = selfplay_kernel_pc<MODE, COUNT, FAST, PROF>;                                                      \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PcLds::BYTES);         \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL(k, dim3(pgrid), dim3(PcGeom::NT), PcLds::BYTES, h->stream, PL);                         \
    }
            if (fast) SYN_LAUNCH_PC(true) else SYN_LAUNCH_PC(false)
#undef SYN_LAUNCH_PC
            h->last_shape = 5; h->last_grid = pgrid; h->last_threads = PcGeom::NT;
            if (out_grid) *out_grid = pgrid;
            if (out_nt) *out_nt = PcGeom::NT;
            return hipGetLastError();
        }
    }
#endif
    // At most 16 trees per CU in the f16x2 arithmetic: four free-running waves of four trees, each evaluating its own leaves in a tile
    // of its own (free_kernel.cuh). The draws of Fpu::Func / PolicyNoise::Dirichlet live in the lane-per-tree kernels only.
    // SYN_DEBUG=1 SYN_FREE=0 switches it off (the lane kernel at 4 waves then plays these games).
    if (h->net_kind == 0 && h->net_arith == SYN_NET_ARITH_F16X2 && P.wimg == reinterpret_cast<const float*>(h->d_wimg16) &&
        want_slots <= 16 * h->num_cus && P.mcts.fpu != 2 && P.mcts.noise != 2 && debug_env("SYN_LANES") == nullptr &&
        !(debug_env("SYN_FREE") && std::atoi(debug_env("SYN_FREE")) == 0)
#ifdef SYN_DEBUG_SHAPES
        && !(debug_env("SYN_POOL") && std::atoi(debug_env("SYN_POOL")) > 64)
#endif
        ) {
        const int mgrid = (want_slots + 15) / 16;
#define SYN_LAUNCH_FR(FAST, PROFV)                                                                                 \
    {                                                                                                              \
        auto k = selfplay_kernel_free<MODE, COUNT, FAST, PROFV>;                                                   \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)FreeLds::BYTES);       \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL(k, dim3(mgrid), dim3(256), FreeLds::BYTES, h->stream, P);                               \
    }
        constexpr bool PROFFR = PROF && MODE == MODE_SELFPLAY && !COUNT;
        if (fast) SYN_LAUNCH_FR(true, PROFFR) else SYN_LAUNCH_FR(false, false)
#undef SYN_LAUNCH_FR
        h->last_shape = 7; h->last_grid = mgrid; h->last_threads = 256;
        if (out_grid) *out_grid = mgrid;
        if (out_nt) *out_nt = 256;
        return hipGetLastError();
    }
#ifdef SYN_DEBUG_SHAPES
    // Two trees per lane (lane2_kernel.cuh): 8 waves per workgroup, 1,024 trees per CU. SYN_DEBUG=1 SYN_LANES2=8 forces it,
    // SYN_LANES2=0 switches it off.
    {
        int nw2 = 0;
        const bool needs_noise2 = P.mcts.fpu == 2 || P.mcts.noise == 2;
        if (const char* ev = debug_env("SYN_LANES2")) nw2 = std::atoi(ev);
        if (PROF) nw2 = 0;
        if ((nw2 == 8 || nw2 == 12) && h->cap <= LANE_MAX_CAP) {
            (void)needs_noise2;
            const int per_wg = 128 * nw2;
            const int lgrid = (want_slots + per_wg - 1) / per_wg;
            const size_t need_path = (size_t)lgrid * nw2 * 2 * PATH_ENTRIES * sizeof(uint4);
            if (need_path > h->path_bytes) {
                if (h->d_path) (void)hipFree(h->d_path);
                h->d_path = nullptr;
                h->path_bytes = 0;
                hipError_t pe = hipMalloc(&h->d_path, need_path);
                if (pe != hipSuccess) return pe;
                h->path_bytes = need_path;
            }
            EngineParams PL = P;
            PL.path = h->d_path;
            PL.lane_thresh = 48;
            if (const char* ev = debug_env("SYN_LANE_THRESH")) PL.lane_thresh = std::atoi(ev);
            if (PL.lane_thresh < 16 || PL.lane_thresh > 64) PL.lane_thresh = 48;
            PL.lane_thresh &= ~15;  // whole tiles
#define SYN_LAUNCH_L2(NW, FAST, POL)                                                                               \
    {                                                                                                              \
        auto k = selfplay_kernel_lanes2<MODE, COUNT, FAST, NW, POL>;                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lane2Lds<NW>::BYTES);  \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL(k, dim3(lgrid), dim3(64 * NW), Lane2Lds<NW>::BYTES, h->stream, PL);                     \
    }
            if (nw2 == 12) {
                if (h->net_kind == 1) { if (fast) SYN_LAUNCH_L2(12, true, 2) else SYN_LAUNCH_L2(12, false, 2) }
                else { if (fast) SYN_LAUNCH_L2(12, true, 0) else SYN_LAUNCH_L2(12, false, 0) }
            }
            else if (h->net_kind == 1) { if (fast) SYN_LAUNCH_L2(8, true, 2) else SYN_LAUNCH_L2(8, false, 2) }
            else if (MODE == MODE_SELFPLAY && !COUNT && fast && debug_env("SYN_L2_TILE")) {
                auto k = selfplay_kernel_lanes2<MODE_SELFPLAY, false, true, 8, 0, 1>;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lane2Lds<8>::BYTES);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k, dim3(lgrid), dim3(512), Lane2Lds<8>::BYTES, h->stream, PL);
            }
            else { if (fast) SYN_LAUNCH_L2(8, true, 0) else SYN_LAUNCH_L2(8, false, 0) }
#undef SYN_LAUNCH_L2
            h->last_shape = 6; h->last_grid = lgrid; h->last_threads = 64 * nw2;
            if (out_grid) *out_grid = -lgrid;
            if (out_nt) *out_nt = 64 * nw2;
            return hipGetLastError();
        }
    }
#endif
#ifdef SYN_DEBUG_SHAPES
    // The pool kernel (pool_kernel.cuh): a wave's 64 lanes work on a pool of M trees (64 < M <= 128) — a lane whose descent arrives
    // binds the next READY tree in the same iteration, a round fires on 64 leaves — for the two compile-time-folded configuration
    // families of Connect4Net (f32 and f16x2), 12 or 8 waves x M trees per CU. Measured slower than the lane kernel on every leg in
    // three same-box A/B runs (profiles/r06_pool_unbinding_ab.txt, NOTES round 6), so it is never chosen automatically and ships only
    // in DEBUG_SHAPES=1 builds: SYN_DEBUG=1 SYN_POOL=<M> selects it (parity tests, re-measurement).
    {
        int pm = h->pool_trees;
        if (const char* ev = debug_env("SYN_POOL")) pm = std::atoi(ev);
        // 1 = as many trees per wave as the engine's slots give 12 waves on every CU (at most 128)
        if (pm == 1) pm = (want_slots + h->num_cus * 12 - 1) / (h->num_cus * 12) > PoolGeom::M_MAX ? PoolGeom::M_MAX
                                                                                                   : (want_slots + h->num_cus * 12 - 1) / (h->num_cus * 12);
        const int fam = cfg_family(P.mcts);
        const bool f16 = h->net_kind == 0 && h->net_arith == SYN_NET_ARITH_F16X2 && P.wimg == reinterpret_cast<const float*>(h->d_wimg16);
        const bool lanes_forced = debug_env("SYN_LANES") != nullptr || debug_env("SYN_QUADS") != nullptr;
        if (pm > 64 && pm <= PoolGeom::M_MAX && (fam == 1 || fam == 2) && h->net_kind == 0 && h->cap <= LANE_MAX_CAP && !PROF && !lanes_forced &&
            (debug_env("SYN_POOL") != nullptr || want_slots >= h->num_cus * 768)) {
            int nw = 12;   // waves per workgroup: 12 (three per SIMD, 168 registers) or 8 (two per SIMD, 256 registers)
            if (const char* ev = debug_env("SYN_POOL_NW")) nw = std::atoi(ev) == 8 ? 8 : 12;
            const int per_wg = nw * pm;
            int pgrid = (want_slots + per_wg - 1) / per_wg;
            if (pgrid > h->num_cus) pgrid = h->num_cus;   // one workgroup per CU: a larger engine only holds idle slabs
            if (pgrid < 1) pgrid = 1;
            const size_t nwv = (size_t)pgrid * nw;
            const size_t need_path = nwv * 2 * PATH_ENTRIES * sizeof(uint4);
            if (need_path > h->path_bytes) {
                if (h->d_path) (void)hipFree(h->d_path);
                h->d_path = nullptr;
                h->path_bytes = 0;
                hipError_t pe = hipMalloc(&h->d_path, need_path);
                if (pe != hipSuccess) return pe;
                h->path_bytes = need_path;
            }
            const size_t need_vw = nwv * PoolGeom::WAVE_BYTES;
            if (need_vw > h->vw_bytes) {
                if (h->d_vw) (void)hipFree(h->d_vw);
                h->d_vw = nullptr;
                h->vw_bytes = 0;
                hipError_t pe = hipMalloc(&h->d_vw, need_vw);
                if (pe != hipSuccess) return pe;
                h->vw_bytes = need_vw;
            }
            EngineParams PL = P;
            PL.path = h->d_path;
            PL.vw_buf = h->d_vw;
            PL.nv = pm;
            PL.lane_thresh = 24;   // Fpu::Func: waiting lanes that trigger a scan iteration
            if (const char* ev = debug_env("SYN_POOL_SCAN")) PL.lane_thresh = std::atoi(ev);
            if (PL.lane_thresh < 1 || PL.lane_thresh > 64) PL.lane_thresh = 24;
            PL.debug_prio = 64;    // leaves that fire a round
            if (const char* ev = debug_env("SYN_POOL_FIRE")) PL.debug_prio = std::atoi(ev);
            if (PL.debug_prio < 16 || PL.debug_prio > 64) PL.debug_prio = 64;
#define SYN_LAUNCH_P(FASTV, POL) { if (nw == 8) SYN_LAUNCH_PN(FASTV, POL, 8) else SYN_LAUNCH_PN(FASTV, POL, 12) }
#define SYN_LAUNCH_PN(FASTV, POL, NWV)                                                                             \
    {                                                                                                              \
        auto k = selfplay_kernel_pool<MODE, COUNT, FASTV, NWV, POL>;                                               \
        using PLds = PoolLds<NWV, FASTV>;                                                                          \
        const size_t plds = PLds::BYTES;                                                                           \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);                 \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL(k, dim3(pgrid), dim3(64 * NWV), plds, h->stream, PL);                                   \
    }
            if (f16) { if (fam == 1) SYN_LAUNCH_P(1, 3) else SYN_LAUNCH_P(2, 3) }
            else { if (fam == 1) SYN_LAUNCH_P(1, 0) else SYN_LAUNCH_P(2, 0) }
#undef SYN_LAUNCH_P
#undef SYN_LAUNCH_PN
            h->last_shape = 8; h->last_grid = pgrid; h->last_threads = 64 * nw;
            h->last_pool_trees = pm;
            if (out_grid) *out_grid = pgrid;
            if (out_nt) *out_nt = 64 * nw;
            return hipGetLastError();
        }
    }
#endif
    {
        int nw = 0;
        // 4 waves per workgroup up to 256 trees per CU, 8 up to 512, 12 up to 768, 16 (hand-pipelined network tile that fits
        // the 128-VGPR budget: mlp_tile16_pipe) beyond
        if (want_slots >= h->num_cus * 256)
            nw = want_slots > h->num_cus * 768 ? 16 : (want_slots > h->num_cus * 512 ? 12 : (want_slots > h->num_cus * 256 ? 8 : 4));
        // the random draws of Fpu::Func / PolicyNoise::Dirichlet (noise.cuh) exist in the lane-per-tree kernels only
        const bool needs_noise = P.mcts.fpu == 2 || P.mcts.noise == 2;
        if (needs_noise && nw == 0) nw = 4;
        if (const char* ev = debug_env("SYN_LANES")) nw = std::atoi(ev);
        if (needs_noise && !(nw == 4 || nw == 8 || nw == 12 || nw == 16)) nw = 4;
        if (needs_noise && h->cap > LANE_MAX_CAP) return hipErrorInvalidValue;
        // Connect4ConvNet (convnet.cuh) is evaluated by the lane-per-tree kernels only: 4 waves per workgroup up to 256 trees
        // per CU, 8 up to 512, 16 beyond
        const bool conv = h->net_kind == 1;
        // Connect4Net in the f16x2 arithmetic (f16x2_tile.cuh) is evaluated by the lane-per-tree kernels only, at every size
        const bool f16x2 = !conv && h->net_arith == SYN_NET_ARITH_F16X2 && P.wimg == reinterpret_cast<const float*>(h->d_wimg16);
        if (f16x2) {
            if (h->cap > LANE_MAX_CAP) return hipErrorInvalidValue;
            if (debug_env("SYN_LANES") == nullptr || !(nw == 4 || nw == 8 || nw == 12 || nw == 16))
                nw = want_slots > h->num_cus * 768 ? 16 : (want_slots > h->num_cus * 512 ? 12 : (want_slots > h->num_cus * 256 ? 8 : 4));
        }
        if (conv) {
            if (h->cap > LANE_MAX_CAP) return hipErrorInvalidValue;
            nw = want_slots > h->num_cus * 512 ? 16 : (want_slots > h->num_cus * 256 ? 8 : 4);
        }
        // The runtime-switched (general) instantiations need 300-450 more registers than the 128 of a 16-wave workgroup and are
        // bound by their own scratch traffic there (PMC: 8.8x the algorithmic bytes; Fpu::ParentQ 29.5k games/s against 42.4k,
        // Fpu::Func 15.9k against 31.2k): they run 8 waves of 256 registers on at most 512 trees per CU, whatever the capacity.
        if (!fast && cfg_family(P.mcts) != 2 && nw > 8 && debug_env("SYN_LANES") == nullptr) {
            nw = 8;
            if (want_slots > h->num_cus * 512) want_slots = h->num_cus * 512;
        }
        // Tree-bound regimes — PolicyWithCache on (most leaf evaluations are table hits) or the reference's own Fpu::Func
        // configuration — run 12 waves of 168 registers (17 spilled) on at most 768 trees per CU rather than 16 x 128 (55 spilled):
        // measured 90.5k against 84.8k games/s with the cache, 61.2k against 55.8k with the trained checkpoint and the cache,
        // 51.2k against 47.7k for the reference configuration; without the cache the two shapes are equal (70.6k / 71.2k).
        // The f16x2 arithmetic makes every regime tree-bound (its network tile is a quarter of the f32 one): 102k games/s at 16 x 1,024
        // against 111k at 12 x 768 (random-init), 65.0k against 72.4k (trained checkpoint).
        // Round 5: the headline (parity family, f32, no cache) takes the same shape — it measures the same or better there (75.9k against
        // 74.6k at 1,048,576 games, 77.5k against 76.9k at a full step) and 12 x 164 registers spill nothing where 16 x 128 spills 35.
        if (!conv && nw == 16 && (fast || cfg_family(P.mcts) == 2) &&
            debug_env("SYN_LANES") == nullptr) {
            nw = 12;
            if (want_slots > h->num_cus * 768) want_slots = h->num_cus * 768;
        }
        if ((nw == 4 || nw == 8 || nw == 12 || nw == 16) && h->cap <= LANE_MAX_CAP) {
            int lgrid = (want_slots + 64 * nw - 1) / (64 * nw);
            // (slots are rounded up to whole workgroups; the pool was allocated for a multiple of 1024 slabs)
            const size_t need_path = (size_t)lgrid * nw * PATH_ENTRIES * sizeof(uint4);
            if (need_path > h->path_bytes) {
                if (h->d_path) (void)hipFree(h->d_path);
                h->d_path = nullptr;
                h->path_bytes = 0;
                hipError_t pe = hipMalloc(&h->d_path, need_path);
                if (pe != hipSuccess) return pe;
                h->path_bytes = need_path;
            }
            EngineParams PL = P;
            PL.path = h->d_path;
            // a round ends once this many lanes of a wave stand on a leaf: 48 with the f32 tile; with the f16x2 tile (a quarter of the cost) waiting
            // for all 64 measures +1-2 % (same box: 107.5k -> 108.7k games/s random-init, 65.3k -> 66.7k trained; 32: 100.5k / 60.8k)
            PL.lane_thresh = f16x2 ? 64 : 48;
            if (const char* ev = debug_env("SYN_LANE_THRESH")) PL.lane_thresh = std::atoi(ev);
            if (PL.lane_thresh < 16 || PL.lane_thresh > 64) PL.lane_thresh = f16x2 ? 64 : 48;
            PL.lane_thresh &= ~15;  // whole tiles
            PL.debug_stub = 0;
            if (PROF) { if (const char* ev = debug_env("SYN_ABLATE")) PL.debug_stub = std::atoi(ev); }
            // Fpu::Func: 0 = every level takes its draws on the spot (a scan is ~220 issue slots since round 6); 1..64 = scans are
            // deferred until that many lanes wait for one or nobody can move without one (rounds 4-5, when a scan was ~600: 64)
            PL.nv = 0;
            if (const char* ev = debug_env("SYN_SCAN_MIN")) PL.nv = std::atoi(ev);
            if (PL.nv < 0 || PL.nv > 64) PL.nv = 0;
#define SYN_LAUNCH_L(NW, FAST)                                                                                     \
    {                                                                                                              \
        auto k = selfplay_kernel_lanes<MODE, COUNT, FAST, NW, PROF>;                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LaneLds<NW>::BYTES);   \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL(k, dim3(lgrid), dim3(64 * NW), LaneLds<NW>::BYTES, h->stream, PL);                       \
    }
#define SYN_LAUNCH_LR(NW)                                                                                          \
    {                                                                                                              \
        auto k = selfplay_kernel_lanes<MODE, COUNT, 2, NW, PROF && MODE == MODE_SELFPLAY && !COUNT, 0>;                                              \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LaneLds<NW>::BYTES);   \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL(k, dim3(lgrid), dim3(64 * NW), LaneLds<NW>::BYTES, h->stream, PL);                      \
    }
#define SYN_LAUNCH_LC(NW, FAST)                                                                                    \
    {                                                                                                              \
        auto k = selfplay_kernel_lanes<MODE, COUNT, FAST, NW, false, 2>;                                           \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LaneLds<NW>::BYTES);   \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL(k, dim3(lgrid), dim3(64 * NW), LaneLds<NW>::BYTES, h->stream, PL);                      \
    }
            // the reference's own self-play configuration (Fpu::Func folded at compile time: mcts.cuh cfg_family) has instantiations of
            // its own at 8 and 16 waves
            const bool ref_family = !conv && (!PROF || (MODE == MODE_SELFPLAY && !COUNT)) && cfg_family(P.mcts) == 2 && (nw == 8 || nw == 12 || nw == 16);
            if (f16x2) {
                // family 1 / 2 at every wave count, the runtime-switched configurations at 4 and 8 waves (nw was capped above)
                const int fam = cfg_family(P.mcts);
#define SYN_LAUNCH_LH(NW, FAST, PROFV)                                                                             \
    {                                                                                                              \
        auto k = selfplay_kernel_lanes<MODE, COUNT, FAST, NW, PROFV, 3>;                                           \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LaneLds<NW>::BYTES);   \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL(k, dim3(lgrid), dim3(64 * NW), LaneLds<NW>::BYTES, h->stream, PL);                      \
    }
                constexpr bool PROFH = PROF && MODE == MODE_SELFPLAY && !COUNT;
                if (fam == 2) {
                    if (nw == 4) SYN_LAUNCH_LH(4, 2, false) else if (nw == 8) SYN_LAUNCH_LH(8, 2, false)
                    else if (nw == 12) SYN_LAUNCH_LH(12, 2, false) else SYN_LAUNCH_LH(16, 2, false)
                } else if (fast) {
                    if (nw == 4) SYN_LAUNCH_LH(4, true, false) else if (nw == 8) SYN_LAUNCH_LH(8, true, false)
                    else if (nw == 12) SYN_LAUNCH_LH(12, true, PROFH) else SYN_LAUNCH_LH(16, true, PROFH)
                } else {
                    if (nw == 4) SYN_LAUNCH_LH(4, false, false) else SYN_LAUNCH_LH(8, false, false)
                }
#undef SYN_LAUNCH_LH
            } else
            if (ref_family) {
                if (nw == 8) SYN_LAUNCH_LR(8) else if (nw == 12) SYN_LAUNCH_LR(12) else SYN_LAUNCH_LR(16)
            } else
            if (conv) {
                if (nw == 4) { if (fast) SYN_LAUNCH_LC(4, true) else SYN_LAUNCH_LC(4, false) }
                else if (nw == 8) { if (fast) SYN_LAUNCH_LC(8, true) else SYN_LAUNCH_LC(8, false) }
                else { if (fast) SYN_LAUNCH_LC(16, true) else SYN_LAUNCH_LC(16, false) }
            } else
            if (nw == 4) { if (fast) SYN_LAUNCH_L(4, true) else SYN_LAUNCH_L(4, false) }
            else if (nw == 8) { if (fast) SYN_LAUNCH_L(8, true) else SYN_LAUNCH_L(8, false) }
            else if (nw == 12) { if (fast) SYN_LAUNCH_L(12, true) else SYN_LAUNCH_L(12, false) }
            else { if (fast) SYN_LAUNCH_L(16, true) else SYN_LAUNCH_L(16, false) }
#undef SYN_LAUNCH_L
#undef SYN_LAUNCH_LC
#undef SYN_LAUNCH_LR
            h->last_shape = 4; h->last_grid = lgrid; h->last_threads = 64 * nw;
            if (out_grid) *out_grid = -lgrid;  // negative: lane kernel (profile layout differs)
            if (out_nt) *out_nt = 64 * nw;
            return hipGetLastError();
        }
    }
    int nq = 0;
    {
        int per_cu = (grid + h->num_cus - 1) / h->num_cus;
        nq = per_cu <= 2 ? 0 : (per_cu >= 4 ? 4 : 3);
        if (const char* ev = debug_env("SYN_QUADS")) nq = std::atoi(ev);
        if (nq < 0 || nq == 1 || nq > 4) nq = 0;
    }
    if (nq >= 2) {
        int qgrid = (grid + nq - 1) / nq;
#define SYN_LAUNCH_Q(NQ, FAST)                                                                                     \
    {                                                                                                              \
        auto k = selfplay_kernel_quads<MODE, COUNT, FAST, NQ, PROF>;                                                    \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)QuadLds<NQ>::BYTES);   \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL(k, dim3(qgrid), dim3(256 * NQ), QuadLds<NQ>::BYTES, h->stream, P);                      \
    }
        if (nq == 2) { if (fast) SYN_LAUNCH_Q(2, true) else SYN_LAUNCH_Q(2, false) }
        else if (nq == 3) { if (fast) SYN_LAUNCH_Q(3, true) else SYN_LAUNCH_Q(3, false) }
        else { if (fast) SYN_LAUNCH_Q(4, true) else SYN_LAUNCH_Q(4, false) }
#undef SYN_LAUNCH_Q
        h->last_shape = 3; h->last_grid = qgrid; h->last_threads = 256 * nq;
        if (out_grid) *out_grid = qgrid;
        if (out_nt) *out_nt = 256 * nq;
        return hipGetLastError();
    }
    if (grid <= h->num_cus) {
        if (fast) SYN_LAUNCH(1, true) else SYN_LAUNCH(1, false)
    } else {
        if (fast) SYN_LAUNCH(2, true) else SYN_LAUNCH(2, false)
    }
#undef SYN_LAUNCH
    h->last_shape = grid <= h->num_cus ? 1 : 2; h->last_grid = grid; h->last_threads = 256;
    if (out_grid) *out_grid = grid;
    if (out_nt) *out_nt = 256;
    return hipGetLastError();
}

// MCTS over RolloutPolicy: always the lane-per-tree kernel (8 waves per workgroup), RolloutPolicy instead of the network
static hipError_t launch_rollout_search(syn_engine* h, const EngineParams& P, int jobs) {
    if (h->cap > LANE_MAX_CAP) return hipErrorInvalidValue;
    const int nw = 8;
    int want_slots = h->slots < jobs ? h->slots : jobs;
    int lgrid = (want_slots + 64 * nw - 1) / (64 * nw);
    if (lgrid < 1) lgrid = 1;
    const size_t need_path = (size_t)lgrid * nw * PATH_ENTRIES * sizeof(uint4);
    if (need_path > h->path_bytes) {
        if (h->d_path) (void)hipFree(h->d_path);
        h->d_path = nullptr;
        h->path_bytes = 0;
        hipError_t pe = hipMalloc(&h->d_path, need_path);
        if (pe != hipSuccess) return pe;
        h->path_bytes = need_path;
    }
    EngineParams PL = P;
    PL.path = h->d_path;
    PL.lane_thresh = 64;
    PL.cache = nullptr;
    auto k = selfplay_kernel_lanes<MODE_SEARCH, false, false, 8, false, 1>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LaneLds<8>::BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(lgrid), dim3(64 * nw), LaneLds<8>::BYTES, h->stream, PL);
    h->last_shape = 4; h->last_grid = lgrid; h->last_threads = 64 * nw;
    return hipGetLastError();
}

extern "C" {

void syn_default_rollout_config(syn_rollout_config* cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->num_explores = 800;
    cfg->random_actions_until = 1;
    cfg->sample_actions_until = 30;
    cfg->stop_games_when_solved = 0;
    cfg->value_target = SYN_VALUE_Q;
    cfg->action = SYN_ACTION_NUM_VISITS;
    cfg->mcts_cfg.exploration = SYN_EXPLORATION_POLYNOMIAL_UCT;
    cfg->mcts_cfg.c = 3.0f;
    cfg->mcts_cfg.solve = 1;
    cfg->mcts_cfg.correct_values_on_solve = 1;
    cfg->mcts_cfg.select_solved_nodes = 1;
    cfg->mcts_cfg.auto_extend = 1;
    cfg->mcts_cfg.fpu = SYN_FPU_CONST;
    cfg->mcts_cfg.fpu_value = 1.0f;
    cfg->mcts_cfg.root_policy_noise = SYN_NOISE_NONE;
}

const char* syn_last_error(const syn_engine* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int syn_engine_create(const syn_engine_config* cfg, int device, syn_engine** out) {
    if (!cfg || !out) return fail(nullptr, SYN_ERR_INVALID_ARGUMENT, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->concurrent_games < 1 || cfg->max_explores < 1)
        return fail(nullptr, SYN_ERR_INVALID_ARGUMENT, "concurrent_games and max_explores must be >= 1");
    if (cfg->policy_cache_log2 != 0 && (cfg->policy_cache_log2 < 10 || cfg->policy_cache_log2 > 30))
        return fail(nullptr, SYN_ERR_INVALID_ARGUMENT, "policy_cache_log2 must be 0 (off) or 10..30");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, SYN_ERR_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= ndev)
        return fail(nullptr, SYN_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
    syn_engine* h = new (std::nothrow) syn_engine();
    if (!h) return fail(nullptr, SYN_ERR_HIP, "out of host memory");
    h->device = device;
    auto bail = [&](const char* what, hipError_t e) {
        int rc = fail(nullptr, SYN_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        syn_engine_destroy(h);
        return rc;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        int rc = fail(nullptr, SYN_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                      prop.gcnArchName);
        syn_engine_destroy(h);
        return rc;
    }
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->slots = ((cfg->concurrent_games + 15) / 16) * 16;
    h->max_explores = cfg->max_explores;
    if (const char* ev = debug_env("SYN_EVAL_POLL_MAX")) h->eval_poll_max = (size_t)std::atoll(ev);
    if (const char* ev = debug_env("SYN_EVAL_ZC_OUT")) h->eval_zero_copy_out = (size_t)std::atoll(ev);
    // nodes.len() <= 1 + 9*(explores+1), rounded up to keep slabs 16-byte-record aligned per 4 nodes
    h->cap = (uint32_t)(1 + 9 * (cfg->max_explores + 1));
    h->cap = (h->cap + 3u) & ~3u;
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    {
        // the progress / cancel stream must never share a hardware queue with the launch stream (it would wait behind the running
        // kernel): the runtime deals streams of one priority round-robin over a few queues, so after enough streams in a process
        // two of them coincide — a different priority class has its own queues
        int lo = 0, hi = 0;
        if ((e = hipDeviceGetStreamPriorityRange(&lo, &hi)) != hipSuccess) return bail("hipDeviceGetStreamPriorityRange", e);
        if ((e = hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, hi)) != hipSuccess) return bail("hipStreamCreate", e);
    }
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h->h_pin), 64)) != hipSuccess) return bail("hipHostMalloc", e);
    if ((e = hipEventCreate(&h->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&h->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    // the launch may round the slot count up to a whole workgroup (<= 1024 trees: lane kernel)
    // + 48: the 3-quad launch rounds to multiples of 48 slots
    // (768: the 12-wave lane kernel rounds to multiples of 768 slots)
    // (2304: the producer/consumer kernel rounds to multiples of 768 x NV slots, NV <= 3)
    // (1536: the two-trees-per-lane kernel with 12 waves rounds to multiples of 1,536 slots)
    h->pool_slots = ((h->slots + 1023) / 1024) * 1024 + 2304;
    size_t nodes = (size_t)h->pool_slots * h->cap;
    if ((e = hipMalloc(&h->d_stat, nodes * 32)) != hipSuccess) return bail("hipMalloc(node pool)", e);
    h->d_edge = reinterpret_cast<uint4*>(h->d_stat);  // same records, edge half = odd 16-byte elements
    if (cfg->policy_cache_log2 != 0) {
        h->cache_log2 = cfg->policy_cache_log2;
        const size_t bytes = (size_t)64 << h->cache_log2;
        if ((e = hipMalloc(&h->d_cache, bytes)) != hipSuccess) return bail("hipMalloc(policy cache)", e);
        // empty: an all-zero entry never verifies. (On the engine's own stream: a memset on the null stream is asynchronous to the host
        // and not ordered against a non-blocking stream's launches.)
        if ((e = hipMemsetAsync(h->d_cache, 0, bytes, h->stream)) != hipSuccess) return bail("hipMemsetAsync(policy cache)", e);
    }
    if ((e = hipMalloc(&h->d_cache_stats, 16)) != hipSuccess) return bail("hipMalloc(cache stats)", e);
    if ((e = hipMalloc(&h->d_wimg, MlpGeom::IMG_FLOATS * sizeof(float))) != hipSuccess) return bail("hipMalloc(wimg)", e);
    if ((e = hipMalloc(&h->d_job_next, 64)) != hipSuccess) return bail("hipMalloc(job)", e);
    if ((e = hipMemsetAsync(h->d_job_next, 0, 64, h->stream)) != hipSuccess) return bail("hipMemsetAsync(job)", e);  // syn_progress before the first launch reads zeros
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return bail("hipStreamSynchronize", e);
    if ((e = hipMalloc(&h->d_counters, sizeof(DevCounters))) != hipSuccess) return bail("hipMalloc(counters)", e);
    *out = h;
    return SYN_OK;
}

int syn_engine_destroy(syn_engine* h) {
    if (!h) return SYN_OK;
    hipSetDevice(h->device);
    if (h->eval_ctx) syn_eval_ctx_destroy(h->eval_ctx);
    if (h->stream) hipStreamSynchronize(h->stream);
    hipFree(h->d_stat);
    hipFree(h->d_wimg);
    hipFree(h->d_wimg16);
    hipFree(h->d_job_next);
    hipFree(h->d_path);
    hipFree(h->d_vw);
    hipFree(h->d_train_data);
    hipFree(h->d_cache);
    hipFree(h->d_cache_stats);
    hipFree(h->d_counters);
    hipFree(h->d_plies);
    hipFree(h->d_states);
    hipFree(h->d_pis);
    hipFree(h->d_vs);
    hipFree(h->d_actions);
    hipFree(h->d_root_nodes);
    hipFree(h->d_final);
    hipFree(h->d_scratch);
    hipFree(h->d_tw);
    hipFree(h->d_tm);
    hipFree(h->d_tv);
    hipFree(h->d_tgrad);
    hipFree(h->d_tloss);
    hipFree(h->d_twimg);
    hipFree(h->d_ttimg);
    hipFree(h->d_timg2);
    hipFree(h->d_tsync);
    hipFree(h->d_tsnap);
    hipFree(h->d_cxbuf);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    if (h->aux_stream) hipStreamDestroy(h->aux_stream);
    if (h->h_pin) hipHostFree(h->h_pin);
    delete h;
    return SYN_OK;
}

// The f16x2 image of the engine's Connect4Net (f16x2_tile.cuh: build_f16x2_image). State is committed only after the image exists:
// a blob without an f16x2 plan (non-finite values, scales outside the window) leaves the engine exactly as it was — arithmetic,
// parameters, both images and the policy cache — so nothing can evaluate a stale or missing image afterwards.
static int upload_f16x2_image(syn_engine* h, const F16Image& im) {
    HIP_TRY(h, hipSetDevice(h->device));   // (reached through common_params before the entry point's own hipSetDevice)
    if (!h->d_wimg16) HIP_TRY(h, hipMalloc(&h->d_wimg16, (size_t)F16Geom::IMG_WORDS * 4));
    HIP_TRY(h, hipMemcpyAsync(h->d_wimg16, im.words.data(), (size_t)F16Geom::IMG_WORDS * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SYN_OK;
}
static int no_f16x2_plan(syn_engine* h) {
    return fail(h, SYN_ERR_UNSUPPORTED, "these parameters have no f16x2 plan (non-finite values or scales outside the f32-safe window); "
                                        "the engine keeps its previous network and arithmetic");
}
// (re)builds the image from the engine's own copy of the parameters when it is not current (after syn_trainer_publish_weights)
static int ensure_f16x2_image(syn_engine* h) {
    if (h->img16_current) return SYN_OK;
    if (h->net_kind != 0 || h->host_blob.size() != (size_t)MlpGeom::NUM_PARAMS)
        return fail(h, SYN_ERR_UNSUPPORTED, "the f16x2 arithmetic exists for Connect4Net only");
    F16Image im;
    if (!build_f16x2_image(h->host_blob.data(), im))
        return fail(h, SYN_ERR_UNSUPPORTED, "these parameters have no f16x2 plan (non-finite values or scales outside the f32-safe window)");
    const int rc = upload_f16x2_image(h, im);
    if (rc != SYN_OK) return rc;
    h->img16_current = true;
    return SYN_OK;
}
// every path that evaluates Connect4Net in the f16x2 arithmetic checks this first
static int require_f16x2_image(syn_engine* h) {
    if (h->net_arith != SYN_NET_ARITH_F16X2 || h->net_kind != 0) return SYN_OK;
    if (!h->img16_current || !h->d_wimg16)
        return fail(h, SYN_ERR_UNSUPPORTED, "the engine is in the f16x2 arithmetic but holds no f16x2 image of its parameters");
    return SYN_OK;
}

int syn_load_weights(syn_engine* h, const float* blob, size_t n_floats) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!blob) return fail(h, SYN_ERR_INVALID_ARGUMENT, "blob is NULL");
    if (n_floats != (size_t)MlpGeom::NUM_PARAMS)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "Connect4Net has %d parameters, got %zu", MlpGeom::NUM_PARAMS, n_floats);
    HIP_TRY(h, hipSetDevice(h->device));
    F16Image im16;
    const bool want16 = h->net_arith == SYN_NET_ARITH_F16X2;
    if (want16 && !build_f16x2_image(blob, im16)) return no_f16x2_plan(h);   // (before anything of the engine changes)
    std::vector<float> img;
    build_weight_image(blob, img);
    if (want16) {
        h->img16_current = false;
        const int rc = upload_f16x2_image(h, im16);
        if (rc != SYN_OK) return rc;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_wimg, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    // PolicyWithCache entries belong to the network that produced them (the reference builds a fresh cache per
    // run_n_games, alpha_zero.rs:196-198): a new network starts with an empty table
    if (h->d_cache) HIP_TRY(h, hipMemsetAsync(h->d_cache, 0, (size_t)64 << h->cache_log2, h->stream));
    h->has_weights = true;
    h->net_kind = 0;
    h->host_blob.assign(blob, blob + n_floats);
    h->img16_current = want16;
    return SYN_OK;
}

int syn_set_network_arithmetic(syn_engine* h, int arithmetic) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (arithmetic != SYN_NET_ARITH_F32 && arithmetic != SYN_NET_ARITH_F16X2)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown network arithmetic %d", arithmetic);
    if (arithmetic == h->net_arith) return SYN_OK;
    if (arithmetic == SYN_NET_ARITH_F16X2) {
        if (h->cap > LANE_MAX_CAP)
            return fail(h, SYN_ERR_UNSUPPORTED, "the f16x2 arithmetic runs in the lane-per-tree kernels: max_explores must be <= %u",
                        (LANE_MAX_CAP - 1u) / 9u - 1u);
        if (h->has_weights && h->net_kind != 0)
            return fail(h, SYN_ERR_UNSUPPORTED, "the f16x2 arithmetic exists for Connect4Net only (the engine holds Connect4ConvNet)");
    }
    HIP_TRY(h, hipSetDevice(h->device));
    if (arithmetic == SYN_NET_ARITH_F16X2 && h->has_weights && !h->img16_current) {
        // the image first; the switch is committed only when it exists
        if (h->net_kind != 0 || h->host_blob.size() != (size_t)MlpGeom::NUM_PARAMS)
            return fail(h, SYN_ERR_UNSUPPORTED, "the f16x2 arithmetic exists for Connect4Net only");
        F16Image im;
        if (!build_f16x2_image(h->host_blob.data(), im)) return no_f16x2_plan(h);
        const int rc = upload_f16x2_image(h, im);
        if (rc != SYN_OK) return rc;
        h->img16_current = true;
    }
    h->net_arith = arithmetic;
    // PolicyWithCache entries belong to the arithmetic that produced them: the two differ in the last bits
    if (h->d_cache) {
        HIP_TRY(h, hipMemsetAsync(h->d_cache, 0, (size_t)64 << h->cache_log2, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return SYN_OK;
}

int syn_get_network_arithmetic(syn_engine* h, int* arithmetic, syn_f16x2_plan* plan) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (arithmetic) *arithmetic = h->net_arith;
    if (plan) {
        std::memset(plan, 0, sizeof(*plan));
        if (h->has_weights && h->net_kind == 0 && !h->host_blob.empty()) {
            F16Image im;
            if (build_f16x2_image(h->host_blob.data(), im)) {
                plan->valid = 1;
                for (int l = 0; l < 5; l++) { plan->activation_exp[l] = im.s[l]; plan->weight_exp[l] = im.t[l]; plan->bound[l] = im.bound[l]; }
                plan->out_exp = im.out_exp;
            }
        }
    }
    return SYN_OK;
}

int syn_f16x2_plan_of_blob(const float* blob, size_t n_floats, syn_f16x2_plan* plan) {
    if (!blob || !plan || n_floats != (size_t)MlpGeom::NUM_PARAMS) return SYN_ERR_INVALID_ARGUMENT;
    std::memset(plan, 0, sizeof(*plan));
    F16Image im;
    if (build_f16x2_image(blob, im)) {
        plan->valid = 1;
        for (int l = 0; l < 5; l++) { plan->activation_exp[l] = im.s[l]; plan->weight_exp[l] = im.t[l]; plan->bound[l] = im.bound[l]; }
        plan->out_exp = im.out_exp;
    }
    return SYN_OK;
}

int syn_load_weights_conv(syn_engine* h, const float* blob, size_t n_floats) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!blob) return fail(h, SYN_ERR_INVALID_ARGUMENT, "blob is NULL");
    if (n_floats != (size_t)ConvGeom::NUM_PARAMS)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "Connect4ConvNet has %d parameters, got %zu", ConvGeom::NUM_PARAMS, n_floats);
    if (h->cap > LANE_MAX_CAP)
        return fail(h, SYN_ERR_UNSUPPORTED, "Connect4ConvNet runs in the lane-per-tree kernels: max_explores must be <= %d",
                    (LANE_MAX_CAP - 1) / 9 - 1);
    HIP_TRY(h, hipSetDevice(h->device));
    std::vector<float> img((size_t)ConvGeom::IMG_FLOATS);
    build_conv_image(blob, img.data());
    HIP_TRY(h, hipMemcpyAsync(h->d_wimg, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->d_cache) HIP_TRY(h, hipMemsetAsync(h->d_cache, 0, (size_t)64 << h->cache_log2, h->stream));  // a new network: empty cache
    h->has_weights = true;
    h->net_kind = 1;
    h->host_blob.clear();
    h->img16_current = false;
    h->net_arith = SYN_NET_ARITH_F32;   // (Connect4ConvNet has the f32 arithmetic only)
    return SYN_OK;
}

constexpr size_t EVAL_TILE_LDS = 14 * 64 * 16;   // policy_eval_tile_kernel: the two activation-exchange buffers
constexpr size_t EVAL_TILE_MAX = 4096;           // ... is used up to this many positions (256 tiles: one per CU)

// The evaluation kernel of the engine's network on `st` (any stream of the engine's device): n positions, pointers the device can
// read / write (device memory or pinned, device-mapped host memory).
static hipError_t launch_policy_eval(syn_engine* h, hipStream_t st, const uint64_t* d_my, const uint64_t* d_op, int n,
                                     float* d_logits, float* d_value) {
    // Two waves per SIMD (512 threads): one wave's LDS reads / feature math overlap the other's MFMAs. Large batches (every
    // wave gets several tiles) run three waves per SIMD, which also hides the loads and stores around the tiles.
    const int ntiles = (n + 15) / 16;
    hipError_t e;
#define SYN_LAUNCH_EVAL(KERNEL, NT, LDS, SLOT)                                                                                 \
    {                                                                                                                          \
        auto k = KERNEL<NT>;                                                                                                   \
        if (!h->eval_attr_set[SLOT].load(std::memory_order_acquire)) { /* once per engine: the call costs a microsecond */     \
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                         (int)(LDS))) != hipSuccess)                                                           \
                return e;                                                                                                      \
            h->eval_attr_set[SLOT].store(true, std::memory_order_release);                                                     \
        }                                                                                                                      \
        int grid = (ntiles + NT / 64 - 1) / (NT / 64);                                                                         \
        if (grid > h->num_cus) grid = h->num_cus;                                                                              \
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), (LDS), st, h->d_wimg, reinterpret_cast<const unsigned long long*>(d_my),    \
                           reinterpret_cast<const unsigned long long*>(d_op), n, d_logits, d_value);                           \
    }
    if (h->net_kind == 0 && h->net_arith == SYN_NET_ARITH_F16X2) {
        // Connect4Net in the f16x2 arithmetic: the throughput kernel at every size (its tile is 3x shorter than the f32 one's)
#define SYN_LAUNCH_EVAL16(NT, SLOT)                                                                                            \
    {                                                                                                                          \
        auto k = policy_eval_f16x2_kernel<NT>;                                                                                 \
        if (!h->eval_attr_set[SLOT].load(std::memory_order_acquire)) {                                                         \
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                         (int)(F16Geom::IMG_WORDS * 4))) != hipSuccess)                                        \
                return e;                                                                                                      \
            h->eval_attr_set[SLOT].store(true, std::memory_order_release);                                                     \
        }                                                                                                                      \
        int grid = (ntiles + NT / 64 - 1) / (NT / 64);                                                                         \
        if (grid > h->num_cus) grid = h->num_cus;                                                                              \
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), (size_t)F16Geom::IMG_WORDS * 4, st, h->d_wimg16,                           \
                           reinterpret_cast<const unsigned long long*>(d_my), reinterpret_cast<const unsigned long long*>(d_op), n, \
                           d_logits, d_value);                                                                                 \
    }
        if (ntiles >= h->num_cus * 16 * 4) SYN_LAUNCH_EVAL16(1024, 3) else SYN_LAUNCH_EVAL16(512, 4)
#undef SYN_LAUNCH_EVAL16
    } else
    if (h->net_kind == 0 && (size_t)n <= EVAL_TILE_MAX) {
        // at most a tile per CU: the latency kernel (eval_small.cuh), one workgroup per tile, no weight staging
        hipLaunchKernelGGL(policy_eval_tile_kernel, dim3((unsigned)ntiles), dim3(256), EVAL_TILE_LDS, st, h->d_wimg,
                           reinterpret_cast<const unsigned long long*>(d_my), reinterpret_cast<const unsigned long long*>(d_op), n, d_logits,
                           d_value, nullptr, nullptr, 0u);
    } else if (h->net_kind == 1) {
        // (16 waves per CU measured the same 46 % of the MFMA peak as 8: the tile is issue-bound, not latency-bound)
        SYN_LAUNCH_EVAL(policy_eval_conv_kernel, 512, (size_t)ConvGeom::IMG_FLOATS * 4, 0)
    } else if (ntiles >= h->num_cus * 12 * 4) {
        SYN_LAUNCH_EVAL(policy_eval_kernel, 768, (size_t)MlpGeom::IMG_FLOATS * 4, 1)
    } else {
        SYN_LAUNCH_EVAL(policy_eval_kernel, 512, (size_t)MlpGeom::IMG_FLOATS * 4, 2)
    }
#undef SYN_LAUNCH_EVAL
    return hipGetLastError();
}

int syn_policy_eval_batch_device(syn_engine* h, const uint64_t* d_my, const uint64_t* d_op, int n, float* d_logits,
                                 float* d_value, int sync) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (n < 0 || (n > 0 && (!d_my || !d_op || !d_logits || !d_value)))
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_policy_eval_batch_device");
    if (!h->has_weights) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_load_weights first");
    if (n == 0) return SYN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    { const int rc16 = require_f16x2_image(h); if (rc16 != SYN_OK) return rc16; }
    HIP_TRY(h, launch_policy_eval(h, h->stream, d_my, d_op, n, d_logits, d_value));
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    h->last_launches = 1;
    if (sync) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipEventElapsedTime(&h->last_kernel_ms, h->ev0, h->ev1));
    }
    return SYN_OK;
}

// ---- evaluation contexts: one worker's policy (alpha_zero.rs:192-198 gives every worker thread of gather_experience its own) ----
// A context owns a stream, pinned staging and device scratch, and reads the engine's weight image; contexts of one engine run
// side by side from different host threads. Errors stay in the context (the engine's error slot belongs to the engine's thread).
// one polite spin-wait step on the host CPU
static inline void spin_pause() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

struct syn_eval_ctx {
    syn_engine* h = nullptr;
    hipStream_t stream = nullptr;
    void* h_stage = nullptr;   // [my n][op n][logits 9n][value 3n]
    void* d_out = nullptr;     // results of batches too large to be written across the host link in place
    size_t cap = 0;            // positions the two buffers hold
    int pending = 0;           // positions of the submitted batch (0 = none)
    bool out_in_place = false;
    unsigned* d_done = nullptr;   // the latency kernels' completion protocol (eval_small.cuh): workgroups finished, device word
    unsigned* h_flag = nullptr;   // ... and the word in pinned host memory the last one stores the call's sequence number into
    unsigned seq = 0;
    bool polled = false;          // the submitted batch signals through h_flag
    bool poll_broken = false;     // a completion word went missing once: stream synchronisation from then on
    std::string err;
};
static int ctx_fail(syn_eval_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
    c->err = what;
    if (e != hipSuccess) c->err += std::string(": ") + hipGetErrorString(e);
    return code;
}
#define CTX_TRY(c, call)                                                  \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return ctx_fail(c, SYN_ERR_HIP, #call, e_); \
    } while (0)

int syn_eval_ctx_create(syn_engine* h, syn_eval_ctx** out) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!out) return fail(h, SYN_ERR_INVALID_ARGUMENT, "syn_eval_ctx_create: out is NULL");
    *out = nullptr;
    HIP_TRY(h, hipSetDevice(h->device));
    syn_eval_ctx* c = new (std::nothrow) syn_eval_ctx;
    if (!c) return fail(h, SYN_ERR_HIP, "out of host memory");
    c->h = h;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_done), 64);
    // (stream-ordered in front of the context's kernels; SYN_DEBUG=1 SYN_EVAL_BREAK_COUNTER=1 starts the counter off wrong: the test of
    // syn_eval_ctx_wait's way out when a completion word goes missing)
    if (e == hipSuccess) e = hipMemsetAsync(c->d_done, debug_env("SYN_EVAL_BREAK_COUNTER") ? 1 : 0, 64, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->h_flag), 64, hipHostMallocDefault);
    if (e != hipSuccess) {
        syn_eval_ctx_destroy(c);
        return fail(h, SYN_ERR_HIP, "syn_eval_ctx_create: %s", hipGetErrorString(e));
    }
    *c->h_flag = 0u;
    *out = c;
    return SYN_OK;
}

int syn_eval_ctx_destroy(syn_eval_ctx* c) {
    if (!c) return SYN_OK;
    hipSetDevice(c->h->device);
    if (c->stream) {
        hipStreamSynchronize(c->stream);
        hipStreamDestroy(c->stream);
    }
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->d_out) hipFree(c->d_out);
    if (c->d_done) hipFree(c->d_done);
    if (c->h_flag) hipHostFree(c->h_flag);
    delete c;
    return SYN_OK;
}

const char* syn_eval_ctx_last_error(const syn_eval_ctx* c) { return c ? c->err.c_str() : "context is NULL"; }

int syn_eval_ctx_submit(syn_eval_ctx* c, const uint64_t* my_bb, const uint64_t* op_bb, int n) {
    if (!c) return SYN_ERR_INVALID_ARGUMENT;
    if (n < 0 || (n > 0 && (!my_bb || !op_bb))) return ctx_fail(c, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_eval_ctx_submit");
    if (c->pending != 0) return ctx_fail(c, SYN_ERR_INVALID_ARGUMENT, "syn_eval_ctx_submit: the previous batch has not been waited for");
    syn_engine* h = c->h;
    if (!h->has_weights) return ctx_fail(c, SYN_ERR_NO_WEIGHTS, "call syn_load_weights first");
    if (h->net_arith == SYN_NET_ARITH_F16X2 && h->net_kind == 0 && (!h->img16_current || !h->d_wimg16))
        return ctx_fail(c, SYN_ERR_UNSUPPORTED, "the engine is in the f16x2 arithmetic but holds no f16x2 image of its parameters");
    if (n == 0) return SYN_OK;
    CTX_TRY(c, hipSetDevice(h->device));
    const size_t nb = (size_t)n;
    if (nb > c->cap) {
        CTX_TRY(c, hipStreamSynchronize(c->stream));
        if (c->h_stage) CTX_TRY(c, hipHostFree(c->h_stage));
        if (c->d_out) CTX_TRY(c, hipFree(c->d_out));
        c->h_stage = c->d_out = nullptr;
        c->cap = 0;
        const size_t want = nb + nb / 4 + 256;
        CTX_TRY(c, hipHostMalloc(&c->h_stage, want * 64, hipHostMallocDefault));
        CTX_TRY(c, hipMalloc(&c->d_out, want * 48));
        c->cap = want;
    }
    uint64_t* s_my = static_cast<uint64_t*>(c->h_stage);
    uint64_t* s_op = s_my + nb;
    float* s_logits = reinterpret_cast<float*>(s_op + nb);
    std::memcpy(s_my, my_bb, nb * 8);
    std::memcpy(s_op, op_bb, nb * 8);
    // the positions are read across the host link in place (16 B each); small results are written in place, larger ones come back
    // with one DMA (syn_policy_eval_batch, measured)
    c->out_in_place = nb <= h->eval_zero_copy_out;
    float* o_logits = c->out_in_place ? s_logits : static_cast<float*>(c->d_out);
    c->polled = c->out_in_place && h->net_kind == 0 && h->net_arith == SYN_NET_ARITH_F32 && nb <= h->eval_poll_max && !c->poll_broken;
    if (c->polled) {
        // the latency kernels: their last workgroup stores this call's number into pinned memory, syn_eval_ctx_wait polls it
        c->seq += 1u;
        const unsigned long long* k_my = reinterpret_cast<const unsigned long long*>(s_my);
        const unsigned long long* k_op = reinterpret_cast<const unsigned long long*>(s_op);
        hipLaunchKernelGGL(policy_eval_tile_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), EVAL_TILE_LDS, c->stream, h->d_wimg, k_my, k_op,
                           n, o_logits, o_logits + nb * 9, c->d_done, c->h_flag, c->seq);
        CTX_TRY(c, hipGetLastError());
    } else {
        CTX_TRY(c, launch_policy_eval(h, c->stream, s_my, s_op, n, o_logits, o_logits + nb * 9));
        if (!c->out_in_place) CTX_TRY(c, hipMemcpyAsync(s_logits, c->d_out, nb * 48, hipMemcpyDeviceToHost, c->stream));
    }
    c->pending = n;
    return SYN_OK;
}

int syn_eval_ctx_wait(syn_eval_ctx* c, float* logits, float* value) {
    if (!c) return SYN_ERR_INVALID_ARGUMENT;
    if (c->pending == 0) return SYN_OK;
    if (!logits || !value) return ctx_fail(c, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_eval_ctx_wait");
    const size_t nb = (size_t)c->pending;
    c->pending = 0;
    // (under CombiningPolicy the wait runs on whichever worker thread finishes the batch: that thread may never have selected the device)
    CTX_TRY(c, hipSetDevice(c->h->device));
    bool done = false;
    if (c->polled) {
        // at most ~300 us of polling (a call is tens of microseconds; behind a long self-play launch the kernel is far away and a
        // burning core buys nothing); a kernel that has not reported by then falls through to the stream, where a fault shows as an error
        const volatile unsigned* flag = c->h_flag;
        const auto t_poll = std::chrono::steady_clock::now();
        for (int spin = 0; !done; spin++) {
            done = *flag == c->seq;
            if (done) break;
            spin_pause();
            if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t_poll > std::chrono::microseconds(300)) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!done) {
        CTX_TRY(c, hipStreamSynchronize(c->stream));
        if (c->polled && *static_cast<const volatile unsigned*>(c->h_flag) != c->seq) {
            // The stream is drained, so the kernel has finished and its results are complete — only the completion word is missing
            // (the workgroup counter was disturbed). Not an error for the caller: note it, put the counter back and let this context
            // wait on its stream from now on.
            unsigned done_word = 0xFFFFFFFFu;
            if (hipMemcpy(&done_word, c->d_done, 4, hipMemcpyDeviceToHost) != hipSuccess) done_word = 0xFFFFFFFFu;   // (diagnostic only)
            char msg[256];
            std::snprintf(msg, sizeof msg, "note: an evaluation kernel finished without reporting completion (n %zu, call %u, word in pinned "
                          "memory %u, workgroups counted %u); this context now waits on its stream", nb, c->seq,
                          *static_cast<const volatile unsigned*>(c->h_flag), done_word);
            c->err = msg;
            c->poll_broken = true;
            CTX_TRY(c, hipMemsetAsync(c->d_done, 0, 64, c->stream));
            CTX_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    const float* s_logits = reinterpret_cast<const float*>(static_cast<const uint64_t*>(c->h_stage) + 2 * nb);
    std::memcpy(logits, s_logits, nb * 36);
    std::memcpy(value, s_logits + nb * 9, nb * 12);
    return SYN_OK;
}

int syn_eval_ctx_eval(syn_eval_ctx* c, const uint64_t* my_bb, const uint64_t* op_bb, int n, float* logits, float* value) {
    const int rc = syn_eval_ctx_submit(c, my_bb, op_bb, n);
    return rc != SYN_OK ? rc : syn_eval_ctx_wait(c, logits, value);
}

int syn_policy_eval_batch(syn_engine* h, const uint64_t* my_bb, const uint64_t* op_bb, int n, float* logits,
                          float* value) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (n < 0 || (n > 0 && (!my_bb || !op_bb || !logits || !value)))
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_policy_eval_batch");
    if (!h->has_weights) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_load_weights first");
    if (n == 0) return SYN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nb = (size_t)n;
    constexpr size_t EVAL_STAGE_MAX = 32768;  // beyond it two host copies cost more than the runtime's pageable path (measured)
    if (nb <= EVAL_STAGE_MAX && debug_env("SYN_EVAL_PAGEABLE") == nullptr) {
        // A call of this size is latency, not bandwidth (a Rust `impl Policy` adaptor calls with n = 1, a host-tree driver with the
        // leaves of one round): the engine's own evaluation context — positions read and small results written in pinned host
        // memory in place, the latency kernel, completion polled (syn_eval_ctx_* above). Pageable transfers cost a staging copy and
        // a synchronisation each inside the runtime, four per call.
        if (!h->eval_ctx) {
            const int rc = syn_eval_ctx_create(h, &h->eval_ctx);
            if (rc != SYN_OK) return rc;
        }
        const int rc = syn_eval_ctx_eval(h->eval_ctx, my_bb, op_bb, n, logits, value);
        if (rc != SYN_OK) return fail(h, rc, "%s", h->eval_ctx->err.c_str());
        h->last_launches = 1;
        h->last_kernel_ms = 0.0f;   // (not measured on this path: no events in a latency call)
        return SYN_OK;
    }
    // large batches: the runtime's own chunked transfers from / to the pageable buffers around the throughput kernel
    int rc = ensure_scratch(h, nb * (8 + 8 + 36 + 12) + 256);
    if (rc != SYN_OK) return rc;
    char* base = static_cast<char*>(h->d_scratch);
    uint64_t* d_my = reinterpret_cast<uint64_t*>(base);
    uint64_t* d_op = d_my + nb;
    float* d_logits = reinterpret_cast<float*>(d_op + nb);
    float* d_value = d_logits + nb * 9;
    HIP_TRY(h, hipMemcpyAsync(d_my, my_bb, nb * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_op, op_bb, nb * 8, hipMemcpyHostToDevice, h->stream));
    rc = syn_policy_eval_batch_device(h, d_my, d_op, n, d_logits, d_value, 0);
    if (rc != SYN_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(logits, d_logits, nb * 36, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(value, d_value, nb * 12, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipEventElapsedTime(&h->last_kernel_ms, h->ev0, h->ev1));
    return SYN_OK;
}

int syn_features_batch(syn_engine* h, const uint64_t* my_bb, const uint64_t* op_bb, int n, float* out) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (n < 0 || (n > 0 && (!my_bb || !op_bb || !out))) return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments");
    if (n == 0) return SYN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    size_t nb = (size_t)n;
    int rc = ensure_scratch(h, nb * (16 + 252) + 256);
    if (rc != SYN_OK) return rc;
    uint64_t* d_my = reinterpret_cast<uint64_t*>(h->d_scratch);
    uint64_t* d_op = d_my + nb;
    float* d_out = reinterpret_cast<float*>(d_op + nb);
    HIP_TRY(h, hipMemcpyAsync(d_my, my_bb, nb * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_op, op_bb, nb * 8, hipMemcpyHostToDevice, h->stream));
    size_t total = nb * 63;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    if (n <= (1 << 25)) {
        grid = (int)((total / 4 + 256) / 256);
        if (grid > 16 * h->num_cus) grid = 16 * h->num_cus;
        hipLaunchKernelGGL(features4_kernel, dim3(grid), dim3(256), 0, h->stream,   // four features per thread, 16-byte stores
                           reinterpret_cast<const unsigned long long*>(d_my),
                           reinterpret_cast<const unsigned long long*>(d_op), n, d_out);
    } else {
        hipLaunchKernelGGL(features_kernel, dim3(grid), dim3(256), 0, h->stream,
                           reinterpret_cast<const unsigned long long*>(d_my),
                           reinterpret_cast<const unsigned long long*>(d_op), n, d_out);
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out, d_out, total * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipEventElapsedTime(&h->last_kernel_ms, h->ev0, h->ev1));
    h->last_launches = 1;
    return SYN_OK;
}

int syn_linear_forward(syn_engine* h, int I, int O, const float* W, const float* b, const float* x, int batch,
                       float* y, int relu) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (I < 1 || O < 1 || batch < 0 || !W || !b || (batch > 0 && (!x || !y)))
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_linear_forward");
    if (batch == 0) return SYN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    size_t nW = (size_t)I * O, nx = (size_t)batch * I, ny = (size_t)batch * O;
    int rc = ensure_scratch(h, (nW + O + nx + ny) * 4 + 256);
    if (rc != SYN_OK) return rc;
    float* dW = static_cast<float*>(h->d_scratch);
    float* db = dW + nW;
    float* dx = db + O;
    float* dy = dx + nx;
    HIP_TRY(h, hipMemcpyAsync(dW, W, nW * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(db, b, (size_t)O * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(dx, x, nx * 4, hipMemcpyHostToDevice, h->stream));
    int grid = (int)((ny + 255) / 256);
    if (grid > 2048) grid = 2048;
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    if (linear_tiled_lds_bytes(I, O) + 16 <= 64 * 1024 && batch >= LIN_SB) {
        // weights and a 64-sample input tile in LDS, 8 samples per thread (layer_kernels.cuh): VALU-bound, as slimnn's two-rounding
        // multiply-add demands
        int g2 = (batch + LIN_SB - 1) / LIN_SB;
        if (g2 > 4 * h->num_cus) g2 = 4 * h->num_cus;
        hipLaunchKernelGGL(linear_tiled_kernel, dim3(g2), dim3(256), linear_tiled_lds_bytes(I, O) + 16, h->stream, I, O, dW, db, dx,
                           batch, dy, relu);
    } else if (nW * 4 <= 64 * 1024) {
        // weights in LDS, transposed (the usual case: every layer of Connect4Net is <= 48 KB); few, long-lived workgroups so
        // that the staging is amortised
        if (grid > 4 * h->num_cus) grid = 4 * h->num_cus;
        hipLaunchKernelGGL(linear_kernel_lds, dim3(grid), dim3(256), nW * 4, h->stream, I, O, dW, db, dx, batch, dy, relu);
    } else {
        hipLaunchKernelGGL(linear_kernel, dim3(grid), dim3(256), 0, h->stream, I, O, dW, db, dx, batch, dy, relu);
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipMemcpyAsync(y, dy, ny * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipEventElapsedTime(&h->last_kernel_ms, h->ev0, h->ev1));
    h->last_launches = 1;
    return SYN_OK;
}

int syn_conv2d_forward(syn_engine* h, int CIN, int COUT, int K, int RP, int CP, int S, int H_IN, int W_IN, int H_OUT,
                       int W_OUT, const float* W, const float* b, const float* x, int batch, float* y, int relu) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (CIN < 1 || COUT < 1 || K < 1 || RP < 0 || CP < 0 || S < 1 || H_IN < 1 || W_IN < 1 || batch < 0 || !W || !b ||
        (batch > 0 && (!x || !y)))
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_conv2d_forward");
    // slimnn asserts these (conv.rs:50-51)
    if (W_IN + 2 * CP < K || H_IN + 2 * RP < K || W_OUT != ((W_IN + 2 * CP - K) / S) + 1 ||
        H_OUT != ((H_IN + 2 * RP - K) / S) + 1)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "output dims must be ((IN + 2*PAD - K) / STRIDE) + 1");
    if (batch == 0) return SYN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    size_t nW = (size_t)COUT * CIN * K * K, nx = (size_t)batch * CIN * H_IN * W_IN,
           ny = (size_t)batch * COUT * H_OUT * W_OUT;
    int rc = ensure_scratch(h, (nW + COUT + nx + ny) * 4 + 256);
    if (rc != SYN_OK) return rc;
    float* dW = static_cast<float*>(h->d_scratch);
    float* db = dW + nW;
    float* dx = db + COUT;
    float* dy = dx + nx;
    HIP_TRY(h, hipMemcpyAsync(dW, W, nW * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(db, b, (size_t)COUT * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(dx, x, nx * 4, hipMemcpyHostToDevice, h->stream));
    int grid = (int)((ny + 255) / 256);
    if (grid > 2048) grid = 2048;
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    {
        // input planes of SB samples + the weights in LDS (layer_kernels.cuh) when they fit 64 KB, else element-per-thread from global
        const size_t per_in = (size_t)CIN * H_IN * W_IN;
        size_t sb = (60 * 1024 - nW * 4) / (per_in * 4);
        if (nW * 4 < 48 * 1024 && sb >= 4) {
            if (sb > 64) sb = 64;
            int g2 = (int)(((size_t)batch + sb - 1) / sb);
            if (g2 > 8 * h->num_cus) g2 = 8 * h->num_cus;
            hipLaunchKernelGGL(conv2d_tiled_kernel, dim3(g2), dim3(256), (nW + sb * per_in) * 4, h->stream, CIN, COUT, K, RP, CP, S,
                               H_IN, W_IN, H_OUT, W_OUT, (int)sb, dW, db, dx, batch, dy, relu);
        } else {
            hipLaunchKernelGGL(conv2d_kernel, dim3(grid), dim3(256), 0, h->stream, CIN, COUT, K, RP, CP, S, H_IN, W_IN, H_OUT,
                               W_OUT, dW, db, dx, batch, dy, relu);
        }
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipMemcpyAsync(y, dy, ny * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipEventElapsedTime(&h->last_kernel_ms, h->ev0, h->ev1));
    h->last_launches = 1;
    return SYN_OK;
}

// A searchable Connect4 position (connect4.rs:3-13): disjoint bitboards inside the 63 cells, every column filled
// from the bottom without holes, at least one free column. (The reference would panic in best_action().unwrap() on a
// root without legal moves, mcts.rs:293.)
static bool valid_root(uint64_t my, uint64_t op) {
    if ((my & op) != 0 || ((my | op) >> 63) != 0) return false;
    uint64_t occ = my | op;
    bool any_free = false;
    for (int c = 0; c < 9; c++) {
        unsigned col = (unsigned)((occ >> (7 * c)) & 0x7F);
        if ((col & (col + 1)) != 0) return false;  // must be 0b0..01..1
        any_free |= col != 0x7F;
    }
    return any_free;
}

static int common_params(syn_engine* h, EngineParams& P, int explores, bool need_weights = true) {
    if (need_weights && !h->has_weights) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_load_weights first");
    if (need_weights && h->net_arith == SYN_NET_ARITH_F16X2) { int rc16 = ensure_f16x2_image(h); if (rc16 != SYN_OK) return rc16; }
    if (explores < 0) return fail(h, SYN_ERR_INVALID_ARGUMENT, "explores must be >= 0");
    if (explores > h->max_explores)
        return fail(h, SYN_ERR_CAPACITY, "explores %d exceeds the engine's max_explores %d", explores, h->max_explores);
    std::memset(&P, 0, sizeof(P));
    P.wimg = (h->net_arith == SYN_NET_ARITH_F16X2 && h->net_kind == 0) ? reinterpret_cast<const float*>(h->d_wimg16) : h->d_wimg;
    P.stat = h->d_stat;
    P.edge = h->d_edge;
    P.cap = h->cap;
    P.job_next = h->d_job_next;
    P.error = h->d_job_next + 8;  // same 64-byte block, zeroed before every launch
    P.path = nullptr;
    P.lane_thresh = 48;
    P.cache = h->d_cache;
    P.cache_shift = (uint32_t)(64 - h->cache_log2);
    P.cache_stats = h->d_cache_stats;
    P.counters = h->d_counters;
    return SYN_OK;
}

// Fpu::Func, PolicyNoise::Dirichlet and Connect4ConvNet exist in the lane-per-tree kernels only, whose block ids are 14 bits: an
// engine created for more than 7,280 explores cannot run them. Said before anything is enqueued.
static int check_lane_only(syn_engine* h, const DevMctsCfg& m) {
    if ((m.fpu == 2 || m.noise == 2 || h->net_kind == 1) && h->cap > LANE_MAX_CAP)
        return fail(h, SYN_ERR_UNSUPPORTED, "Fpu::Func / PolicyNoise::Dirichlet / Connect4ConvNet run in the lane-per-tree kernels only: "
                    "max_explores must be <= %u for them (this engine: %d)", (LANE_MAX_CAP - 1u) / 9u - 1u, h->max_explores);
    return SYN_OK;
}

static int mcts_search_impl(syn_engine* h, const syn_mcts_config* cfg, const uint64_t* my_bb, const uint64_t* op_bb, int n,
                            int explores, int action_selection, syn_search_result* results, bool rollout, uint64_t seed) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (n < 0 || (n > 0 && (!my_bb || !op_bb || !results)))
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_mcts_search");
    if (action_selection != SYN_ACTION_Q && action_selection != SYN_ACTION_NUM_VISITS)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown action selection %d", action_selection);
    EngineParams P;
    int rc = common_params(h, P, explores, /*need_weights=*/!rollout);
    if (rc != SYN_OK) return rc;
    rc = convert_mcts(h, cfg, P.mcts);
    if (rc != SYN_OK) return rc;
    if (!rollout && (rc = check_lane_only(h, P.mcts)) != SYN_OK) return rc;
    if (n == 0) return SYN_OK;
    for (int i = 0; i < n; i++)
        if (!valid_root(my_bb[i], op_bb[i]))
            return fail(h, SYN_ERR_INVALID_ARGUMENT, "root %d is not a searchable Connect4 position "
                        "(overlapping / floating stones, bit 63 set, or no free column)", i);
    HIP_TRY(h, hipSetDevice(h->device));
    size_t nb = (size_t)n;
    rc = ensure_scratch(h, nb * (16 + sizeof(DevSearchResult)) + 256);
    if (rc != SYN_OK) return rc;
    unsigned long long* d_my = static_cast<unsigned long long*>(h->d_scratch);
    unsigned long long* d_op = d_my + nb;
    DevSearchResult* d_res = reinterpret_cast<DevSearchResult*>(d_op + nb);
    CallScope scope(h, n);
    HIP_TRY(h, hipMemcpyAsync(d_my, my_bb, nb * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_op, op_bb, nb * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_job_next, 0, 64, h->stream));
    HIP_TRY(h, scope.armed());
    HIP_TRY(h, hipMemsetAsync(h->d_cache_stats, 0, 16, h->stream));
    // a root that syn_cancel kept from being searched reads as all zeros (num_nodes == 0: a searched root has at least its own node)
    HIP_TRY(h, hipMemsetAsync(d_res, 0, nb * sizeof(DevSearchResult), h->stream));
    P.roll.num_explores = explores;
    P.n_jobs = n;
    P.in_my = d_my;
    P.in_op = d_op;
    P.results = d_res;
    P.action_selection = action_selection;
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    P.base_seed = seed;
    P.first_game = 0;
    if (rollout) HIP_TRY(h, launch_rollout_search(h, P, n));
    else HIP_TRY(h, (launch_engine<MODE_SEARCH, false>(h, P, n)));
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipMemcpyAsync(results, d_res, nb * sizeof(DevSearchResult), hipMemcpyDeviceToHost, h->stream));
    int kerr = 0;
    unsigned long long cstats[2] = {0, 0};
    HIP_TRY(h, hipMemcpyAsync(&kerr, h->d_job_next + 8, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(cstats, h->d_cache_stats, 16, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->last_cache_hits = cstats[0];
    h->last_cache_misses = cstats[1];
    if (kerr == 2)
        return fail(h, SYN_ERR_CAPACITY, "a tree ran out of node blocks (lane-per-tree kernel: %u blocks per tree for max_explores %d)",
                    h->cap / 4, h->max_explores);
    if (kerr) return fail(h, SYN_ERR_HIP, "kernel reported a synchronisation timeout (bounded spin gave up)");
    HIP_TRY(h, hipEventElapsedTime(&h->last_kernel_ms, h->ev0, h->ev1));
    h->last_launches = 1;
    if (scope.cancelled()) {
        int missing = 0;
        for (int i = 0; i < n; i++) missing += results[i].num_nodes == 0u ? 1 : 0;
        if (missing)
            return fail(h, SYN_ERR_CANCELLED, "cancelled by syn_cancel: %d of %d roots were not searched (their results are all zero)",
                        missing, n);
    }
    return SYN_OK;
}

int syn_mcts_search(syn_engine* h, const syn_mcts_config* cfg, const uint64_t* my_bb, const uint64_t* op_bb, int n,
                    int explores, int action_selection, syn_search_result* results) {
    return mcts_search_impl(h, cfg, my_bb, op_bb, n, explores, action_selection, results, false, 0);
}

int syn_mcts_search_rollout(syn_engine* h, const syn_mcts_config* cfg, uint64_t seed, const uint64_t* my_bb,
                            const uint64_t* op_bb, int n, int explores, int action_selection, syn_search_result* results) {
    return mcts_search_impl(h, cfg, my_bb, op_bb, n, explores, action_selection, results, true, seed);
}

// evaluator.rs:308-319 FrozenMCTS::exploit over RolloutPolicy for n roots (frozen_kernel.cuh)
int syn_frozen_search_rollout(syn_engine* h, const syn_mcts_config* cfg, const uint64_t* seeds, uint64_t* rng_words,
                              const uint64_t* my_bb, const uint64_t* op_bb, const int32_t* explores, int n,
                              int action_selection, syn_frozen_result* results) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!cfg || n < 0 || (n > 0 && (!seeds || !rng_words || !my_bb || !op_bb || !explores || !results)))
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_frozen_search_rollout");
    if (action_selection != SYN_ACTION_Q && action_selection != SYN_ACTION_NUM_VISITS)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown action selection %d", action_selection);
    // the baseline panics on anything else (evaluator.rs:418-421, 429-436)
    if (cfg->exploration != SYN_EXPLORATION_UCT)
        return fail(h, SYN_ERR_UNSUPPORTED, "FrozenMCTS supports Exploration::Uct only (evaluator.rs:429-436)");
    if (cfg->fpu != SYN_FPU_CONST)
        return fail(h, SYN_ERR_UNSUPPORTED, "FrozenMCTS supports Fpu::Const only (evaluator.rs:418-421)");
    if (n == 0) return SYN_OK;
    // The baseline's trees live in the engine's node pool, re-partitioned for this call: every tree gets the record capacity
    // the largest search of the batch can need (1 + 9 nodes per visit), and the pool holds as many trees at once as fit.
    size_t max_need = 0;
    for (int i = 0; i < n; i++) {
        if (!valid_root(my_bb[i], op_bb[i]))
            return fail(h, SYN_ERR_INVALID_ARGUMENT, "root %d is not a searchable Connect4 position", i);
        if (explores[i] < 0) return fail(h, SYN_ERR_INVALID_ARGUMENT, "explores[%d] must be >= 0", i);
        const size_t worst = 1 + 9 * ((size_t)explores[i] + 1);
        if (worst > max_need) max_need = worst;
        if (rng_words[i] > 0xC0000000ull)
            return fail(h, SYN_ERR_INVALID_ARGUMENT, "rng_words[%d] is beyond the supported stream length", i);
    }
    const size_t nodes_per_tree = (max_need + 7) & ~(size_t)7;
    const size_t pool_records = (size_t)h->pool_slots * h->cap * 2;  // 16-byte records in the 32-byte-per-node pool
    if (max_need > 0x1FFFFFu || nodes_per_tree > pool_records)
        return fail(h, SYN_ERR_CAPACITY, "a baseline search of %zu explores needs up to %zu node records; the engine's pool holds %zu "
                    "and a tree at most 2,097,151", (max_need - 1) / 9 - 1, max_need, pool_records);
    size_t n_lanes = pool_records / nodes_per_tree;
    if (n_lanes > (size_t)n) n_lanes = (size_t)n;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nb = (size_t)n;
    int rc = ensure_scratch(h, nb * (8 * 4 + 4 + sizeof(FrozenResult)) + 256);
    if (rc != SYN_OK) return rc;
    unsigned long long* d_my = static_cast<unsigned long long*>(h->d_scratch);
    unsigned long long* d_op = d_my + nb;
    unsigned long long* d_seed = d_op + nb;
    unsigned long long* d_words = d_seed + nb;
    FrozenResult* d_res = reinterpret_cast<FrozenResult*>(d_words + nb);
    int* d_expl = reinterpret_cast<int*>(d_res + nb);
    const int grid = (int)((n_lanes + 63) / 64);  // one wave per workgroup
    const size_t need_path = (size_t)grid * 4096 * sizeof(uint32_t);
    if (need_path > h->path_bytes) {
        if (h->d_path) (void)hipFree(h->d_path);
        h->d_path = nullptr;
        h->path_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->d_path, need_path));
        h->path_bytes = need_path;
    }
    HIP_TRY(h, hipMemcpyAsync(d_my, my_bb, nb * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_op, op_bb, nb * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_seed, seeds, nb * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_words, rng_words, nb * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_expl, explores, nb * 4, hipMemcpyHostToDevice, h->stream));
    CallScope scope(h, n);
    HIP_TRY(h, hipMemsetAsync(h->d_job_next, 0, 64, h->stream));
    HIP_TRY(h, scope.armed());  // (this kernel takes its roots in a grid-stride loop, not from the job counter: a syn_cancel
                                //  during the call is accepted and has no effect — every root is searched)
    FrozenParams P;
    P.pool = reinterpret_cast<uint4*>(h->d_stat);
    P.nodes_per_tree = nodes_per_tree;
    P.path = reinterpret_cast<uint32_t*>(h->d_path);
    P.in_my = d_my; P.in_op = d_op; P.seeds = d_seed; P.rng_words = d_words; P.explores = d_expl;
    P.n_roots = n;
    P.n_lanes = (int)n_lanes;
    P.c = cfg->c;
    P.fpu_value = cfg->fpu_value;
    P.solve = cfg->solve ? 1 : 0;
    P.action_selection = action_selection;
    P.results = d_res;
    P.error = h->d_job_next + 8;
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(frozen_rollout_kernel, dim3(grid), dim3(64), 0, h->stream, P);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    h->last_shape = 5; h->last_grid = grid; h->last_threads = 64;
    static_assert(sizeof(FrozenResult) == sizeof(syn_frozen_result), "device and ABI result records must match");
    int kerr = 0;
    HIP_TRY(h, hipMemcpyAsync(results, d_res, nb * sizeof(FrozenResult), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(rng_words, d_words, nb * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&kerr, h->d_job_next + 8, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (kerr) return fail(h, SYN_ERR_CAPACITY, "a baseline tree ran out of node records");
    HIP_TRY(h, hipEventElapsedTime(&h->last_kernel_ms, h->ev0, h->ev1));
    h->last_launches = 1;
    return SYN_OK;
}

static int ensure_outputs(syn_engine* h, int n_games) {
    if (n_games <= h->out_games) return SYN_OK;
    hipFree(h->d_plies); hipFree(h->d_states); hipFree(h->d_pis); hipFree(h->d_vs);
    hipFree(h->d_actions); hipFree(h->d_root_nodes); hipFree(h->d_final);
    h->d_plies = nullptr; h->d_states = nullptr; h->d_pis = nullptr; h->d_vs = nullptr;
    h->d_actions = nullptr; h->d_root_nodes = nullptr; h->d_final = nullptr;
    h->out_games = 0;
    size_t g = (size_t)n_games, p = g * 63;
    HIP_TRY(h, hipMalloc(&h->d_plies, g * 4));
    HIP_TRY(h, hipMalloc(&h->d_states, p * 16));
    HIP_TRY(h, hipMalloc(&h->d_pis, p * 36));
    HIP_TRY(h, hipMalloc(&h->d_vs, p * 12));
    HIP_TRY(h, hipMalloc(&h->d_actions, p));
    HIP_TRY(h, hipMalloc(&h->d_root_nodes, p * 4));
    HIP_TRY(h, hipMalloc(&h->d_final, g));
    h->out_games = n_games;
    return SYN_OK;
}

int syn_selfplay_run(syn_engine* h, const syn_rollout_config* cfg, uint64_t base_seed, uint64_t first_game,
                     int n_games, int32_t* plies, uint64_t* states_bb, float* pis, float* vs, uint8_t* actions,
                     uint32_t* root_nodes, uint8_t* final_kind, syn_counters* counters) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!cfg) return fail(h, SYN_ERR_INVALID_ARGUMENT, "rollout config is NULL");
    if (n_games < 0) return fail(h, SYN_ERR_INVALID_ARGUMENT, "n_games must be >= 0");
    if (cfg->value_target < SYN_VALUE_Z || cfg->value_target > SYN_VALUE_Q_TO_Z)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown value target %d", cfg->value_target);
    if (cfg->action != SYN_ACTION_Q && cfg->action != SYN_ACTION_NUM_VISITS)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown action selection %d", cfg->action);
    if (cfg->random_actions_until < 0 || cfg->sample_actions_until < 0)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "random_actions_until / sample_actions_until must be >= 0");
    EngineParams P;
    int rc = common_params(h, P, cfg->num_explores);
    if (rc != SYN_OK) return rc;
    rc = convert_mcts(h, &cfg->mcts_cfg, P.mcts);
    if (rc != SYN_OK) return rc;
    if ((rc = check_lane_only(h, P.mcts)) != SYN_OK) return rc;
    if (counters) std::memset(counters, 0, sizeof(*counters));
    if (n_games == 0) return SYN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    rc = ensure_outputs(h, n_games);
    if (rc != SYN_OK) return rc;
    P.roll.num_explores = cfg->num_explores;
    P.roll.random_until = cfg->random_actions_until;
    P.roll.sample_until = cfg->sample_actions_until;
    P.roll.stop_when_solved = cfg->stop_games_when_solved != 0;
    P.roll.value_target = cfg->value_target;
    P.roll.vt_p = cfg->value_target_p;
    P.roll.vt_from = cfg->value_target_from;
    P.roll.vt_to = cfg->value_target_to;
    P.roll.action = cfg->action;
    P.n_jobs = n_games;
    P.base_seed = base_seed;
    P.first_game = first_game;
    P.plies = h->d_plies;
    P.states_bb = h->d_states;
    P.pis = h->d_pis;
    P.vs = h->d_vs;
    P.actions = h->d_actions;
    P.root_nodes = h->d_root_nodes;
    P.final_kind = h->d_final;
    CallScope scope(h, n_games);
    HIP_TRY(h, hipMemsetAsync(h->d_job_next, 0, 64, h->stream));
    HIP_TRY(h, scope.armed());
    HIP_TRY(h, hipMemsetAsync(h->d_cache_stats, 0, 16, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_counters, 0, sizeof(DevCounters), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_plies, 0, (size_t)n_games * 4, h->stream));  // plies = 0: a game that never started (syn_cancel)
    // SYN_PROFILE=1: diagnostic build of the kernel with s_memtime stamps around each phase (never timed/benched)
    const bool prof = !counters && debug_env("SYN_PROFILE") != nullptr;
    int pgrid = 0, pnt = 0;
    unsigned long long* d_prof = nullptr;
    if (prof) {
        HIP_TRY(h, hipMalloc(&d_prof, (size_t)4096 * 16 * 6 * 8));
        HIP_TRY(h, hipMemsetAsync(d_prof, 0, (size_t)4096 * 16 * 6 * 8, h->stream));
        P.prof = d_prof;
    }
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    if (prof) HIP_TRY(h, (launch_engine<MODE_SELFPLAY, false, true>(h, P, n_games, &pgrid, &pnt)));
    else if (counters) HIP_TRY(h, (launch_engine<MODE_SELFPLAY, true>(h, P, n_games)));
    else HIP_TRY(h, (launch_engine<MODE_SELFPLAY, false>(h, P, n_games)));
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    if (prof) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->last_shape == 7) {  // free-running rows: per wave [A, B, C, iterations, tiles, leaves] (free_kernel.cuh)
            const int nwv = pgrid * 4;
            std::vector<unsigned long long> hp((size_t)nwv * FP_FIELDS);
            HIP_TRY(h, hipMemcpy(hp.data(), d_prof, hp.size() * 8, hipMemcpyDeviceToHost));
            HIP_TRY(h, hipFree(d_prof));
            d_prof = nullptr;
            double s7[FP_FIELDS] = {0};
            for (int w = 0; w < nwv; w++)
                for (int j = 0; j < FP_FIELDS; j++) s7[j] += (double)hp[(size_t)w * FP_FIELDS + j];
            const double it7 = s7[FP_ITERS] + 1e-9;
            fprintf(stderr, "[syn profile free] grid=%d waves=%d rounds/wave=%.0f | cycles per round (one explore on each of a wave's four trees): "
                            "A=%.0f B=%.0f C=%.0f total=%.0f | tiles per round=%.3f, leaves per tile=%.2f, cycles per tile=%.0f\n",
                    pgrid, nwv, it7 / nwv, s7[FP_A] / it7, s7[FP_B] / it7, s7[FP_C] / it7, (s7[FP_A] + s7[FP_B] + s7[FP_C]) / it7,
                    s7[FP_TILES] / it7, s7[FP_LEAVES] / (s7[FP_TILES] + 1e-9), s7[FP_B] / (s7[FP_TILES] + 1e-9));
        } else
        if (h->last_shape == 5) {  // producer/consumer kernel: per wave [role, ...] (pc_kernel.cuh)
            const int nwv = pgrid * 16;
            std::vector<unsigned long long> hp((size_t)nwv * 8);
            HIP_TRY(h, hipMemcpy(hp.data(), d_prof, hp.size() * 8, hipMemcpyDeviceToHost));
            HIP_TRY(h, hipFree(d_prof));
            d_prof = nullptr;
            double mw = 0, mb = 0, mt = 0, nm = 0, tw = 0, ta = 0, tc = 0, tr = 0, tf = 0, ntw = 0;
            for (int w = 0; w < nwv; w++) {
                const unsigned long long* o = &hp[(size_t)w * 8];
                if (o[0] == 1) { nm++; mw += (double)o[1]; mb += (double)o[2]; mt += (double)o[3]; }
                else if (o[0] == 2) { ntw++; tw += (double)o[1]; ta += (double)o[2]; tc += (double)o[3]; tr += (double)o[4]; tf += (double)o[5]; }
            }
            fprintf(stderr, "[syn profile pc] grid=%d | matrix waves=%.0f: busy %.1f%% of (busy+wait), %.0f cycles per tile, %.0f tiles per wave | "
                            "tree waves=%.0f: per visit: wait=%.0f C=%.0f A+submit=%.0f cycles, explores finished=%.2f, visits per wave=%.0f\n",
                    pgrid, nm, 100.0 * mb / (mb + mw + 1e-9), mb / (mt + 1e-9), mt / (nm + 1e-9), ntw, tw / (tr + 1e-9), tc / (tr + 1e-9),
                    ta / (tr + 1e-9), tf / (tr + 1e-9), tr / (ntw + 1e-9));
        } else if (pgrid < 0) {  // lane kernel: per wave [A, B, C, move, rounds, tiles, active lanes, evals, then the LP_* fields]
            int nwv = -pgrid * (pnt / 64);
            constexpr int F = LP_FIELDS;
            std::vector<unsigned long long> hp((size_t)nwv * F);
            HIP_TRY(h, hipMemcpy(hp.data(), d_prof, hp.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long* d_prof_keep = d_prof;
            double s[F];
            for (int j = 0; j < F; j++) s[j] = 0;
            for (int w = 0; w < nwv; w++)
                for (int j = 0; j < F; j++) s[j] += (double)hp[(size_t)w * F + j];
            fprintf(stderr, "[syn profile lanes] grid=%d nt=%d waves=%d rounds/wave=%.0f | cycles per round: A=%.0f B=%.0f C: children=%.0f "
                            "solver walk=%.0f sweep=%.0f move=%.0f total=%.0f | per round: tiles=%.3f explores finished=%.2f evals=%.2f "
                            "(%.2f per tile) | cycles per finished explore=%.0f\n",
                    -pgrid, pnt, nwv, s[4] / nwv, s[0] / s[4], s[1] / s[4], s[6] / s[4], s[7] / s[4], s[2] / s[4], s[3] / s[4],
                    (s[0] + s[1] + s[2] + s[3] + s[6] + s[7]) / s[4], s[5] / s[4], s[8] / s[4], s[9] / s[4], s[9] / s[5],
                    (s[0] + s[1] + s[2] + s[3] + s[6] + s[7]) / s[8]);
            const double R = s[4];  // rounds (all waves)
            fprintf(stderr, "[syn profile lanes, inside the phases; cycles per round and wave] A: %.2f iterations (%.1f lanes each), line wait %.0f "
                            "(%.0f per iteration), level arithmetic %.0f (%.0f per iteration), arrive %.0f | B: tiles %.0f (%.0f per tile), "
                            "scatter %.0f | C children: softmaxes %.0f, records %.0f (%.1f lanes) | solver walk: %.2f iterations (%.1f lanes each), "
                            "line wait %.0f (%.0f per iteration) | sweep: %.2f steps, %.1f lane-levels per step, log wait %.0f (%.0f per step), "
                            "arithmetic + stores %.0f (%.0f per step) | end of search: %.3f calls per round, %.1f lanes per call, cycles per call %.0f = "
                            "root line + generator %.0f, targets %.0f, sample %.0f, step %.0f, game end / reset %.0f\n",
                    s[LP_A_ITERS] / R, s[LP_A_LANES] / (s[LP_A_ITERS] + 1e-9), s[LP_A_WAIT] / R, s[LP_A_WAIT] / (s[LP_A_ITERS] + 1e-9),
                    s[LP_A_ALU] / R, s[LP_A_ALU] / (s[LP_A_ITERS] + 1e-9), s[LP_A_ARRIVE] / R, s[LP_B_TILE] / R,
                    s[LP_B_TILE] / (s[5] + 1e-9), s[LP_B_SCATTER] / R, s[LP_C_SOFT] / R, (s[LP_C_WRITE] - s[LP_C_SOFT]) / R,
                    s[LP_C_LANES] / R, s[LP_W_ITERS] / R, s[LP_W_LANES] / (s[LP_W_ITERS] + 1e-9), s[LP_W_WAIT] / R,
                    s[LP_W_WAIT] / (s[LP_W_ITERS] + 1e-9), s[LP_S_STEPS] / R, s[LP_S_LANES] / (s[LP_S_STEPS] + 1e-9), s[LP_S_WAIT] / R,
                    s[LP_S_WAIT] / (s[LP_S_STEPS] + 1e-9), s[LP_S_ALU] / R, s[LP_S_ALU] / (s[LP_S_STEPS] + 1e-9), s[LP_M_CALLS] / R,
                    s[LP_M_LANES] / (s[LP_M_CALLS] + 1e-9), s[LP_M_TOTAL] / (s[LP_M_CALLS] + 1e-9), s[LP_M_T1] / (s[LP_M_CALLS] + 1e-9),
                    s[LP_M_T2] / (s[LP_M_CALLS] + 1e-9), s[LP_M_T3] / (s[LP_M_CALLS] + 1e-9), s[LP_M_T4] / (s[LP_M_CALLS] + 1e-9),
                    s[LP_M_T5] / (s[LP_M_CALLS] + 1e-9));
            {
                std::vector<unsigned long long> tlv(4 * 16 * 3);
                (void)hipMemcpy(tlv.data(), d_prof_keep + PROF_TIMELINE_OFF, tlv.size() * 8, hipMemcpyDeviceToHost);
                unsigned long long t0 = ~0ull;
                for (auto v : tlv) if (v && v < t0) t0 = v;
                for (int w = 0; w < pnt / 256; w++) {
                    fprintf(stderr, "[timeline wave %d of SIMD 0] (B start, B end, round end) kilo-cycles:", w * 4);
                    for (int r = 0; r < 8; r++)
                        fprintf(stderr, " (%.0f %.0f %.0f)", (tlv[(w * 16 + r) * 3] - t0) / 1e3, (tlv[(w * 16 + r) * 3 + 1] - t0) / 1e3,
                                (tlv[(w * 16 + r) * 3 + 2] - t0) / 1e3);
                    fprintf(stderr, "\n");
                }
                (void)hipFree(d_prof_keep);
            }
            d_prof = nullptr;
        }
    }
    if (prof && d_prof) {
        int nw = pgrid * (pnt / 64);
        std::vector<unsigned long long> hp((size_t)nw * 6);
        HIP_TRY(h, hipMemcpy(hp.data(), d_prof, hp.size() * 8, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipFree(d_prof));
        double sum[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long max_it = 0;
        for (int w = 0; w < nw; w++) {
            for (int j = 0; j < 6; j++) sum[j] += (double)hp[(size_t)w * 6 + j];
            if (hp[(size_t)w * 6 + 5] > max_it) max_it = hp[(size_t)w * 6 + 5];
        }
        double its = sum[5] / nw;
        fprintf(stderr, "[syn profile] grid=%d nt=%d waves=%d iterations avg=%.0f max=%llu | cycles/iteration per wave (100 MHz s_memtime ticks x?): "
                        "A=%.0f wait1=%.0f B=%.0f wait2=%.0f C=%.0f total=%.0f\n",
                pgrid, pnt, nw, its, max_it, sum[0] / sum[5], sum[1] / sum[5], sum[2] / sum[5], sum[3] / sum[5],
                sum[4] / sum[5], (sum[0] + sum[1] + sum[2] + sum[3] + sum[4]) / sum[5]);
    }
    size_t g = (size_t)n_games, p = g * 63;
    if (plies) HIP_TRY(h, hipMemcpyAsync(plies, h->d_plies, g * 4, hipMemcpyDeviceToHost, h->stream));
    if (states_bb) HIP_TRY(h, hipMemcpyAsync(states_bb, h->d_states, p * 16, hipMemcpyDeviceToHost, h->stream));
    if (pis) HIP_TRY(h, hipMemcpyAsync(pis, h->d_pis, p * 36, hipMemcpyDeviceToHost, h->stream));
    if (vs) HIP_TRY(h, hipMemcpyAsync(vs, h->d_vs, p * 12, hipMemcpyDeviceToHost, h->stream));
    if (actions) HIP_TRY(h, hipMemcpyAsync(actions, h->d_actions, p, hipMemcpyDeviceToHost, h->stream));
    if (root_nodes) HIP_TRY(h, hipMemcpyAsync(root_nodes, h->d_root_nodes, p * 4, hipMemcpyDeviceToHost, h->stream));
    if (final_kind) HIP_TRY(h, hipMemcpyAsync(final_kind, h->d_final, g, hipMemcpyDeviceToHost, h->stream));
    if (counters)
        HIP_TRY(h, hipMemcpyAsync(counters, h->d_counters, sizeof(DevCounters), hipMemcpyDeviceToHost, h->stream));
    int kerr = 0, games_finished = 0;
    unsigned long long cstats[2] = {0, 0};
    HIP_TRY(h, hipMemcpyAsync(&kerr, h->d_job_next + 8, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&games_finished, h->d_job_next + 1, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(cstats, h->d_cache_stats, 16, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->last_cache_hits = cstats[0];
    h->last_cache_misses = cstats[1];
    if (kerr == 2)
        return fail(h, SYN_ERR_CAPACITY, "a tree ran out of node blocks (lane-per-tree kernel: %u blocks per tree for max_explores %d)",
                    h->cap / 4, h->max_explores);
    if (kerr) return fail(h, SYN_ERR_HIP, "kernel reported a synchronisation timeout (bounded spin gave up)");
    HIP_TRY(h, hipEventElapsedTime(&h->last_kernel_ms, h->ev0, h->ev1));
    h->last_launches = 1;
    // decided by what finished, not by the flag: a cancel that arrives after the last game was handed out cancels nothing
    if (scope.cancelled() && games_finished < n_games)
        return fail(h, SYN_ERR_CANCELLED, "cancelled by syn_cancel: %d of %d games were played (to the end); the others have plies == 0",
                    games_finished, n_games);
    return SYN_OK;
}

int syn_progress(syn_engine* h, int* started, int* finished) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> g(h->call_mu);
    if (hipSetDevice(h->device) != hipSuccess) return SYN_ERR_HIP;
    if (hipMemcpyAsync(h->h_pin, h->d_job_next, 8, hipMemcpyDeviceToHost, h->aux_stream) != hipSuccess) return SYN_ERR_HIP;
    if (hipStreamSynchronize(h->aux_stream) != hipSuccess) return SYN_ERR_HIP;
    const int jobs = h->running_jobs;
    int st = h->h_pin[0];
    if (st >= CANCEL_WORD) st = h->started_at_cancel;  // the counter was raised by syn_cancel: what had been handed out before
    if (jobs > 0 && st > jobs) st = jobs;              // (every lane's last fetch overshoots the job count)
    if (started) *started = st;
    if (finished) *finished = h->h_pin[1];
    return SYN_OK;
}

int syn_cancel(syn_engine* h) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> g(h->call_mu);
    if (h->call_state == 0) return SYN_ERR_INVALID_ARGUMENT;  // no call in flight: nothing to cancel (the error string belongs to the calling thread's entry points and is not touched here)
    if (h->call_state == 1) {  // the call has not reset the job counter yet: it applies the cancel itself, right behind the reset
        h->cancel_pending = 1;
        return SYN_OK;
    }
    if (h->cancel_applied) return SYN_OK;
    if (hipSetDevice(h->device) != hipSuccess) return SYN_ERR_HIP;
    // jobs handed out so far (a lower bound of what will have started: a fetch between this read and the write below still counts)
    if (hipMemcpyAsync(h->h_pin + 9, h->d_job_next, 4, hipMemcpyDeviceToHost, h->aux_stream) != hipSuccess) return SYN_ERR_HIP;
    // every later job fetch (an atomic add on this word) now returns an index past the call's job count: no new game starts
    h->h_pin[8] = CANCEL_WORD;
    if (hipMemcpyAsync(h->d_job_next, h->h_pin + 8, 4, hipMemcpyHostToDevice, h->aux_stream) != hipSuccess) return SYN_ERR_HIP;
    if (hipStreamSynchronize(h->aux_stream) != hipSuccess) return SYN_ERR_HIP;
    h->started_at_cancel = h->h_pin[9] < h->running_jobs ? h->h_pin[9] : h->running_jobs;
    h->cancel_applied = 1;
    return SYN_OK;
}

// The learner's device buffers (shared by both networks' trainers; sized for Connect4Net, the larger one): all of them or none —
// a failed allocation frees what was taken, so that a retry starts from scratch instead of skipping the allocation block.
static int alloc_trainer_buffers(syn_engine* h) {
    if (h->d_tw) return SYN_OK;
    const size_t bytes = (size_t)TrainGeom::NUM_PARAMS * 4;
    const size_t img = (size_t)MlpGeom::IMG_FLOATS * 4, timg = (size_t)TrainImg::T_FLOATS * 4;
    struct { void** p; size_t n; } want[] = {
        {reinterpret_cast<void**>(&h->d_tw), bytes},      {reinterpret_cast<void**>(&h->d_tm), bytes},
        {reinterpret_cast<void**>(&h->d_tv), bytes},      {reinterpret_cast<void**>(&h->d_tgrad), bytes},
        {reinterpret_cast<void**>(&h->d_tloss), 64},      {reinterpret_cast<void**>(&h->d_twimg), img},
        {reinterpret_cast<void**>(&h->d_ttimg), timg},    {reinterpret_cast<void**>(&h->d_timg2), img + timg},
        {reinterpret_cast<void**>(&h->d_tsync), 256},     {reinterpret_cast<void**>(&h->d_tsnap), 3 * bytes + img + timg},
        {reinterpret_cast<void**>(&h->d_cxbuf), (size_t)ConvMwGeom::FLOATS * 4},
    };
    for (auto& w : want) {
        hipError_t e = hipMalloc(w.p, w.n);
        if (e != hipSuccess) {
            for (auto& u : want) {
                if (*u.p) (void)hipFree(*u.p);
                *u.p = nullptr;
            }
            h->has_trainer = false;
            return fail(h, SYN_ERR_HIP, "hipMalloc(trainer buffers) failed: %s", hipGetErrorString(e));
        }
    }
    return SYN_OK;
}

// ------------------------------------------------------------------------------------------------ learner step
// (re)load the Connect4Net learner's state: parameters, zero moments, both fragment images, step counter
static int trainer_load_state(syn_engine* h, const float* blob, const std::vector<float>& img, const std::vector<float>& timg) {
    const size_t bytes = (size_t)TrainGeom::NUM_PARAMS * 4;
    HIP_TRY(h, hipMemcpyAsync(h->d_twimg, img.data(), img.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_ttimg, timg.data(), timg.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_tw, blob, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_tm, 0, bytes, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_tv, 0, bytes, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->train_step = 0;
    return SYN_OK;
}

int syn_trainer_init(syn_engine* h, const float* blob, size_t n_floats, const syn_train_config* cfg) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!blob || !cfg) return fail(h, SYN_ERR_INVALID_ARGUMENT, "blob/cfg is NULL");
    if (n_floats != (size_t)TrainGeom::NUM_PARAMS)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "Connect4Net has %d parameters, got %zu", TrainGeom::NUM_PARAMS, n_floats);
    HIP_TRY(h, hipSetDevice(h->device));
    {
        const int rc = alloc_trainer_buffers(h);
        if (rc != SYN_OK) return rc;
    }
    // the two fragment-order images the matrix-core learner reads its A operands from (train_mfma.cuh); adam_image_kernel
    // keeps them in step with the canonical weights afterwards
    std::vector<float> img, timg((size_t)TrainImg::T_FLOATS, 0.0f);
    build_weight_image(blob, img);
    for (int p = 0; p < TrainGeom::NUM_PARAMS; p++) {
        int fwd, tr;
        train_image_slots(p, fwd, tr);
        if (img[(size_t)fwd] != blob[p]) return fail(h, SYN_ERR_HIP, "internal: image slot table disagrees with build_weight_image at %d", p);
        if (tr >= 0) timg[(size_t)tr] = blob[p];
    }
    int rc = trainer_load_state(h, blob, img, timg);
    if (rc != SYN_OK) return rc;
    h->train_hp = DevTrainHyper{cfg->weight_decay, cfg->policy_weight, cfg->value_weight, cfg->beta1, cfg->beta2, cfg->eps};
    h->has_trainer = true;
    h->trainer_kind = 0;
    h->train_bf16 = 0;
    // ---- the epoch kernel's one-XCD step barrier rests on observed hardware behaviour (train_epoch.cuh: `buffer_inv sc0` empties
    //      the vector L1 outside threadgroup-split mode). Once per engine: eight steps on a synthetic batch through that barrier and
    //      through the device-scope barrier from the same state; any differing bit switches this engine to the device-scope barrier.
    if (!h->epoch_barrier_checked) {
        h->epoch_barrier_checked = true;
        const int n = 64, steps = 8, B = 8;
        std::vector<uint64_t> my(n), op(n);
        std::vector<float> tpi((size_t)n * 9, 1.0f / 9.0f), tv((size_t)n * 3, 0.0f);
        std::vector<int32_t> perm(steps * B);
        for (int i = 0; i < n; i++) {
            my[i] = (0x0000040810204081ull * (uint64_t)(i % 7 + 1)) & 0x00003F7EFDFBF7EFull & ~(0x7Full << (7 * (i % 9)));
            op[i] = (0x7Full << (7 * (i % 9))) & (0x0101010101010101ull * (uint64_t)(i % 5 + 1));
            op[i] &= ~my[i];
            tv[(size_t)i * 3 + i % 3] = 1.0f;
        }
        for (int i = 0; i < steps * B; i++) perm[i] = (i * 37) % n;
        std::vector<float> wa((size_t)TrainGeom::NUM_PARAMS), wb((size_t)TrainGeom::NUM_PARAMS);
        bool ok = true;
        for (int mode = 0; mode < 2 && ok; mode++) {
            h->epoch_device_scope = mode == 1;
            ok = syn_train_set_data(h, my.data(), op.data(), tpi.data(), tv.data(), n) == SYN_OK &&
                 syn_train_epoch(h, perm.data(), steps, B, 1e-3f, nullptr) == SYN_OK &&
                 syn_trainer_get_state(h, mode == 0 ? wa.data() : wb.data(), nullptr, nullptr, nullptr, nullptr) == SYN_OK &&
                 trainer_load_state(h, blob, img, timg) == SYN_OK;
        }
        h->epoch_device_scope = !(ok && std::memcmp(wa.data(), wb.data(), wa.size() * 4) == 0);
        h->train_data_n = 0;   // the synthetic batch replaced whatever syn_train_set_data had uploaded: the caller uploads again (once per engine)
        if (!ok) return fail(h, SYN_ERR_HIP, "the learner's start-up self-check could not run: %s", h->err.c_str());
    }
    return SYN_OK;
}

int syn_trainer_init_conv(syn_engine* h, const float* blob, size_t n_floats, const syn_train_config* cfg) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!blob || !cfg) return fail(h, SYN_ERR_INVALID_ARGUMENT, "blob/cfg is NULL");
    if (n_floats != (size_t)ConvGeom::NUM_PARAMS)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "Connect4ConvNet has %d parameters, got %zu", ConvGeom::NUM_PARAMS, n_floats);
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t cap_bytes = (size_t)TrainGeom::NUM_PARAMS * 4;  // the buffers are shared with the Connect4Net trainer (larger)
    static_assert(ConvGeom::NUM_PARAMS <= TrainGeom::NUM_PARAMS, "trainer buffers are sized for Connect4Net");
    {
        const int rc = alloc_trainer_buffers(h);
        if (rc != SYN_OK) return rc;
    }
    const size_t bytes = (size_t)ConvGeom::NUM_PARAMS * 4;
    HIP_TRY(h, hipMemcpyAsync(h->d_tw, blob, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_tm, 0, cap_bytes, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_tv, 0, cap_bytes, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_tgrad, 0, cap_bytes, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->train_hp = DevTrainHyper{cfg->weight_decay, cfg->policy_weight, cfg->value_weight, cfg->beta1, cfg->beta2, cfg->eps};
    h->train_step = 0;
    h->has_trainer = true;
    h->trainer_kind = 1;
    h->train_bf16 = 0;
    // ---- the four-workgroup epoch kernel exchanges its intermediates through L2 behind the one-XCD barrier of train_epoch.cuh (observed
    //      hardware behaviour, see there). Once per engine: eight steps on a synthetic batch through it and through the one-workgroup
    //      kernel from the same state; any differing bit (or a launch that cannot run) keeps this engine on the one-workgroup kernel.
    if (!h->conv_mw_checked) {
        h->conv_mw_checked = true;
        const int n = 64, steps = 8, B = 8;
        std::vector<uint64_t> my(n), op(n);
        std::vector<float> tpi((size_t)n * 9, 1.0f / 9.0f), tv((size_t)n * 3, 0.0f);
        std::vector<int32_t> perm(steps * B);
        for (int i = 0; i < n; i++) {
            my[i] = (0x0000040810204081ull * (uint64_t)(i % 7 + 1)) & 0x00003F7EFDFBF7EFull & ~(0x7Full << (7 * (i % 9)));
            op[i] = (0x7Full << (7 * (i % 9))) & (0x0101010101010101ull * (uint64_t)(i % 5 + 1));
            op[i] &= ~my[i];
            tv[(size_t)i * 3 + i % 3] = 1.0f;
        }
        for (int i = 0; i < steps * B; i++) perm[i] = (i * 37) % n;
        std::vector<float> wa((size_t)ConvGeom::NUM_PARAMS), wb((size_t)ConvGeom::NUM_PARAMS);
        const long long fallbacks0 = h->epoch_fallbacks;
        bool ok = true;
        for (int mode = 0; mode < 2 && ok; mode++) {
            h->conv_mw_force = mode;   // 0: one workgroup, 1: four
            ok = syn_train_set_data(h, my.data(), op.data(), tpi.data(), tv.data(), n) == SYN_OK &&
                 syn_train_epoch(h, perm.data(), steps, B, 1e-3f, nullptr) == SYN_OK &&
                 syn_trainer_get_state(h, mode == 0 ? wa.data() : wb.data(), nullptr, nullptr, nullptr, nullptr) == SYN_OK;
            // back to the caller's state
            ok = ok && hipMemcpyAsync(h->d_tw, blob, bytes, hipMemcpyHostToDevice, h->stream) == hipSuccess &&
                 hipMemsetAsync(h->d_tm, 0, cap_bytes, h->stream) == hipSuccess && hipMemsetAsync(h->d_tv, 0, cap_bytes, h->stream) == hipSuccess &&
                 hipMemsetAsync(h->d_tgrad, 0, cap_bytes, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess;
            h->train_step = 0;
        }
        h->conv_mw_force = -1;
        h->conv_mw_disabled = !(ok && h->epoch_fallbacks == fallbacks0 && std::memcmp(wa.data(), wb.data(), wa.size() * 4) == 0);
        h->epoch_fallbacks = fallbacks0;
        h->train_data_n = 0;   // the synthetic batch replaced whatever syn_train_set_data had uploaded: the caller uploads again (once per engine)
        if (!ok) return fail(h, SYN_ERR_HIP, "the conv learner's start-up self-check could not run: %s", h->err.c_str());
    }
    return SYN_OK;
}

int syn_trainer_set_precision(syn_engine* h, int precision) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!h->has_trainer) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_trainer_init_conv first");
    if (precision != SYN_TRAIN_F32 && precision != SYN_TRAIN_BF16) return fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown training precision %d", precision);
    if (precision == SYN_TRAIN_BF16 && h->trainer_kind != 1)
        return fail(h, SYN_ERR_UNSUPPORTED, "the bf16 training variant exists for Connect4ConvNet only (BASELINE configs[4]: \"bf16 conv\"); "
                                            "Connect4Net trains in f32, bit-exact with the oracle");
    h->train_bf16 = precision == SYN_TRAIN_BF16 ? 1 : 0;
    return SYN_OK;
}

static int launch_grads(syn_engine* h, const unsigned long long* d_my, const unsigned long long* d_op,
                        const float* d_tpi, const float* d_tv, int batch, float* d_grads, float* d_losses = nullptr,
                        const int* d_idx = nullptr, bool on_callers_stream = false, hipStream_t callers = nullptr) {
    // (the *_enqueue entry points run the step on the caller's stream — which may be the null stream: torch's default)
    const hipStream_t st = on_callers_stream ? callers : h->stream;
    if (h->trainer_kind == 1) {
        // Connect4ConvNet (train_conv_mfma.cuh): one workgroup, the minibatch's activations resident in LDS, every chain on the
        // f32 matrix cores
        if (batch > ConvTrainGeom::CHUNK)
            return fail(h, SYN_ERR_UNSUPPORTED, "the Connect4ConvNet learner takes minibatches of at most %d positions (got %d)",
                        ConvTrainGeom::CHUNK, batch);
        const size_t clds = (size_t)ConvMfmaGeom::LDS_FLOATS * 4;
        auto kg = h->train_bf16 ? train_conv_grad_kernel_mfma<true> : train_conv_grad_kernel_mfma<false>;
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));
        hipLaunchKernelGGL(kg, dim3(1), dim3(CONV_TRAIN_THREADS), clds, st, h->d_tw, d_my, d_op, d_tpi, d_tv, batch,
                           h->train_hp, d_grads, d_losses ? d_losses : h->d_tloss, d_idx);
        HIP_TRY(h, hipGetLastError());
        return SYN_OK;
    }
    // the matrix-core kernel (train_mfma.cuh); SYN_DEBUG=1 SYN_TRAIN_VALU=1 runs the VALU kernel it replaced (A/B, same bits)
    static const bool valu = debug_env("SYN_TRAIN_VALU") != nullptr;
    const size_t lds = valu ? (size_t)TrainGeom::LDS_FLOATS * 4 : (size_t)TrainGeom::WL_OFF * 4;
    if (valu) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(train_grad_kernel_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // SYN_TRAIN_PROFILE=1: diagnostic stamps of the kernel's phases (first chunk), printed to stderr
    static const bool prof = debug_env("SYN_TRAIN_PROFILE") != nullptr;
    unsigned long long* d_prof = nullptr;
    if (prof) {
        HIP_TRY(h, hipMalloc(&d_prof, 4096));
        HIP_TRY(h, hipMemsetAsync(d_prof, 0, 4096, st));
    }
    if (valu)
        hipLaunchKernelGGL(train_grad_kernel, dim3(1), dim3(1024), lds, st, h->d_tw, d_my, d_op, d_tpi, d_tv, batch, h->train_hp,
                           d_grads, d_losses ? d_losses : h->d_tloss, d_idx, d_prof);
    else
        hipLaunchKernelGGL(train_grad_kernel_mfma, dim3(1), dim3(1024), lds, st, h->d_tw, h->d_twimg, h->d_ttimg, d_my, d_op,
                           d_tpi, d_tv, batch, h->train_hp, d_grads, d_losses ? d_losses : h->d_tloss, d_idx, d_prof);
    HIP_TRY(h, hipGetLastError());
    if (prof) {
        unsigned long long t[8] = {0};
        HIP_TRY(h, hipStreamSynchronize(st));
        HIP_TRY(h, hipMemcpy(t, d_prof, sizeof(t), hipMemcpyDeviceToHost));
        HIP_TRY(h, hipFree(d_prof));
        fprintf(stderr, "[syn train profile] cycles: features %llu forward %llu heads %llu backward %llu param-grads %llu\n",
                t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]);
    }
    return SYN_OK;
}

static int launch_adam(syn_engine* h, const float* d_grads, float lr, float grad_scale, bool on_callers_stream = false,
                       hipStream_t callers = nullptr) {
    const hipStream_t st = on_callers_stream ? callers : h->stream;
    h->train_step += 1;
    const double bc1 = 1.0 - std::pow((double)h->train_hp.beta1, (double)h->train_step);
    const double bc2 = 1.0 - std::pow((double)h->train_hp.beta2, (double)h->train_step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
    if (h->trainer_kind == 1) {
        const int nc = ConvGeom::NUM_PARAMS;
        hipLaunchKernelGGL(adam_kernel, dim3((nc + 255) / 256), dim3(256), 0, st, h->d_tw, h->d_tm, h->d_tv, d_grads, nc,
                           h->train_hp, step_size, inv_sqrt_bc2, grad_scale);
        HIP_TRY(h, hipGetLastError());
        return SYN_OK;
    }
    const int n = TrainGeom::NUM_PARAMS;
    hipLaunchKernelGGL(adam_image_kernel, dim3((n + 255) / 256), dim3(256), 0, st, h->d_tw, h->d_tm, h->d_tv, d_grads,
                       n, h->train_hp, step_size, inv_sqrt_bc2, grad_scale, h->d_twimg, h->d_ttimg);
    HIP_TRY(h, hipGetLastError());
    return SYN_OK;
}

int syn_train_gradients_device(syn_engine* h, const uint64_t* d_my_bb, const uint64_t* d_op_bb, const float* d_target_pi,
                               const float* d_target_v, int batch, float* d_grads, float* losses) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!h->has_trainer) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_trainer_init first");
    if (batch < 1 || !d_my_bb || !d_op_bb || !d_target_pi || !d_target_v || !d_grads)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_train_gradients_device");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = launch_grads(h, reinterpret_cast<const unsigned long long*>(d_my_bb),
                          reinterpret_cast<const unsigned long long*>(d_op_bb), d_target_pi, d_target_v, batch, d_grads);
    if (rc != SYN_OK) return rc;
    if (losses) HIP_TRY(h, hipMemcpyAsync(losses, h->d_tloss, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SYN_OK;
}

int syn_train_apply_device(syn_engine* h, const float* d_grads, float lr, float grad_scale) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!h->has_trainer) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_trainer_init first");
    if (!d_grads) return fail(h, SYN_ERR_INVALID_ARGUMENT, "d_grads is NULL");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = launch_adam(h, d_grads, lr, grad_scale);
    if (rc != SYN_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SYN_OK;
}

// The same two halves of a data-parallel step WITHOUT the host in between: both are enqueued on the caller's stream (the stream the
// batch was prepared on and the all-reduce runs on), nothing is synchronised, the two loss sums go to device memory — a step is
// gradients -> all-reduce -> Adam in one stream order. The caller keeps other trainer calls of this engine off other streams meanwhile.
int syn_train_gradients_enqueue(syn_engine* h, void* stream, const uint64_t* d_my_bb, const uint64_t* d_op_bb, const float* d_target_pi,
                                const float* d_target_v, int batch, float* d_grads, float* d_losses) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!h->has_trainer) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_trainer_init first");
    if (batch < 1 || !d_my_bb || !d_op_bb || !d_target_pi || !d_target_v || !d_grads)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_train_gradients_enqueue");
    HIP_TRY(h, hipSetDevice(h->device));
    return launch_grads(h, reinterpret_cast<const unsigned long long*>(d_my_bb), reinterpret_cast<const unsigned long long*>(d_op_bb),
                        d_target_pi, d_target_v, batch, d_grads, d_losses, nullptr, true, static_cast<hipStream_t>(stream));
}

int syn_train_apply_enqueue(syn_engine* h, void* stream, const float* d_grads, float lr, float grad_scale) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!h->has_trainer) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_trainer_init first");
    if (!d_grads) return fail(h, SYN_ERR_INVALID_ARGUMENT, "d_grads is NULL");
    HIP_TRY(h, hipSetDevice(h->device));
    return launch_adam(h, d_grads, lr, grad_scale, true, static_cast<hipStream_t>(stream));
}

int syn_train_step(syn_engine* h, const uint64_t* my_bb, const uint64_t* op_bb, const float* target_pi,
                   const float* target_v, int batch, float lr, float* losses) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!h->has_trainer) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_trainer_init first");
    if (batch < 1 || !my_bb || !op_bb || !target_pi || !target_v)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_train_step");
    HIP_TRY(h, hipSetDevice(h->device));
    size_t nb = (size_t)batch;
    int rc = ensure_scratch(h, nb * (16 + 36 + 12) + 256);
    if (rc != SYN_OK) return rc;
    unsigned long long* d_my = static_cast<unsigned long long*>(h->d_scratch);
    unsigned long long* d_op = d_my + nb;
    float* d_tpi = reinterpret_cast<float*>(d_op + nb);
    float* d_tv = d_tpi + nb * 9;
    HIP_TRY(h, hipMemcpyAsync(d_my, my_bb, nb * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_op, op_bb, nb * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_tpi, target_pi, nb * 36, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_tv, target_v, nb * 12, hipMemcpyHostToDevice, h->stream));
    rc = launch_grads(h, d_my, d_op, d_tpi, d_tv, batch, h->d_tgrad);
    if (rc != SYN_OK) return rc;
    rc = launch_adam(h, h->d_tgrad, lr, 1.0f);
    if (rc != SYN_OK) return rc;
    if (losses) HIP_TRY(h, hipMemcpyAsync(losses, h->d_tloss, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SYN_OK;
}

// ---- epochs without the host in the loop: the de-duplicated buffer is uploaded once per iteration, an epoch is one call
int syn_train_set_data(syn_engine* h, const uint64_t* my_bb, const uint64_t* op_bb, const float* target_pi,
                       const float* target_v, size_t n) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!h->has_trainer) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_trainer_init first");
    if (n < 1 || n > 0x7FFFFFFFu || !my_bb || !op_bb || !target_pi || !target_v)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_train_set_data");
    HIP_TRY(h, hipSetDevice(h->device));
    if (n > h->train_data_cap) {
        (void)hipFree(h->d_train_data);
        h->d_train_data = nullptr;
        h->train_data_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_train_data, n * 64));
        h->train_data_cap = n;
    }
    unsigned char* base = static_cast<unsigned char*>(h->d_train_data);
    HIP_TRY(h, hipMemcpyAsync(base, my_bb, n * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(base + n * 8, op_bb, n * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(base + n * 16, target_pi, n * 36, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(base + n * 52, target_v, n * 12, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->train_data_n = n;
    return SYN_OK;
}

int syn_train_epoch(syn_engine* h, const int32_t* perm, size_t n_steps, int batch, float lr, float* step_losses) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!h->has_trainer) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_trainer_init first");
    if (h->train_data_n == 0) return fail(h, SYN_ERR_INVALID_ARGUMENT, "call syn_train_set_data first");
    if (batch < 1 || (n_steps > 0 && !perm) || n_steps * (size_t)batch > 0x7FFFFFFFu)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_train_epoch");
    if (n_steps == 0) return SYN_OK;
    for (size_t i = 0; i < n_steps * (size_t)batch; i++)
        if (perm[i] < 0 || (size_t)perm[i] >= h->train_data_n)
            return fail(h, SYN_ERR_INVALID_ARGUMENT, "perm[%zu] = %d is outside the %zu uploaded states", i, perm[i],
                        h->train_data_n);
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = h->train_data_n, ni = n_steps * (size_t)batch;
    // scratch: [perm ni x 4][losses n_steps x 8][step-ordered batches: my, op (8 B each), pi (36 B), v (12 B) per sample]
    //          [per-step Adam scalars n_steps x 8][diagnostic stamps 4 KB]
    const size_t perm_bytes = (ni * 4 + 255) & ~(size_t)255, loss_bytes = (n_steps * 8 + 255) & ~(size_t)255;
    const size_t batch_bytes = (ni * 64 + 255) & ~(size_t)255;
    int rc = ensure_scratch(h, perm_bytes + loss_bytes + batch_bytes + loss_bytes + 4096 + 256);
    if (rc != SYN_OK) return rc;
    unsigned char* sc = static_cast<unsigned char*>(h->d_scratch);
    int* d_perm = reinterpret_cast<int*>(sc);
    float* d_losses = reinterpret_cast<float*>(sc + perm_bytes);
    unsigned long long* g_my = reinterpret_cast<unsigned long long*>(sc + perm_bytes + loss_bytes);
    unsigned long long* g_op = g_my + ni;
    float* g_tpi = reinterpret_cast<float*>(g_op + ni);
    float* g_tv = g_tpi + ni * 9;
    HIP_TRY(h, hipMemcpyAsync(d_perm, perm, ni * 4, hipMemcpyHostToDevice, h->stream));
    const unsigned char* base = static_cast<const unsigned char*>(h->d_train_data);
    // one gather for the whole epoch (the sampler's index_select), so a step reads its batch from consecutive addresses
    // instead of chasing perm -> sample inside the latency-bound step kernel
    hipLaunchKernelGGL(train_gather_kernel, dim3((unsigned)((ni * 16 + 255) / 256)), dim3(256), 0, h->stream, d_perm, (int)ni,
                       reinterpret_cast<const unsigned long long*>(base),
                       reinterpret_cast<const unsigned long long*>(base + n * 8),
                       reinterpret_cast<const float*>(base + n * 16), reinterpret_cast<const float*>(base + n * 52), g_my,
                       g_op, g_tpi, g_tv);
    HIP_TRY(h, hipGetLastError());
    // One persistent launch for the whole epoch (train_epoch.cuh) when the batch fits one 32-sample chunk — the reference's
    // batch_size. SYN_DEBUG=1 SYN_TRAIN_QUEUED=1 keeps the two launches per step below (A/B, same bits), as do larger batches.
    static const bool queued = debug_env("SYN_TRAIN_QUEUED") != nullptr;
    if (!queued && batch <= TrainGeom::CHUNK && h->trainer_kind == 0) {
        // per-step Adam scalars, in double on the host like libtorch (launch_adam)
        std::vector<float> adam_sc(2 * n_steps);
        for (size_t s = 0; s < n_steps; s++) {
            const double t = (double)(h->train_step + (long long)s + 1);
            const double bc1 = 1.0 - std::pow((double)h->train_hp.beta1, t);
            const double bc2 = 1.0 - std::pow((double)h->train_hp.beta2, t);
            adam_sc[s] = (float)((double)lr / bc1);
            adam_sc[n_steps + s] = (float)(1.0 / std::sqrt(bc2));
        }
        float* d_sc = reinterpret_cast<float*>(sc + perm_bytes + loss_bytes + batch_bytes);
        unsigned long long* d_prof = reinterpret_cast<unsigned long long*>(sc + perm_bytes + loss_bytes + batch_bytes + loss_bytes);
        static const bool prof = debug_env("SYN_TRAIN_PROFILE") != nullptr;
        int rc2 = SYN_OK;
        unsigned status[4] = {0u, 0u, 0u, 0u};
        unsigned long long stamps[16 * EP_WGS] = {0};
        do {
            hipError_t e;
#define EP_TRY(expr) if ((e = (expr)) != hipSuccess) { rc2 = fail(h, SYN_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e)); break; }
            EP_TRY(hipMemcpyAsync(d_sc, adam_sc.data(), adam_sc.size() * 4, hipMemcpyHostToDevice, h->stream));
            // snapshot of the learner (parameters, moments, both images: 0.85 MB of device copies): the launch below needs its 16
            // workgroups resident together; if it gives up (another kernel holds the CUs) the state is put back and the epoch runs
            // through the queued per-step launches instead — same bits, no co-residency requirement
            {
                const size_t pb = (size_t)TrainGeom::NUM_PARAMS * 4, ib = (size_t)MlpGeom::IMG_FLOATS * 4, tb = (size_t)TrainImg::T_FLOATS * 4;
                unsigned char* sn = reinterpret_cast<unsigned char*>(h->d_tsnap);
                EP_TRY(hipMemcpyAsync(sn, h->d_tw, pb, hipMemcpyDeviceToDevice, h->stream));
                EP_TRY(hipMemcpyAsync(sn + pb, h->d_tm, pb, hipMemcpyDeviceToDevice, h->stream));
                EP_TRY(hipMemcpyAsync(sn + 2 * pb, h->d_tv, pb, hipMemcpyDeviceToDevice, h->stream));
                EP_TRY(hipMemcpyAsync(sn + 3 * pb, h->d_twimg, ib, hipMemcpyDeviceToDevice, h->stream));
                EP_TRY(hipMemcpyAsync(sn + 3 * pb + ib, h->d_ttimg, tb, hipMemcpyDeviceToDevice, h->stream));
            }
            EP_TRY(hipMemsetAsync(d_prof, 0, 2048, h->stream));
            EP_TRY(hipMemsetAsync(h->d_tsync, 0, 256, h->stream));
            float* img2 = h->d_timg2;
            float* timg2 = h->d_timg2 + MlpGeom::IMG_FLOATS;
            // both buffers start as the current network: the kernel rewrites every parameter's slot, never the padding
            EP_TRY(hipMemcpyAsync(img2, h->d_twimg, (size_t)MlpGeom::IMG_FLOATS * 4, hipMemcpyDeviceToDevice, h->stream));
            EP_TRY(hipMemcpyAsync(timg2, h->d_ttimg, (size_t)TrainImg::T_FLOATS * 4, hipMemcpyDeviceToDevice, h->stream));
            EpochParams ep{};
            ep.w = h->d_tw; ep.m = h->d_tm; ep.v = h->d_tv;
            ep.img[0] = h->d_twimg; ep.img[1] = img2;
            ep.timg[0] = h->d_ttimg; ep.timg[1] = timg2;
            ep.my_bb = g_my; ep.op_bb = g_op; ep.tpi = g_tpi; ep.tv = g_tv;
            ep.step_size = d_sc; ep.inv_sqrt_bc2 = d_sc + n_steps;
            ep.losses = d_losses; ep.grads = h->d_tgrad; ep.sync = h->d_tsync;
            ep.prof = prof ? d_prof : nullptr;
            ep.n_steps = (int)n_steps; ep.batch = batch; ep.hp = h->train_hp;
            static const bool device_scope = debug_env("SYN_TRAIN_DEVICE_SCOPE") != nullptr;
            ep.force_device_scope = (device_scope || h->epoch_device_scope) ? 1 : 0;
            const size_t lds = (size_t)TrainGeom::WL_OFF * 4;
            EP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(train_epoch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(train_epoch_kernel, dim3(EP_WGS * EP_XCDS), dim3(EP_THREADS), lds, h->stream, ep);
            EP_TRY(hipGetLastError());
            if (n_steps & 1) {  // the final network sits in the second buffer: bring the first one (the published image) up to date
                EP_TRY(hipMemcpyAsync(h->d_twimg, img2, (size_t)MlpGeom::IMG_FLOATS * 4, hipMemcpyDeviceToDevice, h->stream));
                EP_TRY(hipMemcpyAsync(h->d_ttimg, timg2, (size_t)TrainImg::T_FLOATS * 4, hipMemcpyDeviceToDevice, h->stream));
            }
            if (step_losses) EP_TRY(hipMemcpyAsync(step_losses, d_losses, n_steps * 8, hipMemcpyDeviceToHost, h->stream));
            EP_TRY(hipMemcpyAsync(status, h->d_tsync, 16, hipMemcpyDeviceToHost, h->stream));
            if (prof) EP_TRY(hipMemcpyAsync(stamps, d_prof, sizeof(stamps), hipMemcpyDeviceToHost, h->stream));
            EP_TRY(hipStreamSynchronize(h->stream));
#undef EP_TRY
        } while (0);
        if (rc2 != SYN_OK) return rc2;
        const bool force_abort = debug_env("SYN_TRAIN_FORCE_ABORT") != nullptr;  // test hook: take the recovery path
        bool aborted = status[1] != 0u || force_abort;
        if (aborted) {
            // the workers were not resident together: put the learner back and run the epoch through the queued launches below
            const size_t pb = (size_t)TrainGeom::NUM_PARAMS * 4, ib = (size_t)MlpGeom::IMG_FLOATS * 4, tb = (size_t)TrainImg::T_FLOATS * 4;
            const unsigned char* sn = reinterpret_cast<const unsigned char*>(h->d_tsnap);
            HIP_TRY(h, hipMemcpyAsync(h->d_tw, sn, pb, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->d_tm, sn + pb, pb, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->d_tv, sn + 2 * pb, pb, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->d_twimg, sn + 3 * pb, ib, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->d_ttimg, sn + 3 * pb + ib, tb, hipMemcpyDeviceToDevice, h->stream));
            h->epoch_fallbacks++;
        }
        if (!aborted && prof) {
            // stamps of step 2, workgroup g at [16 g ..]: top, features, forward L0..L4, heads, act-grads L4..L1, parameter jobs, step barrier
            fprintf(stderr, "[syn train profile] epoch kernel (%s), step 2, cycles per phase\n", status[3] ? "workers on one XCD" : "device-scope barrier");
            for (int g = 0; g < EP_WGS; g++) {
                fprintf(stderr, "  wg %d (start %+lld):", g, (long long)(stamps[16 * g] - stamps[0]));
                for (int i = 1; i < 14; i++) fprintf(stderr, " %llu", stamps[16 * g + i] - stamps[16 * g + i - 1]);
                fprintf(stderr, " | probe fresh line %llu, cold line %llu\n", stamps[16 * g + 14], stamps[16 * g + 15]);
            }
        }
        if (!aborted) {
            h->train_step += (long long)n_steps;
            return SYN_OK;
        }
    }
    if (!queued && batch <= ConvTrainGeom::CHUNK && h->trainer_kind == 1) {
        // Connect4ConvNet: the persistent one-workgroup epoch kernel (train_conv_mfma.cuh) — gradients and Adam of every step in ONE
        // launch, no co-residency requirement
        std::vector<float> adam_sc(2 * n_steps);
        for (size_t s = 0; s < n_steps; s++) {
            const double t = (double)(h->train_step + (long long)s + 1);
            const double bc1 = 1.0 - std::pow((double)h->train_hp.beta1, t);
            const double bc2 = 1.0 - std::pow((double)h->train_hp.beta2, t);
            adam_sc[s] = (float)((double)lr / bc1);
            adam_sc[n_steps + s] = (float)(1.0 / std::sqrt(bc2));
        }
        float* d_sc = reinterpret_cast<float*>(sc + perm_bytes + loss_bytes + batch_bytes);
        HIP_TRY(h, hipMemcpyAsync(d_sc, adam_sc.data(), adam_sc.size() * 4, hipMemcpyHostToDevice, h->stream));
        ConvEpochParams ep{};
        ep.w = h->d_tw; ep.m = h->d_tm; ep.v = h->d_tv;
        ep.my_bb = g_my; ep.op_bb = g_op; ep.tpi = g_tpi; ep.tv = g_tv;
        ep.step_size = d_sc; ep.inv_sqrt_bc2 = d_sc + n_steps;
        ep.losses = d_losses; ep.grads = h->d_tgrad;
        ep.n_steps = (int)n_steps; ep.batch = batch; ep.hp = h->train_hp;
        unsigned long long* d_cprof = reinterpret_cast<unsigned long long*>(sc + perm_bytes + loss_bytes + batch_bytes + loss_bytes);
        const bool cprof = debug_env("SYN_TRAIN_PROFILE") != nullptr && n_steps > 2;
        if (cprof) HIP_TRY(h, hipMemsetAsync(d_cprof, 0, 128, h->stream));
        ep.prof = cprof ? d_cprof : nullptr;
        const size_t clds = (size_t)ConvMfmaGeom::LDS_FLOATS * 4;
        // the step spread over four workgroups of one XCD (train_conv_epoch_kernel_mw: same chains, same bits, f32 and bf16). They must be
        // resident together: the learner is snapshotted first, and a launch that gives up (or SYN_DEBUG=1 SYN_TRAIN_CONV_MW=0, or a
        // failed start-up self-check) runs the one-workgroup kernel below instead.
        static const bool mw_off = [] { const char* e = debug_env("SYN_TRAIN_CONV_MW"); return e && std::atoi(e) == 0; }();
        const bool want_mw = h->conv_mw_force >= 0 ? h->conv_mw_force == 1 : (!mw_off && !h->conv_mw_disabled);
        if (want_mw) {
            const size_t pb = (size_t)ConvGeom::NUM_PARAMS * 4;
            unsigned char* sn = reinterpret_cast<unsigned char*>(h->d_tsnap);
            HIP_TRY(h, hipMemcpyAsync(sn, h->d_tw, pb, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(sn + pb, h->d_tm, pb, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(sn + 2 * pb, h->d_tv, pb, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemsetAsync(h->d_tsync, 0, 256, h->stream));
            if (cprof) HIP_TRY(h, hipMemsetAsync(d_cprof, 0, 512, h->stream));
            ConvMwParams mp{};
            mp.e = ep;
            mp.xbuf = h->d_cxbuf;
            mp.sync = h->d_tsync;
            static const bool device_scope = debug_env("SYN_TRAIN_DEVICE_SCOPE") != nullptr;
            mp.force_device_scope = device_scope ? 1 : 0;
            const size_t mlds = (size_t)ConvMwGeom::LDS_FLOATS * 4;
            auto km = h->train_bf16 ? train_conv_epoch_kernel_mw<true> : train_conv_epoch_kernel_mw<false>;
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(km), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
            hipLaunchKernelGGL(km, dim3(CONV_MW_WGS * CONV_MW_XCDS), dim3(CONV_TRAIN_THREADS), mlds, h->stream, mp);
            HIP_TRY(h, hipGetLastError());
            unsigned status[4] = {0u, 0u, 0u, 0u};
            HIP_TRY(h, hipMemcpyAsync(status, h->d_tsync, 16, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            const bool aborted = status[1] != 0u || debug_env("SYN_TRAIN_FORCE_ABORT") != nullptr;
            if (!aborted) {
                if (step_losses) HIP_TRY(h, hipMemcpy(step_losses, d_losses, n_steps * 8, hipMemcpyDeviceToHost));
                if (cprof) {
                    unsigned long long t[16 * CONV_MW_WGS] = {0};
                    HIP_TRY(h, hipMemcpy(t, d_cprof, sizeof(t), hipMemcpyDeviceToHost));
                    fprintf(stderr, "[syn train profile] conv epoch kernel on %d workgroups (%s), step 2, cycles: stage | F | barrier | H | G1 | G2 | Adam (own head weights, inside the barrier) | rest of the barrier | dY in | G3 | barrier | G4 + Adam (shared)\n",
                            CONV_MW_WGS, status[3] ? "one XCD" : "device-scope barrier");
                    for (int g = 0; g < CONV_MW_WGS; g++) {
                        fprintf(stderr, "  wg %d (start %+lld):", g, (long long)(t[16 * g] - t[0]));
                        for (int i = 1; i < 13; i++) fprintf(stderr, " %llu", t[16 * g + i] - t[16 * g + i - 1]);
                        fprintf(stderr, " | total %llu\n", t[16 * g + 12] - t[16 * g]);
                    }
                }
                h->train_step += (long long)n_steps;
                return SYN_OK;
            }
            HIP_TRY(h, hipMemcpyAsync(h->d_tw, sn, pb, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->d_tm, sn + pb, pb, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->d_tv, sn + 2 * pb, pb, hipMemcpyDeviceToDevice, h->stream));
            h->epoch_fallbacks++;
            if (cprof) HIP_TRY(h, hipMemsetAsync(d_cprof, 0, 128, h->stream));
        }
        auto ke = h->train_bf16 ? train_conv_epoch_kernel<true> : train_conv_epoch_kernel<false>;
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(ke), hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));
        hipLaunchKernelGGL(ke, dim3(1), dim3(CONV_TRAIN_THREADS), clds, h->stream, ep);
        HIP_TRY(h, hipGetLastError());
        if (step_losses) HIP_TRY(h, hipMemcpyAsync(step_losses, d_losses, n_steps * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (cprof && !h->train_bf16) {
            unsigned long long t[16] = {0};
            HIP_TRY(h, hipMemcpy(t, d_cprof, sizeof(t), hipMemcpyDeviceToHost));
            fprintf(stderr, "[syn train profile] conv epoch kernel, step 2, cycles: stage %llu | F %llu | H %llu | losses+G1 %llu | G2 %llu | G3 %llu | G4 %llu | Adam %llu | total %llu\n",
                    t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6], t[15] - t[7], t[15] - t[0]);
        }
        h->train_step += (long long)n_steps;
        return SYN_OK;
    }
    for (size_t s = 0; s < n_steps; s++) {  // steps are dependent (weights of step s feed step s+1): queued, never synced
        const size_t o = s * (size_t)batch;
        rc = launch_grads(h, g_my + o, g_op + o, g_tpi + o * 9, g_tv + o * 3, batch, h->d_tgrad, d_losses + 2 * s);
        if (rc != SYN_OK) return rc;
        rc = launch_adam(h, h->d_tgrad, lr, 1.0f);
        if (rc != SYN_OK) return rc;
    }
    if (step_losses) HIP_TRY(h, hipMemcpyAsync(step_losses, d_losses, n_steps * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SYN_OK;
}

int syn_trainer_get_state(syn_engine* h, float* blob, float* m, float* v, long long* step, float* grads) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!h->has_trainer) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_trainer_init first");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t bytes = (size_t)(h->trainer_kind == 1 ? ConvGeom::NUM_PARAMS : TrainGeom::NUM_PARAMS) * 4;
    if (blob) HIP_TRY(h, hipMemcpyAsync(blob, h->d_tw, bytes, hipMemcpyDeviceToHost, h->stream));
    if (m) HIP_TRY(h, hipMemcpyAsync(m, h->d_tm, bytes, hipMemcpyDeviceToHost, h->stream));
    if (v) HIP_TRY(h, hipMemcpyAsync(v, h->d_tv, bytes, hipMemcpyDeviceToHost, h->stream));
    if (grads) HIP_TRY(h, hipMemcpyAsync(grads, h->d_tgrad, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (step) *step = h->train_step;
    return SYN_OK;
}

// weight hand-off learner -> self-play (the reference does it through models/model_i.ot, alpha_zero.rs:97,194)
int syn_trainer_publish_weights(syn_engine* h) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!h->has_trainer) return fail(h, SYN_ERR_NO_WEIGHTS, "call syn_trainer_init first");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->trainer_kind == 1) {
        // Connect4ConvNet: the fragment image of convnet.cuh is rebuilt from the canonical parameters on the device
        if (h->cap > LANE_MAX_CAP) return fail(h, SYN_ERR_UNSUPPORTED, "Connect4ConvNet runs in the lane-per-tree kernels only");
        hipLaunchKernelGGL(conv_image_kernel, dim3((ConvGeom::IMG_FLOATS + 255) / 256), dim3(256), 0, h->stream, h->d_tw, h->d_wimg);
        HIP_TRY(h, hipGetLastError());
        if (h->d_cache) HIP_TRY(h, hipMemsetAsync(h->d_cache, 0, (size_t)64 << h->cache_log2, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->has_weights = true;
        h->net_kind = 1;
        h->host_blob.clear();
        h->img16_current = false;
        h->net_arith = SYN_NET_ARITH_F32;
        return SYN_OK;
    }
    // the trainer keeps its weights in the inference fragment order as well (train_mfma.cuh): publishing is one device copy
    HIP_TRY(h, hipMemcpyAsync(h->d_wimg, h->d_twimg, (size_t)MlpGeom::IMG_FLOATS * 4, hipMemcpyDeviceToDevice, h->stream));
    if (h->d_cache) HIP_TRY(h, hipMemsetAsync(h->d_cache, 0, (size_t)64 << h->cache_log2, h->stream));  // new network: empty PolicyWithCache
    // the f16x2 image is built on the host from the canonical parameters: keep a copy of what was published
    h->host_blob.resize((size_t)MlpGeom::NUM_PARAMS);
    HIP_TRY(h, hipMemcpyAsync(h->host_blob.data(), h->d_tw, (size_t)MlpGeom::NUM_PARAMS * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->has_weights = true;
    h->net_kind = 0;
    h->img16_current = false;
    if (h->net_arith == SYN_NET_ARITH_F16X2) return ensure_f16x2_image(h);
    return SYN_OK;
}

// ------------------------------------------------------------------------------------------------ deduplicate
int syn_replay_deduplicate(syn_engine* h, const uint64_t* my_bb, const uint64_t* op_bb, const float* pis,
                           const float* vs, size_t n, uint64_t* out_my, uint64_t* out_op, float* out_pi, float* out_v,
                           uint32_t* out_num, size_t* out_count) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!out_count) return fail(h, SYN_ERR_INVALID_ARGUMENT, "out_count is NULL");
    *out_count = 0;
    if (n == 0) return SYN_OK;
    if (!my_bb || !op_bb || !pis || !vs || !out_my || !out_op || !out_pi || !out_v || !out_num || n > 0x7FFFFFFFu)
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments to syn_replay_deduplicate");
    HIP_TRY(h, hipSetDevice(h->device));
    const int ni = (int)n;
    // device layout: inputs | sort keys/values (double buffers) | heads | scan | seg_start | outputs | cub temp
    size_t tmp_sort = 0, tmp_scan = 0;
    HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, (const unsigned long long*)nullptr,
                                                  (unsigned long long*)nullptr, (const unsigned*)nullptr,
                                                  (unsigned*)nullptr, ni, 0, 64, h->stream));
    HIP_TRY(h, hipcub::DeviceScan::InclusiveSum(nullptr, tmp_scan, (const unsigned*)nullptr, (unsigned*)nullptr, ni,
                                                h->stream));
    size_t tmp = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    size_t o_my = take(n * 8), o_op = take(n * 8), o_pi = take(n * 36), o_v = take(n * 12);
    size_t o_k0 = take(n * 8), o_k1 = take(n * 8), o_i0 = take(n * 4), o_i1 = take(n * 4);
    size_t o_head = take(n * 4), o_scan = take(n * 4), o_start = take(n * 4);
    size_t o_omy = take(n * 8), o_oop = take(n * 8), o_opi = take(n * 36), o_ov = take(n * 12), o_on = take(n * 4);
    size_t o_tmp = take(tmp);
    int rc = ensure_scratch(h, off + 256);
    if (rc != SYN_OK) return rc;
    char* base = static_cast<char*>(h->d_scratch);
    auto P8 = [&](size_t o) { return reinterpret_cast<unsigned long long*>(base + o); };
    auto P4 = [&](size_t o) { return reinterpret_cast<unsigned*>(base + o); };
    auto PF = [&](size_t o) { return reinterpret_cast<float*>(base + o); };
    HIP_TRY(h, hipMemcpyAsync(P8(o_my), my_bb, n * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(P8(o_op), op_bb, n * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(PF(o_pi), pis, n * 36, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(PF(o_v), vs, n * 12, hipMemcpyHostToDevice, h->stream));
    const int blocks = (ni + 255) / 256;
    // stable LSD sort of the buffer indices by the 128-bit key: first by op_bb, then by my_bb
    hipLaunchKernelGGL(iota_kernel, dim3(blocks), dim3(256), 0, h->stream, P4(o_i0), ni);
    size_t t1 = tmp;
    HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(base + o_tmp, t1, P8(o_op), P8(o_k0), P4(o_i0), P4(o_i1), ni, 0, 64,
                                                  h->stream));
    hipLaunchKernelGGL(gather_u64_kernel, dim3(blocks), dim3(256), 0, h->stream, P8(o_my), P4(o_i1), ni, P8(o_k1));
    t1 = tmp;
    HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(base + o_tmp, t1, P8(o_k1), P8(o_k0), P4(o_i1), P4(o_i0), ni, 0, 64,
                                                  h->stream));
    // o_k0 = my_bb sorted, o_i0 = buffer indices in (my, op, index) order; op_bb in that order:
    hipLaunchKernelGGL(gather_u64_kernel, dim3(blocks), dim3(256), 0, h->stream, P8(o_op), P4(o_i0), ni, P8(o_k1));
    hipLaunchKernelGGL(dedup_heads_kernel, dim3(blocks), dim3(256), 0, h->stream, P8(o_k0), P8(o_k1), ni, P4(o_head));
    t1 = tmp;
    HIP_TRY(h, hipcub::DeviceScan::InclusiveSum(base + o_tmp, t1, P4(o_head), P4(o_scan), ni, h->stream));
    hipLaunchKernelGGL(dedup_starts_kernel, dim3(blocks), dim3(256), 0, h->stream, P4(o_head), P4(o_scan), ni,
                       P4(o_start));
    unsigned m_u = 0;
    HIP_TRY(h, hipMemcpyAsync(&m_u, P4(o_scan) + (ni - 1), 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int m = (int)m_u;
    hipLaunchKernelGGL(dedup_reduce_kernel, dim3((m * 16 + 255) / 256), dim3(256), 0, h->stream, P4(o_i0), P4(o_start),
                       m, ni, P8(o_my), P8(o_op), PF(o_pi), PF(o_v), P8(o_omy), P8(o_oop), PF(o_opi), PF(o_ov),
                       P4(o_on));
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out_my, P8(o_omy), (size_t)m * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_op, P8(o_oop), (size_t)m * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_pi, PF(o_opi), (size_t)m * 36, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_v, PF(o_ov), (size_t)m * 12, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_num, P4(o_on), (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *out_count = (size_t)m;
    return SYN_OK;
}

int syn_last_timing(const syn_engine* h, float* kernel_ms, int* n_launches) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (kernel_ms) *kernel_ms = h->last_kernel_ms;
    if (n_launches) *n_launches = h->last_launches;
    return SYN_OK;
}

int syn_last_cache_stats(const syn_engine* h, uint64_t* hits, uint64_t* misses) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (hits) *hits = h->last_cache_hits;
    if (misses) *misses = h->last_cache_misses;
    return SYN_OK;
}

int syn_last_launch_shape(const syn_engine* h, int* shape, int* grid, int* threads) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (shape) *shape = h->last_shape;
    if (grid) *grid = h->last_grid;
    if (threads) *threads = h->last_threads;
    return SYN_OK;
}

int syn_debug_calibrate(syn_engine* h, const void* d_base, const uint32_t* d_span_off, int n_spans, int do_write) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!d_base || !d_span_off || n_spans < 0) return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_scratch(h, 1 << 20);
    if (rc != SYN_OK) return rc;
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(calib_gather_kernel, dim3(2048), dim3(256), 0, h->stream, static_cast<const float4*>(d_base),
                       d_span_off, n_spans, static_cast<float4*>(h->d_scratch), do_write);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipEventElapsedTime(&h->last_kernel_ms, h->ev0, h->ev1));
    return SYN_OK;
}

int syn_debug_stdrng_u32(syn_engine* h, uint64_t seed, int n, uint32_t* out) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (n < 0 || (n > 0 && !out)) return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments");
    if (n == 0) return SYN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_scratch(h, (size_t)n * 4 + 256);
    if (rc != SYN_OK) return rc;
    uint32_t* d = static_cast<uint32_t*>(h->d_scratch);
    hipLaunchKernelGGL(debug_rng_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, (unsigned long long)seed, n, d);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out, d, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SYN_OK;
}

int syn_activation_forward(syn_engine* h, int kind, const float* x, int batch, int n, float* y) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (kind < SYN_ACT_RELU || kind > SYN_ACT_SOFTMAX) return fail(h, SYN_ERR_INVALID_ARGUMENT, "unknown activation %d", kind);
    if (batch < 0 || n < 1 || (batch > 0 && (!x || !y))) return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments");
    if (batch == 0) return SYN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t cnt = (size_t)batch * (size_t)n;
    int rc = ensure_scratch(h, cnt * 8 + 256);
    if (rc != SYN_OK) return rc;
    float* dx = static_cast<float*>(h->d_scratch);
    float* dy = dx + cnt;
    HIP_TRY(h, hipMemcpyAsync(dx, x, cnt * 4, hipMemcpyHostToDevice, h->stream));
    const size_t threads = kind == SYN_ACT_SOFTMAX ? (size_t)batch : cnt;
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(activation_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, h->stream, kind, dx, batch, n, dy);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipMemcpyAsync(y, dy, cnt * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipEventElapsedTime(&h->last_kernel_ms, h->ev0, h->ev1));
    h->last_launches = 1;
    return SYN_OK;
}

int syn_debug_fast_div(syn_engine* h, const float* a, const float* b, int n, float* out_fast, float* out_full) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (n < 0 || (n > 0 && (!a || !b || !out_fast || !out_full))) return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments");
    if (n == 0) return SYN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    size_t nb = (size_t)n;
    int rc = ensure_scratch(h, nb * 16 + 256);
    if (rc != SYN_OK) return rc;
    float* da = static_cast<float*>(h->d_scratch);
    float* db = da + nb;
    float* df = db + nb;
    float* dd = df + nb;
    HIP_TRY(h, hipMemcpyAsync(da, a, nb * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(db, b, nb * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(debug_fast_div_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, da, db, n, df, dd);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out_fast, df, nb * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_full, dd, nb * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SYN_OK;
}

int syn_debug_small_int_math(syn_engine* h, int b_lo, int b_hi, unsigned long long* mismatches3) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (!mismatches3 || b_lo < 1 || b_hi < b_lo || b_hi > 65536) return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_scratch(h, 256);
    if (rc != SYN_OK) return rc;
    unsigned long long* dm = static_cast<unsigned long long*>(h->d_scratch);
    HIP_TRY(h, hipMemsetAsync(dm, 0, 24, h->stream));
    hipLaunchKernelGGL(debug_small_int_math_kernel, dim3((1u << 23) / 256), dim3(256), 0, h->stream, b_lo, b_hi, dm);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(mismatches3, dm, 24, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SYN_OK;
}

int syn_debug_math(syn_engine* h, const float* a, const float* b, int n, float* out_exp_a, float* out_div,
                   float* out_sqrt_a) {
    if (!h) return SYN_ERR_INVALID_ARGUMENT;
    if (n < 0 || (n > 0 && (!a || !b || !out_exp_a || !out_div || !out_sqrt_a)))
        return fail(h, SYN_ERR_INVALID_ARGUMENT, "bad arguments");
    if (n == 0) return SYN_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    size_t nb = (size_t)n;
    int rc = ensure_scratch(h, nb * 20 + 256);
    if (rc != SYN_OK) return rc;
    float* da = static_cast<float*>(h->d_scratch);
    float* db = da + nb;
    float* de = db + nb;
    float* dd = de + nb;
    float* ds = dd + nb;
    HIP_TRY(h, hipMemcpyAsync(da, a, nb * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(db, b, nb * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, da, db, n, de, dd, ds);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out_exp_a, de, nb * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_div, dd, nb * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_sqrt_a, ds, nb * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SYN_OK;
}

}  // extern "C"

// synthesis_amd — Policy::eval (study-connect4/src/policies.rs:47-59) as a CALL: the latency kernels behind syn_eval_ctx_* /
// syn_policy_eval_batch for batches whose answers go straight back into pinned host memory.
//
// policy_eval_kernel (engine_kernels.cuh) is a throughput kernel: every workgroup stages the 122 KB weight image into LDS, then a
// wave computes whole 16-position tiles alone (476 dependent-issue MFMAs = 6.8 us). For a batch of at most a tile per CU — a Rust
// `impl Policy` calls with n = 1, a host-tree worker with a few hundred leaves — that is all latency. policy_eval_tile_kernel is
// the shape the 16-trees-per-CU self-play kernel uses for its phase B (mlp.cuh::mlp_split_tile16): one workgroup of four waves per
// tile, each wave loads its quarter of the weights from L2 straight into registers (a workgroup uses every weight exactly once,
// staging them in LDS first buys nothing) and computes its share of every layer's output columns; the activations cross between
// the waves through 14 KB of LDS. Accumulation order per output is unchanged, so the bits are policy_eval_kernel's.
//
// Completion without a stream synchronisation (host_flag != nullptr): the last workgroup to finish stores the call's sequence
// number into a word of pinned host memory (system-scope release behind the storing wave's own system-scope fence); the host polls
// that word. Saves the runtime's completion-signal path (≈ 10 us per call).
#pragma once
#include "engine_kernels.cuh"

namespace syn {

// Called by the ONE wave of the workgroup that stored results (all of its lanes). done_blocks: device word, zero between calls;
// host_flag: pinned, device-mapped word.
SYN_DEV void eval_signal_completion(unsigned* done_blocks, unsigned* host_flag, unsigned seq) {
    __threadfence_system();   // the wave's results are visible to the host before anything below
    if ((threadIdx.x & 63) == 0) {
        const unsigned prev = __hip_atomic_fetch_add(done_blocks, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1u) {
            __hip_atomic_store(done_blocks, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

SYN_DEV void eval_store_outputs(const f32x4& o_in, int q, size_t pos, float* __restrict__ logits, float* __restrict__ value) {
    f32x4 o = o_in;
    if (q < 2) {
#pragma unroll
        for (int r = 0; r < 4; r++) logits[pos * 9 + q * 4 + r] = o[r];
    } else if (q == 2) {
        logits[pos * 9 + 8] = o[0];
        float v0 = o[1], v1 = o[2], v2 = o[3];
        value_softmax(v0, v1, v2);
        value[pos * 3 + 0] = v0;
        value[pos * 3 + 1] = v1;
        value[pos * 3 + 2] = v2;
    }
}

// one workgroup of four waves per 16-position tile (grid = tiles; meant for batches of at most a tile per CU)
__global__ __launch_bounds__(256) void policy_eval_tile_kernel(const float* __restrict__ g_wimg,
                                                               const unsigned long long* __restrict__ my_bb,
                                                               const unsigned long long* __restrict__ op_bb, int n,
                                                               float* __restrict__ logits, float* __restrict__ value,
                                                               unsigned* done_blocks, unsigned* host_flag, unsigned seq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* exA = reinterpret_cast<f32x4*>(smem_raw);                 // 8 x 64 x 16 B
    f32x4* exB = reinterpret_cast<f32x4*>(smem_raw + 8 * 64 * 16);   // 6 x 64 x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    MlpSplitWeights W;
    mlp_split_load_weights<false>(g_wimg, wave, lane, W);
    const int j = lane & 15, q = lane >> 4;
    const FeatureTable FT = make_feature_table(q);
    const int pos = (int)blockIdx.x * 16 + j;
    const bool valid = pos < n;
    const uint64_t my = valid ? my_bb[pos] : 0ull, op = valid ? op_bb[pos] : 0ull;
    uint64_t hi, lo;
    feature_boards(my, op, hi, lo);
    const f32x4 o = mlp_split_tile16<false>(W, g_wimg + MlpGeom::W_FLOATS, nullptr, exA, exB, wave, lane, FT, hi, lo);
    if (wave == 0) {   // (the last layer lives on wave 0: the only wave with results)
        if (valid) eval_store_outputs(o, q, (size_t)pos, logits, value);
        if (host_flag != nullptr) eval_signal_completion(done_blocks, host_flag, seq);
    }
}

}  // namespace syn

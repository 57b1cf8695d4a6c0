// Microbenchmark: what one CU's address path (TA / vector L1) charges for the tree phases' access shapes, 16 waves per CU.
// Every lane owns a slab of 128-byte blocks (as in lane_kernel.cuh) and walks it with a data-dependent next block, so each
// step is one dependent memory round trip per wave — the descent loop's shape. Variants differ only in HOW a step touches its
// line:  R<n>  n x global_load_dwordx4 per lane, 64 lanes -> 64 different lines (n = 8 is the descent level of today)
//        C8    the same 8 KB per wave-step fetched cooperatively: 8 instructions, each 8 lines x 8 lanes x 16 B
//        W9x3  nine 12-byte stores per lane (child records today);  W7x4  seven 16-byte stores (the same 108 B as whole chunks)
//        W3    backprop level of today: 16 B header + 4 B + 2 B;      W2   16 B + 8 B;   W1  one 16-byte store
// Reported per variant: wall ms, G lane-steps/s, and cycles per wave-step per CU (kernel cycles / steps / waves per CU) — the
// figure a per-CU address-processing bound shows up in. POOL: blocks per lane (4 = 128 MB in all, inside the Infinity Cache;
// 1024 = 32 GB, HBM).   Build: hipcc -O3 --offload-arch=gfx950 -o ta_rate ta_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

enum { R8, R4, R2, R1, C8, W9x3, W7x4, W3, W2, W1, R8W3, NVAR };

template <int V>
__global__ __launch_bounds__(1024) void walk(unsigned char* pool, uint32_t nblocks, int steps, uint32_t* out, unsigned long long* cyc) {
    const size_t lane_global = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned char* slab = pool + lane_global * (size_t)nblocks * 128;
    const int lane = threadIdx.x & 63;
    uint32_t state = (uint32_t)(lane_global * 2654435761u) + 12345u;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; s++) {
        const uint32_t idx = (state >> 8) % nblocks;
        u32x4* p = reinterpret_cast<u32x4*>(slab + (size_t)idx * 128);
        uint32_t h = 0;
        constexpr int NR = V == R8 || V == R8W3 ? 8 : V == R4 ? 4 : V == R2 ? 2 : V == R1 ? 1 : 0;
        if (NR > 0) {
            u32x4 r[8];
#pragma unroll
            for (int i = 0; i < NR; i++) r[i] = p[i];
#pragma unroll
            for (int i = 0; i < NR; i++) h += r[i][0] ^ r[i][3];
        }
        if (V == C8) {
            // instruction i fetches the lines of trees 8i .. 8i+7: lane l takes chunk l & 7 of tree 8i + (l >> 3)
            const uint32_t lo = (uint32_t)(uintptr_t)p, hi = (uint32_t)((uintptr_t)p >> 32);
            u32x4 r[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int src = 8 * i + (lane >> 3);
                const uint32_t l2 = (uint32_t)__shfl((int)lo, src, 64), h2 = (uint32_t)__shfl((int)hi, src, 64);
                const u32x4* q = reinterpret_cast<const u32x4*>(((uintptr_t)h2 << 32) | l2) + (lane & 7);
                r[i] = *q;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) h += r[i][0] ^ r[i][3];
            // every lane's next block depends on the whole wave's data (stands in for the transposition back to tree lanes)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) h += (uint32_t)__shfl_xor((int)h, off, 64);
        }
        acc += h;
        const u32x4 v = u32x4{h, state, acc, 1u};
        if (V == W9x3) {
#pragma unroll
            for (int i = 0; i < 9; i++) *reinterpret_cast<u32x3*>(reinterpret_cast<unsigned char*>(p) + 16 + 12 * i) = u32x3{h, state, acc + i};
        }
        if (V == W7x4) {
#pragma unroll
            for (int i = 1; i < 8; i++) p[i] = v;
        }
        if (V == W3 || V == R8W3) {
            p[0] = v;
            reinterpret_cast<uint32_t*>(p)[16 + (state & 15u)] = h;
            reinterpret_cast<unsigned short*>(p)[36 + (state & 7u) * 2] = (unsigned short)state;
        }
        if (V == W2) {
            p[0] = v;
            *reinterpret_cast<u32x2*>(reinterpret_cast<uint32_t*>(p) + 16 + 2 * (state & 7u)) = u32x2{h, state};
        }
        if (V == W1) p[0] = v;
        if (NR == 0 && V != C8) {
            // stores only: keep one dependent round trip per step like the read variants (a 4-byte read of the line)
            h = reinterpret_cast<volatile uint32_t*>(p)[31];
        }
        state = state * 1664525u + 1013904223u + h;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[lane_global] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char* name, unsigned char* pool, uint32_t nb, uint32_t* out, unsigned long long* cyc, int waves) {
    const int steps = 2000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((walk<V>), dim3(grid), dim3(64 * waves), 0, 0, pool, nb, steps, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; i++) c += (double)h[i];
    c /= 256.0;
    printf("%-6s waves/CU %2d blocks/lane %5u  %8.2f ms  %7.2f G lane-steps/s  %8.0f cycles per step (all waves of a CU)  %6.0f per wave-step\n",
           name, waves, nb, best, (double)grid * 64 * waves * steps / best / 1e6, c / steps, c / steps / waves);
    fflush(stdout);
}

int main() {
    unsigned char* pool; uint32_t* out; unsigned long long* cyc;
    const size_t lanes = 256 * 1024;
    const uint32_t big = 1024;
    if (hipMalloc(&pool, lanes * big * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, lanes * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(pool, 1, lanes * big * 128);
    for (uint32_t nb : {4u, big}) {
        for (int waves : {16, 8, 4}) {
            run<R8>("R8", pool, nb, out, cyc, waves);
            run<R4>("R4", pool, nb, out, cyc, waves);
            run<R2>("R2", pool, nb, out, cyc, waves);
            run<R1>("R1", pool, nb, out, cyc, waves);
            run<C8>("C8", pool, nb, out, cyc, waves);
            if (waves != 16) continue;
            run<W9x3>("W9x3", pool, nb, out, cyc, waves);
            run<W7x4>("W7x4", pool, nb, out, cyc, waves);
            run<W3>("W3", pool, nb, out, cyc, waves);
            run<W2>("W2", pool, nb, out, cyc, waves);
            run<W1>("W1", pool, nb, out, cyc, waves);
            run<R8W3>("R8W3", pool, nb, out, cyc, waves);
        }
    }
    return 0;
}

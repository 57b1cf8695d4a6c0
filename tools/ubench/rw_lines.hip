// Microbenchmark: what does the memory system charge for the WRITE side of a lane-per-tree explore?
// Every lane walks random 128-byte blocks of its own slab (the node-pool access pattern: one block = one cache line) and,
// per visit, reads part of the line and/or dirties part of it. Reports line visits per second for each read/write mix.
// Build: hipcc -O3 --offload-arch=gfx950 -o rw_lines rw_lines.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// RD = number of 16-byte loads of the line (0, 7 or 8); WR: 0 none, 1 = 16 B header, 2 = 16 B header + 4 B + 2 B in another
// sector, 3 = whole line (8 x 16 B), 4 = one aligned 32-byte sector, 5 = 4 B only
template <int RD, int WR>
__global__ __launch_bounds__(1024) void walk(unsigned char* pool, uint32_t nblocks, int steps, uint32_t* out) {
    const size_t lane_global = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned char* slab = pool + lane_global * (size_t)nblocks * 128;
    uint32_t state = (uint32_t)(lane_global * 2654435761u) + 12345u;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        const uint32_t idx = (state >> 8) % nblocks;
        u32x4* p = reinterpret_cast<u32x4*>(slab + (size_t)idx * 128);
        uint32_t h = 0;
        if (RD > 0) {
            u32x4 r[8];
#pragma unroll
            for (int i = 0; i < RD; i++) r[i] = p[(8 - RD) + i];
#pragma unroll
            for (int i = 0; i < RD; i++) h += r[i][0] ^ r[i][3];
        }
        acc += h;
        const u32x4 v = u32x4{h, state, acc, 1u};
        if (WR == 1 || WR == 2) p[0] = v;
        if (WR == 2) {
            reinterpret_cast<uint32_t*>(p)[16 + (state & 15u)] = h;
            reinterpret_cast<unsigned short*>(p)[36 + (state & 7u) * 2] = (unsigned short)state;
        }
        if (WR == 3) {
#pragma unroll
            for (int i = 0; i < 8; i++) p[i] = v;
        }
        if (WR == 4) { p[2] = v; p[3] = v; }
        if (WR == 5) reinterpret_cast<uint32_t*>(p)[state & 31u] = h;
        state = state * 1664525u + 1013904223u + (RD > 0 ? h : 0u);  // dependent next block when the visit reads
    }
    out[lane_global] = acc;
}

template <int RD, int WR>
static void run(const char* name, unsigned char* pool, uint32_t nb, uint32_t* out, size_t lanes) {
    const int steps = 1500;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((walk<RD, WR>), dim3((unsigned)(lanes / 1024)), dim3(1024), 0, 0, pool, nb, steps, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-58s %8.2f ms  %7.2f G line visits/s\n", name, best, (double)lanes * steps / best / 1e6);
    fflush(stdout);
}

int main() {
    const size_t lanes = 256 * 1024;
    const uint32_t nb = 1800;  // 230 KB per lane, 60 GB in total: nothing stays in L2 / Infinity Cache
    unsigned char* pool; uint32_t* out;
    if (hipMalloc(&pool, lanes * nb * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, lanes * 4);
    hipMemset(pool, 1, lanes * nb * 128);
    run<7, 0>("read 7 x 16 B (descent level)", pool, nb, out, lanes);
    run<8, 0>("read 8 x 16 B (whole line)", pool, nb, out, lanes);
    run<1, 0>("read 16 B (header only)", pool, nb, out, lanes);
    run<7, 1>("read 7 x 16 B + write 16 B header (same line)", pool, nb, out, lanes);
    run<7, 2>("read 7 x 16 B + write 16 B + 4 B + 2 B (two sectors)", pool, nb, out, lanes);
    run<7, 3>("read 7 x 16 B + write whole line", pool, nb, out, lanes);
    run<7, 5>("read 7 x 16 B + write 4 B", pool, nb, out, lanes);
    run<0, 1>("write 16 B (no read)", pool, nb, out, lanes);
    run<0, 4>("write one aligned 32-B sector (no read)", pool, nb, out, lanes);
    run<0, 3>("write whole 128-B line (no read)", pool, nb, out, lanes);
    run<0, 5>("write 4 B (no read)", pool, nb, out, lanes);
    run<0, 2>("write 16 B + 4 B + 2 B (no read)", pool, nb, out, lanes);
    return 0;
}

// Microbenchmark: per-lane dependent walks over 288-byte "children blocks" (9 x 32-byte records), the access pattern of the
// lane-per-tree descent. Variants of the cache policy / instruction shape; reports block visits per second.
// Build: hipcc -O3 --offload-arch=gfx950 -o gather_blocks gather_blocks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define LOAD18(POL)                                                                                                  \
    asm volatile(                                                                                                   \
        "global_load_dwordx4 %0, %18, off" POL "\n global_load_dwordx4 %1, %18, off offset:16" POL "\n"           \
        "global_load_dwordx4 %2, %18, off offset:32" POL "\n global_load_dwordx4 %3, %18, off offset:48" POL "\n"   \
        "global_load_dwordx4 %4, %18, off offset:64" POL "\n global_load_dwordx4 %5, %18, off offset:80" POL "\n"   \
        "global_load_dwordx4 %6, %18, off offset:96" POL "\n global_load_dwordx4 %7, %18, off offset:112" POL "\n"  \
        "global_load_dwordx4 %8, %18, off offset:128" POL "\n global_load_dwordx4 %9, %18, off offset:144" POL "\n" \
        "global_load_dwordx4 %10, %18, off offset:160" POL "\n global_load_dwordx4 %11, %18, off offset:176" POL "\n" \
        "global_load_dwordx4 %12, %18, off offset:192" POL "\n global_load_dwordx4 %13, %18, off offset:208" POL "\n" \
        "global_load_dwordx4 %14, %18, off offset:224" POL "\n global_load_dwordx4 %15, %18, off offset:240" POL "\n" \
        "global_load_dwordx4 %16, %18, off offset:256" POL "\n global_load_dwordx4 %17, %18, off offset:272" POL "\n" \
        "s_waitcnt vmcnt(0)"                                                                                        \
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),    \
          "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]),          \
          "=&v"(r[15]), "=&v"(r[16]), "=&v"(r[17])                                                                  \
        : "v"(p)                                                                                                    \
        : "memory")

// 9 loads of 16 B (select halves packed contiguously: 144 B)
#define LOAD9(POL)                                                                                                   \
    asm volatile(                                                                                                   \
        "global_load_dwordx4 %0, %9, off" POL "\n global_load_dwordx4 %1, %9, off offset:16" POL "\n"             \
        "global_load_dwordx4 %2, %9, off offset:32" POL "\n global_load_dwordx4 %3, %9, off offset:48" POL "\n"     \
        "global_load_dwordx4 %4, %9, off offset:64" POL "\n global_load_dwordx4 %5, %9, off offset:80" POL "\n"     \
        "global_load_dwordx4 %6, %9, off offset:96" POL "\n global_load_dwordx4 %7, %9, off offset:112" POL "\n"    \
        "global_load_dwordx4 %8, %9, off offset:128" POL "\n"                                                       \
        "s_waitcnt vmcnt(0)"                                                                                        \
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),    \
          "=&v"(r[8])                                                                                               \
        : "v"(p)                                                                                                    \
        : "memory")

#define LOAD7(POL)                                                                                                   \
    asm volatile(                                                                                                   \
        "global_load_dwordx4 %0, %7, off offset:16" POL "\n global_load_dwordx4 %1, %7, off offset:32" POL "\n"    \
        "global_load_dwordx4 %2, %7, off offset:48" POL "\n global_load_dwordx4 %3, %7, off offset:64" POL "\n"   \
        "global_load_dwordx4 %4, %7, off offset:80" POL "\n global_load_dwordx4 %5, %7, off offset:96" POL "\n"   \
        "global_load_dwordx4 %6, %7, off offset:112" POL "\n"                                                      \
        "s_waitcnt vmcnt(0)"                                                                                        \
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6])                  \
        : "v"(p)                                                                                                    \
        : "memory")

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(512) void walk(const unsigned char* pool, uint32_t nblocks_per_lane, int steps, uint32_t* out) {
    const size_t lane_global = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned char* slab = pool + lane_global * (size_t)nblocks_per_lane * 288;
    uint32_t state = (uint32_t)(lane_global * 2654435761u) + 12345u;
    uint32_t idx = (state >> 8) % nblocks_per_lane;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        const unsigned char* p = VARIANT == 8 ? slab + (size_t)idx * 128 : slab + (size_t)idx * 288;
        u32x4 r[18];
        for (int i = 0; i < 18; i++) r[i] = u32x4{0, 0, 0, 0};
        if (VARIANT == 0) LOAD18("");
        else if (VARIANT == 1) LOAD18(" sc1");
        else if (VARIANT == 2) LOAD18(" nt");
        else if (VARIANT == 3) LOAD18(" sc0 sc1");
        else if (VARIANT == 4) LOAD9("");
        else if (VARIANT == 5) LOAD9(" sc1");
        else if (VARIANT == 6) LOAD9(" nt");
        else if (VARIANT == 8) LOAD7("");
        else if (VARIANT == 7) {  // touch one dword per 128-byte line first, wait, then the 18 loads
            uint32_t t0, t1, t2;
            asm volatile("global_load_dword %0, %3, off\n global_load_dword %1, %3, off offset:128\n global_load_dword %2, %3, off offset:256\n s_waitcnt vmcnt(0)"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2) : "v"(p) : "memory");
            acc += t0 + t1 + t2;
            LOAD18("");
        }
        uint32_t h = 0;
        for (int i = 0; i < 18; i++) h += r[i][0] ^ r[i][3];
        acc += h;
        state = state * 1664525u + 1013904223u + h;  // dependent next block
        idx = (state >> 8) % nblocks_per_lane;
    }
    out[lane_global] = acc;
}

int main() {
    const int grid = 256, nt = 512;
    const uint32_t nb = 700;  // blocks per lane: 700 * 288 B = 201 KB per lane, 26 GB total
    const size_t lanes = (size_t)grid * nt;
    unsigned char* pool; uint32_t* out;
    if (hipMalloc(&pool, lanes * nb * 288) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, lanes * 4);
    hipMemset(pool, 1, lanes * nb * 288);
    const int steps = 2000;
    const char* names[] = {"18 x 16B plain", "18 x 16B sc1", "18 x 16B nt", "18 x 16B sc0 sc1", "9 x 16B plain", "9 x 16B sc1", "9 x 16B nt", "touch 3 lines, wait, 18 x 16B plain", "7 x 16B of a 128-B aligned block (one line)"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 9; v++) {
        float best = 1e30f;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            switch (v) {
                case 0: hipLaunchKernelGGL(walk<0>, dim3(grid), dim3(nt), 0, 0, pool, nb, steps, out); break;
                case 1: hipLaunchKernelGGL(walk<1>, dim3(grid), dim3(nt), 0, 0, pool, nb, steps, out); break;
                case 2: hipLaunchKernelGGL(walk<2>, dim3(grid), dim3(nt), 0, 0, pool, nb, steps, out); break;
                case 3: hipLaunchKernelGGL(walk<3>, dim3(grid), dim3(nt), 0, 0, pool, nb, steps, out); break;
                case 4: hipLaunchKernelGGL(walk<4>, dim3(grid), dim3(nt), 0, 0, pool, nb, steps, out); break;
                case 5: hipLaunchKernelGGL(walk<5>, dim3(grid), dim3(nt), 0, 0, pool, nb, steps, out); break;
                case 6: hipLaunchKernelGGL(walk<6>, dim3(grid), dim3(nt), 0, 0, pool, nb, steps, out); break;
                case 7: hipLaunchKernelGGL(walk<7>, dim3(grid), dim3(nt), 0, 0, pool, nb, steps, out); break;
                case 8: hipLaunchKernelGGL(walk<8>, dim3(grid), dim3(nt), 0, 0, pool, nb, steps, out); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double visits = (double)lanes * steps;
        printf("%-40s %8.2f ms  %7.2f G block-visits/s  (%.2f TB/s of 288-B blocks)\n", names[v], best, visits / best / 1e6,
               visits * 288 / best / 1e9);
    }
    return 0;
}

// Microbenchmark: do f32 MFMA (16x16x4) from one wave and plain VALU from another wave of the same SIMD overlap?
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap mfma_valu_overlap.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(512) void k(int mode, int iters, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    // mode bit0: waves 0-3 run MFMA; bit1: waves 4-7 run VALU; bit2: waves 4-7 run MFMA too; bit3: waves 0-3 run VALU
    const bool lo = wave < 4;
    const bool do_mfma = lo ? (mode & 1) : (mode & 4);
    const bool do_valu = lo ? (mode & 8) : (mode & 2);
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x0 = threadIdx.x, x1 = 1.0f, x2 = 2.0f, x3 = 3.0f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
    const float w = 1.0001f + threadIdx.x * 1e-7f, b = 0.5f;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    f32x16 c0 = {0}, c1 = c0;
    if (do_mfma && SHAPE == 1) {
        for (int i = 0; i < iters; i++) {  // 2 x 32x32x2 = 128 pipe cycles (16 passes each)
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, c1, 0, 0, 0);
        }
        a0[0] += c0[0] + c1[5];
    } else if (do_mfma) {
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, a3, 0, 0, 0);
        }
    } else if (do_valu) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                x0 = __builtin_fmaf(x0, w, b); x1 = __builtin_fmaf(x1, w, b); x2 = __builtin_fmaf(x2, w, b); x3 = __builtin_fmaf(x3, w, b);
                x4 = __builtin_fmaf(x4, w, b); x5 = __builtin_fmaf(x5, w, b); x6 = __builtin_fmaf(x6, w, b); x7 = __builtin_fmaf(x7, w, b);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main() {
    unsigned long long* d; float* s;
    hipMalloc(&d, 8 * 8); hipMalloc(&s, 512 * 4);
    const int iters = 20000;
    const char* names[] = {"", "MFMA on waves 0-3 only", "VALU on waves 4-7 only", "MFMA(0-3) + VALU(4-7)", "", "MFMA on all 8 waves", "", "", "", "", "VALU on all 8 waves"};
    for (int shape = 0; shape < 2; shape++)
    for (int mode : {1, 2, 3, 5, 10}) {
        if (shape == 1 && (mode == 2 || mode == 10)) continue;
        for (int rep = 0; rep < 2; rep++) {
            if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(512), 0, 0, mode, iters, d, s);
            else hipLaunchKernelGGL(k<1>, dim3(1), dim3(512), 0, 0, mode, iters, d, s);
            hipDeviceSynchronize();
        }
        if (shape == 1) printf("[32x32x2] ");
        unsigned long long h[8];
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("mode %2d %-28s cycles/iter: wave0 %.1f  wave4 %.1f   (iter = 4 MFMA 16x16x4 f32 = 128 pipe cycles | 32 v_fma = 128 issue cycles)\n",
               mode, names[mode], (double)h[0] / iters, (double)h[4] / iters);
    }
    return 0;
}

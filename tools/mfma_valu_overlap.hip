// Micro-benchmark: can fp32 VALU work be issued in the shadow of v_mfma_f32_32x32x16_f16 (8 passes = 32 cycles)?
// Each iteration issues 12 independent-accumulator MFMAs with NV independent v_fma_f32 after each one, pinned in
// program order; 1 or 2 waves per SIMD.  Time per iteration vs NV tells how many VALU ops fit behind one MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NV>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a = {1, 0.5, 0.25, 2, 1, 0.5, 0.25, 2}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = seed + j + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                x[j & 7] = __builtin_fmaf(x[j & 7], 1.0001f, 0.5f);
                asm volatile("" : "+v"(x[j & 7]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
void run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs = 1; wgs <= 2; ++wgs) {
        const int grid = 256 * wgs * 4;
        hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // cycles per MFMA slot per SIMD at 2.4 GHz nominal: waves per SIMD = wgs (4 sequential workgroups per CU slot)
        const double mfma_per_simd = 4.0 * wgs * (double)iters * 12;     // 4 rounds x wgs waves
        printf("VALU per MFMA = %2d, %d wave(s)/SIMD: %.2f ms  -> %.1f ns per MFMA slot per SIMD (pure MFMA = 13.3 ns at 2.4 GHz)\n",
               NV, wgs, ms, ms * 1e6 / mfma_per_simd);
    }
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 20000;
    run<0>(out, iters); run<2>(out, iters); run<4>(out, iters); run<6>(out, iters); run<8>(out, iters); run<12>(out, iters); run<16>(out, iters);
    return 0;
}

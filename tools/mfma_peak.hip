// Micro-benchmark: peak v_mfma_f32_32x32x2_f32 rate on this chip/power state, 1..2 waves per SIMD,
// optionally with a dwordx4 global-load stream and ds_read_b128 traffic beside the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const f32x4* __restrict__ w, float* out, int iters) {
    __shared__ f32x4 lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 a0 = {1.f, 0.5f, 0.25f, 2.f}, a1 = a0, b0 = {1.f, 1.f, 1.f, 1.f}, b1 = b0;
    const f32x4* wp = w + (threadIdx.x >> 6) * 4096 + lane;
    f32x4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 4) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
        if (MODE & 1) { na0 = wp[(it & 31) * 128]; na1 = wp[(it & 31) * 128 + 64]; }
        if (MODE & 2) { nb0 = lds[(lane + it) & 1023]; nb1 = lds[(lane + it + 512) & 1023]; }
        if (!(MODE & 4)) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    f32x4* w; float* out;
    hipMalloc(&w, 4 * 4096 * 16 + 65536); hipMemset(w, 0, 4 * 4096 * 16 + 65536);
    hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 8; ++mode) for (int wgs = 1; wgs <= 2; ++wgs) {
        const int grid = 256 * wgs;
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            if (mode == 4) return;
        };
        if (mode == 4) continue;
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 4 * iters * 16 * 4096.0;
        printf("mode=%d (global=%d lds=%d pipelined=%d) wg/CU=%d: %.2f ms  %.1f TFLOP/s\n", mode, mode & 1, (mode >> 1) & 1, (mode >> 2) & 1, wgs, ms, flop / ms / 1e9);
    }
    return 0;
}

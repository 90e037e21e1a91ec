// Micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate on this chip / power state with the operand delivery
// pattern of one k-step of the split kernels: 12 MFMAs fed by 8 global_load_dwordx4 (2 N-tiles x hi/lo, next
// k-step) and 4 ds_read_b128 (2 M-tiles x hi/lo), 1 or 2 workgroups (of 4 waves) per CU, plus a long run that
// reports the steady-state rate (clock management settles after ~100 ms).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>   // bit0: global weight stream, bit1: LDS activation reads
__global__ __launch_bounds__(256, 2) void k(const h8* __restrict__ w, float* out, int iters) {
    __shared__ h8 lds[2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = h8{1, 2, 3, 4, 1, 2, 3, 4};
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][r] = 0.f;
    h8 ah[2][2], al[2][2], bh[2], bl[2];
    for (int i = 0; i < 2; ++i) {
        bh[i] = h8{1, 1, 1, 1, 1, 1, 1, 1}; bl[i] = bh[i];
        for (int j = 0; j < 2; ++j) { ah[j][i] = h8{1, 0.5, 0.25, 2, 1, 0.5, 0.25, 2}; al[j][i] = ah[j][i]; }
    }
    const h8* wp = w + (threadIdx.x >> 6) * 8192 + lane;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (MODE & 1) {
                const h8* p = wp + ((it + u) & 31) * 256;
                ah[(u + 1) & 1][0] = p[0]; al[(u + 1) & 1][0] = p[64]; ah[(u + 1) & 1][1] = p[128]; al[(u + 1) & 1][1] = p[192];
            }
            if (MODE & 2) {
                bh[0] = lds[(lane + it + u) & 2047]; bl[0] = lds[(lane + it + u + 512) & 2047];
                bh[1] = lds[(lane + it + u + 1024) & 2047]; bl[1] = lds[(lane + it + u + 1536) & 2047];
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u][nt], bh[mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u][nt], bl[mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u][nt], bh[mt], acc[nt][mt], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i >> 1][i & 1][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// pattern B: one N-tile x four M-tiles per wave (128-row tiles, 8 waves, one workgroup per CU): per k-step
// 2 weight loads, 8 ds_read_b128, 12 MFMAs
template <int MODE>
__global__ __launch_bounds__(512, 2) void k2(const h8* __restrict__ w, float* out, int iters) {
    __shared__ h8 lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = h8{1, 2, 3, 4, 1, 2, 3, 4};
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 ah[2], al[2], bh[4], bl[4];
    for (int i = 0; i < 4; ++i) { bh[i] = h8{1, 1, 1, 1, 1, 1, 1, 1}; bl[i] = bh[i]; }
    for (int j = 0; j < 2; ++j) { ah[j] = h8{1, 0.5, 0.25, 2, 1, 0.5, 0.25, 2}; al[j] = ah[j]; }
    const h8* wp = w + (threadIdx.x >> 6) * 4096 + lane;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (MODE & 1) {
                const h8* p = wp + ((it + u) & 31) * 128;
                ah[(u + 1) & 1] = p[0]; al[(u + 1) & 1] = p[64];
            }
            if (MODE & 2) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    bh[m] = lds[(lane + it + u + 1024 * m) & 4095]; bl[m] = lds[(lane + it + u + 1024 * m + 512) & 4095];
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u], bh[mt], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], bl[mt], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], bh[mt], acc[mt], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 40000;
    h8* w; float* out;
    hipMalloc(&w, 4 * 8192 * 16 + 65536 * 16); hipMemset(w, 0, 4 * 8192 * 16 + 65536 * 16);
    hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) for (int wgs = 1; wgs <= 2; ++wgs) {
        const int grid = 256 * wgs * (rep ? 8 : 1);    // rep 1: 8x longer launches (steady-state clocks)
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, w, out, iters);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
        printf("%s mode=%d (global=%d lds=%d) wg/CU=%d: %.2f ms  %.0f executed TFLOP/s = %.1f algorithmic (x1/3)\n",
               rep ? "long " : "short", mode, mode & 1, (mode >> 1) & 1, wgs, ms, flop / ms / 1e9, flop / ms / 1e9 / 3.0);
    }
    for (int mode = 0; mode < 4; ++mode) {
        const int grid = 256 * 8;
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k2<0>, dim3(grid), dim3(512), 0, 0, w, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k2<1>, dim3(grid), dim3(512), 0, 0, w, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k2<2>, dim3(grid), dim3(512), 0, 0, w, out, iters);
            if (mode == 3) hipLaunchKernelGGL(k2<3>, dim3(grid), dim3(512), 0, 0, w, out, iters);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 8 * iters * 12 * 2.0 * 32 * 32 * 16;
        printf("long  pattern B (1 N-tile x 4 M-tiles, 8 waves/WG, 1 WG/CU) mode=%d (global=%d lds=%d): %.2f ms  %.0f executed TFLOP/s = %.1f algorithmic\n",
               mode, mode & 1, (mode >> 1) & 1, ms, flop / ms / 1e9, flop / ms / 1e9 / 3.0);
    }
    return 0;
}

// Micro-benchmark: what does one 16-B-per-lane vector load cost the CU's vector-memory path (TA / L1) on gfx950,
// depending on how many lanes really fetch?  Decides whether zero-weight bilinear taps (half of the NeO-360
// evaluator's gathers for points outside the unit sphere) can be made cheap WITHOUT control flow around the loads
// (control flow would cost the precise vmcnt bookkeeping of the software pipeline).
//   hipcc --offload-arch=gfx950 -O3 tools/ta_cost.hip -o ta_cost && ./ta_cost
// Variants (8 independent loads per iteration, 8 waves per CU, data L2-resident: 8 MB window shared by the chip):
//   0 global_load_dwordx4, 64 lanes x 16 B contiguous (a weight fragment)
//   1 global gather: 4 rows x 256 B at pseudo-random 1-KB texels (a tap item)
//   2 global: all four rows read the SAME 256 B (today's zero-weight taps: offset 0)
//   3 raw buffer load, every lane out of range (returns 0, no fetch)
//   4 raw buffer load, one row of four in range
//   5 raw buffer gather, all in range (= variant 1 through the buffer path)
//   6 global gather under a divergent `if` with one row of four active
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));


template <int V>
__global__ __launch_bounds__(256) void k(const float* __restrict__ base, uint32_t window, int iters, float* out) {
    const int lane = threadIdx.x & 63, row = lane >> 4, col = lane & 15;
    uint32_t seed = blockIdx.x * 2654435761u + (threadIdx.x >> 6) * 40503u;
    f32x4 acc = {0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)window, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            seed = seed * 1664525u + 1013904223u;
            const uint32_t texel = ((seed >> 8) + row * 977u) % (window / 1024u);
            uint32_t off;
            if (V == 0) off = ((seed >> 8) % (window / 1024u)) * 1024u + lane * 16u;
            else if (V == 2) off = col * 16u;
            else off = texel * 1024u + col * 16u;
            if (V == 3) off = 0xFFFFFF00u;
            if (V == 4) off = row == (int)(seed & 3u) ? off : 0xFFFFFF00u;
            if (V <= 2) v[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
            else if (V <= 5) v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
            else {
                v[u] = f32x4{0, 0, 0, 0};
                if (row == (int)(seed & 3u)) v[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <int V>
void run(const float* d, uint32_t window, float* out, const char* name) {
    const int iters = 2000, blocks = 256 * 2;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, window, 50, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, window, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double loads_per_cu = (double)blocks / 256.0 * 4 * iters * 8;     // wave-level load instructions per CU
    printf("variant %d %-58s %8.3f ms  %6.1f ns per wave-load per CU (= %5.1f cycles at 2.4 GHz)\n", V, name, ms,
           ms * 1e6 / loads_per_cu, ms * 1e6 / loads_per_cu * 2.4);
}

int main() {
    const uint32_t window = 8u << 20;
    float *d, *out;
    hipMalloc(&d, window + 4096);
    hipMemset(d, 0, window + 4096);
    hipMalloc(&out, 16);
    run<0>(d, window, out, "global 1 KB contiguous");
    run<1>(d, window, out, "global gather 4 rows x 256 B");
    run<2>(d, window, out, "global, all rows the same 256 B");
    run<3>(d, window, out, "buffer, all lanes out of range");
    run<4>(d, window, out, "buffer, 1 row of 4 in range");
    run<5>(d, window, out, "buffer gather, all in range");
    run<6>(d, window, out, "global gather under divergent if, 1 row of 4");
    return 0;
}

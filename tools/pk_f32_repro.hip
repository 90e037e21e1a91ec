// Stand-alone reproducer for the packed-fp32 glitch of round 1 (profiles/r01_tp_h_race_bisect.log), found in round 2
// (profiles/r02_pk_f32_repro.log): v_pk_fma_f32 / v_pk_mul_f32 with an op_sel half-broadcast, executed while another
// wave of the same SIMD runs v_mfma_f32_32x32x16_f16, occasionally returns wrong values in lanes 48-63.
//
// This program has no dependency on the library.  One kernel template reproduces the instruction mix of the blend
// phase: per iteration 4 x global_load_dwordx4 taps (pseudo-random 16-B pieces of an L2-resident table), one
// ds_read_b128 of blend weights, 12 fp16 MFMAs on private accumulators, and the 4-tap blend of 4 channels written
//   BLEND 0: scalar v_fma_f32                       (what build.py's -packed-fp32-ops forces)
//   BLEND 1: v_pk_fma_f32, weights splatted into register pairs (no op_sel)
//   BLEND 2: v_pk_fma_f32 with op_sel / op_sel_hi broadcasting one half of the weight pair (the compiler's form)
//   BLEND 3..10: BLEND 2 plus an unrelated second instruction in the same asm statement; they only differ in the
//            schedule the compiler produces.  BLEND 4's schedule FAILS with 2 workgroups per CU; the others pass.
// All are the same IEEE fma chain, so every output must be bitwise identical across BLEND, across launches and
// across occupancy (1 or 2 workgroups per CU, selected by the dynamic LDS size).  The host prints, per configuration,
// the number of differing outputs vs the scalar single-workgroup run and a per-16-lane histogram.
// tools/pk_f32_variants.py perturbs the failing kernel's ISA one ingredient at a time (tools/pk_f32_co_run.cpp runs them).
//   hipcc --offload-arch=gfx950 -O3 -o pk_f32_repro tools/pk_f32_repro.hip && ./pk_f32_repro
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
    f32x2 d;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// both result lanes use the LOW half of w
__device__ __forceinline__ f32x2 pk_fma_lo(f32x2 a, f32x2 w, f32x2 c) {
    f32x2 d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(w), "v"(c));
    return d;
}
// both result lanes use the HIGH half of w
__device__ __forceinline__ f32x2 pk_fma_hi(f32x2 a, f32x2 w, f32x2 c) {
    f32x2 d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(w), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_mul_lo(f32x2 a, f32x2 w) {
    f32x2 d;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(w));
    return d;
}

// BLEND >= 3: a packed fma followed IMMEDIATELY by an instruction that overwrites one of its source pairs
// (write-after-read), as the compiler schedules them in the real kernels (pillar.hip's packed build has
// v_pk_fma_f32 v[46:47], v[86:87], v[42:43], v[46:47] op_sel_hi:[1,0,1] followed by v_cvt_pk_f16_f32 v42, ...).
// The overwritten copy is private to the statement and dead afterwards, so the result must not change.
//   3: op_sel forms, next instruction overwrites the WEIGHT pair (src1)      4: ... overwrites the TAP pair (src0)
//   5: plain v_pk_fma_f32 (splatted weights, no op_sel), overwrites the tap pair
//   6: as 4 with s_nop 0 between the two          7: as 4 with s_nop 1
//   8: as 4, overwriting with v_lshlrev_b64 (a non-packed 64-bit VALU write)  9: as 4, overwriting with v_pk_mul_f32
#define PK_LO "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]"
#define PK_HI "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
#define PK_PLAIN "v_pk_fma_f32 %0, %1, %2, %3"
#define WAR_FN(name, head, tail, WHICH)                                                                   \
    __device__ __forceinline__ f32x2 name(f32x2 a, f32x2 w, f32x2 c, f32x2 junk) {                         \
        f32x2 d, ac = a, wc = w;                                                                           \
        if (WHICH == 0) {                                                                                  \
            asm volatile(head "\n\t" tail : "=&v"(d), "+v"(ac), "+v"(wc) : "v"(c), "v"(junk));             \
        } else {                                                                                           \
            asm volatile(head "\n\t" tail : "=&v"(d), "+v"(ac), "+v"(wc) : "v"(c), "v"(junk));             \
        }                                                                                                  \
        asm volatile("" ::"v"(ac), "v"(wc));                                                               \
        return d;                                                                                          \
    }
WAR_FN(war3_lo, PK_LO, "v_pk_mov_b32 %2, %4, %4", 1)
WAR_FN(war3_hi, PK_HI, "v_pk_mov_b32 %2, %4, %4", 1)
WAR_FN(war10_lo, PK_LO, "v_pk_mov_b32 %1, %4, %4", 0)
WAR_FN(war10_hi, PK_HI, "v_pk_mov_b32 %1, %4, %4", 0)
// 4 is the form that FAILS (r02 log): the weight pair is a plain input, so the packed fma reads the ds_read_b128
// destination directly; in the WAR_FN family the compiler first copies the weights with v_mov_b64 and all variants pass.
__device__ __forceinline__ f32x2 war4_lo(f32x2 a, f32x2 w, f32x2 c, f32x2 junk) {
    f32x2 d, ac = a;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]\n\tv_pk_mov_b32 %1, %4, %4" : "=&v"(d), "+v"(ac) : "v"(w), "v"(c), "v"(junk));
    asm volatile("" :: "v"(ac));
    return d;
}
__device__ __forceinline__ f32x2 war4_hi(f32x2 a, f32x2 w, f32x2 c, f32x2 junk) {
    f32x2 d, ac = a;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\tv_pk_mov_b32 %1, %4, %4" : "=&v"(d), "+v"(ac) : "v"(w), "v"(c), "v"(junk));
    asm volatile("" :: "v"(ac));
    return d;
}
WAR_FN(war5, PK_PLAIN, "v_pk_mov_b32 %1, %4, %4", 0)
WAR_FN(war6_lo, PK_LO, "s_nop 0\n\tv_pk_mov_b32 %1, %4, %4", 0)
WAR_FN(war6_hi, PK_HI, "s_nop 0\n\tv_pk_mov_b32 %1, %4, %4", 0)
WAR_FN(war7_lo, PK_LO, "s_nop 1\n\tv_pk_mov_b32 %1, %4, %4", 0)
WAR_FN(war7_hi, PK_HI, "s_nop 1\n\tv_pk_mov_b32 %1, %4, %4", 0)
WAR_FN(war8_lo, PK_LO, "v_lshlrev_b64 %1, 0, %4", 0)
WAR_FN(war8_hi, PK_HI, "v_lshlrev_b64 %1, 0, %4", 0)
WAR_FN(war9_lo, PK_LO, "v_pk_mul_f32 %1, %4, %4", 0)
WAR_FN(war9_hi, PK_HI, "v_pk_mul_f32 %1, %4, %4", 0)

template <int BLEND>
__device__ __forceinline__ f32x4 blend(const f32x4 (&tap)[4], const f32x4 w) {
    f32x4 v;
    if (BLEND == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = tap[0][e] * w[0];
            a = __builtin_fmaf(tap[1][e], w[1], a);
            a = __builtin_fmaf(tap[2][e], w[2], a);
            a = __builtin_fmaf(tap[3][e], w[3], a);
            v[e] = a;
        }
    } else if (BLEND == 1) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f32x2 a = pk_mul(f32x2{tap[0][2 * p], tap[0][2 * p + 1]}, f32x2{w[0], w[0]});
            a = pk_fma(f32x2{tap[1][2 * p], tap[1][2 * p + 1]}, f32x2{w[1], w[1]}, a);
            a = pk_fma(f32x2{tap[2][2 * p], tap[2][2 * p + 1]}, f32x2{w[2], w[2]}, a);
            a = pk_fma(f32x2{tap[3][2 * p], tap[3][2 * p + 1]}, f32x2{w[3], w[3]}, a);
            v[2 * p] = a[0];
            v[2 * p + 1] = a[1];
        }
    } else if (BLEND >= 3) {
        const f32x2 w01{w[0], w[1]}, w23{w[2], w[3]};
        const f32x2 junk{tap[0][3] * 7.0f, tap[1][2] - 3.0f};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x2 t1{tap[1][2 * p], tap[1][2 * p + 1]}, t2{tap[2][2 * p], tap[2][2 * p + 1]}, t3{tap[3][2 * p], tap[3][2 * p + 1]};
            f32x2 a = BLEND == 5 ? pk_mul(f32x2{tap[0][2 * p], tap[0][2 * p + 1]}, f32x2{w[0], w[0]})
                                 : pk_mul_lo(f32x2{tap[0][2 * p], tap[0][2 * p + 1]}, w01);
            if (BLEND == 3) { a = war3_hi(t1, w01, a, junk); a = war3_lo(t2, w23, a, junk); a = war3_hi(t3, w23, a, junk); }
            if (BLEND == 4) { a = war4_hi(t1, w01, a, junk); a = war4_lo(t2, w23, a, junk); a = war4_hi(t3, w23, a, junk); }
            if (BLEND == 5) { a = war5(t1, f32x2{w[1], w[1]}, a, junk); a = war5(t2, f32x2{w[2], w[2]}, a, junk); a = war5(t3, f32x2{w[3], w[3]}, a, junk); }
            if (BLEND == 6) { a = war6_hi(t1, w01, a, junk); a = war6_lo(t2, w23, a, junk); a = war6_hi(t3, w23, a, junk); }
            if (BLEND == 7) { a = war7_hi(t1, w01, a, junk); a = war7_lo(t2, w23, a, junk); a = war7_hi(t3, w23, a, junk); }
            if (BLEND == 8) { a = war8_hi(t1, w01, a, junk); a = war8_lo(t2, w23, a, junk); a = war8_hi(t3, w23, a, junk); }
            if (BLEND == 10) { a = war10_hi(t1, w01, a, junk); a = war10_lo(t2, w23, a, junk); a = war10_hi(t3, w23, a, junk); }
            if (BLEND == 9) { a = war9_hi(t1, w01, a, junk); a = war9_lo(t2, w23, a, junk); a = war9_hi(t3, w23, a, junk); }
            v[2 * p] = a[0];
            v[2 * p + 1] = a[1];
        }
    } else {
        const f32x2 w01{w[0], w[1]}, w23{w[2], w[3]};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f32x2 a = pk_mul_lo(f32x2{tap[0][2 * p], tap[0][2 * p + 1]}, w01);
            a = pk_fma_hi(f32x2{tap[1][2 * p], tap[1][2 * p + 1]}, w01, a);
            a = pk_fma_lo(f32x2{tap[2][2 * p], tap[2][2 * p + 1]}, w23, a);
            a = pk_fma_hi(f32x2{tap[3][2 * p], tap[3][2 * p + 1]}, w23, a);
            v[2 * p] = a[0];
            v[2 * p + 1] = a[1];
        }
    }
    return v;
}

// table: n16 pieces of 16 B; weights: 64 x f32x4 per workgroup in LDS; out: one f32x4 per thread and iteration block
template <int BLEND, bool MFMA>
__global__ __launch_bounds__(256, 2) void k_mix(const f32x4* __restrict__ table, unsigned n16, int iters,
                                                 f32x4* __restrict__ out, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    f32x4* wl = reinterpret_cast<f32x4*>(smem);
    if (tid < 64) {
        // bilinear-style weights: positive, sum to 1, differ per row
        const float fx = (float)((tid * 37 + blockIdx.x * 11) & 255) / 256.0f, fy = (float)((tid * 91 + 5) & 255) / 256.0f;
        wl[tid] = f32x4{(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy};
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    h8 ha, hb;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(0.001f * (float)((lane + e) & 15)); hb[e] = (_Float16)(0.002f * (float)((lane * 3 + e) & 7)); }
    f32x4 sum{0.f, 0.f, 0.f, 0.f};
    unsigned s = (blockIdx.x * 256u + tid) * 2654435761u + 12345u;
    const int row = tid >> 2;                    // 4 threads share a row of weights, as 16 lanes share one in the kernels
    for (int it = 0; it < iters; ++it) {
        f32x4 tap[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s = s * 1664525u + 1013904223u;
            tap[k] = table[(s >> 8) % n16];
        }
        const f32x4 w = wl[(row + it) & 63];
        if (MFMA) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, ha, acc[a], 0, 0, 0);
            }
        }
        const f32x4 v = blend<BLEND>(tap, w);
        sum = sum + v;
    }
    out[blockIdx.x * 256 + tid] = sum;
    if (MFMA) {
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) t += acc[a][0] + acc[a][7];
        if (t == 12345.678f) sink[0] = t;       // keeps the matrix stream alive
    }
}

template <int BLEND, bool MFMA>
static void run(const f32x4* table, unsigned n16, int iters, int blocks, size_t lds, f32x4* out, float* sink) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mix<BLEND, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_mix<BLEND, MFMA>), dim3(blocks), dim3(256), lds, 0, table, n16, iters, out, sink);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400, blocks = argc > 2 ? atoi(argv[2]) : 4096, reps = argc > 3 ? atoi(argv[3]) : 4;
    const unsigned n16 = 1u << 20;                         // 16 MB table: L2 / MALL resident
    std::vector<float> h(n16 * 4);
    unsigned s = 777u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.3f; }
    f32x4* table; f32x4* out; float* sink;
    CHECK(hipMalloc(&table, n16 * 16));
    CHECK(hipMemcpy(table, h.data(), n16 * 16, hipMemcpyHostToDevice));
    const size_t n_out = (size_t)blocks * 256;
    CHECK(hipMalloc(&out, n_out * 16));
    CHECK(hipMalloc(&sink, 4));
    std::vector<float> ref(n_out * 4), got(n_out * 4);
    const size_t lds1 = 100 * 1024, lds2 = 60 * 1024;     // one / two workgroups per CU
    run<0, false>(table, n16, iters, blocks, lds1, out, sink);
    CHECK(hipMemcpy(ref.data(), out, n_out * 16, hipMemcpyDeviceToHost));
    printf("reference: scalar fma, no MFMA stream, 1 workgroup/CU; %d iterations x %d workgroups\n", iters, blocks);
    auto compare = [&](const char* name) {
        CHECK(hipMemcpy(got.data(), out, n_out * 16, hipMemcpyDeviceToHost));
        size_t bad = 0, hist[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < n_out; ++i)
            if (memcmp(&got[i * 4], &ref[i * 4], 16) != 0) { ++bad; ++hist[(i & 63) >> 4]; }
        printf("  %-44s differing threads %8zu of %zu   by lane group [0-15 16-31 32-47 48-63] = %zu %zu %zu %zu\n", name, bad,
               n_out, hist[0], hist[1], hist[2], hist[3]);
        return bad;
    };
    size_t total = 0;
    for (int r = 0; r < reps; ++r) {
        printf("repetition %d\n", r);
        run<0, true>(table, n16, iters, blocks, lds1, out, sink);  total += compare("scalar fma   + MFMA, 1 WG/CU");
        run<0, true>(table, n16, iters, blocks, lds2, out, sink);  total += compare("scalar fma   + MFMA, 2 WG/CU");
        run<1, true>(table, n16, iters, blocks, lds1, out, sink);  total += compare("pk splat     + MFMA, 1 WG/CU");
        run<1, true>(table, n16, iters, blocks, lds2, out, sink);  total += compare("pk splat     + MFMA, 2 WG/CU");
        run<2, false>(table, n16, iters, blocks, lds2, out, sink); total += compare("pk op_sel, no MFMA, 2 WG/CU");
        run<2, true>(table, n16, iters, blocks, lds1, out, sink);  total += compare("pk op_sel    + MFMA, 1 WG/CU");
        run<2, true>(table, n16, iters, blocks, lds2, out, sink);  total += compare("pk op_sel    + MFMA, 2 WG/CU");
        run<3, true>(table, n16, iters, blocks, lds2, out, sink);  total += compare("op_sel, WAR weights  + MFMA, 2 WG/CU");
        run<4, true>(table, n16, iters, blocks, lds2, out, sink);  total += compare("op_sel, WAR taps     + MFMA, 2 WG/CU");
        run<4, true>(table, n16, iters, blocks, lds1, out, sink);  total += compare("op_sel, WAR taps     + MFMA, 1 WG/CU");
        run<4, false>(table, n16, iters, blocks, lds2, out, sink); total += compare("op_sel, WAR taps, no MFMA, 2 WG/CU");
        run<10, true>(table, n16, iters, blocks, lds2, out, sink); total += compare("op_sel, WAR taps, weights copied + MFMA, 2 WG/CU");
        run<5, true>(table, n16, iters, blocks, lds2, out, sink);  total += compare("plain pk, WAR taps   + MFMA, 2 WG/CU");
        run<6, true>(table, n16, iters, blocks, lds2, out, sink);  total += compare("op_sel, s_nop 0, WAR taps + MFMA, 2 WG/CU");
        run<7, true>(table, n16, iters, blocks, lds2, out, sink);  total += compare("op_sel, s_nop 1, WAR taps + MFMA, 2 WG/CU");
        run<8, true>(table, n16, iters, blocks, lds2, out, sink);  total += compare("op_sel, WAR taps by lshlrev_b64 + MFMA, 2 WG/CU");
        run<9, true>(table, n16, iters, blocks, lds2, out, sink);  total += compare("op_sel, WAR taps by pk_mul + MFMA, 2 WG/CU");
    }
    printf("total differing outputs: %zu (0 = this instruction mix is bitwise stable on this GPU)\n", total);
    return 0;
}

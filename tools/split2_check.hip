// GPU check: split2 (v_cvt_pk_f16_f32 + v_fma_mix_f32) == the scalar split, bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)__builtin_fmaf((float)hi, -1.0f, x);
}
__device__ __forceinline__ void split2(float x0, float x1, h2& hi, h2& lo) {
    const f32x2 v = {x0, x1};
    hi = __builtin_convertvector(v, h2);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
    const f32x2 r = {r0, r1};
    lo = __builtin_convertvector(r, h2);
}
// round 5: the lo pair by v_fma_mixlo_f16 + v_fma_mixhi_f16 (3 instructions per pair of values; build with -DMIXLO=1)
__device__ __forceinline__ void split2_mixlo(float x0, float x1, h2& hi, h2& lo) {
    const f32x2 v = {x0, x1};
    hi = __builtin_convertvector(v, h2);
    unsigned packed;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(packed) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(packed) : "v"(hi), "v"(x1));
    lo = __builtin_bit_cast(h2, packed);
}
#ifdef MIXLO
#define split2 split2_mixlo
#endif
__global__ void k(const float* x, unsigned short* a, unsigned short* b, int n) {
    int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    _Float16 h0, l0, h1, l1;
    split(x[i], h0, l0); split(x[i + 1], h1, l1);
    h2 H, Lo;
    split2(x[i], x[i + 1], H, Lo);
    a[2 * i] = __builtin_bit_cast(unsigned short, h0); a[2 * i + 1] = __builtin_bit_cast(unsigned short, l0);
    a[2 * i + 2] = __builtin_bit_cast(unsigned short, h1); a[2 * i + 3] = __builtin_bit_cast(unsigned short, l1);
    _Float16 t;
    t = H[0]; b[2 * i] = __builtin_bit_cast(unsigned short, t); t = Lo[0]; b[2 * i + 1] = __builtin_bit_cast(unsigned short, t);
    t = H[1]; b[2 * i + 2] = __builtin_bit_cast(unsigned short, t); t = Lo[1]; b[2 * i + 3] = __builtin_bit_cast(unsigned short, t);
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n);
    unsigned s = 12345;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; float u = (s >> 8) / 16777216.0f; s = s * 1664525u + 1013904223u; int e = (int)((s >> 8) % 40) - 30; h[i] = ldexpf(u * 2 - 1, e); }
    h[0] = 0.f; h[1] = -0.f; h[2] = 65504.f; h[3] = 1e-8f; h[4] = 6e-8f; h[5] = -3.3e-5f;
    float* dx; unsigned short *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, da, db, n);
    std::vector<unsigned short> A(2 * n), B(2 * n);
    hipMemcpy(A.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(B.data(), db, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2 * n; ++i) if (A[i] != B[i]) { if (bad < 8) printf("mismatch at value %d (%s): x = %g scalar %04x pair %04x\n", i / 2, i & 1 ? "lo" : "hi", h[i / 2], A[i], B[i]); ++bad; }
    printf("split2 vs split: %d mismatching halves of %d\n", bad, 2 * n);
    return bad != 0;
}

// Shared device helpers for the gfx950 kernels of libneo360_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace neo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;
constexpr float HALF_PI_F32 = 1.57079637050628662109375f;  // fl32(0.5*pi), the reference's phase

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- wave-level scans / reductions (64 lanes) -------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <class Op>
__device__ __forceinline__ float wave_inclusive_scan(float v, Op op) {
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float up = __shfl_up(v, o, 64);
        if (lane >= o) v = op(up, v);
    }
    return v;
}

struct OpMul { __device__ float operator()(float a, float b) const { return a * b; } };
struct OpAdd { __device__ float operator()(float a, float b) const { return a + b; } };
struct OpMax { __device__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct OpMin { __device__ float operator()(float a, float b) const { return fminf(a, b); } };

// ---- activations (torch semantics) --------------------------------------------
// torch.nn.Softplus(beta=1, threshold=20) applied to raw-1
// (vanilla_nerf/model.py:203-204, neo360/model.py:380-381).
__device__ __forceinline__ float density_act(float raw) {
    const float x = raw + (-1.0f);
    return x > 20.0f ? x : log1pf(expf(x));
}
// sigmoid(raw)*(1+2*0.001) - 0.001 (vanilla_nerf/model.py:198-200).
__device__ __forceinline__ float colour_act(float raw) {
    const float s = 1.0f / (1.0f + expf(-raw));
    return s * 1.002f - 0.001f;
}

// torch.nan_to_num(x, nan): nan -> `nan_value`, +inf -> FLT_MAX, -inf -> -FLT_MAX.
__device__ __forceinline__ float nan_to_num(float x, float nan_value) {
    if (x != x) return nan_value;
    if (x == __builtin_inff()) return 3.40282346638528859812e38f;
    if (x == -__builtin_inff()) return -3.40282346638528859812e38f;
    return x;
}

// Positional encoding of one scalar at one octave: sin(x*2^k), sin(x*2^k + fl32(pi/2)).
// x*2^k is exact; the phase add rounds in fp32 exactly as the reference's does
// (neo360/helper.py:123-124).  sinf is OCML's <=1ulp implementation with full
// range reduction (arguments reach 2^9*|x|).
__device__ __forceinline__ void enc_pair(float x, int k, float& s, float& c) {
    const float a = ldexpf(x, k);
    s = sinf(a);
    c = sinf(a + HALF_PI_F32);
}

// Unit-sphere exit depth of one ray (models/neo360/helper.py:253-273):
// d1 = -(d.o)/(d.d), far = d1 + sqrt(1-|o+d1 d|^2)/|d|.  Returns the
// reference's assertion predicate (1-|p|^2 >= 0).
__device__ __forceinline__ bool sphere_exit(const float* o, const float* d, float& far) {
    const float dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const float d1 = -(d[0] * o[0] + d[1] * o[1] + d[2] * o[2]) / dd;
    const float p0 = o[0] + d1 * d[0], p1 = o[1] + d1 * d[1], p2 = o[2] + d1 * d[2];
    const float inv_len = 1.0f / sqrtf(dd);
    const float margin = 1.0f - (p0 * p0 + p1 * p1 + p2 * p2);
    far = d1 + sqrtf(margin) * inv_len;
    return margin >= 0.0f;
}

}  // namespace neo

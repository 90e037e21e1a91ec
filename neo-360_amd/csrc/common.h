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

// torch.sigmoid (PixelNeRF baseline's rgb activation, vanilla_nerf/model_pixel.py:165, :231)
__device__ __forceinline__ float sigmoid_act(float raw) { return 1.0f / (1.0f + expf(-raw)); }

// torch.nan_to_num(x, nan): nan -> `nan_value`, +inf -> FLT_MAX, -inf -> -FLT_MAX.
__device__ __forceinline__ float nan_to_num(float x, float nan_value) {
    if (x != x) return nan_value;
    if (x == __builtin_inff()) return 3.40282346638528859812e38f;
    if (x == -__builtin_inff()) return -3.40282346638528859812e38f;
    return x;
}

// sin(x) for the encodings.  |x| <= 65536 (every argument a scene inside the unit sphere produces: |x 2^k| <= 2^11 |x|):
// 3-term Cody-Waite reduction by pi/2 with FMA + degree-7/8 minimax polynomials on [-pi/4, pi/4]; max abs error 9.2e-8
// (< 1 ulp at 1), checked against fp64 on 1e7 points.  Larger arguments: the same polynomials after a reduction in
// float64 (pi/2 as a double-double: exact to fp32 for |x| < 2^40, where x itself is already spaced wider than the
// period) - a dozen instructions and no extra registers, where the library's Payne-Hanek path costs ~150 instructions
// and ~50 VGPRs at every inlined call site.  inf / NaN -> NaN like sinf.
// fp32 polynomials on [-pi/4, pi/4]: sn ~ sin(r), cs ~ cos(r)
__device__ __forceinline__ void sincos_poly(float r, float& sn, float& cs) {
    const float r2 = r * r;
    float ps = fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f);
    ps = fmaf(ps, r2, -1.6666654611e-1f);
    sn = fmaf(r * r2, ps, r);
    float pc = fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
    pc = fmaf(pc, r2, 4.166664568298827e-2f);
    cs = fmaf(r2 * r2, pc, fmaf(-0.5f, r2, 1.0f));
}

// |x| > 65536: reduction in float64 (exact to fp32 for |x| < 2^40); inf / NaN -> NaN; astronomically large finite -> 0
__device__ __forceinline__ float sin_big(float x) {
    if (!(fabsf(x) < 4.0e18f)) return x - x;
    const double xd = (double)x;
    const double kd = rint(xd * 0.63661977236758138);
    double rd = __builtin_fma(-kd, 1.5707963267948966, xd);
    rd = __builtin_fma(-kd, 6.123233995736766e-17, rd);
    const int q = (int)((long long)kd & 3);
    float sn, cs;
    sincos_poly((float)rd, sn, cs);
    const float res = (q & 1) ? cs : sn;
    return (q & 2) ? -res : res;
}

__device__ __forceinline__ float sin_cw(float x) {
    if (!(fabsf(x) <= 65536.0f)) return sin_big(x);
    const float k = rintf(x * 0.636619772f);
    float r = fmaf(-k, 1.57079625129699707031f, x);
    r = fmaf(-k, 7.54978941586159635335e-8f, r);
    r = fmaf(-k, 5.39030252995776476554e-15f, r);
    const int q = (int)k;
    float sn, cs;
    sincos_poly(r, sn, cs);
    const float res = (q & 1) ? cs : sn;
    return (q & 2) ? -res : res;
}

// The reference's feature PAIR of one encoding argument a (neo360/helper.py:123-124): sin(a) and sin(fl32(a + fl32(pi/2))),
// with ONE argument reduction and one pair of polynomials.  b = fl32(a + HP) is formed exactly as the reference forms it;
// TwoSum gives the rounding error of that add, so b = a + pi/2 + d with d known to fp32 (d = (HP - pi/2) - err, |d| <=
// ulp(b) / 2 + 4.4e-8 <= 4e-3 even at |a| = 65536), and
//     sin(b) = sin((k + 1) pi/2 + (r + d)),   sin(r + d) = sn + d cs + O(d^2),   cos(r + d) = cs - d sn + O(d^2).
// Max abs error of the second value 1.2e-7 for |a| <= 2560, <= 1.5e-7 up to 4096 (d^2 / 2 <= 3e-8 there); 9.2e-8 for
// sin_cw(b), 8.4e-8 for the first value: ~1 ulp at 1 (checked against float64 on 2.2e7 arguments, octaves 0..10 of
// |x| <= 2.5).  Encodings of points inside the unit sphere reach |a| <= 2^9 x 1.7, of vanilla points (far = 3) 2^9 x 3.6.
// Beyond 4096 the d^2 term grows (8e-6 at 65536): those arguments take two full evaluations.
__device__ __forceinline__ void sincos_pair(float a, float& s, float& c) {
    const float b = a + HALF_PI_F32;
    if (!(fabsf(a) <= 4096.0f)) {
        s = sin_cw(a);
        c = sin_cw(b);
        return;
    }
    const float k = rintf(a * 0.636619772f);
    float r = fmaf(-k, 1.57079625129699707031f, a);
    r = fmaf(-k, 7.54978941586159635335e-8f, r);
    r = fmaf(-k, 5.39030252995776476554e-15f, r);
    const int q = (int)k;
    float sn, cs;
    sincos_poly(r, sn, cs);
    const float bb = b - a;                                         // TwoSum: a + HP = b + err exactly
    const float err = (a - (b - bb)) + (HALF_PI_F32 - bb);
    const float d = 4.371139000186243e-08f - err;                   // fl32(pi/2) - pi/2 = +4.3711e-8
    const float sn2 = fmaf(d, cs, sn), cs2 = fmaf(-d, sn, cs);
    const float rs = (q & 1) ? cs : sn;
    s = (q & 2) ? -rs : rs;
    const int q1 = q + 1;
    const float rc = (q1 & 1) ? cs2 : sn2;
    c = (q1 & 2) ? -rc : rc;
}

// Positional encoding of one scalar at one octave: sin(x*2^k), sin(x*2^k + fl32(pi/2)).
// x*2^k is exact; the phase add rounds in fp32 exactly as the reference's does
// (neo360/helper.py:123-124).
__device__ __forceinline__ void enc_pair(float x, int k, float& s, float& c) {
    sincos_pair(ldexpf(x, k), s, c);
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

// Mip-NeRF 360 along-ray kernels: proposal resampling (max-dilation of the previous
// histogram, annealed softmax cdf, deterministic interval sampling) and the
// exp(-cumsum) compositing with an opaque last interval.  One wavefront per ray.
// Reference: models/mipnerf360/helper.py:154-275, :337-394; model.py:258-338.
#include "common.h"
#include "kernels.h"

namespace neo {

namespace {

constexpr int RPB = 4;          // rays (waves) per block
constexpr float EPS32 = 1.1920929e-07f;
constexpr int MAXP = 256;       // max points of a (dilated) histogram: 3*n_prev+1 <= 256

__global__ __launch_bounds__(256) void k_mip_resample(const float* __restrict__ s_prev,
                                                      const float* __restrict__ w_prev, int n_prev, int dilate,
                                                      float dilation, float anneal, const float* __restrict__ u_arr,
                                                      int R, int n, float s_near, float s_far,
                                                      float* __restrict__ sdist, float* __restrict__ tdist,
                                                      const float* __restrict__ jitter) {
    __shared__ float sh_t[RPB][MAXP];     // histogram edges (sorted)
    __shared__ float sh_w[RPB][MAXP];     // histogram weights -> softmax weights
    __shared__ float sh_c[RPB][MAXP];     // cdf
    __shared__ float sh_p[RPB][MAXP];     // pdf of the previous histogram / interval centres
    __shared__ float sh_a[RPB][MAXP];     // t0 (left dilated edges)
    __shared__ float sh_b[RPB][MAXP];     // t1 (right dilated edges)
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int ray_raw = blockIdx.x * RPB + wv;
    const bool live = ray_raw < R;
    const int ray = live ? ray_raw : R - 1;   // surplus waves redo the last ray: uniform barriers
    float* T = sh_t[wv]; float* Wt = sh_w[wv]; float* C = sh_c[wv]; float* Pd = sh_p[wv];
    float* A = sh_a[wv]; float* B = sh_b[wv];
    const float* sp = s_prev + (long)ray * (n_prev + 1);
    const float* wp = w_prev + (long)ray * n_prev;
    int npts, nwt;   // histogram the cdf is built from: npts edges, nwt = npts-1 weights
    if (dilate) {
        // ---- max_dilate_weights (helper.py:154-204), then the [1:-1] trims of model.py:283-284 ----
        const int nd = 3 * n_prev + 1;
        for (int j = lane; j < n_prev; j += 64) {
            const float a = sp[j], b = sp[j + 1];
            Pd[j] = wp[j] / fmaxf(b - a, EPS32);
            A[j] = a - dilation;
            B[j] = b + dilation;
        }
        for (int i = lane; i < MAXP; i += 64) {
            float v = __builtin_inff();
            if (i <= n_prev) v = sp[i];
            else if (i <= 2 * n_prev) v = sp[i - n_prev - 1] - dilation;
            else if (i < nd) v = sp[i - 2 * n_prev] + dilation;
            T[i] = v;
        }
        __syncthreads();
        for (int k = 2; k <= MAXP; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = lane; i < MAXP; i += 64) {
                    const int p = i ^ j;
                    if (p > i) {
                        const float a = T[i], b = T[p];
                        if ((a > b) == ((i & k) == 0)) { T[i] = b; T[p] = a; }
                    }
                }
                __syncthreads();
            }
        for (int i = lane; i < nd; i += 64) T[i] = fminf(fmaxf(T[i], 0.0f), 1.0f);   // clip to the domain (0,1)
        __syncthreads();
        // dilated pdf = max over the source intervals covering each edge; weight = pdf * width; renormalise
        float part = 0.f;
        for (int i = lane; i < nd - 1; i += 64) {
            const float x = T[i];
            float best = 0.f;
            for (int j = 0; j < n_prev; ++j)
                if (A[j] <= x && B[j] > x) best = fmaxf(best, Pd[j]);
            const float wgt = best * (T[i + 1] - x);
            Wt[i] = wgt;
            part += wgt;
        }
        const float total = fmaxf(wave_sum(part), EPS32);
        __syncthreads();
        // trimmed histogram: edges T[1 .. nd-2], weights Wt[1 .. nd-3] / total
        npts = nd - 2;
        nwt = nd - 3;
        for (int i = lane; i < nwt; i += 64) C[i] = Wt[i + 1] / total;
        __syncthreads();
        for (int i = lane; i < nwt; i += 64) Wt[i] = C[i];
        for (int i = lane; i < npts; i += 64) A[i] = T[i + 1];
        __syncthreads();
        for (int i = lane; i < npts; i += 64) T[i] = A[i];
        __syncthreads();
    } else {
        npts = n_prev + 1;
        nwt = n_prev;
        for (int i = lane; i < npts; i += 64) T[i] = sp[i];
        for (int i = lane; i < nwt; i += 64) Wt[i] = wp[i];
        __syncthreads();
    }
    // ---- annealed logits (model.py:292-296) -> softmax -> cdf (helper.py:207-216, :237-243) ----
    float mx = -__builtin_inff();
    for (int k = lane; k < nwt; k += 64) {
        const float lg = (T[k + 1] > T[k]) ? anneal * logf(Wt[k] + 0.0f) : -__builtin_inff();
        C[k] = lg;
        mx = fmaxf(mx, lg);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __syncthreads();
    float se = 0.f;
    for (int k = lane; k < nwt; k += 64) {
        const float e = expf(C[k] - mx);
        Wt[k] = e;
        se += e;
    }
    se = wave_sum(se);
    __syncthreads();
    // cdf: [0, min(1, cumsum(w[:-1])), 1]; running sum carried in fp64 (torch's CPU cumsum accumulates in double)
    double carry = 0.0;
    for (int base = 0; base < nwt - 1; base += 64) {
        const int k = base + lane;
        const bool valid = k < nwt - 1;
        double incl = valid ? (double)(Wt[k] / se) : 0.0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (valid) C[k + 1] = fminf((float)(carry + incl), 1.0f);
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) { C[0] = 0.0f; C[npts - 1] = 1.0f; }
    __syncthreads();
    // ---- interval centres: sorted_interp(u, cdf, edges) (helper.py:219-234) ----
    for (int m = lane; m < n; m += 64) {
        // randomized (helper.py:358-365, single_jitter): the caller's table linspace(0, 1 - u_max, n) + this ray's one draw
        const float u = jitter ? u_arr[m] + jitter[ray] : u_arr[m];
        int lo = 0, hi = npts;   // first index with cdf > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (C[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int J = lo - 1;
        const int J1 = J < npts - 1 ? J + 1 : npts - 1;
        const float x0 = C[J], x1 = C[J1], f0 = T[J], f1 = T[J1];
        float off = nan_to_num((u - x0) / (x1 - x0), 0.0f);
        off = fminf(fmaxf(off, 0.0f), 1.0f);
        Pd[m] = f0 + off * (f1 - f0);
    }
    __syncthreads();
    // ---- interval endpoints (helper.py:373-394) and s -> t (helper.py:171-175) ----
    if (live) {
        float* so = sdist + (long)ray * (n + 1);
        float* to = tdist + (long)ray * (n + 1);
        for (int i = lane; i <= n; i += 64) {
            float sv;
            if (i == 0) sv = fmaxf(2.0f * Pd[0] - (Pd[1] + Pd[0]) / 2.0f, 0.0f);
            else if (i == n) sv = fminf(2.0f * Pd[n - 1] - (Pd[n - 1] + Pd[n - 2]) / 2.0f, 1.0f);
            else sv = (Pd[i] + Pd[i - 1]) / 2.0f;
            so[i] = sv;
            to[i] = 1.0f / (sv * s_far + (1.0f - sv) * s_near);
        }
    }
}

// compute_alpha_weights(opaque_background=True) + volumetric_rendering (helper.py:246-275, :264-274)
__global__ __launch_bounds__(256) void k_mip_composite(const float4* __restrict__ rgbdens,
                                                       const float* __restrict__ tdist,
                                                       const float* __restrict__ rays_d, int R, int n, float bg,
                                                       float* __restrict__ w_out, float* __restrict__ rgb_out) {
    const int ray = blockIdx.x * RPB + (threadIdx.x >> 6);
    if (ray >= R) return;
    const int lane = lane_id();
    const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
    const float* tr = tdist + (long)ray * (n + 1);
    const float4* cs = rgbdens + (long)ray * n;
    double carry = 0.0;   // running sum of density*delta over earlier intervals
    float sr = 0.f, sg = 0.f, sb = 0.f, sa = 0.f;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool valid = i < n;
        float dd = 0.f;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            c = cs[i];
            const float delta = (tr[i + 1] - tr[i]) * dn;
            dd = c.w * delta;
        }
        // exclusive cumsum of dd (the opaque last interval only changes its own alpha)
        double incl = (double)dd;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 0.0;
        const float before = (float)(carry + excl);
        carry += __shfl(incl, 63, 64);
        if (valid) {
            const float alpha = (i == n - 1) ? 1.0f : 1.0f - expf(-dd);   // 1 - exp(-inf) = 1
            const float w = alpha * expf(-before);
            if (w_out) w_out[(long)ray * n + i] = w;
            sr += w * c.x; sg += w * c.y; sb += w * c.z; sa += w;
        }
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sa = wave_sum(sa);
    if (lane == 0 && rgb_out) {
        const float bw = fmaxf(1.0f - sa, 0.0f) * bg;
        rgb_out[ray * 3] = sr + bw;
        rgb_out[ray * 3 + 1] = sg + bw;
        rgb_out[ray * 3 + 2] = sb + bw;
    }
}

}  // namespace

int launch_mip_resample(const float* s_prev, const float* w_prev, int n_prev, int dilate, float dilation,
                        float anneal, const float* u, int R, int n, float s_near, float s_far, float* sdist,
                        float* tdist, hipStream_t s, const float* jitter) {
    if (n_prev < 1 || n < 2 || n > MAXP) return -1;
    if (dilate ? (3 * n_prev + 1 > MAXP || n_prev < 2) : (n_prev + 1 > MAXP)) return -1;
    hipLaunchKernelGGL(k_mip_resample, dim3((R + RPB - 1) / RPB), dim3(256), 0, s, s_prev, w_prev, n_prev, dilate,
                       dilation, anneal, u, R, n, s_near, s_far, sdist, tdist, jitter);
    return 0;
}

void launch_mip_composite(const float* rgbdens, const float* tdist, const float* rays_d, int R, int n, float bg,
                          float* weights, float* rgb, hipStream_t s) {
    hipLaunchKernelGGL(k_mip_composite, dim3((R + RPB - 1) / RPB), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(rgbdens), tdist, rays_d, R, n, bg, weights, rgb);
}

}  // namespace neo

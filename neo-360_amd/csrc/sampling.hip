// Along-ray kernels: positional encoding (stand-alone), alpha compositing and
// inverse-CDF resampling + merge.  One 64-lane wavefront owns one ray; the
// along-ray transmittance product and the cdf sum are wave-shuffle scans with a
// running carry across 64-sample rounds (samples of a ray are contiguous in
// HBM, so every round is one coalesced load per array).  These kernels are
// HBM-bound: 24 B/point for compositing, ~12 B/point for resampling.
#include "common.h"
#include "kernels.h"

namespace neo {

constexpr int RAYS_PER_BLOCK = 4;  // 4 waves of 64

// ---------------------------------------------------------------------------
// pos_enc: out[n][C*(2L+1)] = [x | sin(x 2^k) k-major | sin(x 2^k + pi/2) k-major]
// (neo360/helper.py:121-125 == vanilla_nerf/helper.py:445-449)
// ---------------------------------------------------------------------------
__global__ void k_pos_enc(const float* __restrict__ x, int n, int C, int min_deg, int max_deg,
                          float* __restrict__ out) {
    const int L = max_deg - min_deg;
    const int width = C * (2 * L + 1);
    const long total = (long)n * width;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / width), f = (int)(idx - (long)row * width);
        float v;
        if (f < C) {
            v = x[(long)row * C + f];
        } else {
            const int g = f - C;
            const bool shifted = g >= L * C;
            const int h = shifted ? g - L * C : g;
            const int k = min_deg + h / C, c = h % C;
            const float a = ldexpf(x[(long)row * C + c], k);
            v = shifted ? sin_cw(a + HALF_PI_F32) : sin_cw(a);
        }
        out[idx] = v;
    }
}

void launch_pos_enc(const float* x, int n, int C, int min_deg, int max_deg, float* out, hipStream_t s) {
    const long total = (long)n * C * (2 * (max_deg - min_deg) + 1);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_pos_enc, dim3(blocks), dim3(256), 0, s, x, n, C, min_deg, max_deg, out);
}

// ---------------------------------------------------------------------------
// compositing
//   mode 0: vanilla_nerf/helper.py:521-559   (delta_last 1e10, x|d|, exclusive product)
//   mode 1: neo360/helper.py:128-171 inside  (delta_last = t_far - t_last, x|d|, lambda = T_last)
//   mode 2: neo360/helper.py:128-171 outside (t descending, delta_i = t_i - t_{i+1}, last 1e10)
// The exclusive transmittance product is scanned in fp64 (the reference's CPU
// cumprod accumulates in double) and rounded to fp32 per element.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_composite(int mode, const float4* __restrict__ rgbsigma,
                                                   const float* __restrict__ t, int t_row_stride,
                                                   const float* __restrict__ rays_d,
                                                   const float* __restrict__ t_far, int R, int N, int white_bkgd,
                                                   float* __restrict__ rgb_out, float* __restrict__ acc_out,
                                                   float* __restrict__ depth_out, float* __restrict__ w_out,
                                                   float* __restrict__ lambda_out) {
    const int ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= R) return;
    const int lane = lane_id();
    const float* tr = t + (long)ray * t_row_stride;
    const float4* cs = rgbsigma + (long)ray * N;
    float dnorm = 1.0f;
    if (mode != 2) {
        const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
        dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    }
    const float far = mode == 1 ? t_far[ray] : 0.0f;
    double carry = 1.0;  // product of (1-alpha+eps) over all earlier samples
    float s_r = 0.f, s_g = 0.f, s_b = 0.f, s_acc = 0.f, s_depth = 0.f;
    for (int base = 0; base < N; base += 64) {
        const int i = base + lane;
        const bool valid = i < N;
        float alpha = 0.f, ti = 0.f;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            ti = tr[i];
            c = cs[i];
            float delta;
            if (i < N - 1) {
                const float tn = tr[i + 1];
                delta = mode == 2 ? ti - tn : tn - ti;
            } else {
                delta = mode == 1 ? far - ti : 1e10f;
            }
            if (mode != 2) delta = delta * dnorm;
            alpha = 1.0f - expf(-c.w * delta);
        }
        const float keep = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
        // inclusive product scan in fp64
        double incl = (double)keep;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(incl, o, 64);
            if (lane >= o) incl = up * incl;
        }
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        // per-element rounding to fp32 mirrors storing the fp32 cumprod tensor
        const float trans = (float)(carry * excl);
        const float w = alpha * trans;
        carry = carry * __shfl(incl, 63, 64);
        if (valid) {
            if (w_out) w_out[(long)ray * N + i] = w;
            s_r += w * c.x; s_g += w * c.y; s_b += w * c.z;
            s_acc += w;
            s_depth += w * ti;
        }
    }
    s_r = wave_sum(s_r); s_g = wave_sum(s_g); s_b = wave_sum(s_b);
    s_acc = wave_sum(s_acc); s_depth = wave_sum(s_depth);
    if (lane == 0) {
        if (white_bkgd) { const float bg = 1.0f - s_acc; s_r += bg; s_g += bg; s_b += bg; }
        if (mode == 0) s_depth = nan_to_num(s_depth, __builtin_inff());  // helper.py:546; the clamp at :547 is an identity
        if (rgb_out) { rgb_out[ray * 3] = s_r; rgb_out[ray * 3 + 1] = s_g; rgb_out[ray * 3 + 2] = s_b; }
        if (acc_out) acc_out[ray] = s_acc;
        if (depth_out) depth_out[ray] = s_depth;
        if (lambda_out) lambda_out[ray] = (float)carry;
    }
}

void launch_composite(int mode, const float* rgbsigma, const float* t, int t_row_stride, const float* rays_d,
                      const float* t_far,
                      int R, int N, int white_bkgd, float* rgb, float* acc, float* depth, float* weights,
                      float* lambda, hipStream_t s) {
    hipLaunchKernelGGL(k_composite, dim3((R + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK), dim3(256), 0, s, mode,
                       (const float4*)rgbsigma, t, t_row_stride, rays_d, t_far, R, N, white_bkgd, rgb, acc, depth,
                       weights, lambda);
}

// ---------------------------------------------------------------------------
// resampling: sorted_piecewise_constant_pdf + sort(cat(t_prev, samples))
// (vanilla_nerf/helper.py:567-616, neo360/helper.py:174-249), randomized=False.
//
// The reference brackets every quantile u with a mask/max/min over all bins
// (helper.py:204-210).  The cdf is non-decreasing by construction (running sum
// of non-negative terms, clamped, padded with 0 and 1), so {j : u >= cdf_j} is
// a prefix 0..J and
//    cdf0 = cdf_J                      cdf1 = cdf_{J+1}   (cdf_last if J is last)
//    bin0 = max(bins_0..bins_J)        bin1 = min(bins_{J+1}..bins_last) (bins_last if J is last)
// which is evaluated here with a prefix-max / suffix-min of the bins — valid for
// ascending AND descending bins (the background branch), no monotonicity of the
// bins assumed.  The merged set is sorted with an in-LDS bitonic network
// (a real sort: background samples are not monotone).
// ---------------------------------------------------------------------------
template <int MAXB, int SORT_N>
__global__ __launch_bounds__(256) void k_resample(const float* __restrict__ t_prev, int t_prev_stride,
                                                  const float* __restrict__ weights,
                                                  const float* __restrict__ u_arr, int u_row_stride, int R, int n_prev,
                                                  int n_new, int descending, float* __restrict__ t_out) {
    __shared__ float s_bins[RAYS_PER_BLOCK][MAXB];
    __shared__ float s_pmax[RAYS_PER_BLOCK][MAXB];
    __shared__ float s_smin[RAYS_PER_BLOCK][MAXB];
    __shared__ float s_cdf[RAYS_PER_BLOCK][MAXB];
    __shared__ float s_sort[RAYS_PER_BLOCK][SORT_N];
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int ray_raw = blockIdx.x * RAYS_PER_BLOCK + wv;
    const bool live = ray_raw < R;
    const int ray = live ? ray_raw : R - 1;  // surplus waves redo the last ray so every barrier is uniform
    const int nb = n_prev - 1;   // bins = midpoints
    const int nw = n_prev - 2;   // pdf weights = weights[1:-1]
    const int n_out = n_prev + n_new;
    float* bins = s_bins[wv]; float* pmax = s_pmax[wv]; float* smin = s_smin[wv];
    float* cdf = s_cdf[wv]; float* srt = s_sort[wv];
    {
        const float* tp = t_prev + (long)ray * t_prev_stride;
        const float* wp = weights + (long)ray * n_prev + 1;
        // bins + total weight
        float part = 0.f;
        for (int k = lane; k < nb; k += 64) {
            bins[k] = 0.5f * (tp[k + 1] + tp[k]);
            if (k < nw) part += wp[k];
        }
        float total = wave_sum(part);
        const float pad = fmaxf(0.0f, 1e-5f - total);
        const float add = pad / (float)nw;
        total = total + pad;
        // cdf: [0, min(1, cumsum(pdf[:-1])), 1]; running sum carried in fp64
        double carry = 0.0;
        for (int base = 0; base < nw - 1; base += 64) {
            const int j = base + lane;
            const bool valid = j < nw - 1;
            const float pdf = valid ? (wp[j] + add) / total : 0.f;
            double incl = (double)pdf;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            if (valid) cdf[j + 1] = fminf(1.0f, (float)(carry + incl));
            carry += __shfl(incl, 63, 64);
        }
        if (lane == 0) { cdf[0] = 0.0f; cdf[nb - 1] = 1.0f; }
        __syncthreads();
        // prefix max / suffix min of the bins
        float cmax = -__builtin_inff(), cmin = __builtin_inff();
        for (int base = 0; base < nb; base += 64) {
            const int k = base + lane;
            const float v = k < nb ? bins[k] : -__builtin_inff();
            const float m = fmaxf(cmax, wave_inclusive_scan(v, OpMax()));
            if (k < nb) pmax[k] = m;
            cmax = __shfl(m, 63, 64);
            const int kr = nb - 1 - k;  // mirrored index for the suffix scan
            const float vr = k < nb ? bins[kr] : __builtin_inff();
            const float mr = fminf(cmin, wave_inclusive_scan(vr, OpMin()));
            if (k < nb) smin[kr] = mr;
            cmin = __shfl(mr, 63, 64);
        }
        __syncthreads();
        // previous samples into the sort buffer
        for (int i = lane; i < SORT_N; i += 64) srt[i] = i < n_prev ? tp[i] : __builtin_inff();
        // new samples
        for (int m = lane; m < n_new; m += 64) {
            const float u = u_arr[(long)ray * u_row_stride + m];     // stride 0: one shared row of quantiles
            int lo = 0, hi = nb;  // first index with cdf > u
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
            }
            const int J = lo - 1;
            const bool last = J >= nb - 1;
            const float c0 = cdf[J], c1 = last ? cdf[nb - 1] : cdf[J + 1];
            const float b0 = pmax[J], b1 = last ? bins[nb - 1] : smin[J + 1];
            float frac = nan_to_num((u - c0) / (c1 - c0), 0.0f);
            frac = fminf(fmaxf(frac, 0.0f), 1.0f);
            srt[n_prev + m] = b0 + frac * (b1 - b0);
        }
    }
    __syncthreads();
    // bitonic sort, ascending (padding = +inf stays at the end)
    for (int k = 2; k <= SORT_N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < SORT_N; i += 64) {
                const int p = i ^ j;
                if (p > i) {
                    const float a = srt[i], b = srt[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { srt[i] = b; srt[p] = a; }
                }
            }
            __syncthreads();
        }
    }
    if (live) {
        float* out = t_out + (long)ray * n_out;
        for (int i = lane; i < n_out; i += 64) out[i] = srt[descending ? n_out - 1 - i : i];
    }
}

int launch_resample(const float* t_prev, int t_prev_stride, const float* weights, const float* u, int u_row_stride, int R,
                    int n_prev, int n_new, int descending, float* t_out, hipStream_t s) {
    const int n_out = n_prev + n_new;
    const dim3 grid((R + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK), block(256);
    if (n_prev < 4 || n_prev - 1 > 256) return -1;
    if (n_out <= 256)
        hipLaunchKernelGGL((k_resample<256, 256>), grid, block, 0, s, t_prev, t_prev_stride, weights, u, u_row_stride, R, n_prev, n_new, descending, t_out);
    else if (n_out <= 512)
        hipLaunchKernelGGL((k_resample<256, 512>), grid, block, 0, s, t_prev, t_prev_stride, weights, u, u_row_stride, R, n_prev, n_new, descending, t_out);
    else if (n_out <= 1024)
        hipLaunchKernelGGL((k_resample<256, 1024>), grid, block, 0, s, t_prev, t_prev_stride, weights, u, u_row_stride, R, n_prev, n_new, descending, t_out);
    else
        return -1;
    return 0;
}

// ---------------------------------------------------------------------------
// NeO-360 level-0 sample rows (neo360/helper.py:24-75, randomized=False):
//   inside : t = near*(1-e) + far*e                     (per-ray far, ascending)
//   outside: inverse radius = the edges flipped          (1 -> 0, same for every ray)
// ---------------------------------------------------------------------------
__global__ void k_tp_level0(const float* __restrict__ far, const float* __restrict__ edges, int R, int N, float near,
                            float* __restrict__ fg_t, float* __restrict__ bg_s) {
    const long total = (long)R * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ray = (int)(idx / N), i = (int)(idx - (long)ray * N);
        const float e = edges[i];
        const float lo = near * (1.0f - e);
        const float hi = far[ray] * e;
        fg_t[idx] = lo + hi;
        bg_s[idx] = edges[N - 1 - i];
    }
}

void launch_tp_level0(const float* far, const float* edges, int R, int N, float near, float* fg_t, float* bg_s,
                      hipStream_t s) {
    const long total = (long)R * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_tp_level0, dim3(blocks), dim3(256), 0, s, far, edges, R, N, near, fg_t, bg_s);
}

// rgb = fg + lambda*bg ; depth = fg_depth + lambda*bg_depth (neo360/model.py:521-527)
__global__ void k_tp_merge(const float* __restrict__ fg_rgb, const float* __restrict__ fg_depth,
                           const float* __restrict__ lambda, const float* __restrict__ bg_rgb,
                           const float* __restrict__ bg_depth, int R, float* __restrict__ rgb,
                           float* __restrict__ depth) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float lam = lambda[r];
    if (rgb) {
#pragma unroll
        for (int a = 0; a < 3; ++a) rgb[r * 3 + a] = fg_rgb[r * 3 + a] + lam * bg_rgb[r * 3 + a];
    }
    if (depth) depth[r] = fg_depth[r] + lam * bg_depth[r];
}

void launch_tp_merge(const float* fg_rgb, const float* fg_depth, const float* lambda, const float* bg_rgb,
                     const float* bg_depth, int R, float* rgb, float* depth, hipStream_t s) {
    hipLaunchKernelGGL(k_tp_merge, dim3((R + 255) / 256), dim3(256), 0, s, fg_rgb, fg_depth, lambda, bg_rgb, bg_depth,
                       R, rgb, depth);
}

}  // namespace neo

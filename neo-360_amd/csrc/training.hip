// Training-side operators of the NeO-360 path: backward of the alpha compositing,
// stand-alone tri-plane / pixel-aligned feature lookup with its backward (scatter-add into the feature maps),
// the distortion loss of torch_efficient_distloss with its gradient, and the counter-based uniform generator
// behind the stratified (randomized=True) samplers.  All are HBM-bound along-ray / gather kernels: one 64-lane
// wavefront per ray with shuffle scans, or one 16-lane group per 256-byte run of a channels-last texel.
#include "common.h"
#include "kernels.h"
#include "tp_common.h"

namespace neo {

namespace {

constexpr int RPB = 4;            // rays (waves) per 256-thread block
constexpr int MAXN = 1024;        // samples per ray the per-wave LDS rows hold

// ---- Philox4x32-10 (Salmon et al., SC'11), one 128-bit block per (row, col, stream) ------------------------
__device__ __forceinline__ uint32_t philox_u32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
        const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1;
        const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return x0;
}
// uniform in [0, 1) with 24 random bits, as torch.rand produces fp32 (value = k 2^-24)
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t row, uint32_t col, uint32_t stream) {
    return (float)(philox_u32(seed, row, col, stream) >> 8) * 5.9604644775390625e-08f;
}

__global__ void k_uniform(uint64_t seed, uint32_t stream, int rows, int cols, float* __restrict__ out) {
    const long total = (long)rows * cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / cols), c = (int)(idx - (long)r * cols);
        out[idx] = philox_uniform(seed, (uint32_t)r, (uint32_t)c, stream);
    }
}

// ---- stratified level-0 samples (neo360/helper.py:36-51, randomized=True) ------------------------------------
// base row b_k: inside = near (1 - e_k) + far e_k, outside = e_k; mids m_k = (b_k + b_{k+1}) / 2;
// lower = [b_0, m], upper = [m, b_last]; t_k = lower_k + (upper_k - lower_k) u_k; outside row written flipped.
__global__ void k_tp_level0_rand(const float* __restrict__ far, const float* __restrict__ edges, int R, int N, float near,
                                 const float* __restrict__ u_fg, const float* __restrict__ u_bg, float* __restrict__ fg_t,
                                 float* __restrict__ bg_s) {
    const long total = (long)R * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ray = (int)(idx / N), k = (int)(idx - (long)ray * N);
        const float fr = far[ray];
        auto fg_base = [&](int j) { const float e = edges[j]; const float lo = near * (1.0f - e); const float hi = fr * e; return lo + hi; };
        auto strat = [&](float bm, float b0, float bp, float u) {      // b_{k-1}, b_k, b_{k+1}
            const float lower = k == 0 ? b0 : 0.5f * (b0 + bm);
            const float upper = k == N - 1 ? b0 : 0.5f * (bp + b0);
            return lower + (upper - lower) * u;
        };
        const int km = k > 0 ? k - 1 : 0, kp = k < N - 1 ? k + 1 : N - 1;
        fg_t[idx] = strat(fg_base(km), fg_base(k), fg_base(kp), u_fg[idx]);
        bg_s[(long)ray * N + (N - 1 - k)] = strat(edges[km], edges[k], edges[kp], u_bg[idx]);
    }
}

// ---- compositing backward ---------------------------------------------------------------------------------------
// Forward (sampling.hip:k_composite): delta_i, e_i = exp(-sigma_i delta_i), alpha_i = 1 - e_i, a_i = (1 - alpha_i) + 1e-10,
// T_i = prod_{j<=i} a_j, w_i = alpha_i T_{i-1}, acc = sum w, rgb = sum w c (+ 1 - acc), depth = sum w t, lambda = T_last.
// With G_i = dL/dw_i = g_w_i + g_acc + g_rgb.(c_i - white) + g_depth t_i and S_i = sum_{k>i} G_k w_k:
//   dL/dc_i = w_i g_rgb ;  dL/dsigma_i = delta_i e_i [ G_i T_{i-1} - (S_i + g_lambda T_last) / a_i ].
__global__ __launch_bounds__(256) void k_composite_bwd(int mode, const float4* __restrict__ rgbsigma, const float* __restrict__ t,
                                                       int t_row_stride, const float* __restrict__ rays_d,
                                                       const float* __restrict__ t_far, int R, int N, int white_bkgd,
                                                       const float* __restrict__ g_rgb, const float* __restrict__ g_acc,
                                                       const float* __restrict__ g_depth, const float* __restrict__ g_w,
                                                       const float* __restrict__ g_lam, float4* __restrict__ g_out) {
    __shared__ float s_T[RPB][MAXN];       // T_{i-1}
    __shared__ float s_G[RPB][MAXN];       // G_i = dL/dw_i
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int ray = blockIdx.x * RPB + wv;
    if (ray >= R) return;
    const float* tr = t + (long)ray * t_row_stride;
    const float4* cs = rgbsigma + (long)ray * N;
    float dnorm = 1.0f;
    if (mode != 2) {
        const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
        dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    }
    const float far = mode == 1 ? t_far[ray] : 0.0f;
    const float gr = g_rgb ? g_rgb[ray * 3] : 0.f, gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.f, gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.f;
    const float ga = g_acc ? g_acc[ray] : 0.f, gd = g_depth ? g_depth[ray] : 0.f, gl = (g_lam && mode == 1) ? g_lam[ray] : 0.f;
    const float wsub = white_bkgd ? (gr + gg + gb) : 0.0f;
    auto delta_of = [&](int i, float ti) {
        float delta;
        if (i < N - 1) {
            const float tn = tr[i + 1];
            delta = mode == 2 ? ti - tn : tn - ti;
        } else {
            delta = mode == 1 ? far - ti : 1e10f;
        }
        return mode != 2 ? delta * dnorm : delta;
    };
    // forward pass: T_{i-1} and G_i w_i
    double carry = 1.0;
    for (int base = 0; base < N; base += 64) {
        const int i = base + lane;
        const bool valid = i < N;
        float alpha = 0.f, ti = 0.f;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            ti = tr[i];
            c = cs[i];
            alpha = 1.0f - expf(-c.w * delta_of(i, ti));
        }
        const float keep = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
        double incl = (double)keep;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(incl, o, 64);
            if (lane >= o) incl = up * incl;
        }
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        const float trans = (float)(carry * excl);
        carry = carry * __shfl(incl, 63, 64);
        if (valid) {
            s_T[wv][i] = trans;
            s_G[wv][i] = (g_w ? g_w[(long)ray * N + i] : 0.f) + ga + (gr * c.x + gg * c.y + gb * c.z) - wsub + gd * ti;
        }
    }
    const float T_last = (float)carry;
    // reverse pass: S_i = sum_{k>i} G_k w_k (fp64 running carry, like the forward product)
    double suffix = 0.0;
    const int rounds = (N + 63) / 64;
    for (int rd = rounds - 1; rd >= 0; --rd) {
        const int i = rd * 64 + lane;
        const bool valid = i < N;
        float ti = 0.f, delta = 0.f, e = 1.f, trans = 0.f, w = 0.f, G = 0.f;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            ti = tr[i];
            c = cs[i];
            delta = delta_of(i, ti);
            e = expf(-c.w * delta);
            trans = s_T[wv][i];
            w = (1.0f - e) * trans;
            G = s_G[wv][i];
        }
        const double v = (double)(G * w);
        double incl = v;                       // suffix-inclusive within the round: sum over lanes >= lane
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double dn = __shfl_down(incl, o, 64);
            if (lane + o < 64) incl += dn;
        }
        const double S = suffix + (incl - v);  // strictly later samples
        suffix += __shfl(incl, 0, 64);
        if (valid) {
            const float a = (1.0f - (1.0f - e)) + 1e-10f;
            const float gs = delta * e * (G * trans - (float)((S + (double)gl * (double)T_last) / (double)a));
            g_out[(long)ray * N + i] = make_float4(w * gr, w * gg, w * gb, gs);
        }
    }
}

// ---- distortion loss (torch_efficient_distloss.eff_distloss) -------------------------------------------------------
// loss_ray = interval/3 sum w_i^2 + 2 sum_{i>=1} (w_i m_i W_{i-1} - w_i WM_{i-1}), W / WM inclusive prefix sums of w / w m;
// d loss_ray / d w_i = 2 interval w_i / 3 + 2 (m_i (Wpre_i - Wsuf_i) + (WMsuf_i - WMpre_i)).
__global__ __launch_bounds__(256) void k_distloss(const float* __restrict__ w, const float* __restrict__ m, int R, int N,
                                                   float interval, float* __restrict__ loss_rays, float* __restrict__ grad_w) {
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int ray = blockIdx.x * RPB + wv;
    if (ray >= R) return;
    const float* wr = w + (long)ray * N;
    const float* mr = m + (long)ray * N;
    float tw = 0.f, twm = 0.f;
    for (int i = lane; i < N; i += 64) { tw += wr[i]; twm += wr[i] * mr[i]; }
    const double W_total = (double)wave_sum(tw), WM_total = (double)wave_sum(twm);
    double cw = 0.0, cwm = 0.0, loss = 0.0;
    for (int base = 0; base < N; base += 64) {
        const int i = base + lane;
        const bool valid = i < N;
        const float wi = valid ? wr[i] : 0.f, mi = valid ? mr[i] : 0.f;
        double iw = (double)wi, iwm = (double)(wi * mi);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double a = __shfl_up(iw, o, 64), b = __shfl_up(iwm, o, 64);
            if (lane >= o) { iw += a; iwm += b; }
        }
        const double Wpre = cw + iw - (double)wi, WMpre = cwm + iwm - (double)(wi * mi);     // strictly before i
        cw += __shfl(iw, 63, 64);
        cwm += __shfl(iwm, 63, 64);
        if (valid) {
            const double Wsuf = W_total - Wpre - (double)wi, WMsuf = WM_total - WMpre - (double)(wi * mi);
            loss += (double)interval * (double)wi * (double)wi / 3.0 + 2.0 * ((double)(wi * mi) * Wpre - (double)wi * WMpre);
            if (grad_w)
                grad_w[(long)ray * N + i] = (float)(2.0 * (double)interval * (double)wi / 3.0 +
                                                    2.0 * ((double)mi * (Wpre - Wsuf) + (WMsuf - WMpre)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o, 64);
    if (lane == 0 && loss_rays) loss_rays[ray] = (float)loss;
}

// ---- stand-alone feature lookups (encoder_tp_fusion_conv.py:122-209 index_grid, model.py:239-264 get_local_feats) --
// rows are view-major: row = v P + p.  One 16-lane group per row; lane c of the group owns the 16-byte pieces c, c+16, ...
struct RowTaps { tp::TapSet loc, pl[3]; };

__device__ __forceinline__ RowTaps row_taps(const TpScene& sc, const float* rot, const float* trn, const float* p3) {
    const float fx = p3[0], fy = p3[1], fz = p3[2];
    const float cx_ = (rot[0] * fx + rot[1] * fy + rot[2] * fz) + trn[0];
    const float cy_ = (rot[3] * fx + rot[4] * fy + rot[5] * fz) + trn[1];
    const float cz_ = (rot[6] * fx + rot[7] * fy + rot[8] * fz) + trn[2];
    RowTaps r;
    const float den = cz_ + 1e-9f;
    const float u = (-cx_ / den) * sc.focal + sc.cx;
    const float w_ = (-cy_ / den) * (sc.fy_sign * sc.focal) + sc.cy;
    r.loc = tp::bilinear_taps(u * sc.sx - 1.0f, w_ * sc.sy - 1.0f, sc.Wf, sc.Hf);
    r.pl[0] = tp::bilinear_taps(cx_, cz_, sc.Wp, sc.Hp);       // xz
    r.pl[1] = tp::bilinear_taps(cx_, cy_, sc.Wp, sc.Hp);       // xy
    r.pl[2] = tp::bilinear_taps(cy_, cz_, sc.Wp, sc.Hp);       // yz
    return r;
}

template <bool BWD>
__global__ __launch_bounds__(256) void k_gather(TpScene sc, TpViews views, const float* __restrict__ pts, long P,
                                                float* __restrict__ world, float* __restrict__ local,
                                                float* __restrict__ g_plane0, float* __restrict__ g_plane1,
                                                float* __restrict__ g_plane2, float* __restrict__ g_latent) {
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c = threadIdx.x & 15;
    if (row >= P * sc.nv) return;
    const int v = (int)(row / P);
    const long p = row - (long)v * P;
    const float p3[3] = {pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2]};
    const RowTaps tp_ = row_taps(sc, views.rot[v], views.trans[v], p3);
    float* gpl[3] = {g_plane0, g_plane1, g_plane2};
    // tri-planes: 128 channels = 32 pieces of 16 B -> 2 per lane
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        const int piece = c + 16 * k2;
        if (!BWD) {
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 tap[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    tap[k] = *reinterpret_cast<const f32x4*>(sc.plane[j] + ((long)v * sc.Hp * sc.Wp + tp_.pl[j].off[k]) * 128 + piece * 4);
                const f32x4 wv4 = {tp_.pl[j].w[0], tp_.pl[j].w[1], tp_.pl[j].w[2], tp_.pl[j].w[3]};
                const f32x4 b = tp::blend4(tap, wv4);
                sum = j == 0 ? b : sum + b;
            }
            *reinterpret_cast<f32x4*>(world + row * 128 + piece * 4) = sum;
        } else {
            const f32x4 g = *reinterpret_cast<const f32x4*>(world + row * 128 + piece * 4);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float wk = tp_.pl[j].w[k];
                    if (wk == 0.0f) continue;
                    float* dst = gpl[j] + ((long)v * sc.Hp * sc.Wp + tp_.pl[j].off[k]) * 128 + piece * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(dst + e, wk * g[e]);
                }
        }
    }
    // pixel-aligned latent: 512 channels = 128 pieces -> 8 per lane (local == nullptr: planes only - the projected-space training
    // path gathers the latent through neo_tp_gather_map instead)
    if (local == nullptr) return;
#pragma unroll 2
    for (int k8 = 0; k8 < 8; ++k8) {
        const int piece = c + 16 * k8;
        if (!BWD) {
            f32x4 tap[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                tap[k] = *reinterpret_cast<const f32x4*>(sc.latent + ((long)v * sc.Hf * sc.Wf + tp_.loc.off[k]) * 512 + piece * 4);
            const f32x4 wv4 = {tp_.loc.w[0], tp_.loc.w[1], tp_.loc.w[2], tp_.loc.w[3]};
            *reinterpret_cast<f32x4*>(local + row * 512 + piece * 4) = tp::blend4(tap, wv4);
        } else {
            const f32x4 g = *reinterpret_cast<const f32x4*>(local + row * 512 + piece * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float wk = tp_.loc.w[k];
                if (wk == 0.0f) continue;
                float* dst = g_latent + ((long)v * sc.Hf * sc.Wf + tp_.loc.off[k]) * 512 + piece * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dst + e, wk * g[e]);
            }
        }
    }
}

// ---- lookup backward with RUN MERGING (round 5) -----------------------------------------------------------------------------------
// The scatter-add above issues one fp32 atomic per (row, tap, channel): 5.5e9 atomics per training step of 500 rays - 41 % of the
// step (profiles/r05_train_step_kernel_stats.csv).  Consecutive rows are consecutive samples of one ray, and at the fine level
// 3-6 of them fall into the SAME texel cell (profiles/r03_tile_footprint.json: 32-38 distinct plane texels among the 256 taps of 64
// samples).  Here a 16-lane group walks 16 consecutive rows; per map and tap slot it keeps the texel it is accumulating for and
// the running w * g sum in registers, and issues the atomics only when the texel changes (and at the end of the run): the same
// sums, a few times fewer atomics.  The 16 rows' tap descriptors are computed once (lane c computes row c) and shared through LDS.
constexpr int GRUN = 16;
// channel order of a 16-lane group's 64-channel piece: 1 = lane c holds channels c, c + 16, c + 32, c + 48 (one atomic instruction of
// the group = one 64-byte sector of the texel; with 0 = 4 consecutive channels per lane, every one of the 4 instructions touched
// all 4 sectors)
__device__ __forceinline__ int scatter_channel(int piece64, int c, int e) {
    return piece64 * 64 + 16 * e + c;
}
__device__ __forceinline__ void scatter_add(float* dst, float v) {
    atomicAdd(dst, v);
}
__device__ __forceinline__ f32x4 scatter_grad(const float* row, int piece64, int c) {
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = row[piece64 * 64 + 16 * e + c];
    return g;
}
__global__ __launch_bounds__(256) void k_gather_bwd_runs(TpScene sc, TpViews views, const float* __restrict__ pts, long P,
                                                        const float* __restrict__ g_world, const float* __restrict__ g_local,
                                                        float* __restrict__ g_plane0, float* __restrict__ g_plane1,
                                                        float* __restrict__ g_plane2, float* __restrict__ g_latent) {
    __shared__ int s_off[16][GRUN][4][4];        // [group][row of the run][map: 0 latent, 1..3 planes][tap] texel index incl. the view's base, -1 = no weight
    __shared__ float s_w[16][GRUN][4][4];
    const int grp = threadIdx.x >> 4, c = threadIdx.x & 15;
    const long rows = P * sc.nv;
    const long row0 = ((long)blockIdx.x * 16 + grp) * GRUN;
    {
        const long row = row0 + c;
        if (row < rows) {
            const int v = (int)(row / P);
            const long p = row - (long)v * P;
            const float p3[3] = {pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2]};
            const RowTaps t = row_taps(sc, views.rot[v], views.trans[v], p3);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_off[grp][c][0][k] = t.loc.w[k] != 0.0f ? v * sc.Hf * sc.Wf + t.loc.off[k] : -1;
                s_w[grp][c][0][k] = t.loc.w[k];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    s_off[grp][c][1 + j][k] = t.pl[j].w[k] != 0.0f ? v * sc.Hp * sc.Wp + t.pl[j].off[k] : -1;
                    s_w[grp][c][1 + j][k] = t.pl[j].w[k];
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int k = 0; k < 4; ++k) { s_off[grp][c][m][k] = -1; s_w[grp][c][m][k] = 0.0f; }
        }
    }
    __syncthreads();
    if (row0 >= rows) return;
    float* gmap[4] = {g_latent, g_plane0, g_plane1, g_plane2};
#pragma unroll 1
    for (int m = g_local ? 0 : 1; m < 4; ++m) {
        const int ch = m == 0 ? 512 : 128;
        const float* gsrc = m == 0 ? g_local : g_world;
        const int pieces = ch / 64;                       // 16-byte pieces per lane: 8 (latent) or 2 (a plane)
#pragma unroll 1
        for (int q = 0; q < pieces; ++q) {
            int cur[4] = {-1, -1, -1, -1};
            f32x4 acc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto flush = [&](int k) {
                if (cur[k] >= 0) {
                    float* dst = gmap[m] + (long)cur[k] * ch;
#pragma unroll
                    for (int e = 0; e < 4; ++e) scatter_add(dst + scatter_channel(q, c, e), acc[k][e]);
                }
            };
#pragma unroll 1
            for (int r = 0; r < GRUN; ++r) {
                if (row0 + r >= rows) break;
                const f32x4 g = scatter_grad(gsrc + (row0 + r) * ch, q, c);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int o = s_off[grp][r][m][k];
                    if (o < 0) continue;
                    const float w = s_w[grp][r][m][k];
                    if (o != cur[k]) {
                        flush(k);
                        cur[k] = o;
                        acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[k][e] = __builtin_fmaf(w, g[e], acc[k][e]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) flush(k);
        }
    }
}

// ---- lookup in a CALLER-OWNED channels-last map at the latent's taps (round 5: projected-space training) ---------------------------
// map (NV Hf Wf, C) fp32, C a multiple of 64 (256 for the latent projected through [W0_loc | W3_loc]); rows view-major as above, the
// taps those of get_local_feats (row_taps: view 0's intrinsics, the uploaded scene's latent geometry).  Forward: one 16-lane group
// per row; backward: the run-merged scatter of k_gather_bwd_runs for the one map.
__global__ __launch_bounds__(256) void k_map_gather(TpScene sc, TpViews views, const float* __restrict__ pts, long P,
                                                    const float* __restrict__ map, long pitch, int C, float* __restrict__ out) {
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c = threadIdx.x & 15;
    if (row >= P * sc.nv) return;
    const int v = (int)(row / P);
    const long p = row - (long)v * P;
    const float p3[3] = {pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2]};
    const RowTaps t = row_taps(sc, views.rot[v], views.trans[v], p3);
    const f32x4 wv4 = {t.loc.w[0], t.loc.w[1], t.loc.w[2], t.loc.w[3]};
    for (int piece = c; piece < C / 4; piece += 16) {
        f32x4 tap[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            tap[k] = *reinterpret_cast<const f32x4*>(map + ((long)v * sc.Hf * sc.Wf + t.loc.off[k]) * pitch + piece * 4);
        *reinterpret_cast<f32x4*>(out + row * C + piece * 4) = tp::blend4(tap, wv4);
    }
}

__global__ __launch_bounds__(256) void k_map_gather_bwd_runs(TpScene sc, TpViews views, const float* __restrict__ pts, long P,
                                                            const float* __restrict__ g_out, int C, float* __restrict__ g_map, long pitch) {
    __shared__ int s_off[16][GRUN][4];
    __shared__ float s_w[16][GRUN][4];
    const int grp = threadIdx.x >> 4, c = threadIdx.x & 15;
    const long rows = P * sc.nv;
    const long row0 = ((long)blockIdx.x * 16 + grp) * GRUN;
    {
        const long row = row0 + c;
        if (row < rows) {
            const int v = (int)(row / P);
            const long p = row - (long)v * P;
            const float p3[3] = {pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2]};
            const RowTaps t = row_taps(sc, views.rot[v], views.trans[v], p3);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_off[grp][c][k] = t.loc.w[k] != 0.0f ? v * sc.Hf * sc.Wf + t.loc.off[k] : -1;
                s_w[grp][c][k] = t.loc.w[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { s_off[grp][c][k] = -1; s_w[grp][c][k] = 0.0f; }
        }
    }
    __syncthreads();
    if (row0 >= rows) return;
#pragma unroll 1
    for (int q = 0; q < C / 64; ++q) {
        int cur[4] = {-1, -1, -1, -1};
        f32x4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto flush = [&](int k) {
            if (cur[k] >= 0) {
                float* dst = g_map + (long)cur[k] * pitch;
#pragma unroll
                for (int e = 0; e < 4; ++e) scatter_add(dst + scatter_channel(q, c, e), acc[k][e]);
            }
        };
#pragma unroll 1
        for (int r = 0; r < GRUN; ++r) {
            if (row0 + r >= rows) break;
            const f32x4 g = scatter_grad(g_out + (row0 + r) * C, q, c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = s_off[grp][r][k];
                if (o < 0) continue;
                const float w = s_w[grp][r][k];
                if (o != cur[k]) {
                    flush(k);
                    cur[k] = o;
                    acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[k][e] = __builtin_fmaf(w, g[e], acc[k][e]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) flush(k);
    }
}

// ---- sample points and their encodings for the training call (round 5) --------------------------------------------------------
// What the evaluators compute per tile in registers / LDS (tp_common.h:point_setup_row + the camera transform of
// view_descriptors), written out for the operator chain of training.py: per point the world-space lookup point (inside: o + t d;
// outside: o + (far (1 - s) + 3 s) d, neo360/helper.py:59-73) and per source view the reference-order positional encoding of
// the camera-frame point (inside: 63 features of R_v x + t_v; outside: 84 features of [R_v x' + t_v | s] with x' the inverted-sphere
// point, helper.py:401-451, model.py:454-464).  One thread per point, 64 points per block; the same device functions as the
// inference kernels, so both paths see bitwise the same points.  Replaces training.py's torch restatements of both transforms.
template <int PE_C>
__global__ __launch_bounds__(64) void k_tp_train_points(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                        const float* __restrict__ tvals, const float* __restrict__ far_arr, int R, int N,
                                                        TpViews views, int nv, uint32_t* __restrict__ flags, float* __restrict__ look,
                                                        float* __restrict__ x_enc) {
    __shared__ float sm[3 * tp::TM * 4];
    tp::Scratch S{};
    S.pe_world = sm;
    S.feat_world = sm + tp::TM * 4;
    S.vdir_world = sm + 2 * tp::TM * 4;
    const int tid = threadIdx.x;
    const long P = (long)R * N, tile0 = (long)blockIdx.x * tp::TM;
    // viewdirs only feed the direction tiling (unused here): rays_d stands in for the pointer
    tp::point_setup_row<PE_C>(S, tid, tile0, P, N, R, R, rays_o, rays_d, rays_d, tvals, far_arr, flags);
    const long g = tile0 + tid;
    if (g >= P) return;
    const float ex = S.pe_world[tid * 4], ey = S.pe_world[tid * 4 + 1], ez = S.pe_world[tid * 4 + 2], ew = S.pe_world[tid * 4 + 3];
#pragma unroll
    for (int a = 0; a < 3; ++a) look[g * 3 + a] = S.feat_world[tid * 4 + a];
    constexpr int F = 21 * PE_C;
    for (int v = 0; v < nv; ++v) {
        const float* rot = views.rot[v];
        const float* trn = views.trans[v];
        const float xc[4] = {(rot[0] * ex + rot[1] * ey + rot[2] * ez) + trn[0], (rot[3] * ex + rot[4] * ey + rot[5] * ez) + trn[1],
                             (rot[6] * ex + rot[7] * ey + rot[8] * ez) + trn[2], ew};
        float* dst = x_enc + ((long)v * P + g) * F;
#pragma unroll 7
        for (int f = 0; f < F; ++f) dst[f] = tp::pe_feature<PE_C>(xc, f);
    }
}

// ---- the reference's output activations as ONE op with a backward (neo360/model.py:380-385) ----------------------------------
// rgbsigma (P, 4) = (sigmoid(raw_rgb) * 1.002 - 0.001, softplus(raw_sigma + noise - 1)); backward: d rgb = 1.002 s (1 - s),
// d sigma = sigmoid(x) (torch.nn.Softplus: x beyond the threshold 20 passes through with derivative 1)
__global__ void k_tp_activate(const float* __restrict__ raw_rgb, const float* __restrict__ raw_sigma, const float* __restrict__ noise,
                              float noise_scale, long P, float4* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float rs = raw_sigma[i] + (noise ? noise[i] * noise_scale : 0.0f);
    out[i] = make_float4(colour_act(raw_rgb[i * 3]), colour_act(raw_rgb[i * 3 + 1]), colour_act(raw_rgb[i * 3 + 2]), density_act(rs));
}
__global__ void k_tp_activate_bwd(const float* __restrict__ raw_rgb, const float* __restrict__ raw_sigma, const float* __restrict__ noise,
                                  float noise_scale, long P, const float4* __restrict__ g, float* __restrict__ g_rgb,
                                  float* __restrict__ g_sigma) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float4 gi = g[i];
    const float gc[3] = {gi.x, gi.y, gi.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float sg = 1.0f / (1.0f + expf(-raw_rgb[i * 3 + c]));
        g_rgb[i * 3 + c] = gc[c] * (1.002f * sg * (1.0f - sg));
    }
    const float x = raw_sigma[i] + (noise ? noise[i] * noise_scale : 0.0f) + (-1.0f);
    g_sigma[i] = gi.w * (x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x)));
}

}  // namespace

void launch_map_gather(const TpScene& sc, const TpViews& views, const float* pts, long P, const float* map, int C, float* out,
                       hipStream_t s, long pitch) {
    const long rows = P * sc.nv;
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_map_gather, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, sc, views, pts, P, map, pitch > 0 ? pitch : (long)C, C, out);
}

void launch_map_gather_bwd(const TpScene& sc, const TpViews& views, const float* pts, long P, const float* g_out, int C, float* g_map,
                           hipStream_t s, long pitch) {
    const long rows = P * sc.nv;
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_map_gather_bwd_runs, dim3((unsigned)((rows + 16 * GRUN - 1) / (16 * GRUN))), dim3(256), 0, s, sc, views, pts, P,
                       g_out, C, g_map, pitch > 0 ? pitch : (long)C);
}

void launch_tp_train_points(int input_ch, const float* rays_o, const float* rays_d, const float* tvals, const float* far, int R, int N,
                            const TpViews& views, int nv, uint32_t* flags, float* look, float* x_enc, hipStream_t s) {
    const long P = (long)R * N;
    if (P <= 0) return;
    const dim3 grid((unsigned)((P + tp::TM - 1) / tp::TM));
    if (input_ch == 3)
        hipLaunchKernelGGL(k_tp_train_points<3>, grid, dim3(64), 0, s, rays_o, rays_d, tvals, far, R, N, views, nv, flags, look, x_enc);
    else
        hipLaunchKernelGGL(k_tp_train_points<4>, grid, dim3(64), 0, s, rays_o, rays_d, tvals, far, R, N, views, nv, flags, look, x_enc);
}

void launch_tp_activate(const float* raw_rgb, const float* raw_sigma, const float* noise, float noise_scale, long P, float* rgbsigma,
                        hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(k_tp_activate, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, raw_rgb, raw_sigma, noise, noise_scale, P,
                       reinterpret_cast<float4*>(rgbsigma));
}

void launch_tp_activate_bwd(const float* raw_rgb, const float* raw_sigma, const float* noise, float noise_scale, long P,
                            const float* g_rgbsigma, float* g_rgb, float* g_sigma, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(k_tp_activate_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, raw_rgb, raw_sigma, noise, noise_scale, P,
                       reinterpret_cast<const float4*>(g_rgbsigma), g_rgb, g_sigma);
}

// dst[b][c][r] = src[b][r][c]: batched 2-D transpose through a 64 x 64 LDS tile, 256-B segments on both sides.  NCHW <-> channels-last
// of the latent under autograd (training.py: the texel-space projection reads (texels, 512) rows; torch's strided copy moved the
// 472 MB at 0.9 TB/s: 0.66 ms forward + 1.04 ms backward of a 31 ms step)
namespace {
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
    __shared__ float tile[64][65];
    const long b = blockIdx.z;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const float* sp = src + b * (long)rows * cols;
    float* dp = dst + b * (long)rows * cols;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = r0 + ty + 4 * j, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 4 * j][tx] = sp[(long)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = c0 + ty + 4 * j, r = r0 + tx;
        if (r < rows && c < cols) dp[(long)c * rows + r] = tile[tx][ty + 4 * j];
    }
}
}  // namespace

void launch_transpose(const float* src, long batch, int rows, int cols, float* dst, hipStream_t s) {
    if (batch <= 0 || rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(k_transpose, dim3((cols + 63) / 64, (rows + 63) / 64, (unsigned)batch), dim3(256), 0, s, src, rows, cols, dst);
}

void launch_uniform(uint64_t seed, uint32_t stream, int rows, int cols, float* out, hipStream_t s) {
    const long total = (long)rows * cols;
    if (total <= 0) return;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_uniform, dim3(blocks), dim3(256), 0, s, seed, stream, rows, cols, out);
}

void launch_tp_level0_rand(const float* far, const float* edges, int R, int N, float near, const float* u_fg,
                           const float* u_bg, float* fg_t, float* bg_s, hipStream_t s) {
    const long total = (long)R * N;
    if (total <= 0) return;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_tp_level0_rand, dim3(blocks), dim3(256), 0, s, far, edges, R, N, near, u_fg, u_bg, fg_t, bg_s);
}

int launch_composite_bwd(int mode, const float* rgbsigma, const float* t, int t_row_stride, const float* rays_d,
                         const float* t_far, int R, int N, int white_bkgd, const float* g_rgb, const float* g_acc,
                         const float* g_depth, const float* g_w, const float* g_lam, float* g_rgbsigma, hipStream_t s) {
    if (N > MAXN) return -1;
    hipLaunchKernelGGL(k_composite_bwd, dim3((R + RPB - 1) / RPB), dim3(256), 0, s, mode,
                       reinterpret_cast<const float4*>(rgbsigma), t, t_row_stride, rays_d, t_far, R, N, white_bkgd, g_rgb,
                       g_acc, g_depth, g_w, g_lam, reinterpret_cast<float4*>(g_rgbsigma));
    return 0;
}

void launch_distloss(const float* w, const float* m, int R, int N, float interval, float* loss_rays, float* grad_w,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_distloss, dim3((R + RPB - 1) / RPB), dim3(256), 0, s, w, m, R, N, interval, loss_rays, grad_w);
}

void launch_gather(const TpScene& sc, const TpViews& views, const float* pts, long P, float* world, float* local,
                   hipStream_t s) {
    const long rows = P * sc.nv;
    if (rows <= 0) return;
    hipLaunchKernelGGL((k_gather<false>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, sc, views, pts, P, world,
                       local, nullptr, nullptr, nullptr, nullptr);
}

void launch_gather_bwd(const TpScene& sc, const TpViews& views, const float* pts, long P, const float* g_world,
                       const float* g_local, float* g_plane_xz, float* g_plane_xy, float* g_plane_yz, float* g_latent,
                       hipStream_t s) {
    const long rows = P * sc.nv;
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_gather_bwd_runs, dim3((unsigned)((rows + 16 * GRUN - 1) / (16 * GRUN))), dim3(256), 0, s, sc, views, pts, P,
                       g_world, g_local, g_plane_xz, g_plane_xy, g_plane_yz, g_latent);
}

}  // namespace neo

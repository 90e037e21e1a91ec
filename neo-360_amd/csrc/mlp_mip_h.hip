// Mip-NeRF 360 point evaluator on the fp16 matrix cores with hi/lo-split fp32 operands
// (models/mipnerf360/model.py:30-176 + helper.py:33-88, 278-370; structure of mlp_mip.hip, arithmetic of
// split_tile.h): conical frustum -> Gaussian -> contraction -> 21-direction lift -> 504-d integrated
// positional encoding -> PropMLP 4x256 or NeRFMLP 8x1024 (+ bottleneck / view / rgb branch) -> activations.
//
// tile = 32 intervals, workgroup = 8 waves.  The 32 x W activation tile lives in LDS as two fp16 planes
// (128 KB at W = 1024, same footprint as the fp32 tile of mlp_mip.hip), every wave owns W/8 outputs
// (4 accumulator tiles at W = 1024).  The encoding is recomputed 64 features at a time into a
// double-buffered stage tile for layer 0 and for the skip layer.  Weights stream from L2 as split
// fragments (hi | lo per k-step: the same bytes as fp32); per 16-deep k-step a wave issues 3 MFMAs per
// accumulator tile.  Compiled without packed-fp32 VALU ops like mlp_tp_h.hip (build.py:EXTRA_FLAGS).
//
// Algorithmic work: PropMLP 325,888 MAC, NeRFMLP 8,672,000 MAC per interval.
//
// Colour branch (round 4): bottleneck_layer has no activation and feeds views_linear.0 only (model.py:100-108), so it is folded
// into that layer at pack time - [W_v[:, :256] W_b | W_v[:, 256:]] h', one 128 x (1024 + 27) layer instead of 256 x 1024 +
// 128 x 283 - and the layer's K range is split over the workgroup's 16 waves (4 N-tiles x 4 quarters, partial tiles added
// through LDS): all waves stream weights, each for 16-18 k-steps, with a 4-deep fragment ring.
//
// Round 4: the NeRF MLP also runs LAYER BY LAYER (launch_mip_mlp_h_layered): k_mip_ipe_h writes a batch's encodings as MFMA
// fragments, eight k_mip_gemm_h launches (mip_gemm_h.h: 256 x 256 output tiles, weight fragment reused by 4 interval tiles
// instead of 1) run the trunk with the activations in L2 / Infinity Cache between layers, and this file's evaluator in TAIL
// mode (trunk output read back into its LDS tile) adds density head, bottleneck, view layer and rgb head.
#include "mip_gemm_h.h"
#include "mip_layered.h"
#include "split_tile.h"

#ifndef NEO_MIP_H_WAVES
#define NEO_MIP_H_WAVES 16      // waves per workgroup of the 1024-wide evaluator: 16 (4 per SIMD) measured +3-4 % over 8
#endif

namespace neo {

namespace {

constexpr int TMR = 32;          // rows (intervals) per tile
constexpr int NB = 21;           // basis directions
constexpr float EPS32 = 1.1920929e-07f;

// ---- packed layout (h8 units) / bias layout (floats) ------------------------------------------------
struct MipLayoutH {
    int w_layer[8];   // h8 offset of trunk layer i
    int ks_layer[8];  // 16-deep k-steps of layer i
    int w_view;       // views_linear.0 with the bottleneck folded in: 128 outputs, K = W + 27 -> W / 16 + 2 k-steps
    int b_layer[8];
    int total_h8, total_b;
};

__host__ __device__ inline MipLayoutH mip_layout_h(int W, int depth, int rgb) {
    MipLayoutH L{};
    int ow = 0, ob = 0;
    for (int i = 0; i < depth; ++i) {
        const int ks = i == 0 ? 32 : (W / 16 + ((i == 5) ? 32 : 0));     // 504 -> 512 = 32 k-steps
        L.w_layer[i] = ow;
        L.ks_layer[i] = ks;
        L.b_layer[i] = ob;
        ow += (W / 32) * ks * 128;
        ob += W;
    }
    if (rgb) {
        L.w_view = ow; ow += 4 * (W / 16 + 2) * 128;
    }
    L.total_h8 = ow;
    L.total_b = ob;
    return L;
}
// closed forms of the per-layer entries (the kernel indexes layers at run time: no private array in scratch)
__host__ __device__ constexpr int ks_of(int W, int i) { return i == 0 ? 32 : (W / 16 + (i == 5 ? 32 : 0)); }
__host__ __device__ constexpr int woff_of(int W, int i) {
    return i == 0 ? 0 : (W / 32) * 128 * (32 + (i - 1) * (W / 16) + (i > 5 ? 32 : 0));
}
// heads (fp32): density w[W] | density b (4) | rgb w[3][128] | rgb b (4)   (same as mlp_mip.hip)
__host__ __device__ inline int hd_db(int W) { return W; }
__host__ __device__ inline int hd_rw(int W) { return W + 4; }
__host__ __device__ inline int hd_rb(int W) { return W + 4 + 384; }

// ---- per-interval set-up shared by the fused evaluator and the encoding producer of the layer-by-layer path ----
// conical frustum -> Gaussian -> contraction of interval i of `ray`: rz[0..2] = contracted mean, rz[3..11] = covariance
__device__ __forceinline__ void row_gaussian(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                             const float* __restrict__ radii, const float* __restrict__ tdist, int ray, int i,
                                             int n, float* __restrict__ rz) {
        const float t0 = tdist[(long)ray * (n + 1) + i], t1 = tdist[(long)ray * (n + 1) + i + 1];
        float o[3], d[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { o[a] = rays_o[ray * 3 + a]; d[a] = rays_d[ray * 3 + a]; }
        const float rad = radii[ray];
        // conical_frustum_to_gaussian (helper.py:293-306)
        const float mu = (t0 + t1) / 2.0f, hw = (t1 - t0) / 2.0f;
        const float mu2 = mu * mu, hw2 = hw * hw;
        const float denom = fmaxf(3.0f * mu2 + hw2, EPS32);
        const float t_mean = mu + (2.0f * mu * hw2) / denom;
        const float hw4 = hw2 * hw2;
        const float t_var = hw2 / 3.0f - (4.0f / 15.0f) * hw4 * (12.0f * mu2 - hw2) / (denom * denom);
        float r_var = mu2 / 4.0f + (5.0f / 12.0f) * hw2 - (4.0f / 15.0f) * hw4 / denom;
        r_var = r_var * (rad * rad);
        // lift_gaussian, diag=False (helper.py:320-334)
        float x[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) x[a] = d[a] * t_mean + o[a];
        const float dm = fmaxf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], 1e-10f);
        float cov[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float outer = d[a] * d[b];
                const float null_o = (a == b ? 1.0f : 0.0f) - d[a] * (d[b] / dm);
                cov[a][b] = t_var * outer + r_var * null_o;
            }
        // contract (helper.py:33-66): z, J = dz/dx (closed form of the reference's autograd Jacobian)
        const float msq = fmaxf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2], 1e-32f);
        float z[3], J[3][3];
        if (msq <= 1.0f) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                z[a] = x[a];
#pragma unroll
                for (int b = 0; b < 3; ++b) J[a][b] = a == b ? 1.0f : 0.0f;
            }
        } else {
            const float rt = sqrtf(msq);
            const float sc = (2.0f * rt - 1.0f) / msq;
            const float coef = 2.0f / (msq * rt) - 2.0f * sc / msq;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                z[a] = sc * x[a];
#pragma unroll
                for (int b = 0; b < 3; ++b) J[a][b] = (a == b ? sc : 0.0f) + coef * x[a] * x[b];
            }
        }
        float tmp[3][3], cc[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) tmp[a][b] = J[a][0] * cov[0][b] + J[a][1] * cov[1][b] + J[a][2] * cov[2][b];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) cc[a][b] = tmp[a][0] * J[b][0] + tmp[a][1] * J[b][1] + tmp[a][2] * J[b][2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            rz[a] = z[a];
#pragma unroll
            for (int b = 0; b < 3; ++b) rz[3 + a * 3 + b] = cc[a][b];
        }
}

// view-direction encoding of `ray` into row `row` of the [32][32] split tile, append_identity=True (helper.py:92-99):
// [d | sin(d 2^k) | sin(d 2^k + pi/2)], k < 4, zero-padded to 32
__device__ __forceinline__ void row_direnc(const float* __restrict__ viewdirs, int ray, int row, const HT& dsm) {
            auto put = [&](int f, float v) {
                _Float16 h, l;
                split(v, h, l);
                const int o2 = chunk_off<32>(row, f >> 3) + (f & 7);
                dsm.hi[o2] = h;
                dsm.lo[o2] = l;
            };
            float vd[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { vd[a] = viewdirs[ray * 3 + a]; put(a, vd[a]); }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float sn, cs;
                    enc_pair(vd[a], k, sn, cs);
                    put(3 + k * 3 + a, sn);
                    put(15 + k * 3 + a, cs);
                }
#pragma unroll
            for (int f = 27; f < 32; ++f) put(f, 0.0f);
}

// lift_and_diagonalize (helper.py:70-73): mean_j = z . b_j ; var_j = sum_i b_ij (cov b_j)_i  -> lift[row][44]
__device__ __forceinline__ void lift_rows(const float* __restrict__ basis, const float* __restrict__ rowz, float* __restrict__ lift,
                                          int tid, int nthreads, int rows = TMR) {
    for (int idx = tid; idx < rows * NB; idx += nthreads) {
        const int row = idx / NB, j = idx - row * NB;
        const float b0 = basis[j], b1 = basis[NB + j], b2 = basis[2 * NB + j];
        const float* rz = rowz + row * 12;
        const float mj = rz[0] * b0 + rz[1] * b1 + rz[2] * b2;
        float vj = 0.f;
        const float bb[3] = {b0, b1, b2};
#pragma unroll
        for (int a = 0; a < 3; ++a) vj += bb[a] * (rz[3 + a * 3] * b0 + rz[3 + a * 3 + 1] * b1 + rz[3 + a * 3 + 2] * b2);
        lift[row * 44 + j] = mj;
        lift[row * 44 + NB + j] = vj;
    }
}

// integrated_pos_enc (helper.py:77-88) in PAIR ORDER: packed features 2 g, 2 g + 1 = the reference's features g and 252 + g
// (g = octave k * 21 + direction j < 252): exp(-var 4^k / 2) * sin(a) and exp(...) * sin(fl32(a + fl32(pi / 2))), a = mean 2^k.
// The two share the exponential and ONE argument reduction (common.h:sincos_pair reproduces the fp32 rounding of a + pi/2):
// ~50 VALU instructions per pair instead of ~80 for two separate features.  launch_mip_pack_h packs the encoding's weight
// columns in the same order.  g >= 252: the zero padding (504 -> 512).
__device__ __forceinline__ void ipe_pair(const float* __restrict__ lift, int row, int g, float& f0, float& f1) {
    f0 = 0.0f;
    f1 = 0.0f;
    if (g < 252) {
        const int k = g / NB, j = g - k * NB;
        const float mean = lift[row * 44 + j], var = lift[row * 44 + NB + j];
        float sn, cs;
        sincos_pair(ldexpf(mean, k), sn, cs);
        const float ex = expf(-0.5f * ldexpf(var, 2 * k));
        f0 = ex * sn;
        f1 = ex * cs;
    }
}

// acc[nt] += W-stage k-steps [ks0, ks0+n) x tile k-steps [tks0, tks0+n); N-tiles nt0..nt0+NTW-1, the one M-tile.
// wb = byte address of the stage's fragments (uniform); fragments are addressed SGPR base + 32-bit VGPR offset.
template <int NTW, int LDH>
__device__ __forceinline__ void gemm_h(f32x16 (&acc)[NTW], const char* __restrict__ wb, int KS, int nt0, int ks0,
                                       int tks0, int n, const HT& tile, const LaneCtx& L) {
    h8 ah[2][NTW], al[2][NTW];
    uint32_t off[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) off[nt] = (uint32_t)(((nt0 + nt) * KS + ks0) * 128 + L.lane) * 16u;
    auto load_w = [&](int slot, int s) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            ah[slot][nt] = *reinterpret_cast<const h8*>(wb + (off[nt] + 2048u * s));
            al[slot][nt] = *reinterpret_cast<const h8*>(wb + (off[nt] + 2048u * s + 1024u));
        }
    };
    load_w(0, 0);
#pragma unroll 1
    for (int s = 0; s < n; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (s + u < n) {
                if (s + u + 1 < n) load_w((u + 1) & 1, s + u + 1);
                const int o = chunk_off<LDH>(L.l31, ((tks0 + s + u) << 1) + L.half);
                const h8 bh = *reinterpret_cast<const h8*>(tile.hi + o);
                const h8 bl = *reinterpret_cast<const h8*>(tile.lo + o);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    acc[nt] = NEO_MFMA_H(al[u][nt], bh, acc[nt]);
                    acc[nt] = NEO_MFMA_H(ah[u][nt], bl, acc[nt]);
                    acc[nt] = NEO_MFMA_H(ah[u][nt], bh, acc[nt]);
                }
            }
        }
    }
}

// one accumulator tile, N k-steps fully unrolled, weight fragments in a ring D deep (k-step k + D - 1 is requested before
// k-step k is multiplied): the colour branch's short K ranges are latency-bound with one fragment in flight
template <int LDH, int N, int D>
__device__ __forceinline__ void gemm_ring(f32x16& acc, const char* __restrict__ wb, int KS, int nt, int ks0, int tks0,
                                          const HT& tile, const LaneCtx& L) {
    h8 ah[D], al[D];
    const char* p = wb + (size_t)((uint32_t)((nt * KS + ks0) * 128 + L.lane) * 16u);
#pragma unroll
    for (int i = 0; i < D - 1; ++i) {
        ah[i] = *reinterpret_cast<const h8*>(p + 2048 * i);
        al[i] = *reinterpret_cast<const h8*>(p + 2048 * i + 1024);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k + D - 1 < N) {
            ah[(k + D - 1) % D] = *reinterpret_cast<const h8*>(p + 2048 * (k + D - 1));
            al[(k + D - 1) % D] = *reinterpret_cast<const h8*>(p + 2048 * (k + D - 1) + 1024);
        }
        const int o = chunk_off<LDH>(L.l31, ((tks0 + k) << 1) + L.half);
        const h8 bh = *reinterpret_cast<const h8*>(tile.hi + o);
        const h8 bl = *reinterpret_cast<const h8*>(tile.lo + o);
        acc = NEO_MFMA_H(al[k % D], bh, acc);
        acc = NEO_MFMA_H(ah[k % D], bl, acc);
        acc = NEO_MFMA_H(ah[k % D], bh, acc);
    }
}

// NWV waves per workgroup (8, or 16 for the 1024-wide MLP: 4 waves per SIMD, 128 VGPRs, two accumulator tiles each)
// TAIL: the trunk has been run by the layer-by-layer path; its output (fragment order, interval tile blockIdx.x of the
// batch that starts at interval p0) is `yin`, this kernel adds the heads and the colour branch
template <int W, int DEPTH, bool RGB, int NWV, bool TAIL = false>
__global__ __launch_bounds__(NWV * 64, (NWV == 16 ? 4 : (W == 1024 ? 2 : 4))) void k_mip_mlp_h(MipMlpHDev m, const float* __restrict__ rays_o,
                                                                        const float* __restrict__ rays_d,
                                                                        const float* __restrict__ viewdirs,
                                                                        const float* __restrict__ radii,
                                                                        const float* __restrict__ tdist, int R, int n,
                                                                        float4* __restrict__ out,
                                                                        const char* __restrict__ yin = nullptr, long p0 = 0) {
    constexpr int NTW = W / (32 * NWV);      // N-tiles per wave for W-wide layers (NWV waves x NTW x 32 = W)
    constexpr int NT = NWV * 64;             // threads
    constexpr int FPT = 2048 / NT;           // encoding features per thread and stage (32 rows x 64 features)
    constexpr int KSW = W / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* hb = reinterpret_cast<_Float16*>(smem);
    const HT act{hb, hb + TMR * W};                                             // [32][W] x 2 planes
    _Float16* xb = hb + 2 * TMR * W;
    auto xbuf = [&](int b) { return HT{xb + b * (2 * TMR * 64), xb + b * (2 * TMR * 64) + TMR * 64}; };   // 2 x [32][64] x 2 planes
    _Float16* db = xb + 4 * TMR * 64;
    const HT dsm{db, db + TMR * 32};                                            // [32][32] x 2 planes (rgb branch)
    float* lift = reinterpret_cast<float*>(db + 2 * TMR * 32);                  // [32][44]: 21 lifted means | 21 variances
    float* rowz = lift + TMR * 44;                                              // [32][12]: contracted mean | covariance
    LaneCtx L;
    L.init();
    int tid = threadIdx.x;
    const long P = (long)R * n;
    const long tile0 = p0 + (long)blockIdx.x * TMR;
    constexpr int W_VIEW = woff_of(W, DEPTH - 1) + (W / 32) * ks_of(W, DEPTH - 1) * 128;
    const char* wbase = reinterpret_cast<const char*>(m.wpack);

    // ---- per-row Gaussian, contraction (identical arithmetic to mlp_mip.hip) ---------------------------
    if (tid < TMR) {
        long g = tile0 + tid;
        if (g >= P) g = P - 1;
        const int ray = (int)(g / n), i = (int)(g - (long)ray * n);
        if (!TAIL) row_gaussian(rays_o, rays_d, radii, tdist, ray, i, n, rowz + tid * 12);
        if (RGB) row_direnc(viewdirs, ray, tid, dsm);
    }
    __syncthreads();
    if (!TAIL) lift_rows(m.basis, rowz, lift, tid, NT);
    else {
        // trunk output of the layer-by-layer path: fragment (k-step, plane, lane = (half, interval)) -> 16-byte chunk
        // 2 ks + half of row `interval` in the swizzled tile; this workgroup's interval tile is blockIdx.x of the batch
        const char* yt = yin + (size_t)blockIdx.x * (KSW * 2048);
        for (int c = tid; c < KSW * 128; c += NT) {
            const int ks = c >> 7, plane = (c >> 6) & 1, ln = c & 63;
            const h8 v = *reinterpret_cast<const h8*>(yt + (size_t)c * 16);
            *reinterpret_cast<h8*>((plane ? act.lo : act.hi) + chunk_off<W>(ln & 31, 2 * ks + (ln >> 5))) = v;
        }
    }
    __syncthreads();

    // integrated_pos_enc (helper.py:77-88): 64 packed features (32 pairs) of stage s; thread = (row, FPT consecutive features)
    auto produce = [&](int s, const HT& buf) {
        const int row = tid & 31, q = tid >> 5;              // q: group of FPT consecutive features
        static_assert(FPT % 2 == 0, "whole (sin, shifted sin) pairs per thread");
        _Float16 vh[FPT], vl[FPT];
#pragma unroll
        for (int e = 0; e < FPT; e += 2) {
            float f0, f1;
            ipe_pair(lift, row, (s * 64 + q * FPT + e) >> 1, f0, f1);
            split(f0, vh[e], vl[e]);
            split(f1, vh[e + 1], vl[e + 1]);
        }
        const int o = chunk_off<64>(row, (q * FPT) >> 3) + ((q * FPT) & 7);
#pragma unroll
        for (int e = 0; e < FPT; ++e) { buf.hi[o + e] = vh[e]; buf.lo[o + e] = vl[e]; }
    };
    // acc += W_layer[:, ksbase .. ksbase + 32 k-steps] * ipe^T, streamed through the double buffer
    auto stream_ipe = [&](f32x16 (&acc)[NTW], const char* wl, int KS, int ksbase) {
        produce(0, xbuf(0));
        __syncthreads();
#pragma unroll 1
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) produce(s + 1, xbuf((s + 1) & 1));
            gemm_h<NTW, 64>(acc, wl, KS, L.wv * NTW, ksbase + s * 4, 0, 4, xbuf(s & 1), L);
            __syncthreads();
        }
    };

    f32x16 acc[NTW];
    // ---- trunk ----
#pragma unroll 1
    for (int layer = TAIL ? DEPTH : 0; layer < DEPTH; ++layer) {
        // per-lane indices re-derived from an opaque lane id: keeps swizzled LDS addresses out of scratch
        asm volatile("" : "+v"(tid));
        L.lane = tid & 63;
        L.half = L.lane >> 5;
        L.l31 = L.lane & 31;
        L.key = L.lane & 15;
        const char* wl = wbase + (size_t)woff_of(W, layer) * 16;
        const int KS = ks_of(W, layer);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) bias_tile(acc[nt], m.bias + layer * W, L.wv * NTW + nt, L);
        if (layer == 0) {
            stream_ipe(acc, wl, KS, 0);
        } else {
            gemm_h<NTW, W>(acc, wl, KS, L.wv * NTW, 0, 0, KSW, act, L);
            if (layer == 5) stream_ipe(acc, wl, KS, KSW);    // skip concat: [h | ipe]
            __syncthreads();
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) store_tile_h<true, W>(acc[nt], act, L.wv * NTW + nt, 0, L);
        __syncthreads();
    }
    // ---- density head (VALU): 16 lanes per row, W/16 channels each ----
    float raw_density = 0.f;
    if (tid < 512) {
        const int row = tid >> 4, part = tid & 15;
        constexpr int CH = W / 8 / 16;      // 8-half chunks per lane
        float s = 0.f;
#pragma unroll 2
        for (int c = 0; c < CH; ++c) {
            const int chunk = part * CH + ((c + part) % CH);
            const int o = chunk_off<W>(row, chunk);
            const h8 vh = *reinterpret_cast<const h8*>(act.hi + o);
            const h8 vl = *reinterpret_cast<const h8*>(act.lo + o);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(m.heads + chunk * 8);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(m.heads + chunk * 8 + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float w = e < 4 ? w0[e] : w1[e - 4];
                s = __builtin_fmaf((float)vh[e], w, s);
                s = __builtin_fmaf((float)vl[e], w, s);
            }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
        raw_density = s + m.heads[hd_db(W)];
    }
    float r = 0.f, g = 0.f, b = 0.f;
    if (RGB) {
        // ---- views_linear.0 with the bottleneck folded in: [h W | dir enc 27 -> 32] -> 128, ReLU.  Wave = (N-tile wv & 3,
        // K quarter wv >> 2): quarter q multiplies k-steps [16 q, 16 q + 16) of h, quarter 3 the two direction k-steps as
        // well; quarters 1..3 hand their partial tiles to quarter 0 through the (now dead) activation tile as fp32 ----
        if constexpr (NWV == 16) {
            constexpr int KSV = KSW + 2, KQ = KSW / 4;
            const int vnt = L.wv & 3, kq = L.wv >> 2;
            f32x16 ab[1];
            if (kq == 0) bias_tile(ab[0], m.view_bias, vnt, L);
            else
#pragma unroll
                for (int e = 0; e < 16; ++e) ab[0][e] = 0.0f;
            gemm_ring<W, KQ, 4>(ab[0], wbase + (size_t)W_VIEW * 16, KSV, vnt, kq * KQ, kq * KQ, act, L);
            if (kq == 3) gemm_h<1, 32>(ab, wbase + (size_t)W_VIEW * 16, KSV, vnt, KSW, 0, 2, dsm, L);
            __syncthreads();                                   // every read of the trunk output (density head too) is done
            float* part = reinterpret_cast<float*>(hb);        // 12 partial tiles x 4 KB, D layout lane for lane
            if (kq > 0) {
                float* ps = part + ((kq - 1) * 4 + vnt) * 1024 + L.lane * 4;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<f32x4*>(ps + gq * 256) = f32x4{ab[0][4 * gq], ab[0][4 * gq + 1], ab[0][4 * gq + 2], ab[0][4 * gq + 3]};
            }
            __syncthreads();
            const HT vt{xb, xb + TMR * 128};                   // [32][128] x 2 planes in the (idle) encoding stage buffers
            if (kq == 0) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float* ps = part + (j * 4 + vnt) * 1024 + L.lane * 4;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(ps + gq * 256);
#pragma unroll
                        for (int e = 0; e < 4; ++e) ab[0][4 * gq + e] += v[e];
                    }
                }
                store_tile_h<true, 128>(ab[0], vt, vnt, 0, L);
            }
            __syncthreads();
            // ---- rgb head: 16 lanes per row, 8 features each ----
            const int row = (tid >> 4) & 31, part16 = tid & 15;     // (threads >= 512 repeat rows; only tid < 512 writes)
            const float* wr = m.heads + hd_rw(W);
            const int o = chunk_off<128>(row, part16);
            const h8 vh = *reinterpret_cast<const h8*>(vt.hi + o);
            const h8 vl = *reinterpret_cast<const h8*>(vt.lo + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float h = (float)vh[e] + (float)vl[e];
                r += h * wr[part16 * 8 + e];
                g += h * wr[128 + part16 * 8 + e];
                b += h * wr[256 + part16 * 8 + e];
            }
        }
#pragma unroll
        for (int o2 = 1; o2 < 16; o2 <<= 1) {
            r += __shfl_xor(r, o2, 64);
            g += __shfl_xor(g, o2, 64);
            b += __shfl_xor(b, o2, 64);
        }
    }
    range_commit(L, m.flags);        // split range guard (split_tile.h): IPE features are bounded by 1, activations are not
    {
        const int row = tid >> 4, part = tid & 15;
        const long gi = tile0 + row;
        if (tid < 512 && part == 0 && gi < P) {
            float4 o4;
            if (RGB) {
                o4.x = colour_act(r + m.heads[hd_rb(W)]);
                o4.y = colour_act(g + m.heads[hd_rb(W) + 1]);
                o4.z = colour_act(b + m.heads[hd_rb(W) + 2]);
            } else {
                o4.x = o4.y = o4.z = 0.0f;       // disable_rgb: zeros_like(means) (model.py:131-136)
            }
            o4.w = density_act(raw_density);     // softplus(raw + density_bias), density_bias = -1
            out[gi] = o4;
        }
    }
}

// Encoding producer of the layer-by-layer path: one interval tile (32 intervals) of the batch that starts at interval p0 per
// workgroup; the 504 features (+ 8 zeros) as fragments of 32 k-steps.  Thread (row, q) makes features 64 s + 8 q + 0..7 of
// stage s = chunk q of that stage = k-step 4 s + q/2, fragment half q & 1: one 16-byte store per plane.  Same feature
// order, same arithmetic as `produce` in the fused evaluator.
__global__ __launch_bounds__(256) void k_mip_ipe_h(const float* __restrict__ basis, const float* __restrict__ rays_o,
                                                   const float* __restrict__ rays_d, const float* __restrict__ radii,
                                                   const float* __restrict__ tdist, int R, int n, long p0,
                                                   char* __restrict__ x0) {
    __shared__ float rowz[TMR * 12];
    __shared__ float lift[TMR * 44];
    const int tid = threadIdx.x;
    const long P = (long)R * n;
    if (tid < TMR) {
        long g = p0 + (long)blockIdx.x * TMR + tid;
        if (g >= P) g = P - 1;                                   // padding intervals repeat the last one (finite values)
        const int ray = (int)(g / n), i = (int)(g - (long)ray * n);
        row_gaussian(rays_o, rays_d, radii, tdist, ray, i, n, rowz + tid * 12);
    }
    __syncthreads();
    lift_rows(basis, rowz, lift, tid, 256);
    __syncthreads();
    const int row = tid & 31, q = tid >> 5;
    char* xt = x0 + (size_t)blockIdx.x * (32 * 2048) + (size_t)((q & 1) * 32 + row) * 16;
#pragma unroll 1
    for (int s = 0; s < 8; ++s) {
        h8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float f0, f1;
            ipe_pair(lift, row, (s * 64 + q * 8 + e) >> 1, f0, f1);
            _Float16 h, l;
            split(f0, h, l);
            vh[e] = h;
            vl[e] = l;
            split(f1, h, l);
            vh[e + 1] = h;
            vl[e + 1] = l;
        }
        char* p = xt + (size_t)(s * 4 + (q >> 1)) * 2048;
        *reinterpret_cast<h8*>(p) = vh;
        *reinterpret_cast<h8*>(p + 1024) = vl;
    }
}

// Trunk-layer fragments with the encoding's 504 source columns [col0, col0 + 504) packed in pair order at packed k in
// [k_enc, k_enc + 512): packed k_enc + 2 g + h <- column col0 + 252 h + g (g < 252), zeros behind; every other packed k < kin
// maps to itself (the hidden features of the skip layer).  Same fragment layout as pack_h (split_tile.h).
__global__ void k_mip_pack_layer_h(const float* __restrict__ src, int ld, int rows, int KS, int k_enc, int col0, int k_plain,
                                   _Float16* __restrict__ dst) {
    const int total = (rows / 32) * KS * 512;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, blk = idx >> 9;
        const int ks = blk % KS, ntl = blk / KS;
        const int n = ntl * 32 + (lane & 31);
        const int k = ks * 16 + 8 * (lane >> 5) + e;
        float w = 0.0f;
        if (k >= k_enc && k < k_enc + 504) {
            const int p = k - k_enc;
            w = src[(long)n * ld + col0 + 252 * (p & 1) + (p >> 1)];
        } else if (k < k_plain) {
            w = src[(long)n * ld + k];
        }
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        const long base = ((long)(ntl * KS + ks) * 2) * 512 + lane * 8 + e;
        dst[base] = hi;
        dst[base + 512] = lo;
    }
}

template <int W>
size_t lds_bytes() {
    return (size_t)(2 * TMR * W + 4 * TMR * 64 + 2 * TMR * 32) * sizeof(_Float16) + (size_t)(TMR * 44 + TMR * 12) * sizeof(float);
}

}  // namespace

size_t mip_wpack_h_bytes(int width, int depth, int rgb) { return (size_t)mip_layout_h(width, depth, rgb).total_h8 * 16; }

size_t mip_fold_floats() { return (size_t)128 * (1024 + 27); }

void launch_mip_pack_h(int width, int depth, int rgb, const float* const* w, const float* const* b, void* wpack_h, float* fold_ws,
                       float* view_bias, hipStream_t s) {
    const MipLayoutH lay = mip_layout_h(width, depth, rgb);
    _Float16* base = reinterpret_cast<_Float16*>(wpack_h);
    const PackSegs none = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < depth; ++i) {
        const int kin = i == 0 ? 504 : (i == 5 ? width + 504 : width);
        if (i == 0 || i == 5) {
            // the encoding's columns in pair order (ipe_pair): layer 0 = [enc], layer 5 = [h | enc] (model.py:76-79)
            const int k_enc = i == 0 ? 0 : width;
            const int total = (width / 32) * lay.ks_layer[i] * 512;
            hipLaunchKernelGGL(k_mip_pack_layer_h, dim3((total + 255) / 256), dim3(256), 0, s, w[i], kin, width, lay.ks_layer[i], k_enc,
                               k_enc, k_enc, base + (size_t)lay.w_layer[i] * 8);
        } else {
            PackSegs sg = none;
            sg.len[0] = kin;
            pack_h(w[i], kin, width, lay.ks_layer[i], 0, sg, base + (size_t)lay.w_layer[i] * 8, s);
        }
    }
    if (rgb) {
        // w / b order: trunk, density_layer, bottleneck_layer (256 x width), views_linear.0 (128 x 283), rgb_layer
        launch_fold_bottleneck(w[depth + 2], w[depth + 1], b[depth + 1], b[depth + 2], 128, 256, width, 27, fold_ws, view_bias, s);
        PackSegs sv = none;
        sv.len[0] = width + 27;
        pack_h(fold_ws, width + 27, 128, width / 16 + 2, 0, sv, base + (size_t)lay.w_view * 8, s);
    }
}

int launch_mip_mlp_h(int width, int depth, int rgb, const MipMlpHDev& m, const float* rays_o, const float* rays_d,
                     const float* viewdirs, const float* radii, const float* tdist, int R, int n, float* out,
                     hipStream_t s) {
    const long P = (long)R * n;
    if (P <= 0) return 0;
    const long tiles = (P + TMR - 1) / TMR;
    // the attribute is per device: set it on every launch (a host-side table write) rather than once per process
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mip_mlp_h<1024, 8, true, NEO_MIP_H_WAVES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<1024>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mip_mlp_h<256, 4, false, 8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<256>());
    if (width == 1024 && depth == 8 && rgb)
        hipLaunchKernelGGL((k_mip_mlp_h<1024, 8, true, NEO_MIP_H_WAVES>), dim3((unsigned)tiles), dim3(NEO_MIP_H_WAVES * 64), lds_bytes<1024>(), s, m,
                           rays_o, rays_d, viewdirs, radii, tdist, R, n, reinterpret_cast<float4*>(out));
    else if (width == 256 && depth == 4 && !rgb) {
        hipLaunchKernelGGL((k_mip_mlp_h<256, 4, false, 8>), dim3((unsigned)tiles), dim3(512), lds_bytes<256>(), s, m,
                           rays_o, rays_d, viewdirs, radii, tdist, R, n, reinterpret_cast<float4*>(out));
    }
    else
        return -1;
    return 0;
}

int launch_mip_mlp_h_layered(const MipMlpHDev& m, const MipLayeredWs& ws, const float* rays_o, const float* rays_d,
                             const float* viewdirs, const float* radii, const float* tdist, int R, int n, float* out,
                             hipStream_t s) {
    constexpr int W = 1024, DEPTH = 8;
    const long P = (long)R * n;
    if (P <= 0) return 0;
    if (!ws.x0 || !ws.ya || !ws.yb || ws.cap < 2048 || ws.cap % 2048) return -1;
    auto tail = k_mip_mlp_h<W, DEPTH, true, NEO_MIP_H_WAVES, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tail), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<W>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mip_gemm_h<true>), hipFuncAttributeMaxDynamicSharedMemorySize, MG_LDS_BYTES);
    const char* wbase = reinterpret_cast<const char*>(m.wpack);
    char* bufs[2] = {ws.ya, ws.yb};
    for (long p0 = 0; p0 < P; p0 += ws.cap) {
        const long pb = P - p0 < ws.cap ? P - p0 : ws.cap;
        const int n_it = (int)((pb + 2047) / 2048) * 64;            // the GEMM's grid wants whole groups of 64 interval tiles
        hipLaunchKernelGGL(k_mip_ipe_h, dim3((unsigned)n_it), dim3(256), 0, s, m.basis, rays_o, rays_d, radii, tdist, R, n, p0, ws.x0);
        for (int l = 0; l < DEPTH; ++l) {
            MipGemmArgs a{};
            a.w = wbase + (size_t)woff_of(W, l) * 16;
            a.bias = m.bias + l * W;
            a.x0 = l == 0 ? ws.x0 : bufs[(l + 1) & 1];
            a.ks0 = l == 0 ? 32 : W / 16;
            a.x1 = l == 5 ? ws.x0 : nullptr;                        // skip concat [h | encoding] (model.py:76-79)
            a.ks1 = l == 5 ? 32 : 0;
            a.y = bufs[l & 1];
            a.n_it = n_it;
            a.flags = m.flags;
            hipLaunchKernelGGL(k_mip_gemm_h<true>, dim3(mip_gemm_grid(n_it)), dim3(MG_THREADS), MG_LDS_BYTES, s, a);
        }
        hipLaunchKernelGGL(tail, dim3((unsigned)((pb + TMR - 1) / TMR)), dim3(NEO_MIP_H_WAVES * 64), lds_bytes<W>(), s, m, rays_o,
                           rays_d, viewdirs, radii, tdist, R, n, reinterpret_cast<float4*>(out),
                           static_cast<const char*>(bufs[(DEPTH - 1) & 1]), p0);
    }
    return 0;
}

}  // namespace neo

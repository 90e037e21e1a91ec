// NeO-360 decoder point evaluator, split-fp16 arithmetic, ALL THREE source views resident (the reference default
// NV = 3): the arithmetic and data flow of mlp_tp_h.hip with the view loop turned inside out.  One workgroup =
// 8 waves = one 64-point tile x 3 views; each view has its own LDS stage / activation tile, direction encoding and
// tap descriptors (49 KB per view, 151 KB in all, one workgroup per CU).  Every weight fragment is fetched once per
// tile and multiplied against the three views' tiles (18 MFMAs per fragment pair instead of 6), the streamed input
// needs one barrier per stage per TILE instead of per tile-view, and the gathers / blends / encodings of the three
// views are spread over 512 threads.
//   wave w: N-tile w&3 (and 4 + (w&3) = the L3 skip half in the streamed GEMM), M-tile w>>2, all three views.
// Opt-in ($NEO_TP_BATCHED=1): on MI355X it measured 5-10 % SLOWER than the view-loop kernel (17.8 vs 17.0 ms on
// 8192 rays x 385 points, fg; 17.0 vs 15.4 ms bg) — with one workgroup per CU all eight waves sit in the same phase
// (gather/blend VALU vs MFMA), and the saved weight traffic was not the limiter.
// Numerics are those of mlp_tp_h.hip (same products, same summation order per accumulator); the two kernels agree
// bit for bit, which tests/test_gpu_repeatable.py checks.  Compiled without packed-fp32 VALU ops (build.py).
#include <type_traits>

#include "split_tile.h"
#include "tp_common.h"

namespace neo {

namespace {

using tp::TM;
using tp::blend4;
using tp::pe_feature;

constexpr int NVT = 3;

// packed weight layout: identical to mlp_tp_h.hip (the same wpack_h buffer is used)
__host__ __device__ constexpr int pe_ksteps(int pe_c) { return pe_c == 3 ? 4 : 6; }
__host__ __device__ constexpr int ks_x(int pe_c) { return 32 + 8 + pe_ksteps(pe_c); }
__host__ __device__ constexpr int hoff_1(int pe_c) { return 8 * ks_x(pe_c) * 128; }
__host__ __device__ constexpr int hoff_2(int pe_c) { return hoff_1(pe_c) + 4 * 8 * 128; }
__host__ __device__ constexpr int hoff_3a(int pe_c) { return hoff_2(pe_c) + 4 * 8 * 128; }
__host__ __device__ constexpr int hoff_b(int pe_c) { return hoff_3a(pe_c) + 4 * 8 * 128; }
__host__ __device__ constexpr int hoff_v0(int pe_c) { return hoff_b(pe_c) + 4 * 8 * 128; }
__host__ __device__ constexpr int hoff_v1(int pe_c) { return hoff_v0(pe_c) + 2 * 10 * 128; }
constexpr int B_0 = 0, B_3 = 128, B_1 = 256, B_2 = 384, B_B = 512, B_V0 = 640, B_V1 = 704;
constexpr int HD_DW = 0, HD_DB = 128, HD_RW = 132, HD_RB = 324;

// LDS carve (bytes)
constexpr int V_ACT = 0, V_DIR = 32768, V_LOC_OFF = 40960, V_LOC_W = 41984, V_PL_OFF = 43008, V_PL_W = 46080,
              V_CAM = 49152, VIEW_BYTES = 50176;
constexpr int S_PE = NVT * VIEW_BYTES, S_FEAT = S_PE + 1024, S_VDIR = S_FEAT + 1024, S_DENSW = S_VDIR + 1024,
              LDS_BYTES = S_DENSW + 512;
constexpr int VIEW_HALVES = VIEW_BYTES / 2;     // view stride of the fp16 tiles

// acc[v] += W (N-tile nt, k-steps [ks0, ks0+n)) x tile_v^T (M-tile mt, tile k-steps [0, n)) for the three views
template <int LDH>
__device__ __forceinline__ void gemm_v(f32x16 (&acc)[NVT], const char* __restrict__ wb, int KS, int nt, int mt, int ks0,
                                       int n, const HT& tile0, const LaneCtx& L) {
    h8 ah[2], al[2];
    const uint32_t off = (uint32_t)((nt * KS + ks0) * 128 + L.lane) * 16u;
    auto load_w = [&](int slot, int s) {
        ah[slot] = *reinterpret_cast<const h8*>(wb + (off + 2048u * s));
        al[slot] = *reinterpret_cast<const h8*>(wb + (off + 2048u * s + 1024u));
    };
    load_w(0, 0);
#pragma unroll 1
    for (int s = 0; s < n; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (s + u < n) {
                if (s + u + 1 < n) load_w((u + 1) & 1, s + u + 1);
                const int o = chunk_off<LDH>(mt * 32 + L.l31, ((s + u) << 1) + L.half);
#pragma unroll
                for (int v = 0; v < NVT; ++v) {
                    const h8 bh = *reinterpret_cast<const h8*>(tile0.hi + v * VIEW_HALVES + o);
                    const h8 bl = *reinterpret_cast<const h8*>(tile0.lo + v * VIEW_HALVES + o);
                    acc[v] = NEO_MFMA_H(al[u], bh, acc[v]);
                    acc[v] = NEO_MFMA_H(ah[u], bl, acc[v]);
                    acc[v] = NEO_MFMA_H(ah[u], bh, acc[v]);
                }
            }
        }
    }
}

template <int PE_C>
__global__ __launch_bounds__(512, 2) void k_tp_mlp_hv(TpMlpHDev m, TpScene sc, TpViews views,
                                                       const float* __restrict__ rays_o,
                                                       const float* __restrict__ rays_d,
                                                       const float* __restrict__ viewdirs,
                                                       const float* __restrict__ tvals,
                                                       const float* __restrict__ far_arr, int R, int N, int chunk,
                                                       uint32_t* __restrict__ flags, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // per-view LDS objects as pure functions of the view index (no pointer arrays: a run-time index would put
    // them in scratch)
    auto act = [&](int v) {
        _Float16* hb = reinterpret_cast<_Float16*>(lds + v * VIEW_BYTES + V_ACT);
        return HT{hb, hb + TM * 128};
    };
    auto dsm = [&](int v) {
        _Float16* db = reinterpret_cast<_Float16*>(lds + v * VIEW_BYTES + V_DIR);
        return HT{db, db + TM * 32};
    };
    auto scratch = [&](int v) {
        char* vb = lds + v * VIEW_BYTES;
        tp::Scratch S;
        S.loc_off = reinterpret_cast<int*>(vb + V_LOC_OFF);
        S.loc_w = reinterpret_cast<float*>(vb + V_LOC_W);
        S.pl_off = reinterpret_cast<int*>(vb + V_PL_OFF);
        S.pl_w = reinterpret_cast<float*>(vb + V_PL_W);
        S.cam_enc = reinterpret_cast<float*>(vb + V_CAM);
        S.pe_world = reinterpret_cast<float*>(lds + S_PE);
        S.feat_world = reinterpret_cast<float*>(lds + S_FEAT);
        S.vdir_world = reinterpret_cast<float*>(lds + S_VDIR);
        return S;
    };
    float* dens_w = reinterpret_cast<float*>(lds + S_DENSW);
    // stage buffer b of view v: [64][64] x 2 planes, aliasing that view's activation tile
    auto xbuf = [&](int v, int b) {
        _Float16* hb = act(v).hi + b * (2 * TM * 64);
        return HT{hb, hb + TM * 64};
    };

    LaneCtx L;
    L.init();
    const int tid = threadIdx.x;
    const long P = (long)R * N;
    const long tile0 = tp::xcd_tile(blockIdx.x, (P + TM - 1) / TM) * TM;      // contiguous tile range per XCD (tp_common.h)
    if (tile0 >= P) return;       // surplus workgroup of the rounded-up grid (uniform exit before any barrier)
    const char* wb = reinterpret_cast<const char*>(m.wpack);
    constexpr int KSX = ks_x(PE_C);
    constexpr int NST = PE_C == 3 ? 11 : 12;
    const int nt = L.wv & 3, mt = L.wv >> 2;       // this wave's N-tile / M-tile in every 128-wide layer

    tp::point_setup<PE_C>(scratch(0), tid, tile0, P, N, R, chunk, rays_o, rays_d, viewdirs, tvals, far_arr, flags);
    if (tid < 128) dens_w[tid] = m.heads[HD_DW + tid];
    __syncthreads();
    // ---- per-view descriptors: waves 0-3 take views 0 and 2, waves 4-7 view 1 ----
    {
        LaneCtx Lg = L;
        Lg.wv = L.wv & 3;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int v = 2 * pass + (L.wv >> 2);
            if (v < NVT) {
                const HT d = dsm(v);
                const tp::Scratch Sv = scratch(v);
                // constant indices into the kernel-argument struct (a run-time index would copy it to scratch)
                const float* rot = v == 0 ? views.rot[0] : v == 1 ? views.rot[1] : views.rot[2];
                const float* trn = v == 0 ? views.trans[0] : v == 1 ? views.trans[1] : views.trans[2];
                tp::view_descriptors(Sv, Lg, sc, rot, trn, v, [&](int p, int f, float val) {
                    _Float16 h, l;
                    split(val, h, l);
                    const int o = chunk_off<32>(p, f >> 3) + (f & 7);
                    d.hi[o] = h;
                    d.lo[o] = l;
                });
            }
        }
    }
    __syncthreads();

    // ---- streamed-input GEMM: [L0 | L3 skip half] over 703 / 724 features, three views per weight fragment ----
    f32x16 acc0[NVT], acc3[NVT];
    bias_tile(acc0[0], m.bias + B_0, nt, L);
    bias_tile(acc3[0], m.bias + B_3, nt, L);
#pragma unroll
    for (int v = 1; v < NVT; ++v) { acc0[v] = acc0[0]; acc3[v] = acc3[0]; }
    {
        const int col4 = tid & 15, rg = tid >> 4;          // rg 0..31: one row per thread per half stage, all views
        const uint32_t lane_b = 16u * col4;
        f32x4 tap[NVT][4];
        auto issue_local = [&](int s, int hf) {
            const int row = rg + 32 * hf;
#pragma unroll
            for (int v = 0; v < NVT; ++v)
#pragma unroll
                for (int k = 0; k < 4; ++k) tap[v][k] = tp::load_tap(sc.latent, (uint32_t)scratch(v).loc_off[row * 4 + k] + lane_b + 256u * s);
        };
        auto issue_plane = [&](int j, int s2, int hf) {
            const int row = rg + 32 * hf;
#pragma unroll
            for (int v = 0; v < NVT; ++v)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    tap[v][k] = tp::load_tap(sc.plane[j], (uint32_t)scratch(v).pl_off[(j * TM + row) * 4 + k] + lane_b + 256u * s2);
        };
        auto write_x = [&](const HT& buf, int row, const f32x4 val) {
            h4 vh, vl;
            split4(val, vh, vl);
            const int o = chunk_off<64>(row, col4 >> 1) + 4 * (col4 & 1);
            *reinterpret_cast<h4*>(buf.hi + o) = vh;
            *reinterpret_cast<h4*>(buf.lo + o) = vl;
        };
        auto finish_local = [&](int b, int hf) {
            const int row = rg + 32 * hf;
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                write_x(xbuf(v, b), row, blend4(tap[v], *reinterpret_cast<const f32x4*>(scratch(v).loc_w + row * 4)));
        };
        auto finish_planes = [&](int b, int s2, int hf) {
            const int row = rg + 32 * hf;
            f32x4 sum[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) sum[v] = blend4(tap[v], *reinterpret_cast<const f32x4*>(scratch(v).pl_w + row * 4));
#pragma unroll
            for (int j = 1; j < 3; ++j) {
                issue_plane(j, s2, hf);
#pragma unroll
                for (int v = 0; v < NVT; ++v)
                    sum[v] = sum[v] + blend4(tap[v], *reinterpret_cast<const f32x4*>(scratch(v).pl_w + (j * TM + row) * 4));
            }
#pragma unroll
            for (int v = 0; v < NVT; ++v) write_x(xbuf(v, b), row, sum[v]);
        };
        // pos_enc of the camera-frame point, 64 features of pos_enc stage `pstage`: thread = (row, chunk) of one view;
        // half 0 produces views 0 and 1, half 1 view 2
        auto pe_view = [&](int b, int pstage, int v) __attribute__((always_inline)) {
            const int row = tid & 63, ch = tid >> 6;
            const float* ce = scratch(v).cam_enc;
            const float xc[4] = {ce[row * 4], ce[row * 4 + 1], ce[row * 4 + 2], ce[row * 4 + 3]};
            h8 vh, vl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 h, l;
                split(pe_feature<PE_C>(xc, pstage * 64 + ch * 8 + e), h, l);
                vh[e] = h;
                vl[e] = l;
            }
            const HT buf = xbuf(v, b);
            const int o = chunk_off<64>(row, ch);
            *reinterpret_cast<h8*>(buf.hi + o) = vh;
            *reinterpret_cast<h8*>(buf.lo + o) = vl;
        };
        auto finish_pe = [&](int b, int pstage, int hf) {
            if (hf == 0) { pe_view(b, pstage, 0); pe_view(b, pstage, 1); }
            else pe_view(b, pstage, 2);
        };
        // weights of one half stage: 2 k-steps x 2 N-tiles (nt, 4 + nt), hi + lo
        h8 wh[2][2], wl[2][2];
        uint32_t wx_off[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) wx_off[q] = (uint32_t)((nt + 4 * q) * KSX * 128 + L.lane) * 16u;
        auto load_wx = [&](int h) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    wh[u][q] = *reinterpret_cast<const h8*>(wb + (wx_off[q] + 4096u * h + 2048u * u));
                    wl[u][q] = *reinterpret_cast<const h8*>(wb + (wx_off[q] + 4096u * h + 2048u * u + 1024u));
                }
        };
        auto mma_x = [&](int b, int tks0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int o = chunk_off<64>(mt * 32 + L.l31, ((tks0 + u) << 1) + L.half);
#pragma unroll
                for (int v = 0; v < NVT; ++v) {
                    const HT t = xbuf(v, b);
                    const h8 bh = *reinterpret_cast<const h8*>(t.hi + o);
                    const h8 bl = *reinterpret_cast<const h8*>(t.lo + o);
                    acc0[v] = NEO_MFMA_H(wl[u][0], bh, acc0[v]);
                    acc0[v] = NEO_MFMA_H(wh[u][0], bl, acc0[v]);
                    acc0[v] = NEO_MFMA_H(wh[u][0], bh, acc0[v]);
                    acc3[v] = NEO_MFMA_H(wl[u][1], bh, acc3[v]);
                    acc3[v] = NEO_MFMA_H(wh[u][1], bl, acc3[v]);
                    acc3[v] = NEO_MFMA_H(wh[u][1], bh, acc3[v]);
                }
            }
        };
        constexpr int K_LOCAL = 0, K_PLANE = 1, K_PE = 2, K_NONE = 3;
        auto half_stage = [&](int s, int hf, auto kind_c) {
            constexpr int kind = decltype(kind_c)::value;
            const int cur = s & 1, nxt = (s + 1) & 1, sn = s + 1;
            if constexpr (kind == K_LOCAL) issue_local(sn, hf);
            if constexpr (kind == K_PLANE) issue_plane(0, sn - 8, hf);
            __builtin_amdgcn_sched_barrier(0);
            mma_x(cur, 2 * hf);
            __builtin_amdgcn_sched_barrier(0);
            if (2 * (2 * s + hf + 1) < KSX) load_wx(2 * s + hf + 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kind == K_LOCAL) finish_local(nxt, hf);
            if constexpr (kind == K_PLANE) finish_planes(nxt, sn - 8, hf);
            if constexpr (kind == K_PE) finish_pe(nxt, sn - 10, hf);
        };
        using std::integral_constant;
        load_wx(0);
        issue_local(0, 0);
        finish_local(0, 0);
        issue_local(0, 1);
        finish_local(0, 1);
        __syncthreads();
#pragma unroll 1
        for (int s = 0; s < 7; ++s) {
#pragma unroll 1
            for (int hf = 0; hf < 2; ++hf) half_stage(s, hf, integral_constant<int, K_LOCAL>());
            __syncthreads();
        }
#pragma unroll 1
        for (int s = 7; s < 9; ++s) {
#pragma unroll 1
            for (int hf = 0; hf < 2; ++hf) half_stage(s, hf, integral_constant<int, K_PLANE>());
            __syncthreads();
        }
#pragma unroll 1
        for (int s = 9; s < NST - 1; ++s) {
#pragma unroll 1
            for (int hf = 0; hf < 2; ++hf) half_stage(s, hf, integral_constant<int, K_PE>());
            __syncthreads();
        }
        half_stage(NST - 1, 0, integral_constant<int, K_NONE>());
        if (PE_C == 3) half_stage(NST - 1, 1, integral_constant<int, K_NONE>());   // (the 84-wide encoding ends with a 2-k-step stage)
        __syncthreads();
    }

    auto store_all = [&](const f32x16 (&a)[NVT], auto relu_c) {
        constexpr bool relu = decltype(relu_c)::value;
#pragma unroll
        for (int v = 0; v < NVT; ++v) store_tile_h<relu>(a[v], act(v), nt, mt, L);
    };
    using TrueT = std::integral_constant<bool, true>;
    using FalseT = std::integral_constant<bool, false>;
    // ---- L0 epilogue, L1, L2 ----
    store_all(acc0, TrueT());
    __syncthreads();
#pragma unroll 1
    for (int layer = 0; layer < 2; ++layer) {
        bias_tile(acc0[0], m.bias + (layer == 0 ? B_1 : B_2), nt, L);
#pragma unroll
        for (int v = 1; v < NVT; ++v) acc0[v] = acc0[0];
        gemm_v<128>(acc0, wb + (size_t)(layer == 0 ? hoff_1(PE_C) : hoff_2(PE_C)) * 16, 8, nt, mt, 0, 8, act(0), L);
        __syncthreads();
        store_all(acc0, TrueT());
        __syncthreads();
    }
    // ---- L3 = skip half + W3[:, :128] h2; ReLU ----
    gemm_v<128>(acc3, wb + (size_t)hoff_3a(PE_C) * 16, 8, nt, mt, 0, 8, act(0), L);
    __syncthreads();
    store_all(acc3, TrueT());
    __syncthreads();
    // density head: linear in the view mean of relu(L3) -> sum of per-view dot products (waves 0-3: 4 lanes per point)
    float sig_part = 0.f;
    if (L.wv < 4) {
#pragma unroll
        for (int v = 0; v < NVT; ++v) sig_part += density_partial(act(v), dens_w, L);
    }
    // ---- per-view bottleneck (no activation) ----
    bias_tile(acc0[0], m.bias + B_B, nt, L);
#pragma unroll
    for (int v = 1; v < NVT; ++v) acc0[v] = acc0[0];
    gemm_v<128>(acc0, wb + (size_t)hoff_b(PE_C) * 16, 8, nt, mt, 0, 8, act(0), L);
    __syncthreads();
    store_all(acc0, FalseT());
    __syncthreads();
    // ---- view layer 0: [bottleneck | dir enc] -> 64, summed over views; waves 0-3, one 32x32 tile each ----
    const int vnt = L.wv & 1, vmt = (L.wv >> 1) & 1;
    const h8* wp = reinterpret_cast<const h8*>(m.wpack);
    f32x16 ysum;
#pragma unroll
    for (int r = 0; r < 16; ++r) ysum[r] = 0.f;
    if (L.wv < 4) {
#pragma unroll
        for (int v = 0; v < NVT; ++v) {
            f32x16 y;
            bias_tile(y, m.bias + B_V0, vnt, L);
            gemm1h<128>(y, wp + hoff_v0(PE_C), 10, vnt, vmt, 0, 8, act(v), L);
            gemm1h<32>(y, wp + hoff_v0(PE_C), 10, vnt, vmt, 8, 2, dsm(v), L);
#pragma unroll
            for (int r = 0; r < 16; ++r) ysum[r] += y[r];
        }
    }
    __syncthreads();
    const float nvf = (float)NVT;
    float raw_sigma = 0.f;
    {
        float sg = sig_part;
        sg += __shfl_xor(sg, 1, 64);
        sg += __shfl_xor(sg, 2, 64);
        raw_sigma = sg / nvf + m.heads[HD_DB];
    }
    // ---- view mean of the view branch -> ReLU -> 64x64 -> ReLU -> rgb head (waves 0-3, view 0's tile) ----
    if (L.wv < 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ysum[r] = ysum[r] / nvf;
        store_tile_h<true>(ysum, act(0), vnt, vmt, L);
    }
    __syncthreads();
    f32x16 y2;
    if (L.wv < 4) {
        bias_tile(y2, m.bias + B_V1, vnt, L);
        gemm1h<128>(y2, wp + hoff_v1(PE_C), 4, vnt, vmt, 0, 4, act(0), L);
    }
    __syncthreads();
    if (L.wv < 4) store_tile_h<true>(y2, act(0), vnt, vmt, L);
    __syncthreads();
    if (L.wv < 4) {
        const int pt = L.wv * 16 + (L.lane >> 2), part = L.lane & 3;
        const float* wr = m.heads + HD_RW;
        float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int chunk_i = part * 2 + ((c + part) & 1);
            const int o = chunk_off<128>(pt, chunk_i);
            const h8 vh = *reinterpret_cast<const h8*>(act(0).hi + o);
            const h8 vl = *reinterpret_cast<const h8*>(act(0).lo + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float h = (float)vh[e] + (float)vl[e];
                r += h * wr[chunk_i * 8 + e];
                g += h * wr[64 + chunk_i * 8 + e];
                b += h * wr[128 + chunk_i * 8 + e];
            }
        }
        r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64);
        g += __shfl_xor(g, 1, 64); g += __shfl_xor(g, 2, 64);
        b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64);
        const long gi = tile0 + pt;
        if (part == 0 && gi < P) {
            out[gi] = make_float4(colour_act(r + m.heads[HD_RB]), colour_act(g + m.heads[HD_RB + 1]),
                                  colour_act(b + m.heads[HD_RB + 2]), density_act(raw_sigma));
        }
    }
}

}  // namespace

bool tp_views_batched_supported(int nv) { return nv == NVT; }

void launch_tp_mlp_hv(int input_ch, const TpMlpHDev& m, const TpScene& sc, const TpViews& views, const float* rays_o,
                      const float* rays_d, const float* viewdirs, const float* tvals, const float* far, int R, int N,
                      int chunk, uint32_t* flags, float* out, hipStream_t s) {
    const long P = (long)R * N;
    if (P <= 0) return;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_mlp_hv<3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_mlp_hv<4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr = true;
    }
    const long tiles = tp::xcd_grid((P + TM - 1) / TM);
    if (input_ch == 3)
        hipLaunchKernelGGL(k_tp_mlp_hv<3>, dim3((unsigned)tiles), dim3(512), LDS_BYTES, s, m, sc, views, rays_o, rays_d,
                           viewdirs, tvals, far, R, N, chunk, flags, reinterpret_cast<float4*>(out));
    else
        hipLaunchKernelGGL(k_tp_mlp_hv<4>, dim3((unsigned)tiles), dim3(512), LDS_BYTES, s, m, sc, views, rays_o, rays_d,
                           viewdirs, tvals, far, R, N, chunk, flags, reinterpret_cast<float4*>(out));
}

}  // namespace neo

// PixelNeRF baseline decoder point evaluator (models/vanilla_nerf/model_pixel.py:35-131, :195-237) on the fp16
// matrix cores with hi/lo-split fp32 operands (split_tile.h).  Same tile / streaming design as mlp_tp_h.hip
// (tp_common.h supplies the per-point set-up and the per-view descriptors) for the simpler network:
//   per 64-point tile, per source view: descriptors -> the 575-wide input [512 pixel-aligned latent | 63 pos_enc of
//   the camera-frame point] streamed 64 features at a time (9 stages) -> L0..L3 (128, ReLU; the skip never fires at
//   depth 4) -> per-view bottleneck -> view layer 0 on [bottleneck | 27 camera-frame direction encoding] (128 wide)
//   -> view means -> density head (ReLU), 128x128 ReLU, rgb head (sigmoid).
// The view direction of row (view, ray b, sample s) is that of ray (b*N+s) mod B of the reference chunk
// (model_pixel.py:219-222), as in NeRF_TP.  Compiled without packed-fp32 VALU ops (build.py:EXTRA_FLAGS).
//
// Algorithmic work per point-view: 575*128 + 3*128*128 + 128*128 + 155*128 = 158,976 MAC; per point 128 + 128*128
// + 384 = 16,896 MAC.
#include <type_traits>

#include "split_tile.h"
#include "tp_common.h"

#ifndef NEO_GATHER_WAVES_PER_SIMD
// 3 workgroups per CU (53.8 KB LDS each): caps the kernel at 168 VGPRs; the 16-register view-branch sum is then
// spilled once per view, and the third wave per SIMD still buys 4-6 % (tools/bench_tp_kernel.py: 17.40 -> 16.63 ms)
#define NEO_GATHER_WAVES_PER_SIMD 3
#endif


namespace neo {

namespace {

using tp::TM;
using tp::blend4;
using tp::pe_feature;

// ---- packed weight layout (h8 units) ---------------------------------------------------------------------------
constexpr int KSX = 36;                                   // 512 latent + 64 (63 pos_enc + pad) = 36 k-steps
constexpr int PX_X = 0;
constexpr int PX_1 = 4 * KSX * 128;
constexpr int PX_2 = PX_1 + 4 * 8 * 128;
constexpr int PX_3 = PX_2 + 4 * 8 * 128;
constexpr int PX_B = PX_3 + 4 * 8 * 128;
constexpr int PX_V0 = PX_B + 4 * 8 * 128;                 // 155 -> 160 = 10 k-steps
constexpr int PX_V1 = PX_V0 + 4 * 10 * 128;
constexpr int PX_TOTAL = PX_V1 + 4 * 8 * 128;
constexpr int B_0 = 0, B_1 = 128, B_2 = 256, B_3 = 384, B_B = 512, B_V0 = 640, B_V1 = 768, BIAS_FLOATS = 896;
constexpr int HD_DW = 0, HD_DB = 128, HD_RW = 132, HD_RB = 516, HEADS_FLOATS = 520;
constexpr int NST = 9;                                    // streamed stages: 8 latent, 1 pos_enc

// PROJ: the latent is gathered PRE-PROJECTED through pts_linears.0's latent columns (W . bilerp(F) = bilerp(W . F), as
// in mlp_tp_hp.hip): G = F . W0_loc^T, 128 channels = 512 B per texel instead of 2 KB; its two 64-channel chunks are
// blended into an fp32 LDS tile and ADDED to the L0 accumulators, and only the pos_enc stage (4 of the 36 k-steps of
// the first layer) is still multiplied per point.
// Chunk ch (8 packed features) of the 63-wide positional encoding in PAIR order (as mlp_tp_hp.hip): packed positions 2 p, 2 p + 1 =
// sin(a), sin(fl32(a + fl32(pi / 2))) of pair p = octave * 3 + coordinate, a = x 2^octave (reference columns 3 + p and 33 + p:
// model_pixel.py / helper.py:121-125), both from ONE argument reduction (common.h:sincos_pair); positions 60..62 = x, y, z, 63 = 0.
// launch_pix_pack_h packs the weight columns in the same order.
__device__ __forceinline__ void pe_chunk_pairs(const float (&xc)[4], int ch, h8& vh, h8& vl) {
    float f[8];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        if (ch == 7 && jj >= 2) {
            f[2 * jj] = jj == 2 ? xc[0] : xc[2];
            f[2 * jj + 1] = jj == 2 ? xc[1] : 0.0f;
            continue;
        }
        const int p = ch * 4 + jj;                 // wave-uniform
        const int oct = p / 3, a = p - 3 * oct;
        const float x = a == 0 ? xc[0] : a == 1 ? xc[1] : xc[2];
        sincos_pair(ldexpf(x, oct), f[2 * jj], f[2 * jj + 1]);
    }
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        h2 h, l;
        split2(f[e], f[e + 1], h, l);
        vh[e] = h[0]; vh[e + 1] = h[1];
        vl[e] = l[0]; vl[e + 1] = l[1];
    }
}

// first-layer fragments: packed k = [latent 512 | pos_enc 63 in pair order | 0] <- source columns [pos_enc 63 | latent 512]
__global__ void k_pix_pack_x_h(const float* __restrict__ src, _Float16* __restrict__ dst) {
    const int total = 4 * KSX * 512;               // 128 outputs = 4 N-tiles
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, blk = idx >> 9;
        const int ks = blk % KSX, ntl = blk / KSX;
        const int n = ntl * 32 + (lane & 31);
        const int k = ks * 16 + 8 * (lane >> 5) + e;
        int col = -1;
        if (k < 512) col = 63 + k;
        else {
            const int q = k - 512;
            if (q < 60) col = 3 + (q >> 1) + ((q & 1) ? 30 : 0);
            else if (q < 63) col = q - 60;
        }
        const float w = col >= 0 ? src[(long)n * 575 + col] : 0.0f;
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        const long base = ((long)(ntl * KSX + ks) * 2) * 512 + lane * 8 + e;
        dst[base] = hi;
        dst[base + 512] = lo;
    }
}

template <bool PROJ>
__global__ __launch_bounds__(256, NEO_GATHER_WAVES_PER_SIMD) void k_pix_mlp_h(TpMlpHDev m, const float* __restrict__ proj,
                                                      TpScene sc, TpViews views,
                                                      const float* __restrict__ rays_o,
                                                      const float* __restrict__ rays_d,
                                                      const float* __restrict__ viewdirs,
                                                      const float* __restrict__ tvals, int t_shared, int R, int N,
                                                      int chunk, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* hbase = reinterpret_cast<_Float16*>(smem + tp::OFF_ACT);
    const HT act{hbase, hbase + TM * 128};                                   // [64][128] x 2 planes (32 KB)
    auto xbuf = [&](int b) { return HT{hbase + b * (2 * TM * 64), hbase + b * (2 * TM * 64) + TM * 64}; };   // aliases act
    _Float16* dbase = reinterpret_cast<_Float16*>(smem + tp::OFF_DIR);
    const HT dsm{dbase, dbase + TM * 32};                                    // [64][32] x 2 planes
    const tp::Scratch S = tp::carve(smem);
    int* loc_off = S.loc_off;
    float* loc_w = S.loc_w;
    float* cam_enc = S.cam_enc;

    LaneCtx L;
    L.init();
    int tid = threadIdx.x;
    const long P = (long)R * N;
    const long tile0 = tp::xcd_tile(blockIdx.x, (P + TM - 1) / TM) * TM;      // contiguous tile range per XCD (tp_common.h)
    if (tile0 >= P) return;       // surplus workgroup of the rounded-up grid (uniform exit before any barrier)
    const h8* wp = reinterpret_cast<const h8*>(m.wpack);

    tp::point_setup<3>(S, tid, tile0, P, N, R, chunk, rays_o, rays_d, viewdirs, tvals, nullptr, nullptr, t_shared != 0);
    float* dens_w = smem + tp::OFF_DENSW;
    if (tid < 128) dens_w[tid] = m.heads[HD_DW + tid];
    __syncthreads();

    // View means by linearity (see mlp_tp_h.hip): the loop accumulates hsum = sum_v relu(L3_v) and the direction
    // encodings; density head, bottleneck and view layer 0 run once per tile on the view means.
    f32x16 hsum[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = 0.f; hsum[1][r] = 0.f; }
    float* dsum = smem + tp::OFF_DIR;          // [64][32] fp32 running sum of the direction encodings (same 8 KB as dsm)
    const int nts_1[1] = {L.wv};

#pragma unroll 1
    for (int v = 0; v < sc.nv; ++v) {
        // per-lane indices re-derived from an opaque lane id (keeps swizzled LDS addresses out of scratch, mlp_tp_h.hip)
        asm volatile("" : "+v"(tid));
        L.lane = tid & 63;
        L.half = L.lane >> 5;
        L.l31 = L.lane & 31;
        L.key = L.lane & 15;
        const float* rot = views.rot[v];
        const float* trn = views.trans[v];
        tp::view_descriptors<PROJ ? 512 : 2048>(S, L, sc, rot, trn, v, [&](int p, int f, float val) {
            const int di = p * 32 + (f ^ (p & 31));     // lane = p: XOR keeps the 64 lanes on distinct banks
            dsum[di] = v == 0 ? val : dsum[di] + val;   // (p, f) is owned by one thread in every view
        });
        __syncthreads();

        // ---- streamed-input GEMM: L0 (128 outputs; N-tile = wave) over 575 features ----
        f32x16 accx[1][2];
        bias_tile(accx[0][0], m.bias + B_0, L.wv, L);
        accx[0][1] = accx[0][0];
        if constexpr (PROJ) {
            const int col4 = tid & 15, rg = tid >> 4;
            const uint32_t lane_b = 16u * col4;
            float* fb = smem + tp::OFF_ACT;                       // two fp32 [64][64] chunk tiles = the 32 KB of `act`
            f32x4 tap[2][2][4];                                   // two tap sets in flight (half a chunk each)
            auto issue = [&](int set, int c, int hf) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = rg + 16 * (2 * hf + i);
                    const int4 off = *reinterpret_cast<const int4*>(loc_off + row * 4);
                    tap[set][i][0] = tp::load_tap(proj, (uint32_t)off.x + lane_b + 256u * c);
                    tap[set][i][1] = tp::load_tap(proj, (uint32_t)off.y + lane_b + 256u * c);
                    tap[set][i][2] = tp::load_tap(proj, (uint32_t)off.z + lane_b + 256u * c);
                    tap[set][i][3] = tp::load_tap(proj, (uint32_t)off.w + lane_b + 256u * c);
                }
            };
            auto finish = [&](int set, int c, int hf) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = rg + 16 * (2 * hf + i);
                    const f32x4 val = blend4(tap[set][i], *reinterpret_cast<const f32x4*>(loc_w + row * 4));
                    *reinterpret_cast<f32x4*>(fb + c * (TM * 64) + row * 64 + ((col4 ^ (row & 15)) << 2)) = val;
                }
            };
            // weights of the pos_enc stage: k-steps 32..35 of the packed first layer, both halves resident
            h8 wh[4], wl[4];
            {
                const char* wxb = reinterpret_cast<const char*>(wp + PX_X);
                const uint32_t wx_off = (uint32_t)(L.wv * KSX * 2 * 64 + L.lane) * 16u;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    wh[u] = *reinterpret_cast<const h8*>(wxb + (wx_off + 2048u * (32 + u)));
                    wl[u] = *reinterpret_cast<const h8*>(wxb + (wx_off + 2048u * (32 + u) + 1024u));
                }
            }
            issue(0, 0, 0);
            issue(1, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            finish(0, 0, 0);
            issue(0, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            finish(1, 0, 1);
            issue(1, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            finish(0, 1, 0);
            finish(1, 1, 1);
            __syncthreads();
            // this wave's pieces of both chunks (channel order: tp_hp_layout.h:proj_index) -> accumulators
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int row = mt * 32 + L.l31;
                        const int piece = L.wv * 4 + gg * 2 + L.half;
                        const f32x4 val = *reinterpret_cast<const f32x4*>(fb + c * (TM * 64) + row * 64 + ((piece ^ (row & 15)) << 2));
#pragma unroll
                        for (int e = 0; e < 4; ++e) accx[0][mt][4 * (2 * c + gg) + e] += val[e];
                    }
            __syncthreads();
            // pos_enc of the camera-frame point into the first stage tile, then its 4 k-steps
            {
                const HT buf = xbuf(0);
                const int row = tid & 63, q = tid >> 6;
                const float xc[4] = {cam_enc[row * 4], cam_enc[row * 4 + 1], cam_enc[row * 4 + 2], 0.0f};
                range_see(L, xc[0]); range_see(L, xc[1]); range_see(L, xc[2]);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int ch = hf * 4 + q;
                    h8 vh, vl;
                    pe_chunk_pairs(xc, ch, vh, vl);
                    const int o = chunk_off<64>(row, ch);
                    *reinterpret_cast<h8*>(buf.hi + o) = vh;
                    *reinterpret_cast<h8*>(buf.lo + o) = vl;
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const HT tile = xbuf(0);
                h8 bh[2], bl[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int o = chunk_off<64>(mt * 32 + L.l31, (u << 1) + L.half);
                    bh[mt] = *reinterpret_cast<const h8*>(tile.hi + o);
                    bl[mt] = *reinterpret_cast<const h8*>(tile.lo + o);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    accx[0][mt] = NEO_MFMA_H(wl[u], bh[mt], accx[0][mt]);
                    accx[0][mt] = NEO_MFMA_H(wh[u], bl[mt], accx[0][mt]);
                    accx[0][mt] = NEO_MFMA_H(wh[u], bh[mt], accx[0][mt]);
                }
            }
            __syncthreads();
        } else {
            const int col4 = tid & 15, rg = tid >> 4;
            const uint32_t lane_b = 16u * col4;
            f32x4 tap[2][4];
            auto issue_local = [&](int s, int hf) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = rg + 16 * (2 * hf + i);
#pragma unroll
                    for (int k = 0; k < 4; ++k) tap[i][k] = tp::load_tap(sc.latent, (uint32_t)loc_off[row * 4 + k] + lane_b + 256u * s);
                }
            };
            auto finish_local = [&](const HT& buf, int hf) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = rg + 16 * (2 * hf + i);
                    const f32x4 v4 = blend4(tap[i], *reinterpret_cast<const f32x4*>(loc_w + row * 4));
                    h4 vh, vl;
                    split4(v4, vh, vl);
                    const int o = chunk_off<64>(row, col4 >> 1) + 4 * (col4 & 1);
                    *reinterpret_cast<h4*>(buf.hi + o) = vh;
                    *reinterpret_cast<h4*>(buf.lo + o) = vl;
                }
            };
            // pos_enc of the camera-frame point: half hf of the stage = chunks 4hf..4hf+3, one per wave
            auto finish_pe = [&](const HT& buf, int hf) {
                const int row = tid & 63, q = tid >> 6;
                const float xc[4] = {cam_enc[row * 4], cam_enc[row * 4 + 1], cam_enc[row * 4 + 2], 0.0f};
                range_see(L, xc[0]); range_see(L, xc[1]); range_see(L, xc[2]);     // identity features (the rest are sines)
                const int ch = hf * 4 + q;
                h8 vh, vl;
                pe_chunk_pairs(xc, ch, vh, vl);
                const int o = chunk_off<64>(row, ch);
                *reinterpret_cast<h8*>(buf.hi + o) = vh;
                *reinterpret_cast<h8*>(buf.lo + o) = vl;
            };
            // weights of one half stage (2 k-steps x 1 N-tile, hi + lo): SGPR base + 32-bit VGPR offset, 4 KB per half
            h8 wh[2], wl[2];
            const char* wxb = reinterpret_cast<const char*>(wp + PX_X);
            const uint32_t wx_off = (uint32_t)(L.wv * KSX * 2 * 64 + L.lane) * 16u;
            auto load_wx = [&](int h) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    wh[u] = *reinterpret_cast<const h8*>(wxb + (wx_off + 4096u * h + 2048u * u));
                    wl[u] = *reinterpret_cast<const h8*>(wxb + (wx_off + 4096u * h + 2048u * u + 1024u));
                }
            };
            auto mma_x = [&](const HT& tile, int tks0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    h8 bh[2], bl[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int o = chunk_off<64>(mt * 32 + L.l31, ((tks0 + u) << 1) + L.half);
                        bh[mt] = *reinterpret_cast<const h8*>(tile.hi + o);
                        bl[mt] = *reinterpret_cast<const h8*>(tile.lo + o);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        accx[0][mt] = NEO_MFMA_H(wl[u], bh[mt], accx[0][mt]);
                        accx[0][mt] = NEO_MFMA_H(wh[u], bl[mt], accx[0][mt]);
                        accx[0][mt] = NEO_MFMA_H(wh[u], bh[mt], accx[0][mt]);
                    }
                }
            };
            constexpr int K_LOCAL = 0, K_PE = 2, K_NONE = 3;
            auto half_stage = [&](int s, int hf, auto kind_c) {
                constexpr int kind = decltype(kind_c)::value;
                const HT cur = xbuf(s & 1), nxt = xbuf((s + 1) & 1);
                if constexpr (kind == K_LOCAL) issue_local(s + 1, hf);
                __builtin_amdgcn_sched_barrier(0);
                mma_x(cur, 2 * hf);
                __builtin_amdgcn_sched_barrier(0);
                if (2 * (2 * s + hf + 1) < KSX) load_wx(2 * s + hf + 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (kind == K_LOCAL) finish_local(nxt, hf);
                if constexpr (kind == K_PE) finish_pe(nxt, hf);
            };
            using std::integral_constant;
            load_wx(0);
            issue_local(0, 0);
            finish_local(xbuf(0), 0);
            issue_local(0, 1);
            finish_local(xbuf(0), 1);
            __syncthreads();
#pragma unroll 1
            for (int s = 0; s < 7; ++s) {            // stages 0..6 multiply while latent stages 1..7 are gathered
#pragma unroll 1
                for (int hf = 0; hf < 2; ++hf) half_stage(s, hf, integral_constant<int, K_LOCAL>());
                __syncthreads();
            }
#pragma unroll 1
            for (int hf = 0; hf < 2; ++hf) half_stage(7, hf, integral_constant<int, K_PE>());
            __syncthreads();
#pragma unroll 1
            for (int hf = 0; hf < 2; ++hf) half_stage(NST - 1, hf, integral_constant<int, K_NONE>());
            __syncthreads();
        }

        // ---- L0 epilogue, L1, L2, L3 ----
        f32x16 acc[1][2];
        store_tile_h<true>(accx[0][0], act, L.wv, 0, L);
        store_tile_h<true>(accx[0][1], act, L.wv, 1, L);
        __syncthreads();
#pragma unroll 1
        for (int layer = 0; layer < 3; ++layer) {
            bias_tile(acc[0][0], m.bias + (layer == 0 ? B_1 : layer == 1 ? B_2 : B_3), L.wv, L);
            acc[0][1] = acc[0][0];
            gemm2h<1, 128>(acc, wp + (layer == 0 ? PX_1 : layer == 1 ? PX_2 : PX_3), 8, nts_1, 0, 0, 8, act, L);
            __syncthreads();
            if (layer < 2) {
                store_tile_h<true>(acc[0][0], act, L.wv, 0, L);
                store_tile_h<true>(acc[0][1], act, L.wv, 1, L);
                __syncthreads();
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {         // relu(L3_v) joins the view sum; it is not written back per view
            hsum[0][r] += fmaxf(acc[0][0][r], 0.0f);
            hsum[1][r] += fmaxf(acc[0][1][r], 0.0f);
        }
    }

    // ---- view mean of the trunk -> density head (ReLU, model_pixel.py:232) ----
    const float inv_nv = 1.0f / (float)sc.nv;     // x * (1 / nv): 1 instruction instead of the 11 of an fp32 division, <= 1 ulp (see mlp_tp_hp.hip)
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = hsum[0][r] * inv_nv; hsum[1][r] = hsum[1][r] * inv_nv; }
    store_tile_h<false>(hsum[0], act, L.wv, 0, L);
    store_tile_h<false>(hsum[1], act, L.wv, 1, L);
    float dmean[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dmean[j] = dsum[(tid >> 2) * 32 + ((((tid & 3) << 3) + j) ^ ((tid >> 2) & 31))] * inv_nv;
    __syncthreads();
    {
        h8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 h, l;
            split(dmean[j], h, l);
            vh[j] = h;
            vl[j] = l;
        }
        const int o = chunk_off<32>(tid >> 2, tid & 3);
        *reinterpret_cast<h8*>(dsm.hi + o) = vh;
        *reinterpret_cast<h8*>(dsm.lo + o) = vl;
    }
    float sigma;
    {
        float sg = density_partial(act, dens_w, L);
        sg += __shfl_xor(sg, 1, 64);
        sg += __shfl_xor(sg, 2, 64);
        sigma = fmaxf(sg + m.heads[HD_DB], 0.0f);
    }
    // ---- bottleneck of the view mean (no activation), view layer 0 on [mean bottleneck | mean dir enc] -> 128.
    //      NEO_PIX_FOLDB (default): the bottleneck is folded into view layer 0 at pack time (launch_pix_pack_h; the algebra of
    //      tp_hp_layout.h NEO_TP_FOLDB: model_pixel.py's bottleneck_layer has no activation and feeds views_linear.0 only), so
    //      view layer 0 reads the view-mean trunk itself ----
    f32x16 ysum[1][2];
    {
        bias_tile(ysum[0][0], m.bias + B_V0, L.wv, L);
        ysum[0][1] = ysum[0][0];
        gemm2h<1, 128>(ysum, wp + PX_V0, 10, nts_1, 0, 0, 8, act, L);
        gemm2h<1, 32>(ysum, wp + PX_V0, 10, nts_1, 8, 0, 2, dsm, L);
        __syncthreads();
    }
    // ---- ReLU -> 128x128 -> ReLU -> rgb head -> sigmoid ----
    store_tile_h<true>(ysum[0][0], act, L.wv, 0, L);
    store_tile_h<true>(ysum[0][1], act, L.wv, 1, L);
    __syncthreads();
    {
        f32x16 y[1][2];
        bias_tile(y[0][0], m.bias + B_V1, L.wv, L);
        y[0][1] = y[0][0];
        gemm2h<1, 128>(y, wp + PX_V1, 8, nts_1, 0, 0, 8, act, L);
        __syncthreads();
        store_tile_h<true>(y[0][0], act, L.wv, 0, L);
        store_tile_h<true>(y[0][1], act, L.wv, 1, L);
    }
    __syncthreads();
    {
        const int pt = L.wv * 16 + (L.lane >> 2), part = L.lane & 3;
        const float* wr = m.heads + HD_RW;
        float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int chunk_i = part * 4 + ((c + part) & 3);
            const int o = chunk_off<128>(pt, chunk_i);
            const h8 vh = *reinterpret_cast<const h8*>(act.hi + o);
            const h8 vl = *reinterpret_cast<const h8*>(act.lo + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float h = (float)vh[e] + (float)vl[e];
                r += h * wr[chunk_i * 8 + e];
                g += h * wr[128 + chunk_i * 8 + e];
                b += h * wr[256 + chunk_i * 8 + e];
            }
        }
        r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64);
        g += __shfl_xor(g, 1, 64); g += __shfl_xor(g, 2, 64);
        b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64);
        range_commit(L, m.flags);
        const long gi = tile0 + pt;
        if (part == 0 && gi < P) {
            out[gi] = make_float4(sigmoid_act(r + m.heads[HD_RB]), sigmoid_act(g + m.heads[HD_RB + 1]),
                                  sigmoid_act(b + m.heads[HD_RB + 2]), sigma);
        }
    }
}

__global__ void k_copy_n(const float* __restrict__ src, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

}  // namespace

size_t pix_wpack_h_bytes() { return (size_t)PX_TOTAL * 16; }
size_t pix_bias_floats() { return BIAS_FLOATS; }
size_t pix_heads_floats() { return HEADS_FLOATS; }

size_t pix_fold_floats() { return 128 * 155; }

void launch_pix_pack_h(const float* const* w, const float* const* b, void* wpack_h, float* bias, float* heads,
                       float* fold_ws, hipStream_t s) {
    // w order: pts_linears.0..3, views_linear.0, views_linear.1, bottleneck, density, rgb
    _Float16* base = reinterpret_cast<_Float16*>(wpack_h);
    const PackSegs none = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    // packed [latent | pos_enc in pair order] <- source [pos_enc | latent]
    hipLaunchKernelGGL(k_pix_pack_x_h, dim3((4 * KSX * 512 + 255) / 256), dim3(256), 0, s, w[0], base + (size_t)PX_X * 8);
    PackSegs p128 = none;
    p128.len[0] = 128;
    pack_h(w[1], 128, 128, 8, 0, p128, base + (size_t)PX_1 * 8, s);
    pack_h(w[2], 128, 128, 8, 0, p128, base + (size_t)PX_2 * 8, s);
    pack_h(w[3], 128, 128, 8, 0, p128, base + (size_t)PX_3 * 8, s);
    pack_h(w[6], 128, 128, 8, 0, p128, base + (size_t)PX_B * 8, s);
    PackSegs v0 = none;
    v0.len[0] = 155;
    pack_h(w[4], 155, 128, 10, 0, v0, base + (size_t)PX_V0 * 8, s);       // (re-packed from the folded matrix below when NEO_PIX_FOLDB)
    pack_h(w[5], 128, 128, 8, 0, p128, base + (size_t)PX_V1 * 8, s);
    auto cp = [&](const float* src, int n, float* dst) {
        hipLaunchKernelGGL(k_copy_n, dim3((n + 255) / 256), dim3(256), 0, s, src, n, dst);
    };
    (void)hipMemsetAsync(heads, 0, HEADS_FLOATS * sizeof(float), s);
    cp(b[0], 128, bias + B_0); cp(b[1], 128, bias + B_1); cp(b[2], 128, bias + B_2); cp(b[3], 128, bias + B_3);
    cp(b[6], 128, bias + B_B); cp(b[4], 128, bias + B_V0); cp(b[5], 128, bias + B_V1);
    // after the plain copies: [W_v0[:, :128] . W_b | W_v0[:, 128:]] into fold_ws (packed above - stream order: the fold kernel is
    // enqueued here, so the view layer is packed AGAIN below from the folded matrix), folded bias over B_V0
    launch_fold_bottleneck(w[4], w[6], b[6], b[4], 128, 128, 128, 27, fold_ws, bias + B_V0, s);
    pack_h(fold_ws, 155, 128, 10, 0, v0, base + (size_t)PX_V0 * 8, s);
    cp(w[7], 128, heads + HD_DW); cp(b[7], 1, heads + HD_DB); cp(w[8], 384, heads + HD_RW); cp(b[8], 3, heads + HD_RB);
}

void launch_pix_mlp_h(const TpMlpHDev& m, const float* proj, const TpScene& sc, const TpViews& views, const float* rays_o,
                      const float* rays_d, const float* viewdirs, const float* tvals, int t_shared, int R, int N,
                      int chunk, float* out, hipStream_t s) {
    const long P = (long)R * N;
    if (P <= 0) return;
    const size_t lds = tp::LDS_WORDS * sizeof(float);
    const long tiles = tp::xcd_grid((P + TM - 1) / TM);
    if (proj)
        hipLaunchKernelGGL(k_pix_mlp_h<true>, dim3((unsigned)tiles), dim3(256), lds, s, m, proj, sc, views, rays_o, rays_d,
                           viewdirs, tvals, t_shared, R, N, chunk, reinterpret_cast<float4*>(out));
    else
        hipLaunchKernelGGL(k_pix_mlp_h<false>, dim3((unsigned)tiles), dim3(256), lds, s, m, proj, sc, views, rays_o, rays_d,
                           viewdirs, tvals, t_shared, R, N, chunk, reinterpret_cast<float4*>(out));
}

}  // namespace neo

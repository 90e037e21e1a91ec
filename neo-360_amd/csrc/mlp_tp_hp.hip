// NeO-360 decoder point evaluator, split-fp16 arithmetic, with the pixel-aligned latent PRE-PROJECTED
// through the first-layer weights once per (scene, MLP).
//
// NeRFPPMLP (neo360/model.py:110-158) feeds the 512-channel latent only through two linear maps: the local
// columns of pts_linears.0 and of the skip half of pts_linears.3.  Bilinear interpolation is linear, so
//        W_loc . bilerp(F; taps) = bilerp(W_loc . F; taps)
// and G = F . [W0_loc | W3_loc]^T  (256 channels per texel, k_tp_preproject below, exact fp32 MFMA) can be
// gathered instead of F: 1 KB per tap instead of 2 KB, and 131,072 of the 255,424 MACs per point-view
// (32 of 44 k-steps of the streamed GEMM, their weight fragments, hi/lo splits and LDS writes) disappear from
// the per-point path.  The reassociation moves the pre-activations by ~4e-7.  Algorithmic
// flops in the roofline stay those of the reference formulation.
//
// Per 64-point tile and source view the kernel runs ONE flat software pipeline over 40 gather items
// (16 = 4 chunks x 4 row groups of the pre-projected latent, 24 = 2 stages x 4 row groups x 3 planes),
// a ring of NEO_TP_RING tap register sets deep: item i+RING-1 is requested before item i is blended, and
// loads stay in flight across the stage barriers.  Pre-projected chunks travel through a 16 KB fp32 LDS
// tile (gather layout = one row per 16 lanes, coalesced 256-B runs; accumulator layout = MFMA D fragments)
// and are ADDED to the L0 / L3-skip accumulators; tri-plane and pos_enc stages are split into hi/lo fp16
// planes and multiplied on the matrix cores while the next stage is gathered.  The rest (L1, L2, L3, view-mean
// linearity, heads) is the structure of mlp_tp_h.hip.
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "tp_hp_layout.h"

#ifndef NEO_TP_WPS
#define NEO_TP_WPS 2          // workgroups per CU = waves per SIMD the register allocation is capped for
#endif
#ifndef NEO_TP_RING
#define NEO_TP_RING 3         // tap register sets (16 VGPRs each); prefetch distance = RING - 1 items
#endif
#ifndef NEO_TP_STAGGER_DEFAULT
#define NEO_TP_STAGGER_DEFAULT 0      // x 64 cycles; $NEO_TP_STAGGER overrides (see k_tp_mlp_hp)
#endif
#ifndef NEO_TP_XSTREAM
#define NEO_TP_XSTREAM 2      // streamed-stage weight fragments in a ring this many k-steps ahead
#endif
#ifndef NEO_TP_ABLATE
// timing / energy probes only (results wrong by construction; tools/build_variant.py; profiles/r06_energy_budget.log):
// 1 no latent-chunk gathers, 2 no tri-plane gathers, 4 no pos_enc; 256 every tap reads texel 0 of its map (the loads stay, all L1
// hits on one line: what goes is the divergent-address work and the L2 -> L1 traffic), 1024 / 2048 the same for the tri-plane /
// the latent taps alone; 512 the loads are replaced by undefined registers (blends, LDS transposition and adds stay)
#define NEO_TP_ABLATE 0
#endif
#define TP_SYNC() __syncthreads()
#ifndef NEO_TP_TRACE
#define NEO_TP_TRACE 0        // 1: per-phase s_memtime sums of wave 0 of every workgroup -> g_tp_trace (tools/tp_phase_trace.py)
#endif
#if NEO_TP_TRACE
__device__ unsigned long long g_tp_trace[16];
#define TP_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tr_[k] += now_ - tlast_; tlast_ = now_; } while (0)
#else
#define TP_MARK(k) do { } while (0)
#endif

namespace neo {

namespace {

using namespace hp;
constexpr int RING = NEO_TP_RING;

template <int PE_C>
__global__ __launch_bounds__(256, NEO_TP_WPS) void k_tp_mlp_hp(TpMlpHDev m, const float* __restrict__ proj, TpScene sc,
                                                             TpViews views, const float* __restrict__ rays_o,
                                                             const float* __restrict__ rays_d,
                                                             const float* __restrict__ viewdirs,
                                                             const float* __restrict__ tvals,
                                                             const float* __restrict__ far_arr, int R, int N, int chunk,
                                                             uint32_t* __restrict__ flags, float4* __restrict__ out, int stagger,
                                                             const float* __restrict__ dirsum) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // Two workgroups share a CU.  Launched together they run their phases (gather-bound, matrix-bound) in lockstep and
    // contend for the same pipe all the time; the second generation of resident workgroups (blocks 256..511 on this
    // 256-CU part) therefore starts `stagger` x 64 cycles late, once: later workgroups start whenever a slot frees and
    // inherit the offset.  Placement-independent: a wrong guess about which blocks co-reside only costs the delay.
    if (stagger > 0 && blockIdx.x >= 256u && blockIdx.x < 512u)
        for (int i = 0; i < stagger; i += 100) __builtin_amdgcn_s_sleep(100);
    _Float16* hbase = reinterpret_cast<_Float16*>(smem + tp::OFF_ACT);
    const HT act{hbase, hbase + TM * 128};                                   // [64][128] x 2 planes (32 KB)
    auto xbuf = [&](int b) { return HT{hbase + b * (2 * TM * 64), hbase + b * (2 * TM * 64) + TM * 64}; };   // aliases act
    auto fbuf = [&](int b) { return smem + tp::OFF_ACT + b * (TM * 64); };   // the same 16 KB halves as fp32 [64][64]
    _Float16* dbase = reinterpret_cast<_Float16*>(smem + tp::OFF_DIR);
    const HT dsm{dbase, dbase + TM * 32};                                    // [64][32] x 2 planes
    const tp::Scratch S = tp::carve(smem);
    int* loc_off = S.loc_off;
    float* loc_w = S.loc_w;
    int* pl_off = S.pl_off;
    float* pl_w = S.pl_w;
    float* cam_enc = S.cam_enc;

    LaneCtx L;
    L.init();
    int tid = threadIdx.x;
    const long P = (long)R * N;
    const long tile0 = tp::xcd_tile(blockIdx.x, (P + TM - 1) / TM) * TM;
    if (tile0 >= P) return;       // surplus workgroup of the rounded-up grid (uniform exit before any barrier)
    const h8* wp = reinterpret_cast<const h8*>(m.wpack);
    constexpr int KSX = ks_x(PE_C);
    constexpr int NPE = PE_C == 3 ? 1 : 2;     // pos_enc stages of 64 features

#if NEO_TP_TRACE
    unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = __builtin_amdgcn_s_memtime();
#endif
    tp::point_setup<PE_C>(S, tid, tile0, P, N, R, chunk, rays_o, rays_d, viewdirs, tvals, far_arr, flags, false, sc.grid_w, sc.grid_first, sc.grid_pw, sc.grid_ph);
    float* dens_w = smem + tp::OFF_DENSW;
    if (tid < 128) dens_w[tid] = m.heads[HD_DW + tid];
    // biases and head weights are read from LDS: an accumulator initialisation is on the critical path of every layer
    float* lbias_w = smem + tp::LDS_WORDS;
    for (int i = tid; i < 768; i += 256) lbias_w[i] = m.bias[i];
    for (int i = tid; i < HD_RB + 3; i += 256) lbias_w[768 + i] = m.heads[i];
    const float* lbias = lbias_w;
    const float* lheads = lbias_w + 768;
    float* dsum = smem + tp::OFF_DIR;          // [64][32] fp32: sum over the views of each point's direction encoding (same 8 KB as dsm)
    TP_SYNC();                                 // point_setup has recorded which ray's direction every row carries
    {
        // the sums come ready-made from the per-ray table of this launch (k_tp_dirsum): 8 features per thread
        const int p = tid >> 2, f0 = (tid & 3) << 3;
        const int dray = __float_as_int(S.vdir_world[p * 4 + 3]);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(dirsum + (long)dray * 32 + f0);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(dirsum + (long)dray * 32 + f0 + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) dsum[p * 32 + ((f0 + j) ^ (p & 31))] = j < 4 ? s0[j] : s1[j - 4];
    }
    TP_SYNC();
    TP_MARK(0);

    // view means by linearity (see mlp_tp_h.hip): only sum_v relu(L3_v) and sum_v dir_enc_v are accumulated per view
    f32x16 hsum[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = 0.f; hsum[1][r] = 0.f; }
    [[maybe_unused]] const int nts_1[1] = {L.wv};
    const int vnt = L.wv & 1, vmt = L.wv >> 1;

#pragma unroll 1
    for (int v = 0; v < sc.nv; ++v) {
        // per-lane indices re-derived from an opaque lane id inside the loop: keeps the swizzled LDS addresses of the
        // loop body from being hoisted (and spilled) as loop invariants
        asm volatile("" : "+v"(tid));
        L.lane = tid & 63;
        L.half = L.lane >> 5;
        L.l31 = L.lane & 31;
        L.key = L.lane & 15;
        const float* rot = views.rot[v];
        const float* trn = views.trans[v];
        tp::view_descriptors<PROJ_TEXEL_BYTES, false>(S, L, sc, rot, trn, v, [](int, int, float) {});
        TP_SYNC();
        TP_MARK(1);
        // Does any row carry a non-zero tap weight in the latent?  Samples outside a source image blend to exactly zero
        // (grid_sample's zero padding).  (Skipping single gather items was measured first: loads under a branch cost the software
        // pipeline its exact vmcnt bookkeeping, and zero-weight taps are cheap on the memory path anyway - all lanes read texel 0,
        // one cache line.)  Wave-uniform and identical in all four waves: the barriers stay uniform.
        unsigned long long zm0;
        {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(loc_w + L.lane * 4);
            zm0 = __ballot(w0[0] != 0.0f || w0[1] != 0.0f || w0[2] != 0.0f || w0[3] != 0.0f);
        }

        // ---- [L0 | L3 skip half] pre-activations: bias + pre-projected latent (adds) + world / pos_enc GEMM ----
        f32x16 accx[2][2];
        bias_tile(accx[0][0], lbias + B_0, L.wv, L);
        bias_tile(accx[1][0], lbias + B_3, L.wv, L);
        accx[0][1] = accx[0][0];
        accx[1][1] = accx[1][0];
        {
            const int col4 = tid & 15, rg = tid >> 4;
            const uint32_t lane_b = 16u * col4;
            f32x4 taps[RING][4];
            f32x4 wsum;                                    // running sum over the three planes of one row group
            int4 d_off[2];                                 // tap byte offsets of the item that is REQUESTED next (slot = item & 1)
            f32x4 d_w[2];                                  // tap weights of the item that is BLENDED next
            // item -> its descriptor row in LDS (offsets and weights share the layout)
            auto desc_index = [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 16) return (rg + 16 * (i % 4)) * 4;
                else return (((i - 16) % 3) * TM + rg + 16 * (((i - 16) % 12) / 3)) * 4;
            };
            auto fetch_off = [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 16) d_off[i & 1] = *reinterpret_cast<const int4*>(loc_off + desc_index(ic));
                else if constexpr (i < 40) d_off[i & 1] = *reinterpret_cast<const int4*>(pl_off + desc_index(ic));
            };
            auto fetch_w = [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 16) d_w[i & 1] = *reinterpret_cast<const f32x4*>(loc_w + desc_index(ic));
                else if constexpr (i < 40) d_w[i & 1] = *reinterpret_cast<const f32x4*>(pl_w + desc_index(ic));
            };
            // item i: 0..15 = pre-projected latent (chunk i / 4, row group i % 4); 16..39 = tri-planes
            // (stage (i - 16) / 12, row group ((i - 16) % 12) / 3, plane (i - 16) % 3)
            constexpr int NI = 40;
            auto issue = [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr ((i < 16 && (NEO_TP_ABLATE & 1)) || (i >= 16 && (NEO_TP_ABLATE & 2))) {
                } else if constexpr (i < 16) {
                    constexpr int c = i / 4;
                    [[maybe_unused]] constexpr int q = i % 4;
                    int4 off = d_off[i & 1];
                    if constexpr ((NEO_TP_ABLATE & (256 | 2048)) != 0) off = int4{0, 0, 0, 0};
                    if constexpr ((NEO_TP_ABLATE & 512) != 0) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) asm volatile("" : "=v"(taps[i % RING][t]));
                        return;
                    }
                    taps[i % RING][0] = tp::load_tap(proj, (uint32_t)off.x + lane_b + 256u * c);
                    taps[i % RING][1] = tp::load_tap(proj, (uint32_t)off.y + lane_b + 256u * c);
                    taps[i % RING][2] = tp::load_tap(proj, (uint32_t)off.z + lane_b + 256u * c);
                    taps[i % RING][3] = tp::load_tap(proj, (uint32_t)off.w + lane_b + 256u * c);
                } else if constexpr (i < NI) {
                    constexpr int w = i - 16, s2 = w / 12, j = w % 3;
                    [[maybe_unused]] constexpr int q = (w % 12) / 3;
                    int4 off = d_off[i & 1];
                    if constexpr ((NEO_TP_ABLATE & (256 | 1024)) != 0) off = int4{0, 0, 0, 0};
                    if constexpr ((NEO_TP_ABLATE & 512) != 0) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) asm volatile("" : "=v"(taps[i % RING][t]));
                        return;
                    }
                    taps[i % RING][0] = tp::load_tap(sc.plane[j], (uint32_t)off.x + lane_b + 256u * s2);
                    taps[i % RING][1] = tp::load_tap(sc.plane[j], (uint32_t)off.y + lane_b + 256u * s2);
                    taps[i % RING][2] = tp::load_tap(sc.plane[j], (uint32_t)off.z + lane_b + 256u * s2);
                    taps[i % RING][3] = tp::load_tap(sc.plane[j], (uint32_t)off.w + lane_b + 256u * s2);
                }
            };
            auto write_x = [&](const HT& buf, int row, const f32x4 val) __attribute__((always_inline)) {
                range_see4(L, val);
                h4 vh, vl;
                split4(val, vh, vl);
                const int o = chunk_off<64>(row, col4 >> 1) + 4 * (col4 & 1);
                *reinterpret_cast<h4*>(buf.hi + o) = vh;
                *reinterpret_cast<h4*>(buf.lo + o) = vl;
            };
            auto finish = [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr ((i < 16 && (NEO_TP_ABLATE & 1)) || (i >= 16 && (NEO_TP_ABLATE & 2))) {
                } else if constexpr (i < 16) {
                    constexpr int c = i / 4, q = i % 4;
                    const int row = rg + 16 * q;
                    const f32x4 val = blend4(taps[i % RING], d_w[i & 1]);
                    *reinterpret_cast<f32x4*>(fbuf(c & 1) + row * 64 + ((col4 ^ (row & 15)) << 2)) = val;
                } else {
                    constexpr int w = i - 16, s2 = w / 12, q = (w % 12) / 3, j = w % 3;
                    const int row = rg + 16 * q;
                    const f32x4 wts = d_w[i & 1];
                    const f32x4 val = blend4(taps[i % RING], wts);
                    if constexpr (j == 0) wsum = val; else wsum = wsum + val;
                    if constexpr (j == 2) write_x(xbuf(s2), row, wsum);
                }
            };
            // chunk c of the pre-projected latent -> accumulators (this wave's pieces: 2 per M-tile)
            auto consume_chunk = [&](auto cc) __attribute__((always_inline)) {
                constexpr int c = decltype(cc)::value;
                if constexpr ((NEO_TP_ABLATE & 1) != 0) return;
                const float* buf = fbuf(c & 1);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int row = mt * 32 + L.l31;
                        const int piece = L.wv * 4 + gg * 2 + L.half;
                        const f32x4 val = *reinterpret_cast<const f32x4*>(buf + row * 64 + ((piece ^ (row & 15)) << 2));
                        constexpr int g0 = 2 * (c & 1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) accx[c >> 1][mt][4 * (g0 + gg) + e] += val[e];
                    }
            };
            // pos_enc: half hf of a stage = chunks 4hf..4hf+3, one per wave; a chunk = 8 features = 4 (sin, cos) PAIRS of the
            // pair order (launch_tp_pack_hp): pair p = 4 chunk + jj = octave * C + coordinate.  Both features of a pair
            // come out of one argument reduction (common.h:sincos_pair).  C = 4: a chunk is one octave, coordinate = jj.
            auto finish_pe = [&](const HT& buf, int pstage, int hf) __attribute__((always_inline)) {
                if constexpr ((NEO_TP_ABLATE & 4) != 0) return;
                const int row = tid & 63, q = tid >> 6;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(cam_enc + row * 4);
                const int chs = hf * 4 + q;                    // chunk inside this 64-feature stage (wave-uniform)
                const int ch = pstage * 8 + chs;               // chunk of the whole encoding
                float f[8];
                h8 vh, vl;
                if (ch * 4 < 10 * PE_C) {                      // pairs (C = 3: chunk 7 holds pairs 28, 29 and the identity features)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        if (PE_C == 3 && jj >= 2 && ch == 7) {  // pairs 30, 31 do not exist: positions 60..63 = x, y, z, 0
                            f[2 * jj] = jj == 2 ? xv[0] : xv[2];
                            f[2 * jj + 1] = jj == 2 ? xv[1] : 0.0f;
                            range_see(L, f[2 * jj]); range_see(L, f[2 * jj + 1]);
                            continue;
                        }
                        float x;
                        int oct;
                        if constexpr (PE_C == 4) {
                            x = xv[jj];
                            oct = ch;
                        } else {
                            const int p = ch * 4 + jj;         // wave-uniform: scalar arithmetic
                            oct = p / 3;
                            const int a = p - 3 * oct;
                            x = a == 0 ? xv[0] : a == 1 ? xv[1] : xv[2];
                        }
                        sincos_pair(ldexpf(x, oct), f[2 * jj], f[2 * jj + 1]);
                    }
                } else {                                        // C = 4: chunk 10 = x, y, z, 1/r; chunk 11 = padding
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = 0.0f;
                    if (ch * 4 == 10 * PE_C) {
                        range_see(L, xv[0]); range_see(L, xv[1]); range_see(L, xv[2]); range_see(L, xv[3]);
                        f[0] = xv[0]; f[1] = xv[1]; f[2] = xv[2]; f[3] = xv[3];
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    h2 h, l;
                    split2(f[e], f[e + 1], h, l);
                    vh[e] = h[0]; vh[e + 1] = h[1];
                    vl[e] = l[0]; vl[e + 1] = l[1];
                }
                const int o = chunk_off<64>(row, chs);
                *reinterpret_cast<h8*>(buf.hi + o) = vh;
                *reinterpret_cast<h8*>(buf.lo + o) = vl;
            };
            // streamed-stage weights: k-steps 0..KSX-1 of N-tiles wv (L0) and 4 + wv (L3 skip) in a ring, XD k-steps ahead
            constexpr int XD = NEO_TP_XSTREAM, XS = XD + 1;
            h8 wh[XS][2], wl[XS][2];
            const char* wxb = reinterpret_cast<const char*>(wp + hoff_x());
            uint32_t wx_off[2];
            wx_off[0] = (uint32_t)(L.wv * KSX * 2 * 64 + L.lane) * 16u;
            wx_off[1] = (uint32_t)((4 + L.wv) * KSX * 2 * 64 + L.lane) * 16u;
            auto load_wk = [&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (ks < KSX) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        wh[ks % XS][nt] = *reinterpret_cast<const h8*>(wxb + (wx_off[nt] + 2048u * ks));
                        wl[ks % XS][nt] = *reinterpret_cast<const h8*>(wxb + (wx_off[nt] + 2048u * ks + 1024u));
                    }
                }
            };
            auto mma_k = [&](const HT& tile, auto kc) __attribute__((always_inline)) {      // k-step ks: tile k-step ks % 4
                constexpr int ks = decltype(kc)::value;
                h8 bh[2], bl[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int o = chunk_off<64>(mt * 32 + L.l31, ((ks % 4) << 1) + L.half);
                    bh[mt] = *reinterpret_cast<const h8*>(tile.hi + o);
                    bl[mt] = *reinterpret_cast<const h8*>(tile.lo + o);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        accx[nt][mt] = NEO_MFMA_H_LH(wl[ks % XS][nt], bh[mt], accx[nt][mt]);
                        accx[nt][mt] = NEO_MFMA_H_HL(wh[ks % XS][nt], bl[mt], accx[nt][mt]);
                        accx[nt][mt] = NEO_MFMA_H(wh[ks % XS][nt], bh[mt], accx[nt][mt]);
                    }
                load_wk(std::integral_constant<int, ks + XD>());
            };
            // ---- gathers: two software pipelines (items 0..15 pre-projected latent, 16..39 tri-planes); the latent one is skipped
            //      as a whole when NO row of the tile has a non-zero tap weight in the latent for this view.  That is the normal
            //      case outside the unit sphere (half of all points): the far samples project outside every source image
            //      and lie outside the [-1,1]^3 tri-plane volume of the view (profiles/r03_tile_footprint.json: the median
            //      background tile-view touches ONE texel per map, i.e. only the zero-weight placeholder).  Their features are
            //      exactly zero (grid_sample zero padding), so the latent adds, the 8 world k-steps of the streamed GEMM and
            //      their weight fragments are skipped too.
            const bool any_latent = zm0 != 0ull;
            if (any_latent) {
                static_for<0, RING - 1>([&](auto ic) { fetch_off(ic); issue(ic); });
                fetch_off(std::integral_constant<int, RING - 1>());
                fetch_w(std::integral_constant<int, 0>());
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    // descriptors one item ahead: offsets of the item requested in the NEXT pass, weights of the item blended in it
                    if constexpr (i + RING < 16) fetch_off(std::integral_constant<int, i + RING>());
                    if constexpr (i + 1 < 16) fetch_w(std::integral_constant<int, i + 1>());
                    if constexpr (i + RING - 1 < 16) issue(std::integral_constant<int, i + RING - 1>());
                    if constexpr (i == 4 || i == 8 || i == 12) consume_chunk(std::integral_constant<int, i / 4 - 1>());
                    finish(ic);
                    __builtin_amdgcn_sched_barrier(0);      // keep the ring RING items deep: no hoisting of later items' loads
                    if constexpr (i == 3 || i == 7 || i == 11 || i == 15) TP_SYNC();
                });
                consume_chunk(std::integral_constant<int, 3>());
            }
            TP_MARK(2);
            static_for<0, XD>([&](auto kc) { load_wk(kc); });
            {   // the tri-plane pipeline always runs: world stage 0 is multiplied between the gather items of stage 1 (measured best, round 3)
                static_for<16, 16 + RING - 1>([&](auto ic) { fetch_off(ic); issue(ic); });
                fetch_off(std::integral_constant<int, 16 + RING - 1>());
                fetch_w(std::integral_constant<int, 16>());
                static_for<16, NI>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (i + RING < NI) fetch_off(std::integral_constant<int, i + RING>());
                    if constexpr (i + 1 < NI) fetch_w(std::integral_constant<int, i + 1>());
                    if constexpr (i + RING - 1 < NI) issue(std::integral_constant<int, i + RING - 1>());
                    if constexpr (i >= 28 && (i - 28) % 3 == 0)
                        mma_k(xbuf(0), std::integral_constant<int, (i - 28) / 3>());
                    finish(ic);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (i == 27 || i == 39) TP_SYNC();
                });
            }
            TP_MARK(3);
            // world stage 1 is multiplied while the first pos_enc stage is computed
            static_for<4, 8>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                mma_k(xbuf(1), kc);
                if constexpr (ks & 1) {
                    finish_pe(xbuf(0), 0, (ks - 4) >> 1);
                }
            });
            TP_SYNC();
            // ---- the pos_enc stage(s) ----
            static_for<8, 12>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                mma_k(xbuf(0), kc);
                if constexpr (NPE == 2 && ks == 9) finish_pe(xbuf(1), 1, 0);      // features 64..95 (84..95 are padding)
            });
            TP_SYNC();
            if constexpr (NPE == 2) {
                static_for<12, 14>([&](auto kc) { mma_k(xbuf(1), kc); });
                TP_SYNC();
            }
        }
        TP_MARK(4);
        // ---- L0 epilogue; L1, L2, L3 as ONE weight stream of 24 k-steps (N-tile = wave) requested LD k-steps ahead
        //      across the layer boundaries: the weights of the next layer do not wait for the barriers ----
        constexpr int LD = 4, LS = LD + 1;
        h8 lwh[LS], lwl[LS];
        const char* lwb = reinterpret_cast<const char*>(wp);
        const uint32_t lw_off = (uint32_t)(L.wv * 8 * 128 + L.lane) * 16u;
        auto load_l = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < 24) {
                constexpr int layer = g / 8, ks = g % 8;
                constexpr uint32_t base = (uint32_t)(layer == 0 ? hoff_1(PE_C) : layer == 1 ? hoff_2(PE_C) : hoff_3a(PE_C)) * 16u;
                lwh[g % LS] = *reinterpret_cast<const h8*>(lwb + (base + lw_off + 2048u * ks));
                lwl[g % LS] = *reinterpret_cast<const h8*>(lwb + (base + lw_off + 2048u * ks + 1024u));
            }
        };
        static_for<0, LD>([&](auto gc) { load_l(gc); });
        f32x16 acc[1][2];
        store_tile_h<true>(accx[0][0], act, L.wv, 0, L);
        store_tile_h<true>(accx[0][1], act, L.wv, 1, L);
        TP_SYNC();
        static_for<0, 24>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int layer = g / 8, ks = g % 8;
            if constexpr (ks == 0) {
                if constexpr (layer < 2) {
                    bias_tile(acc[0][0], lbias + (layer == 0 ? B_1 : B_2), L.wv, L);
                    acc[0][1] = acc[0][0];
                } else {
                    acc[0][0] = accx[1][0];
                    acc[0][1] = accx[1][1];
                }
            }
            load_l(std::integral_constant<int, g + LD>());
            h8 bh[2], bl[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int o = chunk_off<128>(mt * 32 + L.l31, (ks << 1) + L.half);
                bh[mt] = *reinterpret_cast<const h8*>(act.hi + o);
                bl[mt] = *reinterpret_cast<const h8*>(act.lo + o);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[0][mt] = NEO_MFMA_H_LH(lwl[g % LS], bh[mt], acc[0][mt]);
                acc[0][mt] = NEO_MFMA_H_HL(lwh[g % LS], bl[mt], acc[0][mt]);
                acc[0][mt] = NEO_MFMA_H(lwh[g % LS], bh[mt], acc[0][mt]);
            }
            if constexpr (ks == 7) {
                if constexpr (layer < 2) {
                    TP_SYNC();
                    store_tile_h<true>(acc[0][0], act, L.wv, 0, L);
                    store_tile_h<true>(acc[0][1], act, L.wv, 1, L);
                    TP_SYNC();
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        hsum[0][r] += relu1(acc[0][0][r]);
                        hsum[1][r] += relu1(acc[0][1][r]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        TP_SYNC();           // every wave is done reading this view's tiles
        TP_MARK(5);
    }

    // ---- view mean of the trunk -> density head ----
    // (x * (1 / nv), not x / nv: a full-precision fp32 division is 11 instructions and there are 40 of them per thread here -
    //  a tenth of the kernel's static VALU instructions, 3 % of its time; the product differs from the quotient by at most
    //  1 ulp of fp32, 2^-13 of the split-fp16 rounding that follows)
    const float inv_nv = 1.0f / (float)sc.nv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = hsum[0][r] * inv_nv; hsum[1][r] = hsum[1][r] * inv_nv; }
    store_tile_h<false>(hsum[0], act, L.wv, 0, L);
    store_tile_h<false>(hsum[1], act, L.wv, 1, L);
    // view mean of the direction encoding: fp32 sums -> hi/lo planes in place (read all, barrier, write)
    float dmean[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dmean[j] = dsum[(tid >> 2) * 32 + ((((tid & 3) << 3) + j) ^ ((tid >> 2) & 31))] * inv_nv;
    TP_SYNC();
    {
        h8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            h2 h, l;
            split2(dmean[j], dmean[j + 1], h, l);
            vh[j] = h[0]; vh[j + 1] = h[1];
            vl[j] = l[0]; vl[j + 1] = l[1];
        }
        const int o = chunk_off<32>(tid >> 2, tid & 3);
        *reinterpret_cast<h8*>(dsm.hi + o) = vh;
        *reinterpret_cast<h8*>(dsm.lo + o) = vl;
    }
    float raw_sigma;
    {
        float sg = density_partial(act, dens_w, L);
        sg += __shfl_xor(sg, 1, 64);
        sg += __shfl_xor(sg, 2, 64);
        raw_sigma = sg + lheads[HD_DB];
    }
    // ---- tail GEMMs as one weight stream of 14 k-steps, TD ahead across the stage boundary: view layer 0 WITH THE BOTTLENECK FOLDED
    //      IN (tp_hp_layout.h) on [mean trunk | mean dir enc] (N-tile vnt, M-tile vmt, 8 + 2 k-steps), then 64 x 64 (4 k-steps) ----
    {
        const char* twb = reinterpret_cast<const char*>(wp);
        constexpr int TD = 6, TS = TD + 1;
        h8 twh[TS], twl[TS];
        auto load_t = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < 14) {
                constexpr bool v0 = g < 10;
                constexpr int ks = v0 ? g : g - 10;
                constexpr int KS = v0 ? 10 : 4;
                constexpr uint32_t base = (uint32_t)(v0 ? hoff_v0(PE_C) : hoff_v1(PE_C)) * 16u;
                const uint32_t off = base + (uint32_t)((vnt * KS + ks) * 128 + L.lane) * 16u;
                twh[g % TS] = *reinterpret_cast<const h8*>(twb + off);
                twl[g % TS] = *reinterpret_cast<const h8*>(twb + off + 1024u);
            }
        };
        static_for<0, TD>([&](auto gc) { load_t(gc); });
        f32x16 y;
        static_for<0, 14>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            load_t(std::integral_constant<int, g + TD>());
            constexpr bool v0 = g < 10;
            constexpr int ks = v0 ? g : g - 10;
            if constexpr (ks == 0) bias_tile(y, lbias + (v0 ? B_V0 : B_V1), vnt, L);
            h8 bh, bl;
            if constexpr (v0 && ks >= 8) {
                const int o = chunk_off<32>(vmt * 32 + L.l31, ((ks - 8) << 1) + L.half);
                bh = *reinterpret_cast<const h8*>(dsm.hi + o);
                bl = *reinterpret_cast<const h8*>(dsm.lo + o);
            } else {
                const int o = chunk_off<128>(vmt * 32 + L.l31, (ks << 1) + L.half);
                bh = *reinterpret_cast<const h8*>(act.hi + o);
                bl = *reinterpret_cast<const h8*>(act.lo + o);
            }
            y = NEO_MFMA_H_LH(twl[g % TS], bh, y);
            y = NEO_MFMA_H_HL(twh[g % TS], bl, y);
            y = NEO_MFMA_H(twh[g % TS], bh, y);
            if constexpr (g == 9) {
                TP_SYNC();          // every wave has read the view-mean trunk (density head, view layer 0)
                store_tile_h<true>(y, act, vnt, vmt, L);
                TP_SYNC();
            }
            if constexpr (g == 13) {
                TP_SYNC();
                store_tile_h<true>(y, act, vnt, vmt, L);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    TP_SYNC();
    {
        const int pt = L.wv * 16 + (L.lane >> 2), part = L.lane & 3;
        const float* wr = lheads + HD_RW;
        float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int chunk_i = part * 2 + ((c + part) & 1);
            const int o = chunk_off<128>(pt, chunk_i);
            const h8 vh = *reinterpret_cast<const h8*>(act.hi + o);
            const h8 vl = *reinterpret_cast<const h8*>(act.lo + o);
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
        range_commit(L, m.flags);
        const long gv = tile0 + pt;
        const long gi = tp::patch_point(gv, N, R, sc.grid_w, sc.grid_first, sc.grid_pw, sc.grid_ph);
        if (part == 0 && gv < P) {
            out[gi] = make_float4(colour_act(r + lheads[HD_RB]), colour_act(g + lheads[HD_RB + 1]),
                                  colour_act(b + lheads[HD_RB + 2]), density_act(raw_sigma));
        }
    }
#if NEO_TP_TRACE
    TP_MARK(6);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) atomicAdd(&g_tp_trace[k], tr_[k]);
        atomicAdd(&g_tp_trace[7], 1ull);
    }
#endif
}

// ---- per-ray sum over the source views of the view-direction encoding (neo360/model.py:339-341, :357-360 feed
// pos_enc(R_v d, 0, 4) of the chunk's rays to every sample; the MLP sees it only through the view mean) ----
// one thread per ray; the arithmetic and the summation order (views 0, 1, .. onto 0.0f) are those the evaluator used
// per sample and view before this table existed
__global__ void k_tp_dirsum(const float* __restrict__ viewdirs, int R, TpViews views, int nv, float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float dx = viewdirs[r * 3], dy = viewdirs[r * 3 + 1], dz = viewdirs[r * 3 + 2];
    float acc[27];
#pragma unroll
    for (int f = 0; f < 27; ++f) acc[f] = 0.0f;
    for (int v = 0; v < nv; ++v) {
        const float* rot = views.rot[v];
        const float dc[3] = {rot[0] * dx + rot[1] * dy + rot[2] * dz, rot[3] * dx + rot[4] * dy + rot[5] * dz,
                             rot[6] * dx + rot[7] * dy + rot[8] * dz};
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] += dc[a];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float sn, cs;
                enc_pair(dc[a], k, sn, cs);
                acc[3 + k * 3 + a] += sn;
                acc[15 + k * 3 + a] += cs;
            }
    }
    float* o = out + (long)r * 32;
#pragma unroll
    for (int f = 0; f < 27; ++f) o[f] = acc[f];
#pragma unroll
    for (int f = 27; f < 32; ++f) o[f] = 0.0f;
}

// ---- G = F . [W0_loc | W3_loc]^T: exact fp32 MFMA, once per (scene, MLP) ----------------------------------
// F: channels-last latent (T texels, 512); wx: the fp32 fragment stream of stage X of mlp_tp.hip's pack
// (8 N-tiles x KC chunks of 8, packed k = [local 512 | world | pe]): its chunks 0..63 are exactly [W0_loc; W3_loc].
// One workgroup = 64 texels (2 M-tiles), wave w = N-tiles 2w, 2w+1.  B fragments come straight from global memory
// (16 B per lane at a 2 KB row pitch: every line is consumed over four consecutive chunks); the kernel is bound by
// the 64-cycle fp32 MFMA (60 GFLOP per MLP at 640x480 sources: ~1 ms).
// The same kernel projects a tri-plane through the WORLD columns (in_ch = 128, chunks c0 = 64..79 of the stream: packed k =
// [local 512 | world 128 | pe]) for mlp_tp_hpp.hip.
template <int NTW, int NC>      // N-tiles per wave: 2 -> 256 outputs per texel (NeRF_TP: [W0 | W3 skip]), 1 -> 128 (PixelNeRF: W0_loc); NC = in_ch / 8
__global__ __launch_bounds__(256, 2) void k_tp_preproject(const float* __restrict__ F, const f32x4* __restrict__ wx, int KC,
                                                           long T, float* __restrict__ G, int c0) {
    constexpr int in_ch = NC * 8;
    LaneCtx L;
    L.init();
    const long t0 = (long)blockIdx.x * 64;
    f32x16 acc[NTW][2];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const float* frow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        long t = t0 + mt * 32 + L.l31;
        if (t >= T) t = T - 1;
        frow[mt] = F + t * in_ch + 4 * L.half;
    }
#pragma unroll 2
    for (int c = 0; c < NC; ++c) {
        f32x4 a[NTW], b[2];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) a[nt] = wx[((NTW * L.wv + nt) * KC + c0 + c) * 64 + L.lane];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) b[mt] = *reinterpret_cast<const f32x4*>(frow[mt] + 8 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = NEO_MFMA(a[nt][e], b[mt][e], acc[nt][mt]);
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const long t = t0 + mt * 32 + L.l31;
            if (t >= T) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int o = (NTW * L.wv + nt) * 32 + 8 * g + 4 * L.half;
                f32x4 val;
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = acc[nt][mt][4 * g + e];
                *reinterpret_cast<f32x4*>(G + t * (NTW * 128) + proj_index(o)) = val;
            }
        }
}

}  // namespace

#if NEO_TP_TRACE
extern "C" void neo_debug_tp_trace(unsigned long long* host16, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_tp_trace), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tp_trace), z, sizeof(z));
    }
}
#endif

size_t tp_wpack_hp_bytes(int input_ch) { return (size_t)hpack_h8(input_ch) * 16; }
size_t tp_proj_bytes(long texels) { return (size_t)texels * PROJ_TEXEL_BYTES; }
size_t tp_proj_pad_bytes() { return 8192; }

size_t tp_fold_floats() { return 64 * 155; }

void launch_tp_pack_hp(int input_ch, const float* const* w, const float* const* b, void* wpack_hp, float* fold_ws,
                       const float* bias_src, float* bias_hp, hipStream_t s) {
    // w / b order: pts_linears.0..3, views_linear.0, views_linear.1, bottleneck, density, rgb
    // biases of the pre-projected evaluators = the shared bias block with view layer 0's entry replaced by the folded one
    (void)hipMemcpyAsync(bias_hp, bias_src, 768 * sizeof(float), hipMemcpyDeviceToDevice, s);
    launch_fold_bottleneck(w[4], w[6], b[6], b[4], 64, 128, 128, 27, fold_ws, bias_hp + B_V0, s);
    const float* w_v0 = fold_ws;
    _Float16* base = reinterpret_cast<_Float16*>(wpack_hp);
    const int pe = input_ch * 21;
    const int x0w = pe + 512 + 128;
    const int ksx = ks_x(input_ch);
    const PackSegs none = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    // streamed stage: packed k = [world 128 | pos_enc in PAIR order | pad] <- x0 columns [pe | local 512 | world 128].
    // Pair order (finish_pe): position 2 p / 2 p + 1 = sin / cos feature of pair p = octave * C + coordinate (reference
    // columns C + p and C + 10 C + p), then the C identity features (columns 0..C-1), zero padding to the stage width.
    const int C = input_ch;
    PackPerm px;
    for (int k = 0; k < 256; ++k) px.col[k] = -1;
    for (int k = 0; k < 128; ++k) px.col[k] = (short)(pe + 512 + k);
    for (int j = 0; j < 20 * C; ++j) px.col[128 + j] = (short)(C + ((j & 1) ? 10 * C : 0) + (j >> 1));
    for (int a = 0; a < C; ++a) px.col[128 + 20 * C + a] = (short)a;
    pack_h_perm(w[0], x0w, 128, ksx, 0, px, base + (long)hoff_x() * 8, s);
    PackPerm px3 = px;
    for (int k = 0; k < 256; ++k)
        if (px3.col[k] >= 0) px3.col[k] = (short)(px3.col[k] + 128);               // L3 input = [h(128) | x0]
    pack_h_perm(w[3], 128 + x0w, 128, ksx, 4, px3, base + (long)hoff_x() * 8, s);
    PackSegs p128 = none;
    p128.len[0] = 128;
    pack_h(w[1], 128, 128, 8, 0, p128, base + (long)hoff_1(input_ch) * 8, s);
    pack_h(w[2], 128, 128, 8, 0, p128, base + (long)hoff_2(input_ch) * 8, s);
    pack_h(w[3], 128 + x0w, 128, 8, 0, p128, base + (long)hoff_3a(input_ch) * 8, s);
    pack_h(w[6], 128, 128, 8, 0, p128, base + (long)hoff_b(input_ch) * 8, s);
    PackSegs v0 = none;
    v0.len[0] = 155;
    pack_h(w_v0, 155, 64, 10, 0, v0, base + (long)hoff_v0(input_ch) * 8, s);
    PackSegs v1 = none;
    v1.len[0] = 64;
    pack_h(w[5], 64, 64, 4, 0, v1, base + (long)hoff_v1(input_ch) * 8, s);
}

void launch_tp_preproject(const float* latent_cl, long texels, const float* wpack_f32_stage_x, int kc_x, float* proj,
                          hipStream_t s, int channels, int in_ch, int first_chunk) {
    if (texels <= 0) return;
    const dim3 grid((unsigned)((texels + 63) / 64));
    const f32x4* wx = reinterpret_cast<const f32x4*>(wpack_f32_stage_x);
    if (channels == 128 && in_ch == 512)
        hipLaunchKernelGGL((k_tp_preproject<1, 64>), grid, dim3(256), 0, s, latent_cl, wx, kc_x, texels, proj, first_chunk);
    else if (in_ch == 512)
        hipLaunchKernelGGL((k_tp_preproject<2, 64>), grid, dim3(256), 0, s, latent_cl, wx, kc_x, texels, proj, first_chunk);
    else          // in_ch == 128: a tri-plane through the world columns
        hipLaunchKernelGGL((k_tp_preproject<2, 16>), grid, dim3(256), 0, s, latent_cl, wx, kc_x, texels, proj, first_chunk);
}

void launch_tp_dirsum(const float* viewdirs, int R, const TpViews& views, int nv, float* dirsum, hipStream_t s) {
    if (R <= 0) return;
    hipLaunchKernelGGL(k_tp_dirsum, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, viewdirs, R, views, nv, dirsum);
}

void launch_tp_mlp_hp(int input_ch, const TpMlpHDev& m, const float* proj, const TpScene& sc, const TpViews& views,
                      const float* rays_o, const float* rays_d, const float* viewdirs, const float* tvals,
                      const float* far, int R, int N, int chunk, uint32_t* flags, float* out, const float* dirsum,
                      hipStream_t s) {
    const long P = (long)R * N;
    if (P <= 0) return;
    static int stagger = -1;
    if (stagger < 0) {
        const char* e = getenv("NEO_TP_STAGGER");
        stagger = e ? atoi(e) : NEO_TP_STAGGER_DEFAULT;
    }
    static size_t lds_pad = ~size_t(0);
    if (lds_pad == ~size_t(0)) {          // occupancy experiments: $NEO_TP_LDS_PAD extra bytes of dynamic LDS per workgroup
        const char* e = getenv("NEO_TP_LDS_PAD");
        lds_pad = e ? (size_t)atol(e) : 0;
    }
    const size_t lds = (tp::LDS_WORDS + 768 + 336) * sizeof(float) + lds_pad;
    if (lds > 65536) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_mlp_hp<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_mlp_hp<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const long tiles = tp::xcd_grid((P + TM - 1) / TM);
    if (input_ch == 3)
        hipLaunchKernelGGL(k_tp_mlp_hp<3>, dim3((unsigned)tiles), dim3(256), lds, s, m, proj, sc, views, rays_o, rays_d,
                           viewdirs, tvals, far, R, N, chunk, flags, reinterpret_cast<float4*>(out), stagger, dirsum);
    else
        hipLaunchKernelGGL(k_tp_mlp_hp<4>, dim3((unsigned)tiles), dim3(256), lds, s, m, proj, sc, views, rays_o, rays_d,
                           viewdirs, tvals, far, R, N, chunk, flags, reinterpret_cast<float4*>(out), stagger, dirsum);
}

}  // namespace neo

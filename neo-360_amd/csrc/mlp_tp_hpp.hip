// NeO-360 decoder point evaluator, split-fp16 arithmetic, with ALL FOUR gathered maps pre-projected through the first-layer
// weights once per (scene, MLP): the pixel-aligned latent (as mlp_tp_hp.hip) AND the three tri-planes.
//
// NeRFPPMLP (neo360/model.py:110-158) consumes the 128-channel tri-plane sum (encoder_tp_fusion_conv.py:180-206: three
// grid_samples, summed) only through two linear maps - the world columns of pts_linears.0 and of the skip half of
// pts_linears.3 (model.py:123-137) - exactly like the 512-channel latent.  Bilinear interpolation is linear, so
//        W_world . sum_j bilerp(P_j) = sum_j bilerp(W_world . P_j)
// and G_j = P_j . [W0_world | W3_world]^T (256 channels per texel, k_tp_preproject on chunks 64..79 of the fp32 fragment
// stream) is gathered instead of P_j.  Per point-view that removes the world GEMM stage - 8 of the 12 (inside) / 14
// (outside) streamed k-steps: 32,768 of 98,304 executed MACs with their hi/lo splits, LDS plane stores, B-fragment reads
// and weight fragments - and replaces 3 x 512 B taps by 3 x 1 KB taps (24 more gather items per tile-view).
//
// Per 64-point tile and source view: descriptors and the pos_enc features (into their OWN LDS tile) first; then a software
// pipeline over gather items - one item = (chunk of 64 output channels, group of 16 rows, map): 4 taps x 256 B per row - a
// ring of RING tap register sets deep with the tap descriptors read from LDS one item ahead; the maps' blends of a
// (chunk, row group) are summed in registers, travel through the 16 KB fp32 transposition tile and are ADDED to the L0 / L3-skip
// accumulators.  The four projected maps live in ONE buffer (one base pointer, 32-bit byte offsets).
// WORK LIST (default): a sample outside a map blends to exactly zero (grid_sample zero padding), and outside the unit
// sphere that is the normal case - so per tile-view the kernel lists the (row group, map) pairs in which ANY row carries a
// weight (6 of 16 on average outside the sphere, 14 inside: tools/footprint_study.py) and the pipeline runs over that list,
// padded to a multiple of RING with zero-weight entries so the ring slots stay compile-time constants; every load of the
// loop is unconditional, so its s_waitcnt bookkeeping stays exact.
// The only matrix work left before the L0 epilogue - the pos_enc k-steps (4 inside / 6 outside the sphere) - is issued
// between the chunks.  L1, L2, L3, view-mean linearity and the heads are those of mlp_tp_hp.hip.
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "tp_hp_layout.h"

#ifndef NEO_TPP_WPS
#define NEO_TPP_WPS 2          // workgroups per CU
#endif
#ifndef NEO_TPP_RING
#define NEO_TPP_RING 3         // tap register sets (16 VGPRs each); prefetch distance = RING - 1 items
#endif
#ifndef NEO_TPP_XD
#define NEO_TPP_XD 1           // pos_enc weight fragments requested this many k-steps ahead (a k-step sits between two chunks of the gather: one ahead is a whole chunk of time)
#endif
#ifndef NEO_TPP_LD
#define NEO_TPP_LD 4           // L1..L3 weight stream prefetch distance (k-steps)
#endif
#ifndef NEO_TPP_TD
#define NEO_TPP_TD 6           // tail weight stream prefetch distance
#endif
#define TPP_SYNC() __syncthreads()
#ifndef NEO_TP_TRACE
#define NEO_TP_TRACE 0         // 1: per-phase s_memtime sums of wave 0 of every workgroup -> g_tpp_trace (tools/bench_tp_kernel.py TRACE=1)
#endif
#if NEO_TP_TRACE
__device__ unsigned long long g_tpp_trace[16];
#define TPP_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tr_[k] += now_ - tlast_; tlast_ = now_; } while (0)
#else
#define TPP_MARK(k) do { } while (0)
#endif

namespace neo {

namespace {

using namespace hp;
constexpr int RING = NEO_TPP_RING;

// LDS carve (4-byte words).  Tap descriptors: 5 blocks of [64 rows][4 taps] - block 0 the latent, 1..3 the planes, block 4
// all zeros (offset 0, weight 0: the padding entries of the work list) - byte offsets first, weights DESC_W words later.
constexpr int PP_OFF_DESC = tp::OFF_DIR + TM * 32;           // after the activation tile (8192) and the direction tile (2048)
constexpr int DESC_BLOCK = TM * 4, DESC_W = 5 * DESC_BLOCK;  // words
constexpr int PP_OFF_LIST = PP_OFF_DESC + 2 * DESC_W;        // [4 waves][LIST_MAX] int4 records
constexpr int LIST_MAX = 18;
constexpr int PP_OFF_CAM = PP_OFF_LIST + 4 * LIST_MAX * 4;
constexpr int PP_OFF_PE = PP_OFF_CAM + TM * 4;
constexpr int PP_OFF_FEAT = PP_OFF_PE + TM * 4;
constexpr int PP_OFF_VDIR = PP_OFF_FEAT + TM * 4;
constexpr int PP_OFF_DENSW = PP_OFF_VDIR + TM * 4;
constexpr int PP_OFF_BIAS = PP_OFF_DENSW + 128;
constexpr int PP_OFF_XPE0 = PP_OFF_BIAS + 768 + 336;
constexpr int PP_LDS_WORDS = PP_OFF_XPE0 + TM * 64;          // [64][64] halves x 2 planes = 16 KB; total 77,760 B: 2 workgroups per CU
static_assert(PP_LDS_WORDS * 4 <= 81920, "two workgroups per CU");

struct TpPlaneBase { int texels[3]; };      // first texel of each projected tri-plane inside the one projected-maps buffer

template <int PE_C>
__global__ __launch_bounds__(256, NEO_TPP_WPS) void k_tp_mlp_hpp(TpMlpHDev m, const float* __restrict__ proj, TpPlaneBase plb,
                                                               TpScene sc, TpViews views, const float* __restrict__ rays_o,
                                                               const float* __restrict__ rays_d,
                                                               const float* __restrict__ viewdirs,
                                                               const float* __restrict__ tvals,
                                                               const float* __restrict__ far_arr, int R, int N, int chunk,
                                                               uint32_t* __restrict__ flags, float4* __restrict__ out,
                                                               const float* __restrict__ dirsum) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* hbase = reinterpret_cast<_Float16*>(smem + tp::OFF_ACT);
    const HT act{hbase, hbase + TM * 128};                                   // [64][128] x 2 planes (32 KB)
    auto fbuf = [&](int b) { return smem + tp::OFF_ACT + b * (TM * 64); };   // the same 32 KB as two fp32 [64][64] tiles
    _Float16* xb0 = reinterpret_cast<_Float16*>(smem + PP_OFF_XPE0);
    const HT xpe0{xb0, xb0 + TM * 64};                                       // pos_enc features 0..63, [64][64] x 2 planes
    _Float16* dbase = reinterpret_cast<_Float16*>(smem + tp::OFF_DIR);
    const HT dsm{dbase, dbase + TM * 32};                                    // [64][32] x 2 planes: view loop = pos_enc features 64..95
    const HT xpe1 = dsm;                                                     //   (outside the sphere); tail = mean direction encoding
    tp::Scratch S;
    S.loc_off = reinterpret_cast<int*>(smem + PP_OFF_DESC);
    S.pl_off = S.loc_off + DESC_BLOCK;
    S.loc_w = smem + PP_OFF_DESC + DESC_W;
    S.pl_w = S.loc_w + DESC_BLOCK;
    S.cam_enc = smem + PP_OFF_CAM;
    S.pe_world = smem + PP_OFF_PE;
    S.feat_world = smem + PP_OFF_FEAT;
    S.vdir_world = smem + PP_OFF_VDIR;
    [[maybe_unused]] int* loc_off = S.loc_off;
    float* loc_w = S.loc_w;
    [[maybe_unused]] int* pl_off = S.pl_off;

    LaneCtx L;
    L.init();
    int tid = threadIdx.x;
    const long P = (long)R * N;
    const long tile0 = tp::xcd_tile(blockIdx.x, (P + TM - 1) / TM) * TM;
    if (tile0 >= P) return;       // surplus workgroup of the rounded-up grid (uniform exit before any barrier)
    const h8* wp = reinterpret_cast<const h8*>(m.wpack);
    constexpr int KSX = ks_x(PE_C);              // k-steps per N-tile in the packed streamed stage: 8 world (unused here) + pos_enc
    constexpr int KSP = pe_ksteps(PE_C);         // pos_enc k-steps: 4 (63 -> 64 features) / 6 (84 -> 96)
    constexpr int NPE = PE_C == 3 ? 1 : 2;

#if NEO_TP_TRACE
    unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = __builtin_amdgcn_s_memtime();
#endif
    tp::point_setup<PE_C>(S, tid, tile0, P, N, R, chunk, rays_o, rays_d, viewdirs, tvals, far_arr, flags, false, sc.grid_w, sc.grid_first, sc.grid_pw, sc.grid_ph);
    float* dens_w = smem + PP_OFF_DENSW;
    if (tid < 128) dens_w[tid] = m.heads[HD_DW + tid];
    if (tid < DESC_BLOCK) {                            // the all-zero descriptor block (offset 0 = the buffer's first texel, weight 0)
        S.loc_off[4 * DESC_BLOCK + tid] = 0;
        S.loc_w[4 * DESC_BLOCK + tid] = 0.0f;
    }
    float* lbias_w = smem + PP_OFF_BIAS;
    for (int i = tid; i < 768; i += 256) lbias_w[i] = m.bias[i];
    for (int i = tid; i < HD_RB + 3; i += 256) lbias_w[768 + i] = m.heads[i];
    const float* lbias = lbias_w;
    const float* lheads = lbias_w + 768;
    TPP_SYNC();

    // view means by linearity (mlp_tp_h.hip): only sum_v relu(L3_v) is accumulated per view; sum_v dir_enc_v comes ready-made
    // per ray from k_tp_dirsum and is only needed in the tail
    f32x16 hsum[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = 0.f; hsum[1][r] = 0.f; }
    const int vnt = L.wv & 1, vmt = L.wv >> 1;

    TPP_MARK(0);
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
        tp::view_descriptors<PROJ_TEXEL_BYTES, false, PROJ_TEXEL_BYTES>(S, L, sc, rot, trn, v, [](int, int, float) {},
                                                                        L.wv == 0 ? 0 : plb.texels[L.wv == 0 ? 0 : L.wv - 1]);
        // ---- pos_enc of this view's camera-frame point into its own tile: row = tid % 64, wave q takes chunks q and 4 + q of
        //      stage 0 (and chunk q of stage 1).  The camera-frame point is computed here with the arithmetic of
        //      view_descriptors (same expression, no contraction), so no barrier separates it from the descriptors. ----
        {
            const int row = tid & 63, q = tid >> 6;
            const float ex = S.pe_world[row * 4], ey = S.pe_world[row * 4 + 1], ez = S.pe_world[row * 4 + 2];
            f32x4 xv;
            xv[0] = (rot[0] * ex + rot[1] * ey + rot[2] * ez) + trn[0];
            xv[1] = (rot[3] * ex + rot[4] * ey + rot[5] * ez) + trn[1];
            xv[2] = (rot[6] * ex + rot[7] * ey + rot[8] * ez) + trn[2];
            xv[3] = S.pe_world[row * 4 + 3];
            // a chunk = 8 features = 4 (sin, cos) PAIRS of the pair order (launch_tp_pack_hp): pair p = 4 chunk + jj = octave * C +
            // coordinate; both features of a pair come out of one argument reduction (common.h:sincos_pair)
            auto pe_chunk = [&](auto ldh, const HT& buf, int pstage, int hf) __attribute__((always_inline)) {
                constexpr int LDH = decltype(ldh)::value;
                const int chs = hf * 4 + q;                    // chunk inside this stage (wave-uniform)
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
                const int o = chunk_off<LDH>(row, chs);
                *reinterpret_cast<h8*>(buf.hi + o) = vh;
                *reinterpret_cast<h8*>(buf.lo + o) = vl;
            };
            pe_chunk(std::integral_constant<int, 64>(), xpe0, 0, 0);
            pe_chunk(std::integral_constant<int, 64>(), xpe0, 0, 1);
            if constexpr (NPE == 2) pe_chunk(std::integral_constant<int, 32>(), xpe1, 1, 0);      // features 64..95 (84..95 padding)
        }
        TPP_SYNC();
        TPP_MARK(1);
        // (the work list below decides per (row group, map) what is gathered)

        // ---- [L0 | L3 skip half] pre-activations: bias + four pre-projected maps (adds) + pos_enc GEMM ----
        f32x16 accx[2][2];
        bias_tile(accx[0][0], lbias + B_0, L.wv, L);
        bias_tile(accx[1][0], lbias + B_3, L.wv, L);
        accx[0][1] = accx[0][0];
        accx[1][1] = accx[1][0];
        {
            const int col4 = tid & 15, rg = tid >> 4;
            const uint32_t lane_b = 16u * col4;
            f32x4 taps[RING][4];
            f32x4 wsum;                                    // running sum over the maps of one (chunk, row group)
            int4 d_off[RING];                              // tap byte offsets of the item REQUESTED next (slot = item & 1; work list: entry % 3)
            [[maybe_unused]] f32x4 d_w[2];                 // tap weights of the item BLENDED next
            f32x4 d_w3[RING];                              // (work-list pipeline: slot = entry % 3, like the taps)
            // pos_enc weight fragments: k-steps 8..8+KSP-1 of N-tiles wv (L0) and 4 + wv (L3 skip) of the packed streamed stage
            constexpr int XD = NEO_TPP_XD, XS = XD + 1;
            h8 wh[XS][2], wl[XS][2];
            const char* wxb = reinterpret_cast<const char*>(wp + hoff_x());
            uint32_t wx_off[2];
            wx_off[0] = (uint32_t)((L.wv * KSX + 8) * 2 * 64 + L.lane) * 16u;
            wx_off[1] = (uint32_t)(((4 + L.wv) * KSX + 8) * 2 * 64 + L.lane) * 16u;
            auto load_wk = [&](auto kc) __attribute__((always_inline)) {
                constexpr int kp = decltype(kc)::value;
                if constexpr (kp < KSP) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        wh[kp % XS][nt] = *reinterpret_cast<const h8*>(wxb + (wx_off[nt] + 2048u * kp));
                        wl[kp % XS][nt] = *reinterpret_cast<const h8*>(wxb + (wx_off[nt] + 2048u * kp + 1024u));
                    }
                }
            };
            auto mma_k = [&](auto kc) __attribute__((always_inline)) {
                constexpr int kp = decltype(kc)::value;
                h8 bh[2], bl[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if constexpr (kp < 4) {
                        const int o = chunk_off<64>(mt * 32 + L.l31, (kp << 1) + L.half);
                        bh[mt] = *reinterpret_cast<const h8*>(xpe0.hi + o);
                        bl[mt] = *reinterpret_cast<const h8*>(xpe0.lo + o);
                    } else {
                        const int o = chunk_off<32>(mt * 32 + L.l31, ((kp - 4) << 1) + L.half);
                        bh[mt] = *reinterpret_cast<const h8*>(xpe1.hi + o);
                        bl[mt] = *reinterpret_cast<const h8*>(xpe1.lo + o);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        accx[nt][mt] = NEO_MFMA_H_LH(wl[kp % XS][nt], bh[mt], accx[nt][mt]);
                        accx[nt][mt] = NEO_MFMA_H_HL(wh[kp % XS][nt], bl[mt], accx[nt][mt]);
                        accx[nt][mt] = NEO_MFMA_H(wh[kp % XS][nt], bh[mt], accx[nt][mt]);
                    }
                load_wk(std::integral_constant<int, kp + XD>());
            };
            // chunk c of the summed projected maps -> accumulators (this wave's pieces: 2 per M-tile)
            auto consume_chunk = [&](auto cc) __attribute__((always_inline)) {
                constexpr int c = decltype(cc)::value;
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
            static_assert(RING == 3 && LIST_MAX % RING == 0, "the work-list pipeline is written for a ring of three");
            // ---- the work list of this tile-view.  Every wave builds its own copy (identical in all four: no barrier). ----
            int n_p;                                         // entries incl. padding: a multiple of RING, 6..18 (wave-uniform)
            bool any_w;                                      // some tap of some map carries weight in this tile-view (wave-uniform, the same in all four waves)
            const int4* list;
            {
                int4* mylist = reinterpret_cast<int4*>(smem + PP_OFF_LIST) + L.wv * LIST_MAX;
                unsigned am = 0;                             // bit 4 q + m: a row of group q (rows 16 q .. 16 q + 15) has a weighted tap in map m
#pragma unroll
                for (int mp = 0; mp < 4; ++mp) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(loc_w + (mp * TM + L.lane) * 4);
                    const unsigned long long z = __ballot(w[0] != 0.0f || w[1] != 0.0f || w[2] != 0.0f || w[3] != 0.0f);
#pragma unroll
                    for (int q = 0; q < 4; ++q) am |= (((z >> (16 * q)) & 0xFFFFull) != 0ull ? 1u : 0u) << (4 * q + mp);
                }
                any_w = am != 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q)                  // an empty group keeps ONE entry: the latent's all-zero weights blend
                    if (((am >> (4 * q)) & 0xFu) == 0u) am |= 1u << (4 * q);      // to the zeros the group's rows must receive
                const int n = __builtin_popcount(am);
                n_p = ((n + RING - 1) / RING) * RING;
                // lane b < 16 with bit b set writes ITS record at its rank (number of set bits below b); lanes 32 + n .. 32 + n_p - 1
                // write the padding records behind them (scatter by rank: a dozen instructions; finding the i-th set bit per
                // lane took ~80)
                {
                    const int b = L.lane;
                    const bool real = b < 16 && ((am >> b) & 1u);
                    const bool pad = b - 32 >= n && b - 32 < n_p;
                    if (real || pad) {
                        int4 rec;                            // x: byte offset of the group's first descriptor row; y: 16 q;
                        int at = b - 32;                     // z: 0.0f for the first map of a group, 1.0f after (wsum * z + blend); w: last map
                        if (real) {
                            const int q = b >> 2, mp = b & 3;
                            const unsigned grp = (am >> (4 * q)) & 0xFu;
                            rec.x = (mp * TM + 16 * q) * 16;
                            rec.y = 16 * q;
                            rec.z = (grp & ((1u << mp) - 1u)) == 0u ? 0 : 0x3f800000;
                            rec.w = (grp >> (mp + 1)) == 0u ? 1 : 0;
                            at = __builtin_popcount(am & ((1u << b) - 1u));
                        } else {
                            rec.x = 4 * DESC_BLOCK * 4;      // padding: the all-zero block - adds 0.0, stores nothing
                            rec.y = 0;
                            rec.z = 0x3f800000;
                            rec.w = 0;
                        }
                        mylist[at] = rec;
                    }
                }
                list = mylist;
            }
            const char* dbase = reinterpret_cast<const char*>(S.loc_off) + rg * 16;     // this thread's row inside a 16-row group
            float* fstore = fbuf(0) + rg * 64 + ((col4 ^ rg) << 2);                     // its 16 B inside a group's rows of the transposition tile
            auto wrap = [&](int idx) __attribute__((always_inline)) { return idx >= n_p ? idx - n_p : idx; };
            int rx[RING];                                    // record offsets: rx[e % 3] (see the step)
            int4 fin[RING];                                  // blend records
#pragma unroll
            for (int e = 0; e < 4; ++e) wsum[e] = 0.0f;
            // prologue: entries 0 and 1 of chunk 0 requested, descriptors of entry 2 and weights of entry 0 on their way
            static_for<0, XD>([&](auto kc) { load_wk(kc); });
            if (any_w) {
                const int r0 = list[0].x, r1 = list[1].x, r2 = list[2].x;
                fin[0] = list[0];
                rx[0] = list[3].x; rx[1] = r1; rx[2] = r2;
                const int4 o0 = *reinterpret_cast<const int4*>(dbase + r0);
                const int4 o1 = *reinterpret_cast<const int4*>(dbase + r1);
                d_off[2] = *reinterpret_cast<const int4*>(dbase + r2);
                d_w3[0] = *reinterpret_cast<const f32x4*>(dbase + r0 + DESC_W * 4);
                taps[0][0] = tp::load_tap(proj, (uint32_t)o0.x + lane_b); taps[0][1] = tp::load_tap(proj, (uint32_t)o0.y + lane_b);
                taps[0][2] = tp::load_tap(proj, (uint32_t)o0.z + lane_b); taps[0][3] = tp::load_tap(proj, (uint32_t)o0.w + lane_b);
                taps[1][0] = tp::load_tap(proj, (uint32_t)o1.x + lane_b); taps[1][1] = tp::load_tap(proj, (uint32_t)o1.y + lane_b);
                taps[1][2] = tp::load_tap(proj, (uint32_t)o1.z + lane_b); taps[1][3] = tp::load_tap(proj, (uint32_t)o1.w + lane_b);
            }
            TPP_MARK(2);
            // One step = one list entry e of chunk c (ring slot j = e % 3: n_p is a multiple of 3, so slots are static):
            //   weights of entry e + 1, records of e + 4 / e + 1, tap offsets of e + 3 are read from LDS (used in the NEXT step);
            //   the taps of entry e + 2 are requested; entry e is blended, added to the group's running sum, and the sum stored
            //   when e is its group's last map.  Indices past the list wrap into the next chunk (same list, next 256 B).
            auto step = [&](auto cc, auto jc, int k) __attribute__((always_inline)) {
                constexpr int c = decltype(cc)::value, j = decltype(jc)::value;
                constexpr int j1 = (j + 1) % RING, j2 = (j + 2) % RING;
                const int e = k + j;
                d_w3[j1] = *reinterpret_cast<const f32x4*>(dbase + rx[j1] + DESC_W * 4);        // rx[j1] still is entry e + 1's
                rx[j1] = list[wrap(e + 4)].x;
                fin[j1] = list[wrap(e + 1)];
                d_off[j] = *reinterpret_cast<const int4*>(dbase + rx[j]);                         // entry e + 3
                {
                    const float* base = proj + 64 * (e + 2 >= n_p ? c + 1 : c);                   // chunk = 64 fp32 channels (scalar select)
                    const int4 off = d_off[j2];
                    taps[j2][0] = tp::load_tap(base, (uint32_t)off.x + lane_b);
                    taps[j2][1] = tp::load_tap(base, (uint32_t)off.y + lane_b);
                    taps[j2][2] = tp::load_tap(base, (uint32_t)off.z + lane_b);
                    taps[j2][3] = tp::load_tap(base, (uint32_t)off.w + lane_b);
                }
                const f32x4 val = blend4(taps[j], d_w3[j]);
                const float keep = __int_as_float(fin[j].z);
#pragma unroll
                for (int t = 0; t < 4; ++t) wsum[t] = __builtin_fmaf(wsum[t], keep, val[t]);
                if (fin[j].w) *reinterpret_cast<f32x4*>(fstore + (c & 1) * (TM * 64) + fin[j].y * 64) = wsum;
                __builtin_amdgcn_sched_barrier(0);          // keep the ring RING entries deep: no hoisting of later entries' loads
            };
            // an empty tile-view runs the same code with a zero-trip gather loop and without the adds: the pos_enc k-steps keep
            // their places between the chunks and stay outside any branch (the barriers stay too: four cheap ones per view)
            if (!any_w) n_p = 0;
            static_for<0, 4>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                // this chunk's share of the pos_enc k-steps (KSP = 4: one per chunk; 6: 2, 1, 2, 1)
                static_for<0, KSP>([&](auto kc) {
                    constexpr int kp = decltype(kc)::value;
                    if constexpr ((KSP == 4 ? kp : (kp < 2 ? 0 : kp < 3 ? 1 : kp < 5 ? 2 : 3)) == c) mma_k(kc);
                });
#pragma unroll 1
                for (int k = 0; k < n_p; k += RING) static_for<0, RING>([&](auto jc) { step(cc, jc, k); });
                TPP_SYNC();
                if (any_w)
                consume_chunk(cc);
            });
        }
        TPP_SYNC();          // the transposition tiles alias the activation tile: every wave has consumed the last chunk
        TPP_MARK(3);
        // ---- L0 epilogue; L1, L2, L3 as ONE weight stream of 24 k-steps (N-tile = wave) requested LD k-steps ahead
        //      across the layer boundaries: the weights of the next layer do not wait for the barriers ----
        {
            constexpr int LD = NEO_TPP_LD, LS = LD + 1;
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
            f32x16 acc[2];
            store_tile_h<true>(accx[0][0], act, L.wv, 0, L);
            store_tile_h<true>(accx[0][1], act, L.wv, 1, L);
            TPP_SYNC();
            static_for<0, 24>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int layer = g / 8, ks = g % 8;
                if constexpr (ks == 0) {
                    if constexpr (layer < 2) {
                        bias_tile(acc[0], lbias + (layer == 0 ? B_1 : B_2), L.wv, L);
                        acc[1] = acc[0];
                    } else {
                        acc[0] = accx[1][0];
                        acc[1] = accx[1][1];
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
                    acc[mt] = NEO_MFMA_H_LH(lwl[g % LS], bh[mt], acc[mt]);
                    acc[mt] = NEO_MFMA_H_HL(lwh[g % LS], bl[mt], acc[mt]);
                    acc[mt] = NEO_MFMA_H(lwh[g % LS], bh[mt], acc[mt]);
                }
                if constexpr (ks == 7) {
                    if constexpr (layer < 2) {
                        TPP_SYNC();
                        store_tile_h<true>(acc[0], act, L.wv, 0, L);
                        store_tile_h<true>(acc[1], act, L.wv, 1, L);
                        TPP_SYNC();
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            hsum[0][r] += relu1(acc[0][r]);
                            hsum[1][r] += relu1(acc[1][r]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        TPP_SYNC();           // every wave is done reading this view's tiles
        TPP_MARK(4);
    }

    // ---- view mean of the trunk -> density head (x * (1 / nv): see mlp_tp_hp.hip) ----
    const float inv_nv = 1.0f / (float)sc.nv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = hsum[0][r] * inv_nv; hsum[1][r] = hsum[1][r] * inv_nv; }
    store_tile_h<false>(hsum[0], act, L.wv, 0, L);
    store_tile_h<false>(hsum[1], act, L.wv, 1, L);
    {
        // view mean of the direction encoding straight from the per-ray table of this launch (k_tp_dirsum: the sum over
        // the views of the ray whose direction this point carries, quirk Q1) -> hi/lo planes; 8 features per thread
        const int p = tid >> 2, f0 = (tid & 3) << 3;
        const int dray = __float_as_int(S.vdir_world[p * 4 + 3]);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(dirsum + (long)dray * 32 + f0);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(dirsum + (long)dray * 32 + f0 + 4);
        h8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const float a = (j < 4 ? s0[j] : s1[j - 4]) * inv_nv, b = (j < 4 ? s0[j + 1] : s1[j - 3]) * inv_nv;
            h2 h, l;
            split2(a, b, h, l);
            vh[j] = h[0]; vh[j + 1] = h[1];
            vl[j] = l[0]; vl[j + 1] = l[1];
        }
        const int o = chunk_off<32>(p, tid & 3);
        *reinterpret_cast<h8*>(dsm.hi + o) = vh;
        *reinterpret_cast<h8*>(dsm.lo + o) = vl;
    }
    TPP_SYNC();
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
        constexpr int TD = NEO_TPP_TD, TS = TD + 1;
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
                TPP_SYNC();          // every wave has read the view-mean trunk (density head, view layer 0)
                store_tile_h<true>(y, act, vnt, vmt, L);
                TPP_SYNC();
            }
            if constexpr (g == 13) {
                TPP_SYNC();
                store_tile_h<true>(y, act, vnt, vmt, L);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    TPP_SYNC();
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
    TPP_MARK(5);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&g_tpp_trace[k], tr_[k]);
        atomicAdd(&g_tpp_trace[7], 1ull);
    }
#endif
}

}  // namespace

#if NEO_TP_TRACE
extern "C" void neo_debug_tpp_trace(unsigned long long* host16, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_tpp_trace), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tpp_trace), z, sizeof(z));
    }
}
#endif

void launch_tp_mlp_hpp(int input_ch, const TpMlpHDev& m, const float* proj_all, const long plane_base_texels[3],
                       const TpScene& sc, const TpViews& views, const float* rays_o, const float* rays_d, const float* viewdirs,
                       const float* tvals, const float* far, int R, int N, int chunk, uint32_t* flags, float* out,
                       const float* dirsum, hipStream_t s) {
    const long P = (long)R * N;
    if (P <= 0) return;
    const float* proj = proj_all;
    const TpPlaneBase pp{{(int)plane_base_texels[0], (int)plane_base_texels[1], (int)plane_base_texels[2]}};
    const size_t lds = (size_t)PP_LDS_WORDS * sizeof(float);
    // the attribute is per device AND per function: a process may hold contexts on several GPUs, so it is cached per device
    // (not in a process-wide flag); the device is current here (ENTER's DeviceGuard)
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_mlp_hpp<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_mlp_hpp<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const long tiles = tp::xcd_grid((P + TM - 1) / TM);
    if (input_ch == 3)
        hipLaunchKernelGGL(k_tp_mlp_hpp<3>, dim3((unsigned)tiles), dim3(256), lds, s, m, proj, pp, sc, views, rays_o, rays_d,
                           viewdirs, tvals, far, R, N, chunk, flags, reinterpret_cast<float4*>(out), dirsum);
    else
        hipLaunchKernelGGL(k_tp_mlp_hpp<4>, dim3((unsigned)tiles), dim3(256), lds, s, m, proj, pp, sc, views, rays_o, rays_d,
                           viewdirs, tvals, far, R, N, chunk, flags, reinterpret_cast<float4*>(out), dirsum);
}

}  // namespace neo

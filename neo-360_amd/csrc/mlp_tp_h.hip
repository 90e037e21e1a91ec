// NeO-360 decoder point evaluator on the fp16 matrix cores with hi/lo-split fp32 operands
// (fp32-equivalent arithmetic, see mlp_vanilla_h.hip for the numerics): the structure of
// mlp_tp.hip — per 64-point tile, looped over the source views: descriptors, the 703/724-wide
// input streamed 64 features at a time into a double-buffered LDS tile while the previous stage
// is multiplied, L0 || L3-skip as one 256-wide GEMM, L1, L2, L3, bottleneck, view layer 0, running
// view means in registers — with every LDS tile stored as two fp16 planes (hi, lo) and every
// product evaluated as a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_f16.
// The gathered features are blended in fp32 and split once when they are written to the stage tile.
// This translation unit is compiled without packed-fp32 VALU ops (build.py:EXTRA_FLAGS): with them, two
// co-resident workgroups produced run-to-run different values in lanes 48-63 of the blends
// (profiles/r01_tp_h_race_bisect.log); tests/test_gpu_repeatable.py guards it.
#include <hip/hip_fp16.h>

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

// ---- packed weight layout (h8 units; one (n_tile, k_step) = hi 64 lanes + lo 64 lanes) -------------
__host__ __device__ constexpr int pe_ksteps(int pe_c) { return pe_c == 3 ? 4 : 6; }
__host__ __device__ constexpr int ks_x(int pe_c) { return 32 + 8 + pe_ksteps(pe_c); }
__host__ __device__ constexpr int hoff_x() { return 0; }
__host__ __device__ constexpr int hoff_1(int pe_c) { return 8 * ks_x(pe_c) * 128; }
__host__ __device__ constexpr int hoff_2(int pe_c) { return hoff_1(pe_c) + 4 * 8 * 128; }
__host__ __device__ constexpr int hoff_3a(int pe_c) { return hoff_2(pe_c) + 4 * 8 * 128; }
__host__ __device__ constexpr int hoff_b(int pe_c) { return hoff_3a(pe_c) + 4 * 8 * 128; }
__host__ __device__ constexpr int hoff_v0(int pe_c) { return hoff_b(pe_c) + 4 * 8 * 128; }
__host__ __device__ constexpr int hoff_v1(int pe_c) { return hoff_v0(pe_c) + 2 * 10 * 128; }
__host__ __device__ constexpr int hpack_h8(int pe_c) { return hoff_v1(pe_c) + 2 * 4 * 128; }
constexpr int B_0 = 0, B_3 = 128, B_1 = 256, B_2 = 384, B_B = 512, B_V0 = 640, B_V1 = 704;
constexpr int HD_DW = 0, HD_DB = 128, HD_RW = 132, HD_RB = 324;

template <int PE_C>
__global__ __launch_bounds__(256, NEO_GATHER_WAVES_PER_SIMD) void k_tp_mlp_h(TpMlpHDev m, TpScene sc, TpViews views,
                                                      const float* __restrict__ rays_o,
                                                      const float* __restrict__ rays_d,
                                                      const float* __restrict__ viewdirs,
                                                      const float* __restrict__ tvals,
                                                      const float* __restrict__ far_arr, int R, int N, int chunk,
                                                      uint32_t* __restrict__ flags, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* hbase = reinterpret_cast<_Float16*>(smem + tp::OFF_ACT);
    const HT act{hbase, hbase + TM * 128};                                   // [64][128] x 2 planes (32 KB)
    auto xbuf = [&](int b) { return HT{hbase + b * (2 * TM * 64), hbase + b * (2 * TM * 64) + TM * 64}; };   // aliases act
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
    constexpr int NST = PE_C == 3 ? 11 : 12;   // streamed stages of 64 features: 8 local, 2 world, 1-2 pos_enc

    tp::point_setup<PE_C>(S, tid, tile0, P, N, R, chunk, rays_o, rays_d, viewdirs, tvals, far_arr, flags, false, sc.grid_w, sc.grid_first, sc.grid_pw, sc.grid_ph);
    __syncthreads();

    // View means by linearity.  Everything after relu(L3_v) is linear up to the view mean: the density head acts on
    // mean_v relu(L3_v); the bottleneck is linear, and so is view layer 0 on [bottleneck_v | dir_enc_v], whose view
    // mean is then V0 [mean_v bottleneck_v | mean_v dir_enc_v] + b.  So the view loop only accumulates
    // hsum = sum_v relu(L3_v) (two accumulator tiles per wave) and the 27 direction features (fp32 in LDS), and the
    // bottleneck / view-layer GEMMs run ONCE per tile instead of once per view (10 % of the MFMA work, two
    // epilogues and three barriers per view).
    f32x16 hsum[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = 0.f; hsum[1][r] = 0.f; }
    float* dens_w = smem + tp::OFF_DENSW;
    if (tid < 128) dens_w[tid] = m.heads[HD_DW + tid];
    float* dsum = smem + tp::OFF_DIR;          // [64][32] fp32 running sum of the direction encodings (same 8 KB as dsm)
    const int nts_x[2] = {L.wv, 4 + L.wv};
    const int nts_1[1] = {L.wv};
    const int vnt = L.wv & 1, vmt = L.wv >> 1;

#pragma unroll 1
    for (int v = 0; v < sc.nv; ++v) {
        // Re-derive the per-lane indices inside the loop from an opaque lane id: otherwise every swizzled LDS
        // address of the loop body is hoisted as a loop invariant and ~40 of them are spilled to scratch
        // (44 KB of scratch stores per tile x 1.2 M tiles per launch shows up as HBM write traffic).
        asm volatile("" : "+v"(tid));
        L.lane = tid & 63;
        L.half = L.lane >> 5;
        L.l31 = L.lane & 31;
        L.key = L.lane & 15;
        const float* rot = views.rot[v];
        const float* trn = views.trans[v];
        tp::view_descriptors(S, L, sc, rot, trn, v, [&](int p, int f, float val) {
            const int di = p * 32 + (f ^ (p & 31));     // lane = p: XOR keeps the 64 lanes on distinct banks
            dsum[di] = v == 0 ? val : dsum[di] + val;   // (p, f) is owned by one thread in every view
        });
        __syncthreads();

        // ---- streamed-input GEMM: [L0 | L3 skip half] (256 outputs) over 703 / 724 features ----
        f32x16 accx[2][2];
        bias_tile(accx[0][0], m.bias + B_0, L.wv, L);
        accx[0][1] = accx[0][0];
        bias_tile(accx[1][0], m.bias + B_3, L.wv, L);
        accx[1][1] = accx[1][0];
        {
            // Quarter-stage schedule: one row per thread and one k-step per pass (4 passes per 64-feature stage):
            // 16 tap registers + 16 weight registers (a half-stage schedule needs 32 + 32), so the 168-VGPR build
            // (three workgroups per CU) spills 14 registers instead of 42: 16.0 -> 15.7 ms fg, 14.3 -> 13.4 ms bg.
            // Per pass: gathers of one row of the NEXT stage | 12 MFMAs of this stage's k-step | next k-step's
            // weights | blend + split + store of the gathered row.
            const int col4 = tid & 15, rg = tid >> 4;
            const uint32_t lane_b = 16u * col4;
            f32x4 tap[4];
            auto issue_local = [&](int s, int q) __attribute__((always_inline)) {
                const int row = rg + 16 * q;
#pragma unroll
                for (int k = 0; k < 4; ++k) tap[k] = tp::load_tap(sc.latent, (uint32_t)loc_off[row * 4 + k] + lane_b + 256u * s);
            };
            auto issue_plane = [&](int j, int s2, int q) __attribute__((always_inline)) {
                const int row = rg + 16 * q;
#pragma unroll
                for (int k = 0; k < 4; ++k) tap[k] = tp::load_tap(sc.plane[j], (uint32_t)pl_off[(j * TM + row) * 4 + k] + lane_b + 256u * s2);
            };
            auto write_x = [&](const HT& buf, int row, const f32x4 v) __attribute__((always_inline)) {
                range_see4(L, v);
                h4 vh, vl;
                split4(v, vh, vl);
                const int o = chunk_off<64>(row, col4 >> 1) + 4 * (col4 & 1);
                *reinterpret_cast<h4*>(buf.hi + o) = vh;
                *reinterpret_cast<h4*>(buf.lo + o) = vl;
            };
            auto finish_local = [&](const HT& buf, int q) __attribute__((always_inline)) {
                const int row = rg + 16 * q;
                write_x(buf, row, blend4(tap, *reinterpret_cast<const f32x4*>(loc_w + row * 4)));
            };
            auto finish_planes = [&](const HT& buf, int s2, int q) __attribute__((always_inline)) {
                const int row = rg + 16 * q;
                f32x4 sum = blend4(tap, *reinterpret_cast<const f32x4*>(pl_w + row * 4));
#pragma unroll
                for (int j = 1; j < 3; ++j) {
                    issue_plane(j, s2, q);
                    sum = sum + blend4(tap, *reinterpret_cast<const f32x4*>(pl_w + (j * TM + row) * 4));
                }
                write_x(buf, row, sum);
            };
            // pos_enc: half hf of a stage = chunks 4hf..4hf+3, one per wave (run in the odd quarters)
            auto finish_pe = [&](const HT& buf, int pstage, int hf) __attribute__((always_inline)) {
                const int row = tid & 63, q = tid >> 6;
                const float xc[4] = {cam_enc[row * 4], cam_enc[row * 4 + 1], cam_enc[row * 4 + 2], cam_enc[row * 4 + 3]};
                range_see(L, xc[0]); range_see(L, xc[1]); range_see(L, xc[2]);     // identity features (the rest are sines)
                const int ch = hf * 4 + q;
                h8 vh, vl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 h, l;
                    split(pe_feature<PE_C>(xc, pstage * 64 + ch * 8 + e), h, l);
                    vh[e] = h;
                    vl[e] = l;
                }
                const int o = chunk_off<64>(row, ch);
                *reinterpret_cast<h8*>(buf.hi + o) = vh;
                *reinterpret_cast<h8*>(buf.lo + o) = vl;
            };
            h8 wh[2], wl[2];                                   // one k-step x 2 N-tiles, hi + lo
            const char* wxb = reinterpret_cast<const char*>(wp + hoff_x());
            uint32_t wx_off[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wx_off[nt] = (uint32_t)(nts_x[nt] * KSX * 2 * 64 + L.lane) * 16u;
            auto load_wq = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    wh[nt] = *reinterpret_cast<const h8*>(wxb + (wx_off[nt] + 2048u * ks));
                    wl[nt] = *reinterpret_cast<const h8*>(wxb + (wx_off[nt] + 2048u * ks + 1024u));
                }
            };
            auto mma_q = [&](const HT& tile, int tks) __attribute__((always_inline)) {
                h8 bh[2], bl[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int o = chunk_off<64>(mt * 32 + L.l31, (tks << 1) + L.half);
                    bh[mt] = *reinterpret_cast<const h8*>(tile.hi + o);
                    bl[mt] = *reinterpret_cast<const h8*>(tile.lo + o);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        accx[nt][mt] = NEO_MFMA_H(wl[nt], bh[mt], accx[nt][mt]);
                        accx[nt][mt] = NEO_MFMA_H(wh[nt], bl[mt], accx[nt][mt]);
                        accx[nt][mt] = NEO_MFMA_H(wh[nt], bh[mt], accx[nt][mt]);
                    }
            };
            constexpr int K_LOCAL = 0, K_PLANE = 1, K_PE = 2, K_NONE = 3;
            auto quarter = [&](int s, int q, auto kind_c) __attribute__((always_inline)) {
                constexpr int kind = decltype(kind_c)::value;
                const HT cur = xbuf(s & 1), nxt = xbuf((s + 1) & 1);
                const int sn = s + 1;
                if constexpr (kind == K_LOCAL) issue_local(sn, q);
                if constexpr (kind == K_PLANE) issue_plane(0, sn - 8, q);
                __builtin_amdgcn_sched_barrier(0);
                mma_q(cur, q);
                __builtin_amdgcn_sched_barrier(0);
                if (4 * s + q + 1 < KSX) load_wq(4 * s + q + 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (kind == K_LOCAL) finish_local(nxt, q);
                if constexpr (kind == K_PLANE) finish_planes(nxt, sn - 8, q);
                if constexpr (kind == K_PE) { if (q & 1) finish_pe(nxt, sn - 10, q >> 1); }
            };
            using std::integral_constant;
            // prologue: stage 0
            load_wq(0);
#pragma unroll 1
            for (int q = 0; q < 4; ++q) { issue_local(0, q); finish_local(xbuf(0), q); }
            __syncthreads();
#pragma unroll 1
            for (int s = 0; s < 7; ++s) {
#pragma unroll 1
                for (int q = 0; q < 4; ++q) quarter(s, q, integral_constant<int, K_LOCAL>());
                __syncthreads();
            }
#pragma unroll 1
            for (int s = 7; s < 9; ++s) {
#pragma unroll 1
                for (int q = 0; q < 4; ++q) quarter(s, q, integral_constant<int, K_PLANE>());
                __syncthreads();
            }
#pragma unroll 1
            for (int s = 9; s < NST - 1; ++s) {
#pragma unroll 1
                for (int q = 0; q < 4; ++q) quarter(s, q, integral_constant<int, K_PE>());
                __syncthreads();
            }
#pragma unroll 1
            for (int q = 0; q < (PE_C == 3 ? 4 : 2); ++q) quarter(NST - 1, q, integral_constant<int, K_NONE>());
            __syncthreads();
        }
        // ---- L0 epilogue, L1, L2 ----
        f32x16 acc[1][2];
        store_tile_h<true>(accx[0][0], act, L.wv, 0, L);
        store_tile_h<true>(accx[0][1], act, L.wv, 1, L);
        __syncthreads();
#pragma unroll 1
        for (int layer = 0; layer < 2; ++layer) {
            bias_tile(acc[0][0], m.bias + (layer == 0 ? B_1 : B_2), L.wv, L);
            acc[0][1] = acc[0][0];
            gemm2h<1, 128>(acc, wp + (layer == 0 ? hoff_1(PE_C) : hoff_2(PE_C)), 8, nts_1, 0, 0, 8, act, L);
            __syncthreads();
            store_tile_h<true>(acc[0][0], act, L.wv, 0, L);
            store_tile_h<true>(acc[0][1], act, L.wv, 1, L);
            __syncthreads();
        }
        // ---- L3 = skip half (in accx[1]) + W3[:, :128] h2; ReLU; accumulate over the views ----
        acc[0][0] = accx[1][0];
        acc[0][1] = accx[1][1];
        gemm2h<1, 128>(acc, wp + hoff_3a(PE_C), 8, nts_1, 0, 0, 8, act, L);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            hsum[0][r] += fmaxf(acc[0][0][r], 0.0f);
            hsum[1][r] += fmaxf(acc[0][1][r], 0.0f);
        }
        __syncthreads();           // every wave is done reading this view's tiles
    }

    // ---- view mean of the trunk -> density head ----
    const float inv_nv = 1.0f / (float)sc.nv;     // x * (1 / nv): 1 instruction instead of the 11 of an fp32 division, <= 1 ulp (see mlp_tp_hp.hip)
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = hsum[0][r] * inv_nv; hsum[1][r] = hsum[1][r] * inv_nv; }
    store_tile_h<false>(hsum[0], act, L.wv, 0, L);
    store_tile_h<false>(hsum[1], act, L.wv, 1, L);
    // view mean of the direction encoding: fp32 sums -> hi/lo planes in place (read all, barrier, write)
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
    float raw_sigma;
    {
        float sg = density_partial(act, dens_w, L);
        sg += __shfl_xor(sg, 1, 64);
        sg += __shfl_xor(sg, 2, 64);
        raw_sigma = sg + m.heads[HD_DB];
    }
    // ---- bottleneck of the view mean (no activation) ----
    {
        f32x16 acc[1][2];
        bias_tile(acc[0][0], m.bias + B_B, L.wv, L);
        acc[0][1] = acc[0][0];
        gemm2h<1, 128>(acc, wp + hoff_b(PE_C), 8, nts_1, 0, 0, 8, act, L);
        __syncthreads();
        store_tile_h<false>(acc[0][0], act, L.wv, 0, L);
        store_tile_h<false>(acc[0][1], act, L.wv, 1, L);
        __syncthreads();
    }
    // ---- view layer 0 on [mean bottleneck | mean dir enc] -> 64 ----
    f32x16 ysum;
    bias_tile(ysum, m.bias + B_V0, vnt, L);
    gemm1h<128>(ysum, wp + hoff_v0(PE_C), 10, vnt, vmt, 0, 8, act, L);
    gemm1h<32>(ysum, wp + hoff_v0(PE_C), 10, vnt, vmt, 8, 2, dsm, L);
    __syncthreads();
    // ---- ReLU -> 64x64 -> ReLU -> rgb head ----
    store_tile_h<true>(ysum, act, vnt, vmt, L);
    __syncthreads();
    {
        f32x16 y;
        bias_tile(y, m.bias + B_V1, vnt, L);
        gemm1h<128>(y, wp + hoff_v1(PE_C), 4, vnt, vmt, 0, 4, act, L);
        __syncthreads();
        store_tile_h<true>(y, act, vnt, vmt, L);
    }
    __syncthreads();
    {
        const int pt = L.wv * 16 + (L.lane >> 2), part = L.lane & 3;
        const float* wr = m.heads + HD_RW;
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
            out[gi] = make_float4(colour_act(r + m.heads[HD_RB]), colour_act(g + m.heads[HD_RB + 1]),
                                  colour_act(b + m.heads[HD_RB + 2]), density_act(raw_sigma));
        }
    }
}

}  // namespace

size_t tp_wpack_h_bytes(int input_ch) { return (size_t)hpack_h8(input_ch) * 16; }

void launch_tp_pack_h(int input_ch, const float* const* w, void* wpack_h, hipStream_t s) {
    // w order: pts_linears.0..3, views_linear.0, views_linear.1, bottleneck, density, rgb
    _Float16* base = reinterpret_cast<_Float16*>(wpack_h);
    const int pe = input_ch * 21;
    const int x0w = pe + 512 + 128;
    const int ksx = ks_x(input_ch);
    const PackSegs none = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    PackSegs sx = {{0, 512, 640}, {512, 128, pe}, {pe, pe + 512, 0}};   // packed [local | world | pe] <- [pe | local | world]
    pack_h(w[0], x0w, 128, ksx, 0, sx, base + (long)hoff_x() * 8, s);
    PackSegs sx3 = sx;
    for (int q = 0; q < 3; ++q) sx3.col[q] += 128;
    pack_h(w[3], 128 + x0w, 128, ksx, 4, sx3, base + (long)hoff_x() * 8, s);
    PackSegs p128 = none;
    p128.len[0] = 128;
    pack_h(w[1], 128, 128, 8, 0, p128, base + (long)hoff_1(input_ch) * 8, s);
    pack_h(w[2], 128, 128, 8, 0, p128, base + (long)hoff_2(input_ch) * 8, s);
    pack_h(w[3], 128 + x0w, 128, 8, 0, p128, base + (long)hoff_3a(input_ch) * 8, s);
    pack_h(w[6], 128, 128, 8, 0, p128, base + (long)hoff_b(input_ch) * 8, s);
    PackSegs v0 = none;
    v0.len[0] = 155;
    pack_h(w[4], 155, 64, 10, 0, v0, base + (long)hoff_v0(input_ch) * 8, s);
    PackSegs v1 = none;
    v1.len[0] = 64;
    pack_h(w[5], 64, 64, 4, 0, v1, base + (long)hoff_v1(input_ch) * 8, s);
}

void launch_tp_mlp_h(int input_ch, const TpMlpHDev& m, const TpScene& sc, const TpViews& views, const float* rays_o,
                     const float* rays_d, const float* viewdirs, const float* tvals, const float* far, int R, int N,
                     int chunk, uint32_t* flags, float* out, hipStream_t s) {
    const long P = (long)R * N;
    if (P <= 0) return;
    const size_t lds = tp::LDS_WORDS * sizeof(float);
    const long tiles = tp::xcd_grid((P + TM - 1) / TM);
    if (input_ch == 3)
        hipLaunchKernelGGL(k_tp_mlp_h<3>, dim3((unsigned)tiles), dim3(256), lds, s, m, sc, views, rays_o, rays_d,
                           viewdirs, tvals, far, R, N, chunk, flags, reinterpret_cast<float4*>(out));
    else
        hipLaunchKernelGGL(k_tp_mlp_h<4>, dim3((unsigned)tiles), dim3(256), lds, s, m, sc, views, rays_o, rays_d,
                           viewdirs, tvals, far, R, N, chunk, flags, reinterpret_cast<float4*>(out));
}

}  // namespace neo

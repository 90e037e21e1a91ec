// Fused NeO-360 decoder point evaluator (neo360/model.py:343-407 `predict`
// + NeRFPPMLP.forward :110-158 + the feature lookups of :409-447), fp32 MFMA.
//
// For a tile of 64 sample points and each of the NV source views in turn:
//   world2camera -> [pos_enc | pixel-aligned 512 | tri-plane 128] (703 / 724 features)
//   -> L0,L1,L2,L3(+skip) -> per-view bottleneck -> view layer 0,
// with the trunk (after L3) and the view branch averaged over the views in
// registers; then density head, 64->64, rgb head, activations.
//
// The 703-wide input is never materialised (the reference builds a rows x 703 fp32
// matrix: 1.1-3.3 GB per chunk).  It is produced 64 features at a time straight into
// a double-buffered LDS tile by the same waves that run the MFMAs:
//   * channels-last (NHWC) feature maps, one bilinear tap = one contiguous run;
//   * per (row, view) tap offsets / weights are computed once into LDS descriptors;
//   * the loads of stage s+1 are issued before the MFMAs of stage s and blended after
//     them (loads in flight under the matrix work), one barrier per 64-feature stage.
// L0 and the skip half of L3 share the operand, so they run as ONE 256-wide GEMM over
// the streamed input; the L3 half stays in accumulator registers until L3's turn.
//
// Work per point-view: 255,424 MAC (fg) / 260,800 MAC (bg); + 4,416 MAC per point.
//
// PROJ (round 4, VERDICT r3 task 9): the exact-fp32 evaluator on the PRE-PROJECTED maps of mlp_tp_hp.hip / mlp_tp_hpp.hip -
// the same algorithm in the reference's arithmetic.  PROJ = 1: the 512-channel latent is gathered as G = F . [W0_loc | W3_loc]^T
// (k_tp_preproject, exact fp32 MFMA) and ADDED to the [L0 | L3 skip] accumulators, the eight local stages of the streamed
// GEMM (131,072 MACs per point-view) are not executed; PROJ = 2: the three tri-planes likewise (32,768 more).
#include "tp_common.h"
#include "tp_hp_layout.h"

#ifndef NEO_TP32_RING
#define NEO_TP32_RING 4       // items (4 tap loads each) in flight per lane in the projected-map gather of k_tp_mlp
#endif
#ifndef NEO_TP32_TRACE
#define NEO_TP32_TRACE 0      // 1: per-phase s_memtime sums of wave 0 of every workgroup -> g_tp32_trace (tools/bench_tp_kernel.py TRACE=f32)
#endif
#if NEO_TP32_TRACE
__device__ unsigned long long g_tp32_trace[16];
#define TP32_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tr_[k] += now_ - tlast_; tlast_ = now_; } while (0)
#else
#define TP32_MARK(k) do { } while (0)
#endif

namespace neo {

namespace {

using tp::TM;
using tp::blend4;
using tp::pe_feature;
constexpr int ACT_LD = 128;   // activation tile [64][128]
constexpr int XB_LD = 64;     // streamed-input tile [64][64], two of them alias the activation tile
constexpr int DIR_LD = 32;

using tp::OFF_ACT;
using tp::OFF_DIR;
constexpr int LDS_FLOATS = tp::LDS_WORDS;

// ---- packed weight layout -----------------------------------------------------
// stage X : N=256 (rows 0-127 = pts_linears.0, rows 128-255 = pts_linears.3[:, 128:]),
//           K order = [local 512 | world 128 | pos_enc padded to 8*pe_chunks]
// stages 1,2 : pts_linears.1/.2 ; 3a : pts_linears.3[:, :128] ; B : bottleneck
// stage V0 : views_linear.0 (64 x (128 + 27->32)) ; V1 : views_linear.1 (64 x 64)
__host__ __device__ constexpr int pe_chunks(int pe_c) { return pe_c == 3 ? 8 : 11; }
__host__ __device__ constexpr int kc_x(int pe_c) { return 64 + 16 + pe_chunks(pe_c); }
__host__ __device__ constexpr int off_x() { return 0; }
__host__ __device__ constexpr int off_1(int pe_c) { return 8 * kc_x(pe_c) * 256; }
__host__ __device__ constexpr int off_2(int pe_c) { return off_1(pe_c) + 4 * 16 * 256; }
__host__ __device__ constexpr int off_3a(int pe_c) { return off_2(pe_c) + 4 * 16 * 256; }
__host__ __device__ constexpr int off_b(int pe_c) { return off_3a(pe_c) + 4 * 16 * 256; }
__host__ __device__ constexpr int off_v0(int pe_c) { return off_b(pe_c) + 4 * 16 * 256; }
__host__ __device__ constexpr int off_v1(int pe_c) { return off_v0(pe_c) + 2 * 20 * 256; }
// stage V0F (round 4): views_linear.0 with the bottleneck folded in, [W_v0[:, :128] W_b | W_v0[:, 128:]] (tp_hp_layout.h NEO_TP_FOLDB)
__host__ __device__ constexpr int off_v0f(int pe_c) { return off_v1(pe_c) + 2 * 8 * 256; }
__host__ __device__ constexpr int wpack_floats(int pe_c) { return off_v0f(pe_c) + 2 * 20 * 256; }
// biases: b0 | b3 | b1 | b2 | bb | bv0 | bv1 | bv0 folded
constexpr int B_0 = 0, B_3 = 128, B_1 = 256, B_2 = 384, B_B = 512, B_V0 = 640, B_V1 = 704, B_V0F = 768, BIAS_FLOATS = 832;
// heads: density w[128] | density b (4) | rgb w[3][64] | rgb b (4)
constexpr int HD_DW = 0, HD_DB = 128, HD_RW = 132, HD_RB = 324, HEADS_FLOATS = 328;

template <int NTW>
__device__ __forceinline__ void mma_chunk2(const f32x4 (&a)[NTW], const f32x4 (&b)[2], f32x16 (&acc)[NTW][2]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = NEO_MFMA(a[nt][e], b[mt][e], acc[nt][mt]);
}

// acc[nt][mt] += W-stage chunks [kc0, kc0+n) x tile chunks [0, n); N-tiles nts[], both M-tiles.
template <int NTW, int LD, int KM>
__device__ __forceinline__ void gemm2(f32x16 (&acc)[NTW][2], const f32x4* __restrict__ wp, int KC,
                                      const int (&nts)[NTW], int kc0, int n, const float* __restrict__ tile,
                                      const LaneCtx& L) {
    f32x4 a[2][NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) a[0][nt] = load_a(wp, KC, nts[nt], kc0, L.lane);
#pragma unroll 1
    for (int c = 0; c < n; c += 2) {
        f32x4 b[2];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) a[1][nt] = load_a(wp, KC, nts[nt], kc0 + (c + 1 < n ? c + 1 : c), L.lane);
        b[0] = load_b<LD, KM>(tile, 0, c, L);
        b[1] = load_b<LD, KM>(tile, 1, c, L);
        mma_chunk2<NTW>(a[0], b, acc);
        if (c + 1 < n) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) a[0][nt] = load_a(wp, KC, nts[nt], kc0 + (c + 2 < n ? c + 2 : c + 1), L.lane);
            b[0] = load_b<LD, KM>(tile, 0, c + 1, L);
            b[1] = load_b<LD, KM>(tile, 1, c + 1, L);
            mma_chunk2<NTW>(a[1], b, acc);
        }
    }
}

// same, reading tile chunks [tc0, tc0+n) against weight chunks [kc0+tc0, ...)
template <int NTW, int LD, int KM>
__device__ __forceinline__ void gemm2x(f32x16 (&acc)[NTW][2], const f32x4* __restrict__ wp, int KC,
                                       const int (&nts)[NTW], int kc0, int tc0, int n,
                                       const float* __restrict__ tile, const LaneCtx& L) {
    f32x4 a[2][NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) a[0][nt] = load_a(wp, KC, nts[nt], kc0 + tc0, L.lane);
#pragma unroll 1
    for (int c = 0; c < n; c += 2) {
        f32x4 b[2];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) a[1][nt] = load_a(wp, KC, nts[nt], kc0 + tc0 + (c + 1 < n ? c + 1 : c), L.lane);
        b[0] = load_b<LD, KM>(tile, 0, tc0 + c, L);
        b[1] = load_b<LD, KM>(tile, 1, tc0 + c, L);
        mma_chunk2<NTW>(a[0], b, acc);
        if (c + 1 < n) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) a[0][nt] = load_a(wp, KC, nts[nt], kc0 + tc0 + (c + 2 < n ? c + 2 : c + 1), L.lane);
            b[0] = load_b<LD, KM>(tile, 0, tc0 + c + 1, L);
            b[1] = load_b<LD, KM>(tile, 1, tc0 + c + 1, L);
            mma_chunk2<NTW>(a[1], b, acc);
        }
    }
}

// single accumulator tile (nt, mt): used by the 64-wide view layers
template <int LD, int KM>
__device__ __forceinline__ void gemm1(f32x16& acc, const f32x4* __restrict__ wp, int KC, int nt, int mt, int kc0,
                                      int n, const float* __restrict__ tile, const LaneCtx& L) {
    f32x4 a = load_a(wp, KC, nt, kc0, L.lane);
#pragma unroll 1
    for (int c = 0; c < n; ++c) {
        const f32x4 an = load_a(wp, KC, nt, kc0 + (c + 1 < n ? c + 1 : c), L.lane);
        const f32x4 b = load_b<LD, KM>(tile, mt, c, L);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = NEO_MFMA(a[e], b[e], acc);
        a = an;
    }
}

#ifndef NEO_TP_F32_WGS
#define NEO_TP_F32_WGS 2     // workgroups per CU the register budget is set for.  3 (<= 168 VGPRs, 43-48 spilled) measured: 180 k against
                             // 188 k rays/s (profiles/r05_exact_f32_experiments.log) - the kernel is phase-bound (MFMA busy 55 % at 2.38 GHz, 1.04 kW)
#endif
template <int PE_C, int PROJ>
__global__ __launch_bounds__(256, NEO_TP_F32_WGS) void k_tp_mlp(TpMlpDev m, const float* __restrict__ proj, TpPlaneProj pp, TpScene sc, TpViews views,
                                                    const float* __restrict__ rays_o,
                                                    const float* __restrict__ rays_d,
                                                    const float* __restrict__ viewdirs,
                                                    const float* __restrict__ tvals,
                                                    const float* __restrict__ far_arr, int R, int N, int chunk,
                                                    uint32_t* __restrict__ flags, float4* __restrict__ out,
                                                    const float* __restrict__ dirsum) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem + OFF_ACT;
    float* dsm = smem + OFF_DIR;
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
    const long tile0 = tp::xcd_tile(blockIdx.x, (P + TM - 1) / TM) * TM;      // contiguous tile range per XCD (tp_common.h)
    if (tile0 >= P) return;       // surplus workgroup of the rounded-up grid (uniform exit before any barrier)
    const f32x4* wp = reinterpret_cast<const f32x4*>(m.wpack);
    constexpr int KCX = kc_x(PE_C);
    constexpr int NST = PE_C == 3 ? 11 : 12;   // streamed-input stages: 8 local, 2 world, 1-2 pos_enc
    // FOLD (with the pre-projected maps, PROJ >= 1): everything after relu(L3_v) is linear up to the view mean, and the bottleneck
    // has no activation: per view only sum_v relu(L3_v) and sum_v dir_enc_v are accumulated, and view layer 0 with the bottleneck
    // folded in (stage V0F) runs ONCE per tile on the means - 26,304 of the 91,584 MACs per point-view leave the view loop.
    // PROJ = 0 keeps the reference's operation order throughout.
    constexpr bool FOLD = PROJ >= 1;

#if NEO_TP32_TRACE
    unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = __builtin_amdgcn_s_memtime();
#endif
    tp::point_setup<PE_C>(S, tid, tile0, P, N, R, chunk, rays_o, rays_d, viewdirs, tvals, far_arr, flags, false, sc.grid_w, sc.grid_first, sc.grid_pw, sc.grid_ph);
    __syncthreads();
    if constexpr (FOLD) {
        // the view branch needs only the SUM over the views of each point's direction encoding, and that depends on the ray alone:
        // it comes ready-made from the per-ray table of this launch (mlp_tp_hp.hip:k_tp_dirsum - the arithmetic and the summation
        // order of the per-view accumulation it replaces, so the outputs are bitwise the same), 8 features per thread
        const int p = tid >> 2, f0 = (tid & 3) << 3;
        const int dray = __float_as_int(S.vdir_world[p * 4 + 3]);      // the ray whose direction this row carries (quirk Q1)
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(dirsum + (long)dray * 32 + f0);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(dirsum + (long)dray * 32 + f0 + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) dsm[swz_index<DIR_LD, 7>(p, f0 + j)] = j < 4 ? s0[j] : s1[j - 4];
    }
    TP32_MARK(0);

    f32x16 hsum[2], ysum;
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = 0.f; hsum[1][r] = 0.f; ysum[r] = 0.f; }
    const int nts_x[2] = {L.wv, 4 + L.wv};
    const int nts_1[1] = {L.wv};
    const int vnt = L.wv & 1, vmt = L.wv >> 1;   // view layers: 2 N-tiles x 2 M-tiles, one per wave

#pragma unroll 1
    for (int v = 0; v < sc.nv; ++v) {
        // per-lane indices re-derived from an opaque lane id each view: keeps the swizzled LDS addresses of
        // the loop body from being hoisted and spilled (see mlp_tp_h.hip)
        asm volatile("" : "+v"(tid));
        L.lane = tid & 63;
        L.half = L.lane >> 5;
        L.l31 = L.lane & 31;
        L.key = L.lane & 15;
        const float* rot = views.rot[v];
        const float* trn = views.trans[v];
        // FOLD: no per-view direction encoding (the table above); otherwise every (point, feature) is written by one thread per view
        tp::view_descriptors<(PROJ >= 1 ? hp::PROJ_TEXEL_BYTES : 2048), !FOLD, (PROJ == 2 ? hp::PROJ_TEXEL_BYTES : 128 * 4)>(
            S, L, sc, rot, trn, v, [&](int p, int f, float val) { dsm[swz_index<DIR_LD, 7>(p, f)] = val; });
        __syncthreads();
        TP32_MARK(1);

        // ---- streamed-input GEMM: [L0 | L3 skip half] (256 outputs) over 703 / 724 features ----
        f32x16 accx[2][2];
        bias_tile(accx[0][0], m.bias + B_0, L.wv, L);
        accx[0][1] = accx[0][0];
        bias_tile(accx[1][0], m.bias + B_3, L.wv, L);
        accx[1][1] = accx[1][0];
        {
            const int col4 = tid & 15, rg = tid >> 4;
            const uint32_t lane_b = 16u * col4;      // this lane's 4 channels inside a 64-channel stage slice
            // Each 64-feature stage is produced in two half-tiles (32 rows each) so only 8 taps
            // (32 VGPR) are in flight: issue half A of stage s+1 | MFMA chunks 0-3 of stage s |
            // blend+write A | issue half B | MFMA chunks 4-7 | blend+write B | barrier.
            f32x4 tap[2][4];
            auto issue_local = [&](int s, int hf) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = rg + 16 * (2 * hf + i);
#pragma unroll
                    for (int k = 0; k < 4; ++k) tap[i][k] = tp::load_tap(sc.latent, (uint32_t)loc_off[row * 4 + k] + lane_b + 256u * s);
                }
            };
            auto issue_plane = [&](int j, int s2, int hf) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = rg + 16 * (2 * hf + i);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        tap[i][k] = tp::load_tap(sc.plane[j], (uint32_t)pl_off[(j * TM + row) * 4 + k] + lane_b + 256u * s2);
                }
            };
            auto write_x = [&](float* buf, int row, const f32x4 v) {
                *reinterpret_cast<f32x4*>(buf + row * XB_LD + ((col4 ^ (row & 15)) << 2)) = v;
            };
            auto finish_local = [&](float* buf, int hf) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = rg + 16 * (2 * hf + i);
                    write_x(buf, row, blend4(tap[i], *reinterpret_cast<const f32x4*>(loc_w + row * 4)));
                }
            };
            // tri-plane sum: (xz + xy) + yz (encoder_tp_fusion_conv.py:204-206); plane 0 was prefetched
            auto finish_planes = [&](float* buf, int s2, int hf) {
                f32x4 sum[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    sum[i] = blend4(tap[i], *reinterpret_cast<const f32x4*>(pl_w + (rg + 16 * (2 * hf + i)) * 4));
#pragma unroll
                for (int j = 1; j < 3; ++j) {
                    issue_plane(j, s2, hf);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        sum[i] = sum[i] + blend4(tap[i], *reinterpret_cast<const f32x4*>(pl_w + (j * TM + rg + 16 * (2 * hf + i)) * 4));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) write_x(buf, rg + 16 * (2 * hf + i), sum[i]);
            };
            // pos_enc of the camera-frame point: 64 features per stage; half hf = feature chunks 8hf..8hf+7
            [[maybe_unused]] auto finish_pe = [&](float* buf, int pstage, int hf) {
                const int row = tid & 63, q = tid >> 6;
                const float xc[4] = {cam_enc[row * 4], cam_enc[row * 4 + 1], cam_enc[row * 4 + 2], cam_enc[row * 4 + 3]};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int ch = hf * 8 + q * 2 + c;
                    f32x4 vv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) vv[e] = pe_feature<PE_C>(xc, pstage * 64 + ch * 4 + e);
                    *reinterpret_cast<f32x4*>(buf + row * XB_LD + ((ch ^ (row & 15)) << 2)) = vv;
                }
            };
            // PROJ == 2 (pos_enc is the only streamed input): ONE argument reduction per (sin, cos) pair (common.h:sincos_pair, as
            // the split kernels do; the sin half is bitwise sin_cw's, the phase-shifted half within 1.5e-7 of the reference's
            // fl32(a + fl32(pi/2))).  Pair p = 4 i + wave: coordinate p % C, octave p / C, features C + p and 11 C + p
            // (neo360/helper.py:121-125).  C = 3: the whole encoding in the prologue (part 0).  C = 4: the pairs whose two features
            // both lie in the first 64-feature stage (octaves 0..4) + the single sines of octaves 5..9 in the prologue (part 0), the
            // phase-shifted halves of octaves 5..9 (features 64..83, second stage's tile `b1`) between the k-steps of the first stage
            // (part 1) - all 40 pairs in the prologue measured 3.7 % SLOWER on the outside launches (profiles/r06_f32_pairs.log).
            [[maybe_unused]] auto encode_pairs = [&](float* b0, float* b1, int part) {
                const int row = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
                const float xc[4] = {cam_enc[row * 4], cam_enc[row * 4 + 1], cam_enc[row * 4 + 2], cam_enc[row * 4 + 3]};
                auto put = [&](int f, float val) {
                    float* b = f < 64 ? b0 : b1;
                    const int g = f & 63;
                    b[row * XB_LD + (((g >> 2) ^ (row & 15)) << 2) + (g & 3)] = val;
                };
                if constexpr (PE_C == 3) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int pr = 4 * i + q;
                        if (pr < 30) {
                            const int oct = pr / 3, ci = pr - 3 * oct;
                            float sn, cs;
                            sincos_pair(ldexpf(ci == 0 ? xc[0] : ci == 1 ? xc[1] : xc[2], oct), sn, cs);
                            put(3 + pr, sn);
                            put(33 + pr, cs);
                        }
                    }
                    if (q == 3) {               // the raw coordinates and the zero pad of the last k-chunk
                        put(0, xc[0]); put(1, xc[1]); put(2, xc[2]); put(63, 0.0f);
                    }
                } else {
                    const float xq = q == 0 ? xc[0] : q == 1 ? xc[1] : q == 2 ? xc[2] : xc[3];      // p % 4 = wave
                    if (part == 0) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            float sn, cs;
                            sincos_pair(ldexpf(xq, i), sn, cs);
                            put(4 + 4 * i + q, sn);
                            put(44 + 4 * i + q, cs);
                        }
#pragma unroll
                        for (int i = 5; i < 10; ++i) put(4 + 4 * i + q, sin_cw(ldexpf(xq, i)));
                        if (q == 3) { put(0, xc[0]); put(1, xc[1]); put(2, xc[2]); put(3, xc[3]); }
                    } else {
#pragma unroll
                        for (int i = 5; i < 10; ++i) put(44 + 4 * i + q, sin_cw(ldexpf(xq, i) + HALF_PI_F32));
                        if (q == 3) { put(84, 0.0f); put(85, 0.0f); put(86, 0.0f); put(87, 0.0f); }
                    }
                }
            };
            // ---- pre-projected maps: 4 chunks of 64 output channels x 4 row groups; the maps' blends of a (chunk, row group)
            //      are summed in registers (latent, then xz, xy, yz), go through an fp32 transposition tile (gather layout ->
            //      MFMA D layout, hp::proj_index) and are added to the accumulators: the structure of mlp_tp_hpp.hip ----
            if constexpr (PROJ >= 1) {
                constexpr int NM = PROJ == 2 ? 4 : 1;
                // One ITEM = the four taps of one (chunk c, row group q, map mp): 4 loads, one blend.  Items run through a ring of
                // NEO_TP32_RING slots: item i + RING is requested as soon as item i's registers are free, so RING - 1 .. RING items
                // (12-16 loads per lane) stay in flight THROUGH the row / chunk boundaries.  The first version issued the 4 NM loads of
                // a row, drained them to zero while blending, and only then issued the next row's: 16 exposed round trips per view
                // (profiles/r05_exact_f32_experiments.log: this phase was 39 % of a tile, 58 k cycles per view; now 36 k).
                constexpr int RING = NEO_TP32_RING, ITEMS = 16 * NM;
                f32x4 ring[RING][4];
                f32x4 sum = {0.0f, 0.0f, 0.0f, 0.0f};
                auto item_addr = [&](int it, int& c, int& q, int& mp) { mp = it % NM; q = (it / NM) & 3; c = it / (4 * NM); };
                auto issue = [&](auto itc) {
                    constexpr int it = decltype(itc)::value;
                    int c, q, mp;
                    item_addr(it, c, q, mp);
                    const int row = rg + 16 * q;
                    const float* base = mp == 0 ? proj : pp.p[mp == 0 ? 0 : mp - 1];
                    const int di = mp == 0 ? row * 4 : ((mp - 1) * TM + row) * 4;
                    const int4 off = *reinterpret_cast<const int4*>((mp == 0 ? loc_off : pl_off) + di);
                    ring[it % RING][0] = tp::load_tap(base, (uint32_t)off.x + lane_b + 256u * c);
                    ring[it % RING][1] = tp::load_tap(base, (uint32_t)off.y + lane_b + 256u * c);
                    ring[it % RING][2] = tp::load_tap(base, (uint32_t)off.z + lane_b + 256u * c);
                    ring[it % RING][3] = tp::load_tap(base, (uint32_t)off.w + lane_b + 256u * c);
                };
                hp::static_for<0, (RING < ITEMS ? RING : ITEMS)>([&](auto itc) { issue(itc); });
                hp::static_for<0, ITEMS>([&](auto itc) {
                    constexpr int it = decltype(itc)::value;
                    int c, q, mp;
                    item_addr(it, c, q, mp);
                    const int row = rg + 16 * q;
                    const int di = mp == 0 ? row * 4 : ((mp - 1) * TM + row) * 4;
                    const f32x4 val = blend4(ring[it % RING], *reinterpret_cast<const f32x4*>((mp == 0 ? loc_w : pl_w) + di));
                    if (mp == 0) sum = val; else sum = sum + val;
                    if constexpr (it + RING < ITEMS) issue(std::integral_constant<int, it + RING>());
                    float* fb = act + (c & 1) * (TM * XB_LD);
                    if (mp == NM - 1) write_x(fb, row, sum);
                    if (mp == NM - 1 && q == 3) {
                        __syncthreads();
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int gg = 0; gg < 2; ++gg) {
                                const int trow = mt * 32 + L.l31;
                                const int piece = L.wv * 4 + gg * 2 + L.half;
                                const f32x4 tv = *reinterpret_cast<const f32x4*>(fb + trow * XB_LD + ((piece ^ (trow & 15)) << 2));
                                const int g0 = 2 * (c & 1);
#pragma unroll
                                for (int e = 0; e < 4; ++e) accx[c >> 1][mt][4 * (g0 + gg) + e] += tv[e];
                            }
                    }
                });
                __syncthreads();      // the transposition tiles are the streamed-input tiles: chunk 3 has been consumed
                TP32_MARK(2);
            }
            constexpr int S0 = PROJ == 0 ? 0 : PROJ == 1 ? 8 : 10;       // first streamed stage: 0 local, 8 world, 10 pos_enc
            // prologue: stage S0
            if constexpr (PROJ == 0) {
                issue_local(0, 0);
                finish_local(act, 0);
                issue_local(0, 1);
                finish_local(act, 1);
            } else if constexpr (PROJ == 1) {
                issue_plane(0, 0, 0);
                finish_planes(act, 0, 0);
                issue_plane(0, 0, 1);
                finish_planes(act, 0, 1);
            } else {
                encode_pairs(act, act + TM * XB_LD, 0);
            }
            __syncthreads();
#pragma unroll 1
            for (int s = S0; s < NST; ++s) {
                float* cur = act + (s & 1) * (TM * XB_LD);
                float* nxt = act + ((s + 1) & 1) * (TM * XB_LD);
                const int sn = s + 1;
                const int nchunks = (PE_C == 4 && s == NST - 1) ? 3 : 8;
                const int first = nchunks < 4 ? nchunks : 4;
#pragma unroll 1
                for (int hf = 0; hf < 2; ++hf) {
                    if (sn < 8) issue_local(sn, hf);
                    else if (sn < 10) issue_plane(0, sn - 8, hf);
                    if (hf == 0) gemm2<2, XB_LD, 15>(accx, wp + off_x() / 4, KCX, nts_x, s * 8, first, cur, L);
                    else if (nchunks > 4) gemm2x<2, XB_LD, 15>(accx, wp + off_x() / 4, KCX, nts_x, s * 8, 4, nchunks - 4, cur, L);
                    if (sn < 8) finish_local(nxt, hf);
                    else if (sn < 10) finish_planes(nxt, sn - 8, hf);
                    else if (sn < NST) {
                        if constexpr (PROJ == 2) { if (hf == 0) encode_pairs(act, act + TM * XB_LD, 1); }
                        else finish_pe(nxt, sn - 10, hf);
                    }
                }
                __syncthreads();
            }
        }

        TP32_MARK(3);
        // ---- L0 epilogue, L1, L2 ------------------------------------------------
        f32x16 acc[1][2];
        store_tile<ACT_LD, 15, true>(accx[0][0], act, L.wv, 0, L);
        store_tile<ACT_LD, 15, true>(accx[0][1], act, L.wv, 1, L);
        __syncthreads();
#pragma unroll 1
        for (int layer = 0; layer < 2; ++layer) {
            bias_tile(acc[0][0], m.bias + (layer == 0 ? B_1 : B_2), L.wv, L);
            acc[0][1] = acc[0][0];
            gemm2<1, ACT_LD, 15>(acc, wp + (layer == 0 ? off_1(PE_C) : off_2(PE_C)) / 4, 16, nts_1, 0, 16, act, L);
            __syncthreads();
            store_tile<ACT_LD, 15, true>(acc[0][0], act, L.wv, 0, L);
            store_tile<ACT_LD, 15, true>(acc[0][1], act, L.wv, 1, L);
            __syncthreads();
        }
        TP32_MARK(4);
        // ---- L3 = skip half (already in accx[1]) + W3[:, :128] h2 ; ReLU; accumulate the view mean ----
        acc[0][0] = accx[1][0];
        acc[0][1] = accx[1][1];
        gemm2<1, ACT_LD, 15>(acc, wp + off_3a(PE_C) / 4, 16, nts_1, 0, 16, act, L);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            hsum[0][r] += fmaxf(acc[0][0][r], 0.0f);
            hsum[1][r] += fmaxf(acc[0][1][r], 0.0f);
        }
        TP32_MARK(5);
        if constexpr (!FOLD) {
        store_tile<ACT_LD, 15, true>(acc[0][0], act, L.wv, 0, L);
        store_tile<ACT_LD, 15, true>(acc[0][1], act, L.wv, 1, L);
        __syncthreads();
        // ---- per-view bottleneck (no activation) ----
        bias_tile(acc[0][0], m.bias + B_B, L.wv, L);
        acc[0][1] = acc[0][0];
        gemm2<1, ACT_LD, 15>(acc, wp + off_b(PE_C) / 4, 16, nts_1, 0, 16, act, L);
        __syncthreads();
        store_tile<ACT_LD, 15, false>(acc[0][0], act, L.wv, 0, L);
        store_tile<ACT_LD, 15, false>(acc[0][1], act, L.wv, 1, L);
        __syncthreads();
        // ---- view layer 0: [bottleneck | dir enc] -> 64, summed over views before the ReLU ----
        {
            f32x16 y;
            bias_tile(y, m.bias + B_V0, vnt, L);
            gemm1<ACT_LD, 15>(y, wp + off_v0(PE_C) / 4, 20, vnt, vmt, 0, 16, act, L);
            gemm1<DIR_LD, 7>(y, wp + off_v0(PE_C) / 4, 20, vnt, vmt, 16, 4, dsm, L);
#pragma unroll
            for (int r = 0; r < 16; ++r) ysum[r] += y[r];
        }
        __syncthreads();   // act / dsm / descriptors are rewritten by the next view
        }                  // !FOLD
    }

    // ---- view mean of the trunk -> density head ----
    const float nvf = (float)sc.nv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = hsum[0][r] / nvf; hsum[1][r] = hsum[1][r] / nvf; }
    store_tile<ACT_LD, 15, false>(hsum[0], act, L.wv, 0, L);
    store_tile<ACT_LD, 15, false>(hsum[1], act, L.wv, 1, L);
    __syncthreads();
    float raw_sigma;
    {
        const int pt = L.wv * 16 + (L.lane >> 2), part = L.lane & 3;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int chunk_i = part * 8 + ((c + 2 * part) & 7);
            const f32x4 h = *reinterpret_cast<const f32x4*>(act + pt * ACT_LD + ((chunk_i ^ (pt & 15)) << 2));
            const f32x4 w = *reinterpret_cast<const f32x4*>(m.heads + HD_DW + chunk_i * 4);
            s += h[0] * w[0] + h[1] * w[1] + h[2] * w[2] + h[3] * w[3];
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        raw_sigma = s + m.heads[HD_DB];
    }
    if constexpr (FOLD) {
        // view layer 0 with the bottleneck folded in, once per tile, on [mean trunk (act) | mean direction encoding (dsm)]
#pragma unroll
        for (int j = 0; j < 8; ++j) dsm[tid * 8 + j] = dsm[tid * 8 + j] / nvf;       // the sums -> means, in place (scaling is layout-blind)
        __syncthreads();
        bias_tile(ysum, m.bias + B_V0F, vnt, L);
        gemm1<ACT_LD, 15>(ysum, wp + off_v0f(PE_C) / 4, 20, vnt, vmt, 0, 16, act, L);
        gemm1<DIR_LD, 7>(ysum, wp + off_v0f(PE_C) / 4, 20, vnt, vmt, 16, 4, dsm, L);
    }
    __syncthreads();
    // ---- view mean of the view branch -> ReLU -> 64x64 -> ReLU -> rgb head ----
    if constexpr (!FOLD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ysum[r] = ysum[r] / nvf;
    }
    store_tile<ACT_LD, 15, true>(ysum, act, vnt, vmt, L);
    __syncthreads();
    {
        f32x16 y;
        bias_tile(y, m.bias + B_V1, vnt, L);
        gemm1<ACT_LD, 15>(y, wp + off_v1(PE_C) / 4, 8, vnt, vmt, 0, 8, act, L);
        __syncthreads();
        store_tile<ACT_LD, 15, true>(y, act, vnt, vmt, L);
    }
    __syncthreads();
    {
        const int pt = L.wv * 16 + (L.lane >> 2), part = L.lane & 3;
        const float* wr = m.heads + HD_RW;
        float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int chunk_i = part * 4 + ((c + part) & 3);
            const f32x4 h = *reinterpret_cast<const f32x4*>(act + pt * ACT_LD + ((chunk_i ^ (pt & 15)) << 2));
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + chunk_i * 4);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + 64 + chunk_i * 4);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 128 + chunk_i * 4);
            r += h[0] * w0[0] + h[1] * w0[1] + h[2] * w0[2] + h[3] * w0[3];
            g += h[0] * w1[0] + h[1] * w1[1] + h[2] * w1[2] + h[3] * w1[3];
            b += h[0] * w2[0] + h[1] * w2[1] + h[2] * w2[2] + h[3] * w2[3];
        }
        r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64);
        g += __shfl_xor(g, 1, 64); g += __shfl_xor(g, 2, 64);
        b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64);
        const long gv = tile0 + pt;
        const long gi = tp::patch_point(gv, N, R, sc.grid_w, sc.grid_first, sc.grid_pw, sc.grid_ph);
        if (part == 0 && gv < P) {
            out[gi] = make_float4(colour_act(r + m.heads[HD_RB]), colour_act(g + m.heads[HD_RB + 1]),
                                  colour_act(b + m.heads[HD_RB + 2]), density_act(raw_sigma));
        }
    }
#if NEO_TP32_TRACE
    TP32_MARK(6);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) atomicAdd(&g_tp32_trace[k], tr_[k]);
        atomicAdd(&g_tp32_trace[7], 1ull);
    }
#endif
}

// ---- weight packing -------------------------------------------------------------
// dst N-tiles [nt0, nt0 + rows/32) of a stage with KC chunks <- src[rows][ld], packed k -> source column
// through up to three segments (k_start, k_len, col_start); everything else is zero.

__global__ void k_pack_block(const float* __restrict__ src, int ld, int rows, int KC, int nt0, PackSegs sg,
                             float* __restrict__ dst) {
    const int total = (rows / 32) * KC * 256;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 3, lane = (idx >> 2) & 63, blk = idx >> 8;
        const int kc = blk % KC, ntl = blk / KC;
        const int n = ntl * 32 + (lane & 31);
        const int k = kc * 8 + 4 * (lane >> 5) + e;
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (k >= sg.k0[q] && k < sg.k0[q] + sg.len[q]) v = src[(long)n * ld + sg.col[q] + (k - sg.k0[q])];
        dst[((long)(nt0 + ntl) * KC + kc) * 256 + (lane << 2) + e] = v;
    }
}

__global__ void k_copy_f(const float* __restrict__ src, int n, float* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// NCHW (NV, C, H, W) -> NHWC (NV, H, W, C)
__global__ void k_to_channels_last(const float* __restrict__ src, int C, int HW, long total, float* __restrict__ dst) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const long rest = idx / C;
        const int hw = (int)(rest % HW);
        const long v = rest / HW;
        dst[idx] = src[(v * C + c) * HW + hw];
    }
}

}  // namespace

#if NEO_TP32_TRACE
extern "C" void neo_debug_tp32_trace(unsigned long long* host16, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_tp32_trace), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tp32_trace), z, sizeof(z));
    }
}
#endif

void pack_block(const float* src, int ld, int rows, int KC, int nt0, PackSegs sg, float* dst, hipStream_t s) {
    const int total = (rows / 32) * KC * 256;
    hipLaunchKernelGGL(k_pack_block, dim3((total + 255) / 256), dim3(256), 0, s, src, ld, rows, KC, nt0, sg, dst);
}

void copy_floats(const float* src, int n, float* dst, hipStream_t s) {
    hipLaunchKernelGGL(k_copy_f, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), dim3(256), 0, s, src, n, dst);
}

size_t tp_wpack_floats(int input_ch) { return wpack_floats(input_ch); }
int tp_kc_x(int input_ch) { return kc_x(input_ch); }
size_t tp_bias_floats() { return BIAS_FLOATS; }
size_t tp_heads_floats() { return HEADS_FLOATS; }

void launch_tp_pack(int input_ch, const float* const* w, const float* const* b, float* wpack, float* bias,
                    float* heads, hipStream_t s, float* fold_ws) {
    // w/b order: pts_linears.0..3, views_linear.0, views_linear.1, bottleneck, density, rgb
    const int pe = input_ch * 21;                 // 63 or 84
    const int x0w = pe + 512 + 128;               // 703 or 724
    const int kcx = kc_x(input_ch);
    const PackSegs none = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    // stage X: packed k = [local 512 | world 128 | pe]; source x0 columns = [pe | local | world]
    PackSegs sx = {{0, 512, 640}, {512, 128, pe}, {pe, pe + 512, 0}};
    pack_block(w[0], x0w, 128, kcx, 0, sx, wpack + off_x(), s);
    PackSegs sx3 = sx;
    for (int q = 0; q < 3; ++q) sx3.col[q] += 128;   // L3 input = [h(128) | x0]
    pack_block(w[3], 128 + x0w, 128, kcx, 4, sx3, wpack + off_x(), s);
    PackSegs plain128 = none;
    plain128.len[0] = 128;
    pack_block(w[1], 128, 128, 16, 0, plain128, wpack + off_1(input_ch), s);
    pack_block(w[2], 128, 128, 16, 0, plain128, wpack + off_2(input_ch), s);
    pack_block(w[3], 128 + x0w, 128, 16, 0, plain128, wpack + off_3a(input_ch), s);
    pack_block(w[6], 128, 128, 16, 0, plain128, wpack + off_b(input_ch), s);
    PackSegs v0 = none;
    v0.len[0] = 155;                              // [bottleneck 128 | dir enc 27], zero padded to 160
    pack_block(w[4], 155, 64, 20, 0, v0, wpack + off_v0(input_ch), s);
    PackSegs v1 = none;
    v1.len[0] = 64;
    pack_block(w[5], 64, 64, 8, 0, v1, wpack + off_v1(input_ch), s);
    auto cp = [&](const float* src, int n, float* dst) {
        hipLaunchKernelGGL(k_copy_f, dim3(1), dim3(256), 0, s, src, n, dst);
    };
    cp(b[0], 128, bias + B_0); cp(b[3], 128, bias + B_3); cp(b[1], 128, bias + B_1); cp(b[2], 128, bias + B_2);
    cp(b[6], 128, bias + B_B); cp(b[4], 64, bias + B_V0); cp(b[5], 64, bias + B_V1);
    // stage V0F: view layer 0 with the bottleneck folded in (fp64 accumulation, pack_h.hip), for the FOLD form of the kernel
    launch_fold_bottleneck(w[4], w[6], b[6], b[4], 64, 128, 128, 27, fold_ws, bias + B_V0F, s);
    pack_block(fold_ws, 155, 64, 20, 0, v0, wpack + off_v0f(input_ch), s);
    (void)hipMemsetAsync(heads, 0, HEADS_FLOATS * sizeof(float), s);
    cp(w[7], 128, heads + HD_DW); cp(b[7], 1, heads + HD_DB); cp(w[8], 192, heads + HD_RW); cp(b[8], 3, heads + HD_RB);
}

void launch_channels_last(const float* src, int NV, int C, int H, int W, float* dst, hipStream_t s) {
    const long total = (long)NV * C * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_to_channels_last, dim3(blocks), dim3(256), 0, s, src, C, H * W, total, dst);
}

void launch_tp_mlp(int input_ch, const TpMlpDev& m, const TpScene& sc, const TpViews& views, const float* rays_o,
                   const float* rays_d, const float* viewdirs, const float* tvals, const float* far, int R, int N,
                   int chunk, uint32_t* flags, float* out, hipStream_t s, const float* proj, const TpPlaneProj* pp,
                   const float* dirsum) {
    const long P = (long)R * N;
    if (P <= 0) return;
    const size_t lds = LDS_FLOATS * sizeof(float);
    const dim3 grid((unsigned)tp::xcd_grid((P + TM - 1) / TM));
    const TpPlaneProj planes = pp ? *pp : TpPlaneProj{{nullptr, nullptr, nullptr}};
    float4* o4 = reinterpret_cast<float4*>(out);
    const int mode = !proj ? 0 : pp ? 2 : 1;      // gather the latent itself | the projected latent | projected latent + planes
#define NEO_TP_F32_LAUNCH(C, PR)                                                                                          \
    hipLaunchKernelGGL((k_tp_mlp<C, PR>), grid, dim3(256), lds, s, m, proj, planes, sc, views, rays_o, rays_d, viewdirs, \
                       tvals, far, R, N, chunk, flags, o4, dirsum)
    if (input_ch == 3) {
        if (mode == 0) NEO_TP_F32_LAUNCH(3, 0); else if (mode == 1) NEO_TP_F32_LAUNCH(3, 1); else NEO_TP_F32_LAUNCH(3, 2);
    } else {
        if (mode == 0) NEO_TP_F32_LAUNCH(4, 0); else if (mode == 1) NEO_TP_F32_LAUNCH(4, 1); else NEO_TP_F32_LAUNCH(4, 2);
    }
#undef NEO_TP_F32_LAUNCH
}

}  // namespace neo

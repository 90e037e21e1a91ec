// Training-side NeRFPPMLP (the reference's training step neo360/model.py:697-820 calls the MLP of
// :110-158 under autograd): forward that keeps the activations, and the backward - gradients of all nine weight matrices
// and biases and of the input rows - on MATERIALISED rows, the formulation the reference trains with
// (x0 = [pos_enc | local 512 | world 128] per point-view, view-major rows).
//
// Everything is dense fp32 GEMM work on the matrix cores in EXACT fp32 (v_mfma_f32_32x32x2_f32): gradients span many
// orders of magnitude, which the fp16 hi/lo split of the inference kernels would have to rescale tile by tile; the
// training step is not the path the headline metric measures, so it takes the simple exact arithmetic.
//
//   k_sgemm<AT, BT, TN>  C[M][N] (+)= op(A)[M][K] . op(B)[N][K]^T, 128 x TN tiles (TN = 64 / 128), 4 waves of 64 x TN / 2, K stepped by 32 through
//                     two K-major LDS tiles; operands in any of the layouts the chain needs:
//                       AT = false: A stored [M][K] (K fastest)      AT = true: A stored [K][M]   (reduction index slowest)
//                       BT = false: B stored [N][K]                  BT = true: B stored [K][N]
//                     forward  Z = X W^T   : (false, false)   B = W (out, in)
//                     dX = dZ W            : (false, true)    B = W (out, in) read as [K = out][N = in]
//                     dW = dZ^T X          : (true,  true)    A = dZ [K = rows][M = out], B = X [K = rows][N = in], split over K
//                     fused epilogue: + bias[n], ReLU, x (mask[m][n] > 0), C += (beta = 1), atomic accumulation (split-K)
//   k_view_mean / k_view_bcast / k_colsum / k_relu_mask: the few elementwise / reduction pieces in between.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "mfma_tile.h"
#include "train_kernels.h"

namespace neo {

namespace {

#ifndef NEO_SGEMM_ABLATE
#define NEO_SGEMM_ABLATE 0     // timing experiments only (wrong results; tools/build_variant.py): 1 no MFMAs, 2 no operand loads in the pipelined loop
#endif
constexpr int GTM = 128;      // C tile: rows of op(A)
constexpr int GTN = 64;       //         rows of op(B): 64, or 128 when N > 64 (template parameter TN; a 128-wide layer is one tile:
                              //         its A rows are read once - at 64 the launch asked the fabric for 8 B per clock and CU)
constexpr int GK = 32;        // K step
constexpr int GP = GK + 4;    // pitch (floats) of a K-major LDS tile [rows][k]: 16-B fragment reads of 16 consecutive rows hit 16 distinct bank groups
constexpr int PRA = GTM + 4;  // pitch of a row-major LDS tile [k][rows] (operands stored with the row index fastest): 4 * pitch = 16 mod 32,
                              //   so the two lane halves of a fragment read (k and k + 4) fall into different bank halves
constexpr int AS_FLOATS = GTM * GP > GK * PRA ? GTM * GP : GK * PRA;

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // 16-B global access at 4-B alignment (rows of 703 floats)

#ifndef NEO_GEMM_XCD
#define NEO_GEMM_XCD 1        // 0: plain blockIdx -> tile map (A/B)
#endif
// XCD-aware workgroup -> tile map (round 6).  Workgroups are dealt to the 8 XCDs round-robin in linear-id order (x fastest), and each
// XCD has its own L2: with the plain map the `inner` tiles that share an operand block - the column tiles of one 128-row block of A in
// k_sgemm, the output tiles of one K slice in k_dw - land on `inner` different XCDs and every one of them pulls that block through the
// fabric again (measured: the texel-space projection moved 8-9 GB per GEMM for 1.4 GB of operands + result).  Here the workgroups with
// linear id = 8 s + x (s = slot, x = XCD) take outer block (s / inner) * 8 + x and inner tile s % inner: the `inner` sharers run on one
// XCD back to back.  Outer blocks beyond the last complete group of 8 keep the plain map (a bijection either way).
__device__ __forceinline__ void xcd_pair(long lin, int inner, long outer_n, long& outer, int& in_tile) {
    const long full = (outer_n / 8) * 8;
    outer = lin / inner;
    in_tile = (int)(lin - outer * inner);
    if (NEO_GEMM_XCD && lin < full * inner) {
        const long slot = lin >> 3;
        outer = (slot / inner) * 8 + (lin & 7);
        in_tile = (int)(slot % inner);
    }
}

struct GemmEpi {
    const float* bias;        // [N] or null
    const float* mask;        // [M][ldm]: output multiplied by (mask > 0) - ReLU backward - or null
    int ldm;
    int relu;                 // max(x, 0), applied AFTER the bias and after the accumulation into C
    int accumulate;           // 0: C = result   1: C = C + result   2: atomicAdd (split-K partials; C zeroed by the caller)
    float scale;              // result multiplied by this first (1 / NV of the view means)
};

// Up to four A operands whose K ranges are concatenated ([h2 | x_enc | local | world] of the NeRFPPMLP skip layer without
// materialising the concatenation): n = 0 -> the plain (A, lda, K) of the launch; else segment i covers the next k[i]
// columns of the reduction index, B advancing along its K axis.  Split-K is not combined with segments.
struct GemmSegs {
    const float* a[4];
    long lda[4];
    int k[4];
    int n;
};

// C[M][N] (+)= op(A) . op(B)^T on 128 x 64 tiles: 4 waves of 64 (m) x 32 (n) = two accumulators sharing the n fragment.
// Operands travel global -> registers (16-B loads along the stored-fast index, guarded element-wise at the edges) -> LDS
// (double-buffered: the next K-step is staged into the other buffer while this one is multiplied, one barrier per step).
template <bool AT, bool BT, int TN>
__global__ __launch_bounds__(256, 2) void k_sgemm(int M, int N, int K, const float* __restrict__ A_in, long lda_in,
                                                  const float* __restrict__ B_in, long ldb, float* __restrict__ C, long ldc,
                                                  GemmEpi ep, int k_per_split, GemmSegs sg) {
    constexpr int PRB = TN + 4;             // pitch of a row-major B tile [k][rows]
    constexpr int NF = TN / 64, JB = TN / 32;   // n fragments per wave; 16-B pieces of the B tile per thread and K step
    constexpr int BS_FLOATS = TN * GP > GK * PRB ? TN * GP : GK * PRB;
    __shared__ __attribute__((aligned(16))) float As[2][AS_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[2][BS_FLOATS];
    LaneCtx L;
    L.init();
    const int tid = threadIdx.x;
    long row_block;
    int col_tile;
    xcd_pair((long)blockIdx.x + (long)gridDim.x * blockIdx.y, (int)gridDim.x, (long)gridDim.y, row_block, col_tile);
    const int m0 = (int)row_block * GTM, n0 = col_tile * TN;
    // the operand range being accumulated (one per segment)
    const float* A = A_in;
    const float* B = B_in;
    long lda = lda_in;
    int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
    const int wm = L.wv & 1, wn = L.wv >> 1;            // this wave: rows m0 + 64 wm .. + 63, columns n0 + (TN / 2) wn .. + TN / 2 - 1
    f32x16 acc[2][NF];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nt = 0; nt < NF; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nt][r] = 0.0f;
    f32x4 ra[4], rb[JB];
    // one 16-B piece of an operand tile: T = stored with the row index fastest ([K][rows]), else K fastest ([rows][K])
    auto piece = [&](const float* __restrict__ P, long ld, bool T, int rows0, int rows_end, int k0, int idx4, int rows_tile) -> f32x4 {
        int r, k, dr, dk;
        if (T) { const int q = rows_tile / 4; r = (idx4 % q) * 4; k = idx4 / q; dr = 1; dk = 0; }
        else { r = idx4 >> 3; k = (idx4 & 7) * 4; dr = 0; dk = 1; }
        const int gr = rows0 + r, gk = k0 + k;
        const float* src = T ? P + (long)gk * ld + gr : P + (long)gr * ld + gk;
        if (gr + 3 * dr < rows_end && gk + 3 * dk < kend) return *reinterpret_cast<const f4u*>(src);
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (gr + e * dr < rows_end && gk + e * dk < kend) v[e] = src[e];
        return v;
    };
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = piece(A, lda, AT, m0, M, k0, tid + 256 * j, GTM);
#pragma unroll
        for (int j = 0; j < JB; ++j) rb[j] = piece(B, ldb, BT, n0, N, k0, tid + 256 * j, TN);
    };
    auto put = [&](float* tile, bool T, int idx4, int rows_tile, int pitch_t, const f32x4 v) {
        if (T) { const int q = rows_tile / 4; *reinterpret_cast<f32x4*>(tile + (idx4 / q) * pitch_t + (idx4 % q) * 4) = v; }
        else *reinterpret_cast<f32x4*>(tile + (idx4 >> 3) * GP + (idx4 & 7) * 4) = v;
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) put(As[buf], AT, tid + 256 * j, GTM, PRA, ra[j]);
#pragma unroll
        for (int j = 0; j < JB; ++j) put(Bs[buf], BT, tid + 256 * j, TN, PRB, rb[j]);
    };
    // fragment of 4 consecutive k (k = 8 c + 4 half + e) of one tile row
    auto frag = [&](const float* tile, bool T, int pitch_t, int row, int c) -> f32x4 {
        if (!T) return *reinterpret_cast<const f32x4*>(tile + row * GP + 8 * c + 4 * L.half);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tile[(8 * c + 4 * L.half + e) * pitch_t + row];
        return v;
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int c = 0; c < GK / 8; ++c) {
            f32x4 a[NF];
#pragma unroll
            for (int nt = 0; nt < NF; ++nt) a[nt] = frag(Bs[buf], BT, PRB, wn * (TN / 2) + 32 * nt + L.l31, c);   // D rows = n
            const f32x4 b0 = frag(As[buf], AT, PRA, wm * 64 + L.l31, c);                   // D cols = m
            const f32x4 b1 = frag(As[buf], AT, PRA, wm * 64 + 32 + L.l31, c);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NF; ++nt) {
                    if (NEO_SGEMM_ABLATE & 1) { acc[0][nt][e] += a[nt][e] * b0[e]; acc[1][nt][e] += a[nt][e] * b1[e]; continue; }
                    acc[0][nt] = NEO_MFMA(a[nt][e], b0[e], acc[0][nt]);
                    acc[1][nt] = NEO_MFMA(a[nt][e], b1[e], acc[1][nt]);
                }
        }
    };
    const int nseg = sg.n > 0 ? sg.n : 1;
    int koff = 0;
    for (int si = 0; si < nseg; ++si) {
    if (sg.n > 0) {
        A = sg.a[si];
        lda = sg.lda[si];
        B = BT ? B_in + (long)koff * ldb : B_in + koff;
        kbeg = 0;
        kend = sg.k[si];
        koff += sg.k[si];
    }
    int kdone = kbeg;            // K steps [kbeg, kdone) are accumulated when the guarded loop below starts
    // ---- interior tiles: every full K step through a branch-free pipeline, operands requested TWO steps ahead (two register
    //      sets; a step's loads have two multiply phases to arrive from HBM), pointers advanced instead of recomputed ----
    const int nfull = (kend - kbeg) / GK;
    if (m0 + GTM <= M && n0 + TN <= N && nfull >= 1) {
        const float* pa[4];
        const float* pb[JB];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx4 = tid + 256 * j;
            pa[j] = AT ? A + (long)(kbeg + idx4 / (GTM / 4)) * lda + m0 + (idx4 % (GTM / 4)) * 4
                       : A + (long)(m0 + (idx4 >> 3)) * lda + kbeg + (idx4 & 7) * 4;
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int idx4 = tid + 256 * j;
            pb[j] = BT ? B + (long)(kbeg + idx4 / (TN / 4)) * ldb + n0 + (idx4 % (TN / 4)) * 4
                       : B + (long)(n0 + (idx4 >> 3)) * ldb + kbeg + (idx4 & 7) * 4;
        }
        const long sa_step = AT ? (long)GK * lda : GK, sb_step = BT ? (long)GK * ldb : GK;
        f32x4 qa[2][4], qb[2][JB];
        auto fetch_q = [&](auto sc) {
            constexpr int S = decltype(sc)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) { if (!(NEO_SGEMM_ABLATE & 2)) qa[S][j] = *reinterpret_cast<const f4u*>(pa[j]); pa[j] += sa_step; }
#pragma unroll
            for (int j = 0; j < JB; ++j) { if (!(NEO_SGEMM_ABLATE & 2)) qb[S][j] = *reinterpret_cast<const f4u*>(pb[j]); pb[j] += sb_step; }
        };
        auto stage_q = [&](auto sc, int buf) {
            constexpr int S = decltype(sc)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) put(As[buf], AT, tid + 256 * j, GTM, PRA, qa[S][j]);
#pragma unroll
            for (int j = 0; j < JB; ++j) put(Bs[buf], BT, tid + 256 * j, TN, PRB, qb[S][j]);
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        int i = 0;
        fetch_q(S0());
        stage_q(S0(), 0);                                  // step 0 -> LDS buffer 0
        if (nfull >= 5) {
            fetch_q(S0());                                 // step 1
            fetch_q(S1());                                 // step 2
            __syncthreads();
            // invariant at the top: LDS buffer 0 = step i, set 0 = step i + 1, set 1 = step i + 2 (both in flight)
            for (; i + 4 < nfull; i += 2) {
                compute(0);
                stage_q(S0(), 1);                          // waits for set 0 only: set 1's loads stay in flight
                fetch_q(S0());                             // step i + 3
                __syncthreads();
                compute(1);
                stage_q(S1(), 0);
                fetch_q(S1());                             // step i + 4
                __syncthreads();
            }
        } else {
            if (nfull >= 2) fetch_q(S0());
            if (nfull >= 3) fetch_q(S1());
            __syncthreads();
        }
        // drain: 1..4 steps left; buffer 0 = step i, set 0 = step i + 1, set 1 = step i + 2 where they exist, step i + 3 unrequested
        const int left = nfull - i;
        compute(0);
        if (left >= 2) {
            stage_q(S0(), 1);
            if (left >= 4) fetch_q(S0());
            __syncthreads();
            compute(1);
        }
        if (left >= 3) {
            stage_q(S1(), 0);
            __syncthreads();
            compute(0);
        }
        if (left >= 4) {
            stage_q(S0(), 1);
            __syncthreads();
            compute(1);
        }
        __syncthreads();                                   // every wave is done with the LDS tiles before the tail step reuses them
        kdone = kbeg + nfull * GK;
    }
    // ---- everything else (edge tiles, the partial last K step): guarded loads, one step ahead ----
    if (kdone < kend) {
        fetch(kdone);
        stage(0);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = kdone; k0 < kend; k0 += GK, buf ^= 1) {
        const bool more = k0 + GK < kend;
        if (more) fetch(k0 + GK);
        compute(buf);
        if (more) stage(buf ^ 1);
        __syncthreads();
    }
    }   // segments
    // ---- epilogue: lane = row m (l31), registers 4 g + e = columns n = 8 g + 4 half + e: 16 B per lane and g ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int gm = m0 + wm * 64 + 32 * t + L.l31;
        if (gm >= M) continue;
#pragma unroll
        for (int nt = 0; nt < NF; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int gn = n0 + wn * (TN / 2) + 32 * nt + 8 * g + 4 * L.half;
            if (gn >= N) continue;
            float* dst = C + (long)gm * ldc + gn;
            const bool full = gn + 3 < N;
            f32x4 v, old = {0.0f, 0.0f, 0.0f, 0.0f}, mk = {1.0f, 1.0f, 1.0f, 1.0f}, bs = {0.0f, 0.0f, 0.0f, 0.0f};
            if (full) {
                if (ep.bias && blockIdx.z == 0) bs = *reinterpret_cast<const f4u*>(ep.bias + gn);
                if (ep.accumulate == 1) old = *reinterpret_cast<const f4u*>(dst);
                if (ep.mask) mk = *reinterpret_cast<const f4u*>(ep.mask + (long)gm * ep.ldm + gn);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gn + e < N) {
                        if (ep.bias && blockIdx.z == 0) bs[e] = ep.bias[gn + e];
                        if (ep.accumulate == 1) old[e] = dst[e];
                        if (ep.mask) mk[e] = ep.mask[(long)gm * ep.ldm + gn + e];
                    }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[t][nt][4 * g + e] * ep.scale + bs[e];
                if (ep.accumulate == 1) x = old[e] + x;
                if (ep.relu) x = fmaxf(x, 0.0f);
                if (!(mk[e] > 0.0f)) x = 0.0f;
                v[e] = x;
            }
            if (ep.accumulate == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gn + e < N) atomicAdd(dst + e, v[e]);
            } else if (full) {
                *reinterpret_cast<f4u*>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gn + e < N) dst[e] = v[e];
            }
        }
    }
}

// Weight gradient of one layer segment, with the bias gradient fused: W[M][N] += Z^T X, db[M] += colsum(Z) over the K = rows
// slice of this workgroup (Z = dZ stored [K][M], X stored [K][N]: both with the reduction index slowest).  128 x NT tiles -
// a 128 x 128 layer is ONE tile, so every row of Z and of X is read once (k_sgemm<true, true> read Z once per 64 columns of X
// and k_colsum once more) - 4 waves of 64 (m) x NT / 2 (n), K stepped by 32 through two [k][rows] LDS tiles, the next step's
// operands in registers while this one is multiplied (2 workgroups per CU: a step is >= 4096 MFMA cycles per wave, the loads
// of one step have that long to arrive).  At 32 flop per byte the launch is balanced between the fp32 matrix peak and HBM.
// Partial tiles of the K slices are accumulated atomically (W, db zeroed by the caller).
template <int NT>
__global__ __launch_bounds__(256, 2) void k_dw(int M, int N, int K, const float* __restrict__ Z, long ldz,
                                               const float* __restrict__ X, long ldx, float* __restrict__ part,
                                               float* __restrict__ part_db, int k_per_split) {
    constexpr int PA = 128 + 4, PB = NT + 4;     // pitches: 4 * pitch = 16 mod 32 (the two lane halves of a fragment read hit different bank halves)
    constexpr int NF = NT / 64;                  // n fragments per wave
    constexpr int JB = NT / 32;                  // 16-B pieces of the X tile per thread and K step (Z: 4)
    constexpr int QB = NT / 4;
    __shared__ __attribute__((aligned(16))) float As[2][GK * PA];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK * PB];
    LaneCtx L;
    L.init();
    const int tid = threadIdx.x;
    // all output tiles of one K slice on one XCD: the slice's rows of Z and X cross the fabric once
    const int tiles_xy = (int)(gridDim.x * gridDim.y);
    long slice;
    int txy;
    xcd_pair((long)blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z), tiles_xy, (long)gridDim.z, slice, txy);
    const int bx = txy % (int)gridDim.x, by = txy / (int)gridDim.x, bz = (int)slice;
    const int m0 = by * 128, n0 = bx * NT;
    const int kbeg = bz * k_per_split, kend = min(K, kbeg + k_per_split);
    const int wm = L.wv & 1, wn = L.wv >> 1;
    f32x16 acc[2][NF];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NF; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    // this thread's pieces: Z piece j = row k (tid >> 5) + 8 j, columns m0 + 4 (tid & 31) ..; X piece j = row k tid / QB + (256 / QB) j
    const int ka = tid >> 5, ra = (tid & 31) * 4, kb = tid / QB, rb = (tid % QB) * 4;
    unsigned mka = 0, mkb = 0;                   // valid elements of a piece (edge tiles: M = 1, 3, 64; N = 27, 63)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (m0 + ra + e < M) mka |= 1u << e;
        if (n0 + rb + e < N) mkb |= 1u << e;
    }
    const float* pa = Z + (long)(kbeg + ka) * ldz + m0 + ra;
    const float* pb = X + (long)(kbeg + kb) * ldx + n0 + rb;
    f32x4 qa[4], qb[JB];
    auto guarded = [&](const float* p, unsigned mask, bool kv) -> f32x4 {
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (kv) {
            if (mask == 15u) v = *reinterpret_cast<const f4u*>(p);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (mask >> e & 1u) v[e] = p[e];
            }
        }
        return v;
    };
    // operands of the K step starting at k0 -> registers (FAST: interior tile and a full step, no guards)
    auto fetch = [&](auto fastc, int k0) {
        constexpr bool FAST = decltype(fastc)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* p = pa + (long)(8 * j) * ldz;
            if constexpr (FAST) qa[j] = *reinterpret_cast<const f4u*>(p);
            else qa[j] = guarded(p, mka, k0 + ka + 8 * j < kend);
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const float* p = pb + (long)((256 / QB) * j) * ldx;
            if constexpr (FAST) qb[j] = *reinterpret_cast<const f4u*>(p);
            else qb[j] = guarded(p, mkb, k0 + kb + (256 / QB) * j < kend);
        }
        pa += (long)GK * ldz;
        pb += (long)GK * ldx;
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(As[buf] + (ka + 8 * j) * PA + ra) = qa[j];
#pragma unroll
        for (int j = 0; j < JB; ++j) *reinterpret_cast<f32x4*>(Bs[buf] + (kb + (256 / QB) * j) * PB + rb) = qb[j];
    };
    const bool bias_wg = part_db != nullptr && bx == 0 && tid < 128;
    float cs = 0.0f;
    auto compute = [&](int buf) {
        if (bias_wg) {
#pragma unroll
            for (int k = 0; k < GK; ++k) cs += As[buf][k * PA + tid];
        }
#pragma unroll
        for (int c = 0; c < GK / 8; ++c) {
            f32x4 a[NF], b[2];                    // 4 consecutive k (8 c + 4 half + e) of rows n / m
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * c + 4 * L.half + e;
#pragma unroll
                for (int nt = 0; nt < NF; ++nt) a[nt][e] = Bs[buf][k * PB + wn * (NT / 2) + 32 * nt + L.l31];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) b[mt][e] = As[buf][k * PA + wm * 64 + 32 * mt + L.l31];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NF; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = NEO_MFMA(b[mt][e], a[nt][e], acc[mt][nt]);   // D rows = m, D cols = n
        }
    };
    // nst K steps from k0: one barrier per step
    auto run = [&](auto fastc, int k0, int nst) {
        if (nst <= 0) return;
        fetch(fastc, k0);
        stage(0);
        __syncthreads();
        for (int i = 0; i < nst; ++i) {
            const bool more = i + 1 < nst;
            if (more) fetch(fastc, k0 + (i + 1) * GK);
            compute(i & 1);
            if (more) stage((i + 1) & 1);
            __syncthreads();
        }
    };
    const bool interior = m0 + 128 <= M && n0 + NT <= N;
    const int nfull = interior ? (kend - kbeg) / GK : 0;
    run(std::true_type(), kbeg, nfull);
    const int kdone = kbeg + nfull * GK;
    run(std::false_type(), kdone, (kend - kdone + GK - 1) / GK);
    // ---- epilogue: this slice's partial tile -> part[slice][tile][m][n]; k_dw_reduce sums the slices.  Adding the slices
    //      atomically into W is what the weight gradients used to cost: ~500 workgroups finish together and add into the same
    //      64 KB (profiles/r05_train_dw.log, dW of a 1.18 M-row training op: k_sgemm<true, true> + k_colsum 13.0 ms, this kernel
    //      with atomics 10.4 ms (lane = m) / 6.1 ms (lane = n), with partial tiles 5.5 ms = 110 TFLOP/s).
    //      lane = n (l31), registers 4 g + e = m = 8 g + 4 half + e: a store instruction covers 32 consecutive floats of two rows ----
    const long tile = ((long)bz * gridDim.y + by) * gridDim.x + bx;
    float* pt = part + tile * (128 * NT);
#pragma unroll
    for (int nt = 0; nt < NF; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    pt[(wm * 64 + 32 * mt + 8 * g + 4 * L.half + e) * NT + wn * (NT / 2) + 32 * nt + L.l31] = acc[mt][nt][4 * g + e];
    if (bias_wg) part_db[((long)bz * gridDim.y + by) * 128 + tid] = cs;
}

// W[m][n] += sum over slices of part[slice][tile][m][n], db[m] += sum of part_db: one float4 of one tile row per thread and
// DW_ZG slices per workgroup row (their loads all in flight), then one atomic per element and slice group (<= 32 per element)
constexpr int DW_ZG = 16;
template <int NT>
__global__ void k_dw_reduce(int M, int N, int nz, const float* __restrict__ part, const float* __restrict__ part_db, int tiles_x,
                            int tiles_y, float* __restrict__ W, long ldw, float* __restrict__ db) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int z0 = blockIdx.y * DW_ZG, z1 = min(nz, z0 + DW_ZG);
    constexpr int Q = 128 * NT / 4;                // float4 pieces of a tile
    const long tiles = (long)tiles_x * tiles_y;
    if (i < tiles * Q) {
        const int t = (int)(i / Q), q = (int)(i - (long)t * Q);
        const int ml = q / (NT / 4), nl = (q % (NT / 4)) * 4;
        const int gm = (t / tiles_x) * 128 + ml, gn = (t % tiles_x) * NT + nl;
        if (gm < M && gn < N) {
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
            for (int z = z0; z < z1; ++z) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(part + ((long)z * tiles + t) * (128 * NT) + ml * NT + nl);
                v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (gn + e < N) atomicAdd(W + (long)gm * ldw + gn + e, v[e]);
        }
    }
    if (db != nullptr && i < (long)tiles_y * 128 && i < M) {
        float sacc = 0.0f;
        for (int z = z0; z < z1; ++z) sacc += part_db[((long)z * tiles_y + i / 128) * 128 + (i & 127)];
        atomicAdd(db + i, sacc);
    }
}

// out[p][c] = (1 / NV) sum_v in[v P + p][c]   (neo360/util.py:599-610 combine_interleaved 'average'), optional ReLU
__global__ void k_view_mean(const float* __restrict__ in, int NV, long P, int C, int relu, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * C) return;
    float s = 0.0f;
    for (int v = 0; v < NV; ++v) s += in[(long)v * P * C + i];
    s = s / (float)NV;
    out[i] = relu ? fmaxf(s, 0.0f) : s;
}

// its backward: out[v P + p][c] (+)= g[p][c] / NV
__global__ void k_view_bcast(const float* __restrict__ g, int NV, long P, int C, int accumulate, float* __restrict__ out, long ldo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * C) return;
    const long p = i / C;
    const int c = (int)(i - p * C);
    const float v = g[i] / (float)NV;
    for (int w = 0; w < NV; ++w) {
        float* d = out + ((long)w * P + p) * ldo + c;
        *d = accumulate ? *d + v : v;
    }
}

// g[r][c] = mask[r][c] > 0 ? g[r][c] : 0   (ReLU backward in place; g with row pitch ldg, mask dense rows x C)
__global__ void k_relu_mask(float* __restrict__ g, long ldg, const float* __restrict__ mask, int C, long rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const long r = i / C;
    if (!(mask[i] > 0.0f)) g[r * ldg + (i - r * C)] = 0.0f;
}

// dst[r][0..cols) = src[r][0..cols) between row-major matrices of different pitch (cols % 4 == 0): the projected-space training path
// initialises a layer's pre-activations with its half of the gathered (rows, 256) projected features and hands the two first-layer
// gradients back as the halves of one (rows, 256) tensor
__global__ void k_copy_cols(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, long rows, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = cols / 4;
    if (i >= rows * c4) return;
    const long r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    *reinterpret_cast<f4u*>(dst + r * ldd + c) = *reinterpret_cast<const f4u*>(src + r * lds_ + c);
}

// out[c] += sum_m g[m][c]: one column per thread, 256 rows per block, atomics across blocks (out zeroed by the caller)
__global__ void k_colsum(const float* __restrict__ g, long M, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const long r0 = (long)blockIdx.y * 256, r1 = r0 + 256 < M ? r0 + 256 : M;
    float s = 0.0f;
    for (long r = r0; r < r1; ++r) s += g[r * C + c];
    atomicAdd(out + c, s);
}

template <bool AT, bool BT>
void gemm(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, const GemmEpi& ep,
          int splits, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    int kps = (K + splits - 1) / splits;
    kps = ((kps + GK - 1) / GK) * GK;
    const int nz = (K + kps - 1) / kps;
    const GemmSegs none{{nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}, {0, 0, 0, 0}, 0};
    hipLaunchKernelGGL((k_sgemm<AT, BT, GTN>), dim3((N + GTN - 1) / GTN, (M + GTM - 1) / GTM, nz), dim3(256), 0, s, M, N, K, A, lda, B,
                           ldb, C, ldc, ep, kps, none);
}

// C[M][N] = [A0 | A1 | ..][M][K0 + K1 + ..] . B[N][K0 + K1 + ..]^T  (B = a weight matrix (out, in), K fastest; dense segments)
struct Seg { const float* a; int k; };
void gemm_cat(int M, int N, const Seg* segs, int nseg, const float* B, long ldb, float* C, long ldc, const GemmEpi& ep,
              hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    GemmSegs sg{{nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}, {0, 0, 0, 0}, nseg};
    int K = 0;
    for (int i = 0; i < nseg; ++i) { sg.a[i] = segs[i].a; sg.lda[i] = segs[i].k; sg.k[i] = segs[i].k; K += segs[i].k; }
    hipLaunchKernelGGL((k_sgemm<false, false, GTN>), dim3((N + GTN - 1) / GTN, (M + GTM - 1) / GTM, 1), dim3(256), 0, s, M, N, K,
                           segs[0].a, (long)segs[0].k, B, ldb, C, ldc, ep, K, sg);
}

inline GemmEpi epi(const float* bias = nullptr, int relu = 0, int accumulate = 0, const float* mask = nullptr, int ldm = 0,
                   float scale = 1.0f) {
    return GemmEpi{bias, mask, ldm, relu, accumulate, scale};
}

inline unsigned blocks(long n) { return (unsigned)((n + 255) / 256); }

// split-K of a weight-gradient GEMM (C (M x N) += A^T B over K = rows): enough K slices that the grid holds >= ~1024 workgroups
// whatever the layer's shape - a 128 x 128 layer is 2 output tiles, and 71 slices of 8192 rows (round 3's rule) left 142 workgroups
// on 256 CUs; each slice keeps >= 1024 rows so the atomically accumulated partial tiles stay few
inline int split_k(int M, int N, long K) {
    const long tiles = (long)((N + GTN - 1) / GTN) * ((M + GTM - 1) / GTM);
    long nz = (1024 + tiles - 1) / tiles;
    const long cap = K / 1024 > 0 ? K / 1024 : 1;
    if (nz > cap) nz = cap;
    const long floor_ = (K + 8191) / 8192;
    return (int)(nz > floor_ ? nz : floor_);
}

// W (M x N, row pitch ldw) += Z^T X over K rows, db (M) += column sums of Z (null: a later segment of the same layer);
// enough K slices for one resident wave of workgroups (2 per CU), each at least 1024 rows
// part: DW_PART_FLOATS of scratch (the slices' partial tiles and bias sums)
constexpr long DW_PART_TILES = 520;        // slices x tiles of one call: <= 512 + tiles - 1, tiles <= 8 (256 x 256: 4)
constexpr long DW_PART_FLOATS = DW_PART_TILES * (128 * 128 + 128);
void dw_gemm(int M, int N, int K, const float* Z, long ldz, const float* X, long ldx, float* W, long ldw, float* db, float* part,
             hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    const int NT = N > 64 ? 128 : 64;
    const int tx = (N + NT - 1) / NT, ty = (M + 127) / 128, tiles = tx * ty;
    long nz = (512 + tiles - 1) / tiles;
    const long cap = K / 1024 > 0 ? K / 1024 : 1;
    if (nz > cap) nz = cap;
    if (nz * tiles > DW_PART_TILES) nz = DW_PART_TILES / tiles;
    int kps = (int)((K + nz - 1) / nz);
    kps = ((kps + GK - 1) / GK) * GK;
    const int slices = (K + kps - 1) / kps;
    float* part_db = part + DW_PART_TILES * (128 * 128);
    const dim3 grid(tx, ty, slices);
    const dim3 rgrid((unsigned)(((long)tiles * 128 * NT / 4 + 255) / 256), (slices + DW_ZG - 1) / DW_ZG);
    if (NT == 128) {
        hipLaunchKernelGGL((k_dw<128>), grid, dim3(256), 0, s, M, N, K, Z, ldz, X, ldx, part, db ? part_db : nullptr, kps);
        hipLaunchKernelGGL((k_dw_reduce<128>), rgrid, dim3(256), 0, s, M, N, slices, part, part_db, tx, ty, W, ldw, db);
    } else {
        hipLaunchKernelGGL((k_dw<64>), grid, dim3(256), 0, s, M, N, K, Z, ldz, X, ldx, part, db ? part_db : nullptr, kps);
        hipLaunchKernelGGL((k_dw_reduce<64>), rgrid, dim3(256), 0, s, M, N, slices, part, part_db, tx, ty, W, ldw, db);
    }
}

#include "train_chain.h"

const int g_heads_fused = [] { const char* e = getenv("NEO360_TRAIN_HEADS"); return e ? atoi(e) : 1; }();    // 0: the P-sized forward tail as separate launches (A/B)
int g_chain_fused = 1;      // 1: the per-row part of the projected-space NeRFPPMLP chain as one kernel each way (train_chain.h); 0: layer by layer

}  // namespace

// y (rows x out_f) (+)= x (rows x in_f) W^T + b, optional ReLU: one linear layer of a training chain composed by the caller
void launch_linear_forward(long rows, int out_f, int in_f, const float* x, long ldx, const float* w, long ldw, const float* bias,
                           int relu, int accumulate, float* y, long ldy, hipStream_t s) {
    gemm<false, false>((int)rows, out_f, in_f, x, ldx, w, ldw, y, ldy, epi(bias, relu, accumulate ? 1 : 0), 1, s);
}
// gx (rows x in_f) (+)= gy (rows x out_f) W
void launch_linear_input_grad(long rows, int in_f, int out_f, const float* gy, long ldy, const float* w, long ldw, int accumulate,
                              float* gx, long ldx, hipStream_t s) {
    gemm<false, true>((int)rows, in_f, out_f, gy, ldy, w, ldw, gx, ldx, epi(nullptr, 0, accumulate ? 1 : 0), 1, s);
}
size_t weight_grad_scratch_floats() { return (size_t)DW_PART_FLOATS; }
void launch_weight_grad(int M, int N, int K, const float* dY, long ldy, const float* X, long ldx, float* dW, long ldw, float* db,
                        float* scratch, hipStream_t s) {
    dw_gemm(M, N, K, dY, ldy, X, ldx, dW, ldw, db, scratch, s);
}

// ---- tape layout (floats): h0, h1, h2, h3, bott (R x 128 each), y0 (R x 64), hm (P x 128), ym (P x 64), y1 (P x 64) ----
size_t tp_train_tape_floats(int NV, long P) {
    const long R = (long)NV * P;
    return (size_t)(R * (5 * 128 + 64) + P * (128 + 64 + 64));
}
// scratch of the backward: three R x 128 gradient buffers (the third: g_bott of the fused chain), one R x 64, three P-sized
size_t tp_train_scratch_floats(int NV, long P) {
    const long R = (long)NV * P;
    return (size_t)(R * (3 * 128 + 64) + P * (128 + 64 + 64) + DW_PART_FLOATS);
}

int train_chain_mode(int mode) {
    const int old = g_chain_fused;
    if (mode >= 0) g_chain_fused = mode ? 1 : 0;
    return old;
}

// w / b order as neo_tp_upload_mlp: pts_linears.0..3, views_linear.0, views_linear.1, bottleneck, density, rgb
// the input rows [x_enc (pe) | local (512) | world (128)] are never concatenated: the two layers that read them take the three
// tensors as segments of one reduction (gemm_cat)
// pre (round 5, the training analogue of the evaluators' pre-projection): (R, 256) = the gathered PROJECTED latent, i.e. the local
// features' contribution [W0_loc f | W3_loc f] to the pre-activations of layer 0 and of the skip half of layer 3, formed in texel space
// by the caller (bilerp is linear: W bilerp(F) = bilerp(W F)).  With pre != null `local` is not read: the two layers start from their
// half of `pre` and reduce over [x_enc | world] only - the 512-wide segments of both GEMMs (and of their dX / dW) are gone.
void launch_tp_train_forward(int pe, const float* const* w, const float* const* b, const float* x_enc, const float* local,
                             const float* world, const float* cond, int NV, long P, float* tape, float* raw_rgb,
                             float* raw_sigma, hipStream_t s, const float* pre) {
    const long R = (long)NV * P;
    const int K0 = pe + 640;
    float* h0 = tape; float* h1 = h0 + R * 128; float* h2 = h1 + R * 128; float* h3 = h2 + R * 128;
    float* bott = h3 + R * 128; float* y0 = bott + R * 128; float* hm = y0 + R * 64; float* ym = hm + P * 128; float* y1 = ym + P * 64;
    if (pre && g_chain_fused && (pe == 63 || pe == 84)) {
        // one kernel for everything per row: layers 0..3 (train_chain.h).  Behind relu(L3_v) the network is linear up to the view means
        // (no activation on the bottleneck; view layer 0 is averaged over the views before its ReLU), so the bottleneck and view
        // layer 0 run on the MEANS, P rows instead of NV P: W4a mean_v(W6 h3_v + b6) + W4c mean_v(cond_v) + b4 = W4a (W6 hm + b6) + W4c cm + b4.
        // The tape's per-row bottleneck / view-layer regions are not written; their first P rows hold bm = W6 hm + b6 and cm = mean_v cond.
        ChainFwdArgs a{w[0], w[1], w[2], w[3], b[0], b[1], b[2], b[3], x_enc, world, pre, h0, h1, h2, h3, R, pe};
        const dim3 grid((unsigned)((R + CH_ROWS - 1) / CH_ROWS));
        if (pe == 63) hipLaunchKernelGGL((k_tp_chain_fwd<8, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_tp_chain_fwd<11, false>), grid, dim3(256), 0, s, a);
        float* bm = bott;                                                                                        // (P, 128)
        float* cm = bott + P * 128;                                                                              // (P, 27)
        if (g_heads_fused) {
            // the whole P-sized tail - two view means, six small products - as one kernel (train_chain.h)
            HeadsFwdArgs ha{h3, cond, w[4], b[4], w[5], b[5], w[6], b[6], w[7], b[7], w[8], b[8], hm, bm, cm, ym, y1, raw_sigma, raw_rgb, P, NV};
            hipLaunchKernelGGL(k_tp_heads_fwd, dim3((unsigned)((P + CH_ROWS - 1) / CH_ROWS)), dim3(256), 0, s, ha);
            return;
        }
        hipLaunchKernelGGL(k_view_mean, dim3(blocks(P * 128)), dim3(256), 0, s, h3, NV, P, 128, 0, hm);
        gemm<false, false>((int)P, 1, 128, hm, 128, w[7], 128, raw_sigma, 1, epi(b[7], 0), 1, s);
        gemm<false, false>((int)P, 128, 128, hm, 128, w[6], 128, bm, 128, epi(b[6], 0), 1, s);                   // mean bottleneck
        hipLaunchKernelGGL(k_view_mean, dim3(blocks(P * 27)), dim3(256), 0, s, cond, NV, P, 27, 0, cm);          // mean direction encoding
        gemm<false, false>((int)P, 64, 128, bm, 128, w[4], 155, ym, 64, epi(b[4], 0), 1, s);                     // view layer 0 on the means
        gemm<false, false>((int)P, 64, 27, cm, 27, w[4] + 128, 155, ym, 64, epi(nullptr, 1, 1), 1, s);           // C = relu(C + ..)
        gemm<false, false>((int)P, 64, 64, ym, 64, w[5], 64, y1, 64, epi(b[5], 1), 1, s);
        gemm<false, false>((int)P, 3, 64, y1, 64, w[8], 64, raw_rgb, 3, epi(b[8], 0), 1, s);
        return;
    }
    if (pre) {
        hipLaunchKernelGGL(k_copy_cols, dim3(blocks(R * 32)), dim3(256), 0, s, pre, 256L, h0, 128L, R, 128);
        gemm<false, false>((int)R, 128, pe, x_enc, pe, w[0], K0, h0, 128, epi(b[0], 0, 1), 1, s);                    // h0 = pre0 + x_enc W0_pe + b0
        gemm<false, false>((int)R, 128, 128, world, 128, w[0] + pe + 512, K0, h0, 128, epi(nullptr, 1, 1), 1, s);     // relu(h0 + world W0_w)
    } else {
        const Seg in0[3] = {{x_enc, pe}, {local, 512}, {world, 128}};
        gemm_cat((int)R, 128, in0, 3, w[0], K0, h0, 128, epi(b[0], 1), s);
    }
    gemm<false, false>((int)R, 128, 128, h0, 128, w[1], 128, h1, 128, epi(b[1], 1), 1, s);
    gemm<false, false>((int)R, 128, 128, h1, 128, w[2], 128, h2, 128, epi(b[2], 1), 1, s);
    if (pre) {
        hipLaunchKernelGGL(k_copy_cols, dim3(blocks(R * 32)), dim3(256), 0, s, pre + 128, 256L, h3, 128L, R, 128);
        const Seg in3p[2] = {{h2, 128}, {x_enc, pe}};                                                                   // columns [0, 128 + pe) of W3
        gemm_cat((int)R, 128, in3p, 2, w[3], 128 + K0, h3, 128, epi(b[3], 0, 1), s);
        gemm<false, false>((int)R, 128, 128, world, 128, w[3] + 128 + pe + 512, 128 + K0, h3, 128, epi(nullptr, 1, 1), 1, s);
    } else {
    // layer 3 on [h2 | x_enc | local | world] (the skip concat after layer index 2): one reduction over four segments
    const Seg in3[4] = {{h2, 128}, {x_enc, pe}, {local, 512}, {world, 128}};
    gemm_cat((int)R, 128, in3, 4, w[3], 128 + K0, h3, 128, epi(b[3], 1), s);
    }
    gemm<false, false>((int)R, 128, 128, h3, 128, w[6], 128, bott, 128, epi(b[6], 0), 1, s);                     // bottleneck, per view
    hipLaunchKernelGGL(k_view_mean, dim3(blocks(P * 128)), dim3(256), 0, s, h3, NV, P, 128, 0, hm);
    gemm<false, false>((int)P, 1, 128, hm, 128, w[7], 128, raw_sigma, 1, epi(b[7], 0), 1, s);
    // view layer 0 on [bott | cond]
    gemm<false, false>((int)R, 64, 128, bott, 128, w[4], 155, y0, 64, epi(b[4], 0), 1, s);
    gemm<false, false>((int)R, 64, 27, cond, 27, w[4] + 128, 155, y0, 64, epi(nullptr, 0, 1), 1, s);
    hipLaunchKernelGGL(k_view_mean, dim3(blocks(P * 64)), dim3(256), 0, s, y0, NV, P, 64, 1, ym);               // mean over views, ReLU
    gemm<false, false>((int)P, 64, 64, ym, 64, w[5], 64, y1, 64, epi(b[5], 1), 1, s);
    gemm<false, false>((int)P, 3, 64, y1, 64, w[8], 64, raw_rgb, 3, epi(b[8], 0), 1, s);
}

// PixelNeRF's late-fusion MLP (vanilla_nerf/model_pixel.py:96-131) as the same kind of chain (round 6): the NeRFPPMLP chain without
// the world features and without the skip (netdepth 4: the skip after layer index 4 never fires).  w / b order as
// neo_pix_upload_mlp: pts_linears.0..3 (128 x 575, 128 x 128 x 3), views_linear.0 (128 x 155), views_linear.1 (128 x 128), bottleneck, density,
// rgb.  pre (R, 128) = the gathered PROJECTED latent W0[:, 63:575] f (formed per texel by the caller); tape layout
// (pix_train_tape_floats): h0 h1 h2 h3 bott y0 (R x 128) hm ym y1 (P x 128).
// tape / scratch of the PixelNeRF chain (view width 128)
size_t pix_train_tape_floats(int NV, long P) {
    const long R = (long)NV * P;
    return (size_t)(R * (5 * 128 + 128) + P * (128 + 128 + 128));
}
size_t pix_train_scratch_floats(int NV, long P) {
    const long R = (long)NV * P;
    return (size_t)(R * (2 * 128 + 128) + P * (128 + 128 + 128) + DW_PART_FLOATS);
}

void launch_pix_train_forward(const float* const* w, const float* const* b, const float* x_enc, const float* pre, const float* cond,
                              int NV, long P, float* tape, float* raw_rgb, float* raw_sigma, hipStream_t s) {
    const long R = (long)NV * P;
    const int pe = 63, K0 = pe + 512, VC = 128;                     // VC: netwidth_condition (model_pixel.py:44)
    float* h0 = tape; float* h1 = h0 + R * 128; float* h2 = h1 + R * 128; float* h3 = h2 + R * 128;
    float* bott = h3 + R * 128; float* y0 = bott + R * 128; float* hm = y0 + R * VC; float* ym = hm + P * 128; float* y1 = ym + P * VC;
    if (g_chain_fused) {
        // the four per-row layers as ONE kernel (train_chain.h, PIX form)
        ChainFwdArgs a{w[0], w[1], w[2], w[3], b[0], b[1], b[2], b[3], x_enc, nullptr, pre, h0, h1, h2, h3, R, pe};
        hipLaunchKernelGGL((k_tp_chain_fwd<8, true>), dim3((unsigned)((R + CH_ROWS - 1) / CH_ROWS)), dim3(256), 0, s, a);
    } else {
    hipLaunchKernelGGL(k_copy_cols, dim3(blocks(R * 32)), dim3(256), 0, s, pre, 128L, h0, 128L, R, 128);
    gemm<false, false>((int)R, 128, pe, x_enc, pe, w[0], K0, h0, 128, epi(b[0], 1, 1), 1, s);                        // relu(pre + x_enc W0_pe + b0)
    gemm<false, false>((int)R, 128, 128, h0, 128, w[1], 128, h1, 128, epi(b[1], 1), 1, s);
    gemm<false, false>((int)R, 128, 128, h1, 128, w[2], 128, h2, 128, epi(b[2], 1), 1, s);
    gemm<false, false>((int)R, 128, 128, h2, 128, w[3], 128, h3, 128, epi(b[3], 1), 1, s);
    }
    if (g_chain_fused) {
        // the bottleneck has no activation and view layer 0 is averaged over the views before its ReLU (:113-126): both run on the view
        // MEANS - P rows instead of NV P - as in the NeRFPPMLP chain; bm / cm live in the first P rows of the tape's per-view regions
        float* bm = bott;
        float* cm = bott + P * 128;
        hipLaunchKernelGGL(k_view_mean, dim3(blocks(P * 128)), dim3(256), 0, s, h3, NV, P, 128, 0, hm);
        gemm<false, false>((int)P, 1, 128, hm, 128, w[7], 128, raw_sigma, 1, epi(b[7], 0), 1, s);
        gemm<false, false>((int)P, 128, 128, hm, 128, w[6], 128, bm, 128, epi(b[6], 0), 1, s);
        hipLaunchKernelGGL(k_view_mean, dim3(blocks(P * 27)), dim3(256), 0, s, cond, NV, P, 27, 0, cm);
        gemm<false, false>((int)P, VC, 128, bm, 128, w[4], 155, ym, VC, epi(b[4], 0), 1, s);
        gemm<false, false>((int)P, VC, 27, cm, 27, w[4] + 128, 155, ym, VC, epi(nullptr, 1, 1), 1, s);               // C = relu(C + ..)
        gemm<false, false>((int)P, VC, VC, ym, VC, w[5], VC, y1, VC, epi(b[5], 1), 1, s);
        gemm<false, false>((int)P, 3, VC, y1, VC, w[8], VC, raw_rgb, 3, epi(b[8], 0), 1, s);
        return;
    }
    gemm<false, false>((int)R, 128, 128, h3, 128, w[6], 128, bott, 128, epi(b[6], 0), 1, s);                         // bottleneck, per view (:113-114)
    hipLaunchKernelGGL(k_view_mean, dim3(blocks(P * 128)), dim3(256), 0, s, h3, NV, P, 128, 0, hm);                   // combine_interleaved "average"
    gemm<false, false>((int)P, 1, 128, hm, 128, w[7], 128, raw_sigma, 1, epi(b[7], 0), 1, s);
    gemm<false, false>((int)R, VC, 128, bott, 128, w[4], 155, y0, VC, epi(b[4], 0), 1, s);                           // view layer 0 on [bott | cond]
    gemm<false, false>((int)R, VC, 27, cond, 27, w[4] + 128, 155, y0, VC, epi(nullptr, 0, 1), 1, s);
    hipLaunchKernelGGL(k_view_mean, dim3(blocks(P * VC)), dim3(256), 0, s, y0, NV, P, VC, 1, ym);                    // mean over views, ReLU
    gemm<false, false>((int)P, VC, VC, ym, VC, w[5], VC, y1, VC, epi(b[5], 1), 1, s);
    gemm<false, false>((int)P, 3, VC, y1, VC, w[8], VC, raw_rgb, 3, epi(b[8], 0), 1, s);
}

// gw / gb ZEROED by the caller; g_pre (R, 128) = dL/dz0 (the lookup's backward carries it into the projected map: the gradient of
// W0's latent columns is formed in texel space); g_x_enc (R, 63) may be null.  scratch: pix_train_scratch_floats(NV, P).
void launch_pix_train_backward(const float* const* w, const float* x_enc, const float* cond, int NV, long P, const float* tape,
                               float* scratch, const float* g_rgb, const float* g_sigma, float* const* gw, float* const* gb,
                               float* g_x_enc, float* g_pre, hipStream_t s) {
    const long R = (long)NV * P;
    const int pe = 63, K0 = pe + 512, VC = 128;                     // VC: netwidth_condition (model_pixel.py:44)
    const float* h0 = tape; const float* h1 = h0 + R * 128; const float* h2 = h1 + R * 128; const float* h3 = h2 + R * 128;
    const float* bott = h3 + R * 128; const float* y0 = bott + R * 128; const float* hm = y0 + R * VC;
    const float* ym = hm + P * 128; const float* y1 = ym + P * VC;
    (void)y0;
    float* ga = scratch; float* gb2 = ga + R * 128; float* gy0 = gb2 + R * 128;                 // R-sized
    float* g_hm = gy0 + R * VC; float* g_y1 = g_hm + P * 128; float* g_ym = g_y1 + P * VC;       // P-sized
    float* part = g_ym + P * VC;                                                                 // DW_PART_FLOATS (dw_gemm)
    dw_gemm(3, VC, (int)P, g_rgb, 3, y1, VC, gw[8], VC, gb[8], part, s);                                         // rgb head
    gemm<false, true>((int)P, VC, 3, g_rgb, 3, w[8], VC, g_y1, VC, epi(nullptr, 0, 0, y1, VC), 1, s);            // x relu'(y1)
    dw_gemm(VC, VC, (int)P, g_y1, VC, ym, VC, gw[5], VC, gb[5], part, s);                                        // view layer 1
    gemm<false, true>((int)P, VC, VC, g_y1, VC, w[5], VC, g_ym, VC, epi(nullptr, 0, 0, ym, VC), 1, s);           // x relu'(mean)
    if (g_chain_fused) {
        // backward of the view branch on the means (see the forward): P-sized; g_h3[v, p] = g_hm[p] / NV
        const float* bm = bott;
        const float* cm = bott + P * 128;
        float* g_bm = gy0;                                                                                        // (P, 128) in the R x VC buffer
        dw_gemm(VC, 128, (int)P, g_ym, VC, bm, 128, gw[4], 155, gb[4], part, s);
        dw_gemm(VC, 27, (int)P, g_ym, VC, cm, 27, gw[4] + 128, 155, nullptr, part, s);
        gemm<false, true>((int)P, 128, VC, g_ym, VC, w[4], 155, g_bm, 128, epi(), 1, s);
        dw_gemm(128, 128, (int)P, g_bm, 128, hm, 128, gw[6], 128, gb[6], part, s);
        dw_gemm(1, 128, (int)P, g_sigma, 1, hm, 128, gw[7], 128, gb[7], part, s);
        gemm<false, true>((int)P, 128, 128, g_bm, 128, w[6], 128, g_hm, 128, epi(), 1, s);
        gemm<false, true>((int)P, 128, 1, g_sigma, 1, w[7], 128, g_hm, 128, epi(nullptr, 0, 1), 1, s);
        // ONE kernel for the input-gradient chain of every row (train_chain.h, PIX form): g_z3 -> gb2, g_z2 -> ga, g_z1 -> gy0, g_z0 -> g_pre
        float* gz1 = gy0;                                                                        // the R x VC buffer: g_bm (its first P rows) is consumed by now
        ChainBwdArgs a{w[0], w[1], w[2], w[3], h0, h1, h2, h3, g_hm, gb2, 128L, g_pre, 128L, ga, gz1, nullptr, R, P, pe, NV};
        hipLaunchKernelGGL(k_tp_chain_bwd<true>, dim3((unsigned)((R + CH_ROWS - 1) / CH_ROWS)), dim3(256), 0, s, a);
        dw_gemm(128, 128, (int)R, gb2, 128, h2, 128, gw[3], 128, gb[3], part, s);                                 // layer 3
        dw_gemm(128, 128, (int)R, ga, 128, h1, 128, gw[2], 128, gb[2], part, s);                                  // layer 2
        dw_gemm(128, 128, (int)R, gz1, 128, h0, 128, gw[1], 128, gb[1], part, s);                                 // layer 1
        dw_gemm(128, pe, (int)R, g_pre, 128, x_enc, pe, gw[0], K0, gb[0], part, s);                               // layer 0, encoding columns
        if (g_x_enc) gemm<false, true>((int)R, pe, 128, g_pre, 128, w[0], K0, g_x_enc, pe, epi(), 1, s);
        return;
    } else {
    hipLaunchKernelGGL(k_view_bcast, dim3(blocks(P * VC)), dim3(256), 0, s, g_ym, NV, P, VC, 0, gy0, (long)VC);        // mean over views -> rows
    dw_gemm(VC, 128, (int)R, gy0, VC, bott, 128, gw[4], 155, gb[4], part, s);                                    // view layer 0 on [bott | cond]
    dw_gemm(VC, 27, (int)R, gy0, VC, cond, 27, gw[4] + 128, 155, nullptr, part, s);
    gemm<false, true>((int)R, 128, VC, gy0, VC, w[4], 155, ga, 128, epi(), 1, s);                                // g_bott
    dw_gemm(128, 128, (int)R, ga, 128, h3, 128, gw[6], 128, gb[6], part, s);                                     // bottleneck
    gemm<false, true>((int)R, 128, 128, ga, 128, w[6], 128, gb2, 128, epi(), 1, s);                              // g_h3 from the bottleneck
    dw_gemm(1, 128, (int)P, g_sigma, 1, hm, 128, gw[7], 128, gb[7], part, s);                                    // density head on the view mean
    gemm<false, true>((int)P, 128, 1, g_sigma, 1, w[7], 128, g_hm, 128, epi(), 1, s);
    hipLaunchKernelGGL(k_view_bcast, dim3(blocks(P * 128)), dim3(256), 0, s, g_hm, NV, P, 128, 1, gb2, 128L);    // g_h3 += g_hm / NV
    }
    hipLaunchKernelGGL(k_relu_mask, dim3(blocks(R * 128)), dim3(256), 0, s, gb2, 128L, h3, 128, R);              // g_z3
    dw_gemm(128, 128, (int)R, gb2, 128, h2, 128, gw[3], 128, gb[3], part, s);                                    // layer 3
    gemm<false, true>((int)R, 128, 128, gb2, 128, w[3], 128, ga, 128, epi(nullptr, 0, 0, h2, 128), 1, s);         // g_z2
    dw_gemm(128, 128, (int)R, ga, 128, h1, 128, gw[2], 128, gb[2], part, s);                                     // layer 2
    gemm<false, true>((int)R, 128, 128, ga, 128, w[2], 128, gb2, 128, epi(nullptr, 0, 0, h1, 128), 1, s);         // g_z1
    dw_gemm(128, 128, (int)R, gb2, 128, h0, 128, gw[1], 128, gb[1], part, s);                                    // layer 1
    gemm<false, true>((int)R, 128, 128, gb2, 128, w[1], 128, g_pre, 128, epi(nullptr, 0, 0, h0, 128), 1, s);      // g_z0 -> g_pre
    dw_gemm(128, pe, (int)R, g_pre, 128, x_enc, pe, gw[0], K0, gb[0], part, s);                                  // layer 0, encoding columns
    if (g_x_enc) gemm<false, true>((int)R, pe, 128, g_pre, 128, w[0], K0, g_x_enc, pe, epi(), 1, s);
}

// gw / gb: nine weight / bias gradients, ZEROED by the caller (split-K partials are accumulated atomically); each of the
// three input gradients (R x pe, R x 512, R x 128) may be null
void launch_tp_train_backward(int pe, const float* const* w, const float* x_enc, const float* local, const float* world,
                              const float* cond, int NV, long P, const float* tape, float* scratch, const float* g_rgb,
                              const float* g_sigma, float* const* gw, float* const* gb, float* g_x_enc, float* g_local,
                              float* g_world, hipStream_t s, float* g_pre) {
    // g_pre != null: the forward ran on `pre` (projected-space path): the local segment has no dW / dX here; instead the two
    // first-layer gradients are returned as g_pre (R, 256) = [g_z0 | g_z3] for the lookup's backward into the projected map
    const long R = (long)NV * P;
    const int K0 = pe + 640;
    const float* h0 = tape; const float* h1 = h0 + R * 128; const float* h2 = h1 + R * 128; const float* h3 = h2 + R * 128;
    const float* bott = h3 + R * 128; const float* y0 = bott + R * 128; const float* hm = y0 + R * 64;
    const float* ym = hm + P * 128; const float* y1 = ym + P * 64;
    (void)y0;
    float* ga = scratch; float* gb2 = ga + R * 128; float* gy0 = gb2 + R * 128;                 // R-sized
    float* g_hm = gy0 + R * 64; float* g_y1 = g_hm + P * 128; float* g_ym = g_y1 + P * 64;       // P-sized
    float* part = g_ym + P * 64;                                                                 // DW_PART_FLOATS (dw_gemm)
    const float* in[3] = {x_enc, local, world};
    float* g_in[3] = {g_x_enc, g_local, g_world};
    const int kin[3] = {pe, 512, 128}, off[3] = {0, pe, pe + 512};
    const bool fused = g_pre && g_chain_fused && (pe == 63 || pe == 84);
    if (fused && g_heads_fused) {
        // the input-gradient part of everything P-sized as one kernel: g_y1, g_ym, g_bm, g_hm (train_chain.h)
        HeadsBwdArgs hb{g_rgb, g_sigma, y1, ym, w[4], w[5], w[6], w[7], w[8], g_y1, g_ym, part + DW_PART_FLOATS, g_hm, P};
        hipLaunchKernelGGL(k_tp_heads_bwd, dim3((unsigned)((P + CH_ROWS - 1) / CH_ROWS)), dim3(256), 0, s, hb);
        dw_gemm(3, 64, (int)P, g_rgb, 3, y1, 64, gw[8], 64, gb[8], part, s);                                     // rgb head
        dw_gemm(64, 64, (int)P, g_y1, 64, ym, 64, gw[5], 64, gb[5], part, s);                                    // view layer 1
    } else {
    // rgb head
    dw_gemm(3, 64, (int)P, g_rgb, 3, y1, 64, gw[8], 64, gb[8], part, s);
    gemm<false, true>((int)P, 64, 3, g_rgb, 3, w[8], 64, g_y1, 64, epi(nullptr, 0, 0, y1, 64), 1, s);            // x relu'(y1)
    // view layer 1
    dw_gemm(64, 64, (int)P, g_y1, 64, ym, 64, gw[5], 64, gb[5], part, s);
    gemm<false, true>((int)P, 64, 64, g_y1, 64, w[5], 64, g_ym, 64, epi(nullptr, 0, 0, ym, 64), 1, s);           // x relu'(mean)
    }
    if (fused) {
        // the forward ran view layer 0 and the bottleneck on the view means (see launch_tp_train_forward): their backward is P-sized too
        const float* bm = bott;
        const float* cm = bott + P * 128;
        float* g_bm = part + DW_PART_FLOATS;                                                     // (P, 128) in the third R x 128 buffer
        dw_gemm(64, 128, (int)P, g_ym, 64, bm, 128, gw[4], 155, gb[4], part, s);                                  // view layer 0
        dw_gemm(64, 27, (int)P, g_ym, 64, cm, 27, gw[4] + 128, 155, nullptr, part, s);
        if (!g_heads_fused) gemm<false, true>((int)P, 128, 64, g_ym, 64, w[4], 155, g_bm, 128, epi(), 1, s);      // g of the mean bottleneck
        dw_gemm(128, 128, (int)P, g_bm, 128, hm, 128, gw[6], 128, gb[6], part, s);                                // bottleneck
        dw_gemm(1, 128, (int)P, g_sigma, 1, hm, 128, gw[7], 128, gb[7], part, s);                                 // density head
        if (!g_heads_fused) {
            gemm<false, true>((int)P, 128, 128, g_bm, 128, w[6], 128, g_hm, 128, epi(), 1, s);                    // g_hm = g_bm W6 + g_sigma W7
            gemm<false, true>((int)P, 128, 1, g_sigma, 1, w[7], 128, g_hm, 128, epi(nullptr, 0, 1), 1, s);
        }
        // ONE kernel for the input-gradient chain of every row: g_h3 = g_hm / NV per view -> g_z3 .. g_z0, g_world (train_chain.h); then
        // the weight-gradient GEMMs on what it wrote: g_z3 | g_z0 (the halves of g_pre), g_z2, g_z1
        ChainBwdArgs a{w[0], w[1], w[2], w[3], h0, h1, h2, h3, g_hm, g_pre + 128, 256L, g_pre, 256L, ga, gb2, g_world, R, P, pe, NV};
        hipLaunchKernelGGL(k_tp_chain_bwd<false>, dim3((unsigned)((R + CH_ROWS - 1) / CH_ROWS)), dim3(256), 0, s, a);
        float* z3 = g_pre + 128;
        float* z0 = g_pre;
        dw_gemm(128, 128, (int)R, z3, 256, h2, 128, gw[3], 128 + K0, gb[3], part, s);                             // layer 3 on [h2 | x_enc | . | world]
        dw_gemm(128, pe, (int)R, z3, 256, x_enc, pe, gw[3] + 128, 128 + K0, nullptr, part, s);
        dw_gemm(128, 128, (int)R, z3, 256, world, 128, gw[3] + 128 + pe + 512, 128 + K0, nullptr, part, s);
        if (g_x_enc) gemm<false, true>((int)R, pe, 128, z3, 256, w[3] + 128, 128 + K0, g_x_enc, pe, epi(), 1, s);
        dw_gemm(128, 128, (int)R, ga, 128, h1, 128, gw[2], 128, gb[2], part, s);                                  // layer 2
        dw_gemm(128, 128, (int)R, gb2, 128, h0, 128, gw[1], 128, gb[1], part, s);                                 // layer 1
        dw_gemm(128, pe, (int)R, z0, 256, x_enc, pe, gw[0], K0, gb[0], part, s);                                  // layer 0 on [x_enc | . | world]
        dw_gemm(128, 128, (int)R, z0, 256, world, 128, gw[0] + pe + 512, K0, nullptr, part, s);
        if (g_x_enc) gemm<false, true>((int)R, pe, 128, z0, 256, w[0], K0, g_x_enc, pe, epi(nullptr, 0, 1), 1, s);
        return;
    }
    // mean over views -> per-view rows; view layer 0 on [bott | cond]
    hipLaunchKernelGGL(k_view_bcast, dim3(blocks(P * 64)), dim3(256), 0, s, g_ym, NV, P, 64, 0, gy0, 64L);
    dw_gemm(64, 128, (int)R, gy0, 64, bott, 128, gw[4], 155, gb[4], part, s);
    dw_gemm(64, 27, (int)R, gy0, 64, cond, 27, gw[4] + 128, 155, nullptr, part, s);
    gemm<false, true>((int)R, 128, 64, gy0, 64, w[4], 155, ga, 128, epi(), 1, s);                                // g_bott (R x 128)
    // bottleneck
    dw_gemm(128, 128, (int)R, ga, 128, h3, 128, gw[6], 128, gb[6], part, s);
    // projected-space path: the two first-layer gradients are produced IN PLACE in the halves of g_pre (R, 256) - g_z3 in
    // columns 128.., g_z0 in columns 0..127, row pitch 256 - instead of being copied there (2 x R x 128 floats read + written)
    float* z3 = g_pre ? g_pre + 128 : gb2;
    float* z0 = g_pre ? g_pre : ga;
    const long lz = g_pre ? 256 : 128;
    gemm<false, true>((int)R, 128, 128, ga, 128, w[6], 128, z3, lz, epi(), 1, s);                                // g_h3 from the bottleneck
    // density head on the view mean of h3
    dw_gemm(1, 128, (int)P, g_sigma, 1, hm, 128, gw[7], 128, gb[7], part, s);
    gemm<false, true>((int)P, 128, 1, g_sigma, 1, w[7], 128, g_hm, 128, epi(), 1, s);
    hipLaunchKernelGGL(k_view_bcast, dim3(blocks(P * 128)), dim3(256), 0, s, g_hm, NV, P, 128, 1, z3, lz);       // g_h3 += g_hm / NV
    hipLaunchKernelGGL(k_relu_mask, dim3(blocks(R * 128)), dim3(256), 0, s, z3, lz, h3, 128, R);                 // g_z3
    // layer 3 on [h2 | x0]
    dw_gemm(128, 128, (int)R, z3, lz, h2, 128, gw[3], 128 + K0, gb[3], part, s);
    for (int i = 0; i < 3; ++i) {
        if (g_pre && i == 1) continue;
        dw_gemm(128, kin[i], (int)R, z3, lz, in[i], kin[i], gw[3] + 128 + off[i], 128 + K0, nullptr, part, s);
        if (g_in[i]) gemm<false, true>((int)R, kin[i], 128, z3, lz, w[3] + 128 + off[i], 128 + K0, g_in[i], kin[i], epi(), 1, s);
    }
    gemm<false, true>((int)R, 128, 128, z3, lz, w[3], 128 + K0, ga, 128, epi(nullptr, 0, 0, h2, 128), 1, s);      // g_z2 = (g_z3 W3a) relu'(h2)
    // layer 2
    dw_gemm(128, 128, (int)R, ga, 128, h1, 128, gw[2], 128, gb[2], part, s);
    gemm<false, true>((int)R, 128, 128, ga, 128, w[2], 128, gb2, 128, epi(nullptr, 0, 0, h1, 128), 1, s);         // g_z1
    // layer 1
    dw_gemm(128, 128, (int)R, gb2, 128, h0, 128, gw[1], 128, gb[1], part, s);
    gemm<false, true>((int)R, 128, 128, gb2, 128, w[1], 128, z0, lz, epi(nullptr, 0, 0, h0, 128), 1, s);          // g_z0
    // layer 0
    for (int i = 0; i < 3; ++i) {
        if (g_pre && i == 1) continue;
        dw_gemm(128, kin[i], (int)R, z0, lz, in[i], kin[i], gw[0] + off[i], K0, i == 0 ? gb[0] : nullptr, part, s);
        if (g_in[i]) gemm<false, true>((int)R, kin[i], 128, z0, lz, w[0] + off[i], K0, g_in[i], kin[i], epi(nullptr, 0, 1), 1, s);
    }
}

// ---- vanilla NeRFMLP (vanilla_nerf/model.py:100-125) ---------------------------------------------------------------------
// rows R = rays x samples; x0 (R, 63) encoded points, cond (R, 27) the ray's direction encoding tiled over its samples.
// w / b order as neo_vanilla_upload_mlp: pts_linears.0..7 (256 wide, layer 5 on [h4 | x0]), views_linear.0 (283 -> 128),
// bottleneck (256 -> 256), density (256 -> 1), rgb (128 -> 3).
// tape (floats): h0..h7 (R x 256 each), bott (R x 256), v (R x 128)
size_t vanilla_train_tape_floats(long R) { return (size_t)(R * (9 * 256 + 128)); }
size_t vanilla_train_scratch_floats(long R) { return (size_t)(R * (2 * 256 + 128) + DW_PART_FLOATS); }

void launch_vanilla_train_forward(const float* const* w, const float* const* b, const float* x0, const float* cond, long R,
                                  float* tape, float* raw_rgb, float* raw_sigma, hipStream_t s) {
    float* h[8];
    for (int i = 0; i < 8; ++i) h[i] = tape + (size_t)i * R * 256;
    float* bott = tape + (size_t)8 * R * 256;
    float* v = bott + (size_t)R * 256;
    const int M = (int)R;
    gemm<false, false>(M, 256, 63, x0, 63, w[0], 63, h[0], 256, epi(b[0], 1), 1, s);
    for (int i = 1; i < 8; ++i) {
        if (i == 5) {              // the skip concat after layer index 4: layer 5 reads [h4 | x0]
            gemm<false, false>(M, 256, 256, h[4], 256, w[5], 319, h[5], 256, epi(b[5], 0), 1, s);
            gemm<false, false>(M, 256, 63, x0, 63, w[5] + 256, 319, h[5], 256, epi(nullptr, 1, 1), 1, s);   // C = relu(C + ..)
        } else {
            gemm<false, false>(M, 256, 256, h[i - 1], 256, w[i], 256, h[i], 256, epi(b[i], 1), 1, s);
        }
    }
    gemm<false, false>(M, 1, 256, h[7], 256, w[10], 256, raw_sigma, 1, epi(b[10], 0), 1, s);
    gemm<false, false>(M, 256, 256, h[7], 256, w[9], 256, bott, 256, epi(b[9], 0), 1, s);
    gemm<false, false>(M, 128, 256, bott, 256, w[8], 283, v, 128, epi(b[8], 0), 1, s);
    gemm<false, false>(M, 128, 27, cond, 27, w[8] + 256, 283, v, 128, epi(nullptr, 1, 1), 1, s);                   // C = relu(C + ..)
    gemm<false, false>(M, 3, 128, v, 128, w[11], 128, raw_rgb, 3, epi(b[11], 0), 1, s);
}

// gw / gb: twelve gradients ZEROED by the caller; g_x0 (R, 63) / g_cond (R, 27) may be null
void launch_vanilla_train_backward(const float* const* w, const float* x0, const float* cond, long R, const float* tape,
                                   float* scratch, const float* g_rgb, const float* g_sigma, float* const* gw, float* const* gb,
                                   float* g_x0, float* g_cond, hipStream_t s) {
    const float* h[8];
    for (int i = 0; i < 8; ++i) h[i] = tape + (size_t)i * R * 256;
    const float* bott = tape + (size_t)8 * R * 256;
    const float* v = bott + (size_t)R * 256;
    float* ga = scratch; float* gb2 = ga + (size_t)R * 256; float* gv = gb2 + (size_t)R * 256;
    float* part = gv + (size_t)R * 128;                                              // DW_PART_FLOATS (dw_gemm)
    const int M = (int)R;
    // rgb head, view layer on [bott | cond]
    dw_gemm(3, 128, M, g_rgb, 3, v, 128, gw[11], 128, gb[11], part, s);
    gemm<false, true>(M, 128, 3, g_rgb, 3, w[11], 128, gv, 128, epi(nullptr, 0, 0, v, 128), 1, s);
    dw_gemm(128, 256, M, gv, 128, bott, 256, gw[8], 283, gb[8], part, s);
    dw_gemm(128, 27, M, gv, 128, cond, 27, gw[8] + 256, 283, nullptr, part, s);
    if (g_cond) gemm<false, true>(M, 27, 128, gv, 128, w[8] + 256, 283, g_cond, 27, epi(), 1, s);
    gemm<false, true>(M, 256, 128, gv, 128, w[8], 283, ga, 256, epi(), 1, s);                                     // g_bott
    // bottleneck + density head -> g_h7
    dw_gemm(256, 256, M, ga, 256, h[7], 256, gw[9], 256, gb[9], part, s);
    gemm<false, true>(M, 256, 256, ga, 256, w[9], 256, gb2, 256, epi(), 1, s);
    dw_gemm(1, 256, M, g_sigma, 1, h[7], 256, gw[10], 256, gb[10], part, s);
    gemm<false, true>(M, 256, 1, g_sigma, 1, w[10], 256, gb2, 256, epi(nullptr, 0, 1), 1, s);
    hipLaunchKernelGGL(k_relu_mask, dim3(blocks(R * 256)), dim3(256), 0, s, gb2, 256L, h[7], 256, R);                  // g_z7
    float* cur = gb2;
    float* nxt = ga;
    for (int i = 7; i >= 1; --i) {
        // cur = g_z_i (R x 256)
        if (i == 5) {
            dw_gemm(256, 256, M, cur, 256, h[4], 256, gw[5], 319, gb[5], part, s);
            dw_gemm(256, 63, M, cur, 256, x0, 63, gw[5] + 256, 319, nullptr, part, s);
            if (g_x0) gemm<false, true>(M, 63, 256, cur, 256, w[5] + 256, 319, g_x0, 63, epi(), 1, s);
            gemm<false, true>(M, 256, 256, cur, 256, w[5], 319, nxt, 256, epi(nullptr, 0, 0, h[4], 256), 1, s);
        } else {
            dw_gemm(256, 256, M, cur, 256, h[i - 1], 256, gw[i], 256, gb[i], part, s);
            gemm<false, true>(M, 256, 256, cur, 256, w[i], 256, nxt, 256, epi(nullptr, 0, 0, h[i - 1], 256), 1, s);
        }
        float* t = cur; cur = nxt; nxt = t;
    }
    dw_gemm(256, 63, M, cur, 256, x0, 63, gw[0], 63, gb[0], part, s);
    if (g_x0) gemm<false, true>(M, 63, 256, cur, 256, w[0], 63, g_x0, 63, epi(nullptr, 0, 1), 1, s);
}

// ---- Mip-NeRF 360 MLP (mipnerf360/model.py:107-176) as one chain each way (round 6) --------------------------------------
// rows = R rays x n intervals; x0 (rows, 504) integrated encodings (data: no gradient), d_enc (R, 27) one direction encoding per
// RAY (the reference tiles it over the intervals; here the direction term of the view layer is formed once per ray and broadcast).
// W = netwidth (256 proposal / 1024 NeRF), D = netdepth (4 / 8): layer i > 0 with (i - 1) % 4 == 0 and i - 1 > 0 reads [h | x0].
// w / b order as neo_mip_upload_mlp: pts_linear.0..D-1, density, then (rgb branch) bottleneck (W -> 256), views_linear.0 (283 -> 128), rgb.
// Outputs ACTIVATED as the reference returns them: rgbdens (rows, 4) = [sigmoid(.) (1 + 2 pad) - pad | softplus(raw - 1)]; the backward
// reads them back (softplus' = 1 - exp(-density), sigmoid' from the colour), so no raw values are kept.
// tape (floats): h0..h(D-1) (rows x W each), bott (rows x 256), y (rows x 128, post-ReLU), the rays' direction terms (R x 128)
namespace {
constexpr int MIP_POS = 504;
constexpr float MIP_PAD = 0.001f;
inline bool mip_skip(int i) { return i > 0 && (i - 1) % 4 == 0 && (i - 1) > 0; }

// y[r n + j][c] = relu(y[..][c] + dterm[r][c])
__global__ void k_ray_bcast_add_relu(float* __restrict__ y, const float* __restrict__ dterm, long rows, int n, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const long row = i / C;
    const int c = (int)(i - row * C);
    y[i] = fmaxf(y[i] + dterm[(row / n) * C + c], 0.0f);
}
// out[r][c] = sum_j g[r n + j][c]
__global__ void k_ray_group_sum(const float* __restrict__ g, long R, int n, int C, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * C) return;
    const long r = i / C;
    const int c = (int)(i - r * C);
    float s = 0.0f;
    for (int j = 0; j < n; ++j) s += g[((long)r * n + j) * C + c];
    out[i] = s;
}
// in place on (rows, 4): columns 0..2 raw colour -> sigmoid(x) (1 + 2 pad) - pad (zeros without the branch), column 3 raw density ->
// softplus(x - 1) with torch's threshold (x > 20: x)
__global__ void k_mip_act(float4* __restrict__ v, long rows, int rgb) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    float4 q = v[i];
    const float x = q.w + (-1.0f);
    q.w = x > 20.0f ? x : log1pf(expf(x));
    if (rgb) {
        q.x = (1.0f / (1.0f + expf(-q.x))) * (1.0f + 2.0f * MIP_PAD) - MIP_PAD;
        q.y = (1.0f / (1.0f + expf(-q.y))) * (1.0f + 2.0f * MIP_PAD) - MIP_PAD;
        q.z = (1.0f / (1.0f + expf(-q.z))) * (1.0f + 2.0f * MIP_PAD) - MIP_PAD;
    } else {
        q.x = q.y = q.z = 0.0f;
    }
    v[i] = q;
}
// graw = g x d(activation): from the activated outputs
__global__ void k_mip_act_bwd(const float4* __restrict__ g, const float4* __restrict__ out, long rows, int rgb, float4* __restrict__ graw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const float4 a = g[i], o = out[i];
    float4 r;
    r.w = a.w * (1.0f - expf(-o.w));
    if (rgb) {
        const float k = 1.0f + 2.0f * MIP_PAD;
        const float sx = (o.x + MIP_PAD) / k, sy = (o.y + MIP_PAD) / k, sz = (o.z + MIP_PAD) / k;
        r.x = a.x * k * sx * (1.0f - sx);
        r.y = a.y * k * sy * (1.0f - sy);
        r.z = a.z * k * sz * (1.0f - sz);
    } else {
        r.x = r.y = r.z = 0.0f;
    }
    graw[i] = r;
}
}  // namespace

size_t mip_train_tape_floats(int W, int D, int rgb, long rows, long R) {
    return (size_t)rows * ((size_t)D * W + (rgb ? 256 + 128 : 0)) + (rgb ? (size_t)R * 128 : 0);
}
size_t mip_train_scratch_floats(int W, int rgb, long rows, long R) {
    return (size_t)rows * (2 * (size_t)W + 4 + (rgb ? 256 + 128 : 0)) + (size_t)R * 128 + DW_PART_FLOATS;
}

void launch_mip_train_forward(int W, int D, int rgb, const float* const* w, const float* const* b, const float* x0, const float* d_enc,
                              long R, int n, float* tape, float* rgbdens, hipStream_t s) {
    const long rows = R * n;
    const int M = (int)rows;
    auto h = [&](int i) { return tape + (size_t)i * rows * W; };
    float* bott = tape + (size_t)D * rows * W;
    float* y = bott + (size_t)rows * 256;
    gemm<false, false>(M, W, MIP_POS, x0, MIP_POS, w[0], MIP_POS, h(0), W, epi(b[0], 1), 1, s);
    for (int i = 1; i < D; ++i) {
        if (mip_skip(i)) {         // the layer after the concatenation [h | x0] (model.py:118-123): two products into one sum
            gemm<false, false>(M, W, W, h(i - 1), W, w[i], W + MIP_POS, h(i), W, epi(b[i], 0), 1, s);
            gemm<false, false>(M, W, MIP_POS, x0, MIP_POS, w[i] + W, W + MIP_POS, h(i), W, epi(nullptr, 1, 1), 1, s);
        } else {
            gemm<false, false>(M, W, W, h(i - 1), W, w[i], W, h(i), W, epi(b[i], 1), 1, s);
        }
    }
    gemm<false, false>(M, 1, W, h(D - 1), W, w[D], W, rgbdens + 3, 4, epi(b[D], 0), 1, s);                          // raw density -> column 3
    if (rgb) {
        float* dterm = y + (size_t)rows * 128;                                                                        // (R, 128)
        gemm<false, false>(M, 256, W, h(D - 1), W, w[D + 1], W, bott, 256, epi(b[D + 1], 0), 1, s);
        gemm<false, false>(M, 128, 256, bott, 256, w[D + 2], 283, y, 128, epi(b[D + 2], 0), 1, s);
        gemm<false, false>((int)R, 128, 27, d_enc, 27, w[D + 2] + 256, 283, dterm, 128, epi(), 1, s);                 // once per ray
        hipLaunchKernelGGL(k_ray_bcast_add_relu, dim3(blocks(rows * 128)), dim3(256), 0, s, y, dterm, rows, n, 128);
        gemm<false, false>(M, 3, 128, y, 128, w[D + 3], 128, rgbdens, 4, epi(b[D + 3], 0), 1, s);                     // raw colour -> columns 0..2
    }
    hipLaunchKernelGGL(k_mip_act, dim3(blocks(rows)), dim3(256), 0, s, reinterpret_cast<float4*>(rgbdens), rows, rgb);
}

// gw / gb ZEROED by the caller (D + 1 or D + 4 tensors); g (rows, 4) = dL/d rgbdens; rgbdens = the forward's output
void launch_mip_train_backward(int W, int D, int rgb, const float* const* w, const float* x0, const float* d_enc, long R, int n,
                               const float* tape, float* scratch, const float* rgbdens, const float* g, float* const* gw,
                               float* const* gb, hipStream_t s) {
    const long rows = R * n;
    const int M = (int)rows;
    auto h = [&](int i) { return tape + (size_t)i * rows * W; };
    const float* bott = tape + (size_t)D * rows * W;
    const float* y = bott + (size_t)rows * 256;
    float* cur = scratch; float* nxt = cur + (size_t)rows * W; float* graw = nxt + (size_t)rows * W;
    float* gbott = graw + (size_t)rows * 4; float* gy = gbott + (rgb ? (size_t)rows * 256 : 0);
    float* gdt = gy + (rgb ? (size_t)rows * 128 : 0); float* part = gdt + (size_t)R * 128;
    hipLaunchKernelGGL(k_mip_act_bwd, dim3(blocks(rows)), dim3(256), 0, s, reinterpret_cast<const float4*>(g),
                       reinterpret_cast<const float4*>(rgbdens), rows, rgb, reinterpret_cast<float4*>(graw));
    int acc = 0;
    if (rgb) {
        dw_gemm(3, 128, M, graw, 4, y, 128, gw[D + 3], 128, gb[D + 3], part, s);                                      // rgb head
        gemm<false, true>(M, 128, 3, graw, 4, w[D + 3], 128, gy, 128, epi(nullptr, 0, 0, y, 128), 1, s);              // x relu'(y)
        dw_gemm(128, 256, M, gy, 128, bott, 256, gw[D + 2], 283, gb[D + 2], part, s);                                 // view layer, bottleneck columns
        hipLaunchKernelGGL(k_ray_group_sum, dim3(blocks(R * 128)), dim3(256), 0, s, gy, R, n, 128, gdt);
        dw_gemm(128, 27, (int)R, gdt, 128, d_enc, 27, gw[D + 2] + 256, 283, nullptr, part, s);                         //             direction columns
        gemm<false, true>(M, 256, 128, gy, 128, w[D + 2], 283, gbott, 256, epi(), 1, s);
        dw_gemm(256, W, M, gbott, 256, h(D - 1), W, gw[D + 1], W, gb[D + 1], part, s);                                // bottleneck
        gemm<false, true>(M, W, 256, gbott, 256, w[D + 1], W, cur, W, epi(), 1, s);
        acc = 1;
    }
    dw_gemm(1, W, M, graw + 3, 4, h(D - 1), W, gw[D], W, gb[D], part, s);                                             // density head
    gemm<false, true>(M, W, 1, graw + 3, 4, w[D], W, cur, W, epi(nullptr, 0, acc), 1, s);
    hipLaunchKernelGGL(k_relu_mask, dim3(blocks(rows * W)), dim3(256), 0, s, cur, (long)W, h(D - 1), W, rows);         // g_z(D-1)
    for (int i = D - 1; i >= 1; --i) {
        if (mip_skip(i)) {
            dw_gemm(W, W, M, cur, W, h(i - 1), W, gw[i], W + MIP_POS, gb[i], part, s);
            dw_gemm(W, MIP_POS, M, cur, W, x0, MIP_POS, gw[i] + W, W + MIP_POS, nullptr, part, s);
            gemm<false, true>(M, W, W, cur, W, w[i], W + MIP_POS, nxt, W, epi(nullptr, 0, 0, h(i - 1), W), 1, s);
        } else {
            dw_gemm(W, W, M, cur, W, h(i - 1), W, gw[i], W, gb[i], part, s);
            gemm<false, true>(M, W, W, cur, W, w[i], W, nxt, W, epi(nullptr, 0, 0, h(i - 1), W), 1, s);
        }
        float* t = cur; cur = nxt; nxt = t;
    }
    dw_gemm(W, MIP_POS, M, cur, W, x0, MIP_POS, gw[0], MIP_POS, gb[0], part, s);
}


}  // namespace neo

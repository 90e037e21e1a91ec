// NeRFPPMLP under autograd, the PER-ROW part of the chain as ONE kernel each way (round 6; neo360/model.py:110-158 inside the training
// step :697-820).  Included by train_mlp.hip (inside namespace neo { namespace { ).
//
// The layer-by-layer chain (k_sgemm per layer) moves every activation through HBM twice per layer - a 128 x 128 layer is 32 flop per
// byte, the launches run at ~110 TFLOP/s / 3.4 TB/s, bound by neither.  Here a workgroup owns 64 rows (point-views) through
//     forward : L0 (pre0 + x_enc W0_pe + world W0_w) -> L1 -> L2 -> L3 (pre3 + h2 W3_h + x_enc W3_pe + world W3_w)
//     backward: (g_hm / NV) relu'(h3) = g_z3 -> g_z2 -> g_z1 -> g_z0, g_world
// Everything after relu(L3_v) is linear up to the view means (the bottleneck has no activation, view layer 0 is averaged over the views
// BEFORE its ReLU: model.py:139-150), so the bottleneck and view layer 0 act on the MEANS - P rows instead of NV P, forward, dX and dW
// (train_mlp.hip; the reassociation the inference kernels make, FOLD in mlp_tp.hip): W4a mean_v(W6 h3_v + b6) + W4c mean_v(cond_v) + b4 =
// W4a (W6 hm + b6) + W4c mean_v(cond_v) + b4.  The per-row chain therefore ends at h3.
// with the current activation (gradient) tile in LDS (64 x 128 floats, XOR-swizzled 16-B chunks: mfma_tile.h) and every layer's output
// written to HBM ONCE (the tape / the operands of the weight-gradient GEMMs), never read back by the chain.  Exact fp32
// (v_mfma_f32_32x32x2_f32), weights straight from L2 in row-major order: the forward's fragment is one 16-B load of W[n][k..k+3], the
// backward's the four dwords W[k..k+3][n] (coalesced over n).  Wave w owns output columns 32 w .. 32 w + 31 of a 128-wide layer for both
// 32-row halves (2 accumulator tiles; a weight fragment feeds 2 MFMAs, fragments requested 4 chunks = 32 MFMAs ahead); 48 KB (forward) /
// 32 KB (backward) of LDS, three workgroups per CU.  The view means and everything P-sized stay separate launches (train_mlp.hip).
#pragma once

#ifndef NEO_CHAIN_ABLATE
#define NEO_CHAIN_ABLATE 0       // timing experiments only (wrong results; tools/build_variant.py): 1 no MFMAs (operands still fetched), 2 no tape /
                                 // gradient stores, 4 one weight fragment per segment instead of one per chunk (no L2 weight stream)
#endif
constexpr int CH_ROWS = 64;      // rows per workgroup
constexpr int CH_HLD = 128;      // activation tile pitch (floats)
constexpr int CH_RING = 4;       // weight fragments in flight per wave

struct ChainFwdArgs {
    const float* w0; const float* w1; const float* w2; const float* w3;
    const float* b0; const float* b1; const float* b2; const float* b3;
    const float* x_enc; const float* world; const float* pre;
    float* h0; float* h1; float* h2; float* h3;
    long R;
    int pe;
};

struct ChainBwdArgs {
    const float* w0; const float* w1; const float* w2; const float* w3;
    const float* h0; const float* h1; const float* h2; const float* h3;
    const float* g_hm;       // (P, 128): gradient of the view mean of h3 (density head + bottleneck / view branch): g_h3[r] = g_hm[r mod P] / NV
    float* gz3; long ldz3;   // out: g_z3 (NeRFPPMLP: columns 128.. of g_pre (R, 256), pitch 256)
    float* gz0; long ldz0;   // out: g_z0 (NeRFPPMLP: columns 0..127 of g_pre)
    float* gz2; float* gz1;  // (R, 128) out
    float* g_world;          // (R, 128) out or null
    long R, P;
    int pe, NV;
};

// ---- fragments ------------------------------------------------------------------------------------------------------------------------
// forward: W[n][k0 .. k0 + 3] of this lane's output row n (p points at W[n][segment start + 4 half]); GUARD: columns >= kvalid read as 0
template <bool GUARD>
__device__ __forceinline__ f32x4 ch_wfrag(const float* __restrict__ p, int c, int kvalid, const LaneCtx& L) {
    if (!GUARD) return *reinterpret_cast<const f4u*>(p + 8 * c);
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    const int k = 8 * c + 4 * L.half;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (k + e < kvalid) v[e] = p[8 * c + e];
    return v;
}
// backward: W[n0 + 8 c + 4 half + e][k] for e = 0..3 (p points at W[4 half][this lane's column k]; ld = row pitch)
__device__ __forceinline__ f32x4 ch_wfrag_t(const float* __restrict__ p, long ld, int c) {
    f32x4 v;
    const float* q = p + (long)(8 * c) * ld;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = q[(long)e * ld];
    return v;
}

// One weight segment of a layer as seen by this lane: the fragment ring can be filled (chunks 0 .. CH_RING - 2) BEFORE the barrier that
// publishes the activation tile - the weights do not depend on it - so a layer's first MFMAs do not wait for an L2 round trip.
template <int NCH, bool TRANS, bool GUARD>
struct ChSeg {
    const float* wp;
    long ld;
    int kvalid;
    f32x4 a[CH_RING];
    __device__ __forceinline__ f32x4 frag(int c, const LaneCtx& L) const {
        if ((NEO_CHAIN_ABLATE & 4) && c > 0) return a[0];
        if (TRANS) return ch_wfrag_t(wp, ld, c);
        if (GUARD && c == NCH - 1) return ch_wfrag<true>(wp, c, kvalid, L);        // only the last chunk of a guarded segment can be partial
        return ch_wfrag<false>(wp, c, kvalid, L);
    }
    __device__ __forceinline__ void prefetch(const float* w, long ld_, int kvalid_, const LaneCtx& L) {
        wp = w; ld = ld_; kvalid = kvalid_;
#pragma unroll
        for (int c = 0; c < CH_RING - 1 && c < NCH; ++c) a[c] = frag(c, L);
    }
    // acc[mt] += sum over the NCH chunks of W-fragment x tile rows (both 32-row halves); tile chunk index = tc0 + c
    template <int LD>
    __device__ __forceinline__ void run(f32x16 (&acc)[2], const float* __restrict__ tile, int tc0, const LaneCtx& Lin) {
        // the swizzled LDS addresses are re-derived from an opaque lane id per segment: hoisted out of the whole kernel they cost
        // ~60 registers (one per chunk and tile) and spill (the same measure as in mlp_tp.hip)
        LaneCtx L = Lin;
        asm volatile("" : "+v"(L.l31), "+v"(L.key), "+v"(L.half));
        // activation fragments one chunk ahead of the MFMAs that use them (the LDS round trip runs under the previous chunk's products)
        f32x4 b[2][2];
        b[0][0] = load_b<LD, 15>(tile, 0, tc0, L);
        b[0][1] = load_b<LD, 15>(tile, 1, tc0, L);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + CH_RING - 1 < NCH) a[(c + CH_RING - 1) % CH_RING] = frag(c + CH_RING - 1, L);
            if (c + 1 < NCH) {
                b[(c + 1) & 1][0] = load_b<LD, 15>(tile, 0, tc0 + c + 1, L);
                b[(c + 1) & 1][1] = load_b<LD, 15>(tile, 1, tc0 + c + 1, L);
            }
            __builtin_amdgcn_sched_barrier(0);            // the ring stays 4 deep: no hoisting of later chunks' loads over these MFMAs
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (NEO_CHAIN_ABLATE & 1) { asm volatile("" ::"v"(a[c % CH_RING][e]), "v"(b[c & 1][0][e]), "v"(b[c & 1][1][e])); continue; }
                acc[0] = NEO_MFMA(a[c % CH_RING][e], b[c & 1][0][e], acc[0]);
                acc[1] = NEO_MFMA(a[c % CH_RING][e], b[c & 1][1][e], acc[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};
template <int LD, int NCH, bool TRANS, bool GUARD>
__device__ __forceinline__ void ch_gemm(f32x16 (&acc)[2], const float* __restrict__ wp, long ld, int kvalid, const float* __restrict__ tile,
                                        int tc0, const LaneCtx& L) {
    ChSeg<NCH, TRANS, GUARD> sg;
    sg.prefetch(wp, ld, kvalid, L);
    sg.template run<LD>(acc, tile, tc0, L);
}
// one accumulator tile (the 64-wide view layers of the heads kernel): rows of half mt
template <int LD, int NCH, bool GUARD>
__device__ __forceinline__ void ch_gemm1(f32x16& acc, const float* __restrict__ wp, int kvalid, const float* __restrict__ tile, int mt,
                                         const LaneCtx& L) {
    f32x4 a[CH_RING];
    auto frag = [&](int c) -> f32x4 {
        if (GUARD && c == NCH - 1) return ch_wfrag<true>(wp, c, kvalid, L);
        return ch_wfrag<false>(wp, c, kvalid, L);
    };
#pragma unroll
    for (int c = 0; c < CH_RING - 1 && c < NCH; ++c) a[c] = frag(c);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + CH_RING - 1 < NCH) a[(c + CH_RING - 1) % CH_RING] = frag(c + CH_RING - 1);
        const f32x4 b = load_b<LD, 15>(tile, mt, c, L);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = NEO_MFMA(a[c % CH_RING][e], b[e], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- D-layout global I/O: acc[4 g + e] <-> M[row][col0 + 8 g + 4 half + e] (row = this lane's row; 4 instructions cover one 128-B line) ----
// (bias indexed by the layer's output column n0 + .., the matrix by col0 + ..: the skip layer's half of `pre` starts at column 128)
__device__ __forceinline__ void ch_d_init(f32x16& acc, const float* __restrict__ M, long ld, long row, bool valid, int col0,
                                          const float* __restrict__ bias, int n0, float scale, const LaneCtx& L) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = col0 + 8 * g + 4 * L.half;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f}, b = {0.0f, 0.0f, 0.0f, 0.0f};
        if (M != nullptr && valid) v = *reinterpret_cast<const f4u*>(M + row * ld + col);
        if (bias != nullptr) b = *reinterpret_cast<const f4u*>(bias + n0 + 8 * g + 4 * L.half);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = v[e] / scale + b[e];
    }
}
template <bool RELU>
__device__ __forceinline__ void ch_d_store(const f32x16& acc, float* __restrict__ M, long ld, long row, bool valid, int col0, const LaneCtx& L) {
    if (!valid) return;
    if (NEO_CHAIN_ABLATE & 2) { if (acc[0] != 12345.678f) return; }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = RELU ? fmaxf(acc[4 * g + e], 0.0f) : acc[4 * g + e];
        *reinterpret_cast<f4u*>(M + row * ld + col0 + 8 * g + 4 * L.half) = v;
    }
}
// ReLU backward against the taped activation: the mask tile (D layout) is requested BEFORE the layer's MFMAs and applied after them
struct ChMask { f32x4 m[4]; };
__device__ __forceinline__ void ch_mask_load(ChMask& k, const float* __restrict__ Hm, long row, bool valid, int col0, const LaneCtx& L) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        k.m[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (valid) k.m[g] = *reinterpret_cast<const f4u*>(Hm + row * 128 + col0 + 8 * g + 4 * L.half);
    }
}
__device__ __forceinline__ void ch_mask_apply(f32x16& acc, const ChMask& k) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (!(k.m[g][e] > 0.0f)) acc[4 * g + e] = 0.0f;
}

// ---- LDS staging --------------------------------------------------------------------------------------------------------------------
// 64 rows x 128 floats (row pitch `ld` floats in global memory, 16-B aligned rows) -> swizzled tile, by LDS-DMA: one
// global_load_lds_dwordx4 per wave moves two rows (1 KB) straight into LDS - no staging registers, no ds_write pass.  The DMA writes
// lane-linearly (wave-uniform base + 16 lane), so the XOR swizzle is applied to the SOURCE: slot p of a row receives logical chunk
// p ^ (row & 15), the involution the fragment reads (load_b) apply.  Rows past the end are clamped to the last row (their results are
// never stored).  The barrier that publishes the tile carries the vmcnt(0) (the compiler knows the DMA is in flight).
__device__ __forceinline__ void ch_stage128(float* __restrict__ tile, const float* __restrict__ M, long ld, long r0, long R) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row_first = 2 * wv + 8 * j;
        const int row = row_first + (lane >> 5);
        long gr = r0 + row;
        gr = gr < R ? gr : R - 1;
        const float* src = M + gr * ld + (((lane & 31) ^ (row & 15)) << 2);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tile + row_first * CH_HLD), 16, 0, 0);
    }
}
// 64 rows x `cols` floats (dense rows of pitch `cols`, any alignment) -> swizzled tile of pitch LD, zero-padded to 8 NCH columns
template <int LD, int NCH>
__device__ __forceinline__ void ch_stage_narrow(float* __restrict__ tile, const float* __restrict__ M, int cols, long r0, long R) {
    constexpr int XW = 8 * NCH;
    for (int idx = threadIdx.x; idx < CH_ROWS * XW; idx += 256) {
        const int row = idx / XW, col = idx - row * XW;
        float v = 0.0f;
        if (col < cols && r0 + row < R) v = M[(r0 + row) * cols + col];
        tile[swz_index<LD, 15>(row, col)] = v;
    }
}

// ---- forward --------------------------------------------------------------------------------------------------------------------------
// PEC: 8-column chunks of the encoding (8: pe = 63, 11: pe = 84); XLD: pitch of the narrow tile (power-of-two chunk count >= 16)
// PIX: PixelNeRF's MLP (vanilla_nerf/model_pixel.py:96-131; 4 x 128, the skip never fires): the same chain without the world features and
// without the skip - h0 = relu(pre + x_enc W0_pe + b0) with pre (R, 128) and W0 (128, pe + 512), layers 1..3 plain.
template <int PEC, bool PIX>
__global__ __launch_bounds__(256, 3) void k_tp_chain_fwd(ChainFwdArgs a) {
    constexpr int XLD = PEC <= 8 ? 64 : 128;
    __shared__ __attribute__((aligned(16))) float H[CH_ROWS * CH_HLD];
    __shared__ __attribute__((aligned(16))) float X[CH_ROWS * XLD];
    LaneCtx L;
    L.init();
    const long r0 = (long)blockIdx.x * CH_ROWS;
    const int pe = a.pe, K0 = PIX ? pe + 512 : pe + 640, K3 = PIX ? 128 : 128 + K0;
    const int n0 = 32 * L.wv;                              // this wave's output columns of a 128-wide layer
    const long row[2] = {r0 + L.l31, r0 + 32 + L.l31};
    const bool ok[2] = {row[0] < a.R, row[1] < a.R};
    const int wrow = n0 + L.l31;                           // the weight row this lane's fragments come from
    const int ko = 4 * L.half;

    ch_stage_narrow<XLD, PEC>(X, a.x_enc, pe, r0, a.R);
    if (!PIX) ch_stage128(H, a.world, 128, r0, a.R);
    __builtin_amdgcn_sched_barrier(0);                     // the staging registers are dead before the 64 accumulator registers fill
    f32x16 acc[2], acc3[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        ch_d_init(acc[mt], a.pre, PIX ? 128 : 256, row[mt], ok[mt], n0, a.b0, n0, 1.0f, L);
        if (!PIX) ch_d_init(acc3[mt], a.pre, 256, row[mt], ok[mt], 128 + n0, a.b3, n0, 1.0f, L);
    }
    __syncthreads();
    // layer 0 and the input segments of layer 3 while x_enc / world are in LDS
    ch_gemm<XLD, PEC, false, true>(acc, a.w0 + (long)wrow * K0 + ko, 0, pe, X, 0, L);
    if (!PIX) {
        ch_gemm<CH_HLD, 16, false, false>(acc, a.w0 + (long)wrow * K0 + pe + 512 + ko, 0, 128, H, 0, L);
        ch_gemm<XLD, PEC, false, true>(acc3, a.w3 + (long)wrow * K3 + 128 + ko, 0, pe, X, 0, L);
        ch_gemm<CH_HLD, 16, false, false>(acc3, a.w3 + (long)wrow * K3 + 128 + pe + 512 + ko, 0, 128, H, 0, L);
        __syncthreads();
    }
    float* const tape[3] = {a.h0, a.h1, a.h2};
    const float* const wl[3] = {a.w1 + (long)wrow * 128 + ko, a.w2 + (long)wrow * 128 + ko, a.w3 + (long)wrow * K3 + ko};   // layer 1, 2, h2 segment of 3
    const float* const bl[3] = {a.b1, a.b2, a.b3};
    ChSeg<16, false, false> sg;
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
        // acc = pre-activation of layer `layer`: ReLU -> tape + LDS, then the next 128 x 128 layer (its first weight fragments
        // requested before the barrier)
        sg.prefetch(wl[layer], 0, 128, L);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            store_tile<CH_HLD, 15, true>(acc[mt], H, L.wv, mt, L);
            ch_d_store<true>(acc[mt], tape[layer], 128, row[mt], ok[mt], n0, L);
        }
        __syncthreads();
        if (layer < 2 || PIX) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) ch_d_init(acc[mt], nullptr, 0, 0, false, n0, bl[layer], n0, 1.0f, L);
            sg.template run<CH_HLD>(acc, H, 0, L);
        } else {
            sg.template run<CH_HLD>(acc3, H, 0, L);
        }
        __syncthreads();
    }
    // h3: the last per-row layer (the view means and everything behind them are P-sized: train_mlp.hip)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) ch_d_store<true>(PIX ? acc[mt] : acc3[mt], a.h3, 128, row[mt], ok[mt], n0, L);
}

// ---- backward (input-gradient chain) -------------------------------------------------------------------------------------------------
template <bool PIX>
__global__ __launch_bounds__(256, 3) void k_tp_chain_bwd(ChainBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float H[CH_ROWS * CH_HLD];
    LaneCtx L;
    L.init();
    const long r0 = (long)blockIdx.x * CH_ROWS;
    const int pe = a.pe, K0 = PIX ? pe + 512 : pe + 640, K3 = PIX ? 128 : 128 + K0;
    const int n0 = 32 * L.wv;
    const long row[2] = {r0 + L.l31, r0 + 32 + L.l31};
    const bool ok[2] = {row[0] < a.R, row[1] < a.R};
    const int col = n0 + L.l31;                            // the output column (input feature of the layer) of this lane's weight fragments
    const int ko = 4 * L.half;
    f32x16 acc[2], accw[2];
    ChSeg<16, true, false> sg;
    ChMask mk[2];
    sg.prefetch(a.w3 + (long)ko * K3 + col, K3, 0, L);
    // g_z3 = (g_hm / NV) relu'(h3)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        ch_d_init(accw[mt], nullptr, 0, 0, false, n0, nullptr, n0, 1.0f, L);
        ch_d_init(acc[mt], a.g_hm, 128, row[mt] % a.P, ok[mt], n0, nullptr, n0, (float)a.NV, L);
        ch_mask_load(mk[mt], a.h3, row[mt], ok[mt], n0, L);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        ch_mask_apply(acc[mt], mk[mt]);
        store_tile<CH_HLD, 15, false>(acc[mt], H, L.wv, mt, L);
        ch_d_store<false>(acc[mt], a.gz3, a.ldz3, row[mt], ok[mt], n0, L);
    }
    __syncthreads();
    // g_z2 = (g_z3 W3[:, :128]) relu'(h2);  g_world = g_z3 W3_w (+ g_z0 W0_w below)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        ch_d_init(acc[mt], nullptr, 0, 0, false, n0, nullptr, n0, 1.0f, L);
        ch_mask_load(mk[mt], a.h2, row[mt], ok[mt], n0, L);
    }
    sg.template run<CH_HLD>(acc, H, 0, L);
    if (!PIX && a.g_world != nullptr) ch_gemm<CH_HLD, 16, true, false>(accw, a.w3 + (long)ko * K3 + 128 + pe + 512 + col, K3, 0, H, 0, L);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) ch_mask_apply(acc[mt], mk[mt]);
    sg.prefetch(a.w2 + (long)ko * 128 + col, 128, 0, L);
    __syncthreads();
    const float* const wnext[3] = {a.w1 + (long)ko * 128 + col, a.w0 + (long)ko * K0 + pe + 512 + col, nullptr};
    const long ldnext[3] = {128, K0, 0};
    const float* const ml[2] = {a.h1, a.h0};
    float* const outp[3] = {a.gz2, a.gz1, a.gz0};
    const long outld[3] = {128, 128, a.ldz0};
#pragma unroll
    for (int step = 0; step < 3; ++step) {
        // acc = g_z(2 - step): -> LDS + HBM, then through the layer below (sg holds that layer's first weight fragments)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            store_tile<CH_HLD, 15, false>(acc[mt], H, L.wv, mt, L);
            ch_d_store<false>(acc[mt], outp[step], outld[step], row[mt], ok[mt], n0, L);
        }
        __syncthreads();
        if (step < 2) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                ch_d_init(acc[mt], nullptr, 0, 0, false, n0, nullptr, n0, 1.0f, L);
                ch_mask_load(mk[mt], ml[step], row[mt], ok[mt], n0, L);
            }
            sg.template run<CH_HLD>(acc, H, 0, L);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) ch_mask_apply(acc[mt], mk[mt]);
            if (step == 0 || (!PIX && a.g_world != nullptr)) sg.prefetch(wnext[step], ldnext[step], 0, L);
            __syncthreads();
        } else if (!PIX && a.g_world != nullptr) {
            sg.template run<CH_HLD>(accw, H, 0, L);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) ch_d_store<false>(accw[mt], a.g_world, 128, row[mt], ok[mt], n0, L);
        }
    }
}

// ---- the P-sized forward tail of the NeRFPPMLP chain as ONE kernel (round 6) ------------------------------------------------------------
// Behind the per-row layers everything acts on the view means (see the top of this file): per 64 points
//     hm = mean_v h3_v, cm = mean_v cond_v -> sigma = hm w7 + b7, bm = hm W6^T + b6, ym = relu(bm W4a^T + cm W4c^T + b4),
//     y1 = relu(ym W5^T + b5), rgb = y1 W8^T + b8
// with hm / bm in the 64 x 128 tile and cm / ym / y1 in the 64 x 64 tile; hm, bm, cm, ym, y1 go to the tape (the backward's operands).  It
// replaces two view-mean launches and six small GEMMs whose time was launches and reading the same P rows again and again; what it
// has to move is h3 once (NV x 512 B per point).  The two heads with 1 / 3 outputs are plain dot products (4 lanes per point).
struct HeadsFwdArgs {
    const float* h3; const float* cond;                       // (NV P, 128), (NV P, 27), view-major rows
    const float* w4; const float* b4; const float* w5; const float* b5; const float* w6; const float* b6;
    const float* w7; const float* b7; const float* w8; const float* b8;
    float* hm; float* bm; float* cm; float* ym; float* y1; float* raw_sigma; float* raw_rgb;
    long P;
    int NV;
};

__global__ __launch_bounds__(256, 3) void k_tp_heads_fwd(HeadsFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float H[CH_ROWS * CH_HLD];
    __shared__ __attribute__((aligned(16))) float X[CH_ROWS * 64];
    LaneCtx L;
    L.init();
    const int tid = threadIdx.x;
    const long p0 = (long)blockIdx.x * CH_ROWS;
    const int n0 = 32 * L.wv;
    const long prow[2] = {p0 + L.l31, p0 + 32 + L.l31};
    const bool ok[2] = {prow[0] < a.P, prow[1] < a.P};
    const int ko = 4 * L.half;
    // view means: the summation order of k_view_mean (v = 0, 1, ..), one division
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
        if (ok[mt])
            for (int v = 0; v < a.NV; ++v) {
                const float* src = a.h3 + ((long)v * a.P + prow[mt]) * 128 + n0 + ko;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 t = *reinterpret_cast<const f4u*>(src + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mt][4 * g + e] += t[e];
                }
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = acc[mt][r] / (float)a.NV;
        store_tile<CH_HLD, 15, false>(acc[mt], H, L.wv, mt, L);
        ch_d_store<false>(acc[mt], a.hm, 128, prow[mt], ok[mt], n0, L);
    }
    for (int idx = tid; idx < CH_ROWS * 32; idx += 256) {
        const int p = idx >> 5, c = idx & 31;
        float sacc = 0.0f;
        const bool valid = c < 27 && p0 + p < a.P;
        if (valid) {
            for (int v = 0; v < a.NV; ++v) sacc += a.cond[((long)v * a.P + p0 + p) * 27 + c];
            sacc = sacc / (float)a.NV;
            a.cm[(p0 + p) * 27 + c] = sacc;
        }
        X[swz_index<64, 15>(p, c)] = sacc;
    }
    __syncthreads();
    // density head: 4 lanes per point, 32 features each
    {
        const int p = tid >> 2, q = tid & 3;
        float d = 0.0f;
#pragma unroll 8
        for (int k = 32 * q; k < 32 * q + 32; ++k) d = __builtin_fmaf(H[swz_index<CH_HLD, 15>(p, k)], a.w7[k], d);
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        if (q == 0 && p0 + p < a.P) a.raw_sigma[p0 + p] = d + a.b7[0];
    }
    // mean bottleneck
    f32x16 bmv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) ch_d_init(bmv[mt], nullptr, 0, 0, false, n0, a.b6, n0, 1.0f, L);
    ch_gemm<CH_HLD, 16, false, false>(bmv, a.w6 + (long)(n0 + L.l31) * 128 + ko, 0, 128, H, 0, L);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        store_tile<CH_HLD, 15, false>(bmv[mt], H, L.wv, mt, L);
        ch_d_store<false>(bmv[mt], a.bm, 128, prow[mt], ok[mt], n0, L);
    }
    __syncthreads();
    // view layer 0 on [bm | cm], view layer 1: wave = (n-tile wv >> 1, m-tile wv & 1) of a 64-wide layer
    const int vnt = L.wv >> 1, vmt = L.wv & 1;
    const long vrow = p0 + 32 * vmt + L.l31;
    const bool vok = vrow < a.P;
    f32x16 y;
    ch_d_init(y, nullptr, 0, 0, false, 32 * vnt, a.b4, 32 * vnt, 1.0f, L);
    const float* w4r = a.w4 + (long)(32 * vnt + L.l31) * 155 + ko;
    ch_gemm1<CH_HLD, 16, false>(y, w4r, 128, H, vmt, L);
    ch_gemm1<64, 4, true>(y, w4r + 128, 27, X, vmt, L);
    ch_d_store<true>(y, a.ym, 64, vrow, vok, 32 * vnt, L);
    __syncthreads();                                       // every wave has read cm
    store_tile<64, 15, true>(y, X, vnt, vmt, L);
    __syncthreads();
    ch_d_init(y, nullptr, 0, 0, false, 32 * vnt, a.b5, 32 * vnt, 1.0f, L);
    ch_gemm1<64, 8, false>(y, a.w5 + (long)(32 * vnt + L.l31) * 64 + ko, 64, X, vmt, L);
    ch_d_store<true>(y, a.y1, 64, vrow, vok, 32 * vnt, L);
    __syncthreads();                                       // every wave has read ym
    store_tile<64, 15, true>(y, X, vnt, vmt, L);
    __syncthreads();
    // rgb head: 4 lanes per point, 16 features each, 3 outputs
    {
        const int p = tid >> 2, q = tid & 3;
        float d[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll 4
        for (int k = 16 * q; k < 16 * q + 16; ++k) {
            const float x = X[swz_index<64, 15>(p, k)];
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = __builtin_fmaf(x, a.w8[c * 64 + k], d[c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d[c] += __shfl_xor(d[c], 1);
            d[c] += __shfl_xor(d[c], 2);
        }
        if (q == 0 && p0 + p < a.P)
#pragma unroll
            for (int c = 0; c < 3; ++c) a.raw_rgb[(p0 + p) * 3 + c] = d[c] + a.b8[c];
    }
}

// ---- the input-gradient part of the P-sized backward as ONE kernel: g_rgb, g_sigma -> g_y1, g_ym, g_bm, g_hm ------------------------------
//     g_y1 = (g_rgb W8) relu'(y1), g_ym = (g_y1 W5) relu'(ym), g_bm = g_ym W4a, g_hm = g_bm W6 + g_sigma w7
// (five small GEMM launches before); the four results are the operands of the weight-gradient launches and of k_tp_chain_bwd.
struct HeadsBwdArgs {
    const float* g_rgb; const float* g_sigma;               // (P, 3), (P, 1)
    const float* y1; const float* ym;                       // (P, 64) post-ReLU (masks)
    const float* w4; const float* w5; const float* w6; const float* w7; const float* w8;
    float* g_y1; float* g_ym; float* g_bm; float* g_hm;     // (P, 64), (P, 64), (P, 128), (P, 128)
    long P;
};

__global__ __launch_bounds__(256, 3) void k_tp_heads_bwd(HeadsBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float H[CH_ROWS * CH_HLD];
    __shared__ __attribute__((aligned(16))) float X[CH_ROWS * 64];
    LaneCtx L;
    L.init();
    const int tid = threadIdx.x;
    const long p0 = (long)blockIdx.x * CH_ROWS;
    const int n0 = 32 * L.wv;
    const long prow[2] = {p0 + L.l31, p0 + 32 + L.l31};
    const bool ok[2] = {prow[0] < a.P, prow[1] < a.P};
    const int ko = 4 * L.half;
    // g_y1: 3-term sums, one (point, 4 features) piece per thread and pass
    for (int idx = tid; idx < CH_ROWS * 16; idx += 256) {
        const int p = idx >> 4, c4 = idx & 15;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (p0 + p < a.P) {
            const float g0 = a.g_rgb[(p0 + p) * 3], g1 = a.g_rgb[(p0 + p) * 3 + 1], g2 = a.g_rgb[(p0 + p) * 3 + 2];
            const f32x4 m = *reinterpret_cast<const f4u*>(a.y1 + (p0 + p) * 64 + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * c4 + e;
                const float x = __builtin_fmaf(g2, a.w8[128 + k], __builtin_fmaf(g1, a.w8[64 + k], g0 * a.w8[k]));
                v[e] = m[e] > 0.0f ? x : 0.0f;
            }
            *reinterpret_cast<f4u*>(a.g_y1 + (p0 + p) * 64 + 4 * c4) = v;
        }
        *reinterpret_cast<f32x4*>(X + p * 64 + ((c4 ^ (p & 15)) << 2)) = v;
    }
    __syncthreads();
    // g_ym = (g_y1 W5) relu'(ym): a 64-wide layer, wave = (n-tile wv >> 1, m-tile wv & 1), transposed fragments of W5 (64 x 64)
    const int vnt = L.wv >> 1, vmt = L.wv & 1;
    const long vrow = p0 + 32 * vmt + L.l31;
    const bool vok = vrow < a.P;
    f32x16 y;
    ch_d_init(y, nullptr, 0, 0, false, 32 * vnt, nullptr, 32 * vnt, 1.0f, L);
    {
        const float* wp = a.w5 + (long)ko * 64 + 32 * vnt + L.l31;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 w = ch_wfrag_t(wp, 64, c);
            const f32x4 b = load_b<64, 15>(X, vmt, c, L);
#pragma unroll
            for (int e = 0; e < 4; ++e) y = NEO_MFMA(w[e], b[e], y);
        }
    }
    {
        ChMask mk;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mk.m[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (vok) mk.m[g] = *reinterpret_cast<const f4u*>(a.ym + vrow * 64 + 32 * vnt + 8 * g + ko);
        }
        ch_mask_apply(y, mk);
    }
    ch_d_store<false>(y, a.g_ym, 64, vrow, vok, 32 * vnt, L);
    __syncthreads();                                       // every wave has read g_y1
    store_tile<64, 15, false>(y, X, vnt, vmt, L);
    __syncthreads();
    // g_bm = g_ym W4[:, :128]
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) ch_d_init(acc[mt], nullptr, 0, 0, false, n0, nullptr, n0, 1.0f, L);
    ch_gemm<64, 8, true, false>(acc, a.w4 + (long)ko * 155 + n0 + L.l31, 155, 0, X, 0, L);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        store_tile<CH_HLD, 15, false>(acc[mt], H, L.wv, mt, L);
        ch_d_store<false>(acc[mt], a.g_bm, 128, prow[mt], ok[mt], n0, L);
    }
    __syncthreads();
    // g_hm = g_bm W6 + g_sigma w7
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const float gs = ok[mt] ? a.g_sigma[prow[mt]] : 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w = *reinterpret_cast<const f4u*>(a.w7 + n0 + 8 * g + ko);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mt][4 * g + e] = gs * w[e];
        }
    }
    ch_gemm<CH_HLD, 16, true, false>(acc, a.w6 + (long)ko * 128 + n0 + L.l31, 128, 0, H, 0, L);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) ch_d_store<false>(acc[mt], a.g_hm, 128, prow[mt], ok[mt], n0, L);
}

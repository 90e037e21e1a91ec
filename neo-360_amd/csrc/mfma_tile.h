// Shared fp32-MFMA tile helpers (gfx950, v_mfma_f32_32x32x2_f32).
//
// GEMM orientation used by every MLP kernel here:
//     D[n][m] (32 outputs x 32 rows)  +=  A[n][k] (weights)  *  B[k][m] (activations^T)
// A fragments come straight from global memory in a pre-packed order
//     wpack[(n_tile*KC + k_chunk)*64 + lane] = float4 of W[n_tile*32 + (lane&31)][k_chunk*8 + 4*(lane>>5) + 0..3]
// so one wave-wide 16-B load (1 KiB contiguous) feeds 4 MFMAs per accumulator tile;
// MFMA e of a chunk contracts k = {8c+e, 8c+4+e}.
// B fragments are ds_read_b128 from an LDS tile [row][LD floats] whose 16-B chunks are
// XOR-swizzled with (row & KM): conflict-free for the reads and for the epilogue's
// ds_write_b128 (D puts 4 consecutive outputs of one row in 4 consecutive registers).
#pragma once
#include "common.h"

namespace neo {

#define NEO_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct LaneCtx {
    int lane, wv, half, l31, key;
    // split-fp16 kernels: running max |x| over everything this thread split into hi/lo fp16 planes (the range guard,
    // see split_tile.h:range_commit); unused by the fp32-MFMA kernels
    mutable float amax;
    __device__ __forceinline__ void init() {
        amax = 0.0f;
        lane = threadIdx.x & 63;
        wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        half = lane >> 5;
        l31 = lane & 31;
        key = lane & 15;
    }
};

__device__ __forceinline__ f32x4 load_a(const f32x4* __restrict__ wp, int KC, int nt, int kc, int lane) {
    return wp[(nt * KC + kc) * 64 + lane];
}

// B fragment of k-chunk c for M-tile mt.
template <int LD, int KM>
__device__ __forceinline__ f32x4 load_b(const float* __restrict__ tile, int mt, int c, const LaneCtx& L) {
    return *reinterpret_cast<const f32x4*>(tile + (mt * 32 + L.l31) * LD + ((((c << 1) + L.half) ^ (L.key & KM)) << 2));
}

// 16 bias values in D layout for N-tile nt: outputs nt*32 + 8g + 4*half + e.
__device__ __forceinline__ void bias_tile(f32x16& acc, const float* __restrict__ bias, int nt, const LaneCtx& L) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + nt * 32 + 8 * g + 4 * L.half);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = b[e];
    }
}

// Store one D tile (N-tile nt, M-tile mt) into a swizzled activation tile.
template <int LD, int KM, bool RELU>
__device__ __forceinline__ void store_tile(const f32x16& acc, float* __restrict__ tile, int nt, int mt,
                                           const LaneCtx& L) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = acc[4 * g + e];
            v[e] = RELU ? fmaxf(x, 0.0f) : x;
        }
        const int chunk = nt * 8 + 2 * g + L.half;
        *reinterpret_cast<f32x4*>(tile + (mt * 32 + L.l31) * LD + ((chunk ^ (L.key & KM)) << 2)) = v;
    }
}

// Address (in floats) of feature f of row p in a swizzled tile.
template <int LD, int KM>
__device__ __forceinline__ int swz_index(int p, int f) {
    return p * LD + ((((f >> 2) ^ (p & KM))) << 2) + (f & 3);
}

}  // namespace neo

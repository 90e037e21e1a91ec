// Shared pieces of the split-fp16 point evaluators (mlp_tp_h.hip, mlp_mip_h.hip): fp32 values live in LDS as two
// fp16 planes (hi = fp16(x), lo = fp16(x - hi)); a product a*b is evaluated on the fp16 matrix cores as
// a_lo b_hi + a_hi b_lo + a_hi b_hi with fp32 accumulation (numerics: mlp_vanilla_h.hip).
#pragma once
#include <hip/hip_fp16.h>

#include "kernels.h"
#include "mfma_tile.h"

namespace neo {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define NEO_MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct HT {   // a swizzled fp16 hi/lo tile
    _Float16* hi;
    _Float16* lo;
};

// 16-B chunk (8 halves) `chunk` of row `row` in a tile with LDH halves per row.  The XOR key uses the
// row bits that do NOT already select the 256-B bank row, so 16 consecutive rows hit 16 different slots.
template <int LDH>
__device__ __forceinline__ int chunk_off(int row, int chunk) {
    constexpr int KEY_SHIFT = LDH >= 128 ? 0 : LDH == 64 ? 1 : 2;
    constexpr int KEY_MASK = LDH / 8 - 1;
    return row * LDH + ((chunk ^ ((row >> KEY_SHIFT) & KEY_MASK)) << 3);
}

__device__ __forceinline__ void split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)__builtin_fmaf((float)hi, -1.0f, x);      // x - hi, exact; one mixed-precision fma
}

__device__ __forceinline__ void split4(const f32x4 v, h4& vh, h4& vl) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        _Float16 h, l;
        split(v[e], h, l);
        vh[e] = h;
        vl[e] = l;
    }
}

// D tile (N-tile nt, M-tile mt) -> (ReLU) -> split -> the two planes of an activation tile with LDH halves per row
template <bool RELU, int LDH = 128>
__device__ __forceinline__ void store_tile_h(const f32x16& acc, const HT& act, int nt, int mt, const LaneCtx& L) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = acc[4 * g + e];
            v[e] = RELU ? fmaxf(x, 0.0f) : x;
        }
        h4 vh, vl;
        split4(v, vh, vl);
        const int o = chunk_off<LDH>(mt * 32 + L.l31, nt * 4 + g) + 4 * L.half;
        *reinterpret_cast<h4*>(act.hi + o) = vh;
        *reinterpret_cast<h4*>(act.lo + o) = vl;
    }
}


// weight re-packing into split fragment order (pack_h.hip): rows [0, rows) of src -> N-tiles [nt0, ...) of a stage
// with KS 16-deep k-steps; h8 index ((nt * KS + ks) * 2 + {hi 0, lo 1}) * 64 + lane; packed k -> source column
// through up to three segments, zero elsewhere
void pack_h(const float* src, int ld, int rows, int KS, int nt0, PackSegs sg, _Float16* dst, hipStream_t s);

}  // namespace neo

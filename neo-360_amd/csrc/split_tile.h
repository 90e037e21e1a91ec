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
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define NEO_MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
// the two cross terms of a split product (a_lo·b_hi, a_hi·b_lo).  NEO_SPLIT_TERMS < 3 drops them: WRONG results, a timing /
// energy experiment only (what a launch costs per matrix instruction); the build never sets it.
#ifndef NEO_SPLIT_TERMS
#define NEO_SPLIT_TERMS 3
#endif
#define NEO_MFMA_H_LH(a, b, c) (NEO_SPLIT_TERMS >= 3 ? NEO_MFMA_H(a, b, c) : (c))
#define NEO_MFMA_H_HL(a, b, c) (NEO_SPLIT_TERMS >= 2 ? NEO_MFMA_H(a, b, c) : (c))

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
    // x is made opaque first.  When x is the direct result of a multiply the compiler folds (float)(_Float16)(a * b) into
    // v_fma_mixlo_f16 - ONE rounding, from the exact product - for the subtraction below, but stores fp16(fl32(a * b)) as
    // hi: two roundings of "the same" value that differ by an ulp of fp16 in rare cases, so hi + lo misses x by 2^-11
    // instead of 2^-22 (measured: 1e-5 outliers in rgb when the view mean became sum * (1 / nv); tests/test_build_cpu.py
    // keeps that instruction out of the library).
    asm("" : "+v"(x));
    hi = (_Float16)x;
    lo = (_Float16)__builtin_fmaf((float)hi, -1.0f, x);      // x - hi, exact; one mixed-precision fma
}

// ---- range guard -------------------------------------------------------------------------------------------
// hi = fp16(x) overflows to inf for |x| >= 65520 and lo = x - hi then poisons the products; ReLU would turn the
// resulting NaN back into a plausible 0.  Every value a split kernel turns into hi/lo planes is therefore folded into
// a per-thread running max (v_max3_f32 with |.| modifiers: half an instruction per element), and a thread that saw
// |x| >= 65504 (or inf / NaN accumulators) raises NEO_FLAG_SPLIT_RANGE once, at the end of the kernel.  Weights
// are checked when they are packed, feature maps when they are uploaded (api*.hip).
constexpr uint32_t FLAG_SPHERE_MISS = 1u, FLAG_SPLIT_RANGE = 2u, FLAG_SPLIT_STATIC = 4u;
constexpr float SPLIT_LIMIT = 65504.0f;
__device__ __forceinline__ void range_see4(const LaneCtx& L, const f32x4 v) {
    L.amax = fmaxf(fmaxf(L.amax, fabsf(v[0])), fabsf(v[1]));
    L.amax = fmaxf(fmaxf(L.amax, fabsf(v[2])), fabsf(v[3]));
}
__device__ __forceinline__ void range_see(const LaneCtx& L, float v) { L.amax = fmaxf(L.amax, fabsf(v)); }
__device__ __forceinline__ void range_commit(const LaneCtx& L, uint32_t* __restrict__ flags) {
    if (!(L.amax < SPLIT_LIMIT)) atomicOr(flags, FLAG_SPLIT_RANGE);
}

// Two values at once, 2 instructions per value: hi pair = v_cvt_pk_f16_f32 (round to nearest, both halves in one
// instruction and already packed); lo = x - hi with the fp16 hi read straight out of the packed pair by v_fma_mix_f32
// (one instruction instead of cvt_f32_f16 + sub); lo pair packed by a second v_cvt_pk_f16_f32.  Same values bit for
// bit as split(): hi = fp16(x), lo = fp16(x - hi) (the subtraction is exact).  The compiler's own lowering of the
// scalar form costs 5 instructions per value (separate conversions, a re-widening and a pack).
__device__ __forceinline__ void split2(float x0, float x1, h2& hi, h2& lo) {
    const f32x2 v = {x0, x1};
    hi = __builtin_convertvector(v, h2);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
    const f32x2 r = {r0, r1};
    lo = __builtin_convertvector(r, h2);
}

__device__ __forceinline__ void split4(const f32x4 v, h4& vh, h4& vl) {
#ifdef NEO_DBG_NO_SPLIT2
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        _Float16 h, l;
        split(v[e], h, l);
        vh[e] = h;
        vl[e] = l;
    }
    return;
#endif
    h2 a, b, c, d;
    split2(v[0], v[1], a, b);
    split2(v[2], v[3], c, d);
    vh = h4{a[0], a[1], c[0], c[1]};
    vl = h4{b[0], b[1], d[0], d[1]};
}

// max(x, 0) in ONE instruction (v_max_i32 on the bit pattern: negative floats are negative integers, -0.0 -> +0.0,
// NaN / inf pass through to the range guard).  fmaxf() costs two: the compiler canonicalises an operand that comes out
// of the matrix pipe first (v_max x, x, x).  (Not inline asm: the hazard recogniser does not see an asm statement's read
// of an in-flight MFMA result - measured: wrong values.)
__device__ __forceinline__ float relu1(float x) {
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, xi > 0 ? xi : 0);
}

// D tile (N-tile nt, M-tile mt) -> (ReLU) -> split -> the two planes of an activation tile with LDH halves per row.
// D holds outputs 8g + 4 half + e (e < 4) of point l31 in 4 consecutive registers: lanes l and l + 32 hold the two 8-byte
// halves of one 16-byte chunk.  The 8-byte form (ds_write_b64: lane groups of 16 CONSECUTIVE lanes, banks (a/4) mod 32)
// is 2-way bank-conflicted - the 16 rows of a group map onto 8 distinct slot residues.  NEO_SPLIT_STORE128:
// v_permlane32_swap exchanges the halves of a PAIR of chunks (2gp, 2gp + 1), lanes 0-31 then own chunk 2gp and lanes 32-63
// chunk 2gp + 1 of their point: one ds_write_b128 per plane and chunk pair (lane groups of 8 consecutive lanes = 8 rows with
// 8 distinct slot residues under the XOR swizzle: conflict-free), as mlp_vanilla_h.hip does since round 2.  Same bits in LDS.
typedef unsigned su32x2 __attribute__((ext_vector_type(2)));
typedef unsigned su32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void swap_halves32(su32x2& x, su32x2& y) {      // x[lanes 32-63] <-> y[lanes 0-31]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const su32x2 r = __builtin_amdgcn_permlane32_swap(x[i], y[i], false, false);   // builtin: the compiler inserts the wait states
        x[i] = r[0];
        y[i] = r[1];
    }
}
template <bool RELU, int LDH = 128>
__device__ __forceinline__ void store_tile_h(const f32x16& acc, const HT& act, int nt, int mt, const LaneCtx& L) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = acc[4 * g + e];
            v[e] = RELU ? relu1(x) : x;
        }
        range_see4(L, v);
        h4 vh, vl;
        split4(v, vh, vl);
        const int o = chunk_off<LDH>(mt * 32 + L.l31, nt * 4 + g) + 4 * L.half;
        *reinterpret_cast<h4*>(act.hi + o) = vh;
        *reinterpret_cast<h4*>(act.lo + o) = vl;
    }
}


// ---- GEMM pieces of the 64-point-tile evaluators (mlp_tp_h.hip, mlp_pix_h.hip) ----------------------------
// acc[nt][mt] += W-stage k-steps [ks0, ks0+n) x tile k-steps [tks0, tks0+n); N-tiles nts[], both M-tiles.
template <int NTW, int LDH>
__device__ __forceinline__ void gemm2h(f32x16 (&acc)[NTW][2], const h8* __restrict__ wp, int KS, const int (&nts)[NTW],
                                       int ks0, int tks0, int n, const HT& tile, const LaneCtx& L) {
    h8 ah[2][NTW], al[2][NTW];
    auto load_w = [&](int slot, int ks) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const h8* p = wp + ((nts[nt] * KS + ks) * 2) * 64 + L.lane;
            ah[slot][nt] = p[0];
            al[slot][nt] = p[64];
        }
    };
    load_w(0, ks0);
#pragma unroll 1
    for (int s = 0; s < n; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (s + u < n) {
                if (s + u + 1 < n) load_w((u + 1) & 1, ks0 + s + u + 1);
                h8 bh[2], bl[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int o = chunk_off<LDH>(mt * 32 + L.l31, ((tks0 + s + u) << 1) + L.half);
                    bh[mt] = *reinterpret_cast<const h8*>(tile.hi + o);
                    bl[mt] = *reinterpret_cast<const h8*>(tile.lo + o);
                }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[nt][mt] = NEO_MFMA_H(al[u][nt], bh[mt], acc[nt][mt]);
                        acc[nt][mt] = NEO_MFMA_H(ah[u][nt], bl[mt], acc[nt][mt]);
                        acc[nt][mt] = NEO_MFMA_H(ah[u][nt], bh[mt], acc[nt][mt]);
                    }
            }
        }
    }
}

// single accumulator tile (nt, mt): the 64-wide view layers
template <int LDH>
__device__ __forceinline__ void gemm1h(f32x16& acc, const h8* __restrict__ wp, int KS, int nt, int mt, int ks0, int n,
                                       const HT& tile, const LaneCtx& L) {
#pragma unroll 1
    for (int s = 0; s < n; ++s) {
        const h8* p = wp + ((nt * KS + ks0 + s) * 2) * 64 + L.lane;
        const h8 ah = p[0], al = p[64];
        const int o = chunk_off<LDH>(mt * 32 + L.l31, (s << 1) + L.half);
        const h8 bh = *reinterpret_cast<const h8*>(tile.hi + o);
        const h8 bl = *reinterpret_cast<const h8*>(tile.lo + o);
        acc = NEO_MFMA_H(al, bh, acc);
        acc = NEO_MFMA_H(ah, bl, acc);
        acc = NEO_MFMA_H(ah, bh, acc);
    }
}

// this lane's share of w_d . h for its point: point = 16 wave + lane/4, lane%4 picks 32 of the 128 channels
__device__ __forceinline__ float density_partial(const HT& act, const float* __restrict__ dens_w, const LaneCtx& L) {
    const int pt = L.wv * 16 + (L.lane >> 2), part = L.lane & 3;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int chunk_i = part * 4 + ((c + part) & 3);
        const int o = chunk_off<128>(pt, chunk_i);
        const h8 vh = *reinterpret_cast<const h8*>(act.hi + o);
        const h8 vl = *reinterpret_cast<const h8*>(act.lo + o);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(dens_w + chunk_i * 8);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(dens_w + chunk_i * 8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float w = e < 4 ? w0[e] : w1[e - 4];
            s = __builtin_fmaf((float)vh[e], w, s);
            s = __builtin_fmaf((float)vl[e], w, s);
        }
    }
    return s;
}


// weight re-packing into split fragment order (pack_h.hip): rows [0, rows) of src -> N-tiles [nt0, ...) of a stage
// with KS 16-deep k-steps; h8 index ((nt * KS + ks) * 2 + {hi 0, lo 1}) * 64 + lane; packed k -> source column
// through up to three segments, zero elsewhere
void pack_h(const float* src, int ld, int rows, int KS, int nt0, PackSegs sg, _Float16* dst, hipStream_t s);
// the same with an arbitrary packed-k -> source-column table (KS * 16 <= 256 entries; -1 = zero)
void pack_h_perm(const float* src, int ld, int rows, int KS, int nt0, const PackPerm& perm, _Float16* dst, hipStream_t s);

}  // namespace neo

// Fused  pos_enc -> NeRFMLP -> activations  for the vanilla NeRF path, fp32-EQUIVALENT
// arithmetic on the fp16 matrix cores ("f16x3 split").
//
// gfx950 has no TF32-class fast path and its exact fp32 MFMA runs at the fp32 vector rate
// (157 TFLOP/s); fp16 MFMA is 16x faster.  Every fp32 operand x is therefore split into two
// fp16 numbers  x = hi + lo  (hi = rn16(x), lo = rn16(x - hi): 22 significand bits) and each
// product is evaluated as
//        a*b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi            (drops a_lo*b_lo ~ 2^-22 |ab|)
// with three v_mfma_f32_32x32x16_f16 accumulating into the same fp32 accumulator.  fp16 x fp16
// products are exact in fp32 and accumulation is fp32, so the result carries fp32-class error
// (measured against the fp32-MFMA kernel and the CPU oracle in tests/test_gpu_vanilla.py); three
// fp16 MFMAs cost 3/16 of one fp32 MFMA of the same shape: a 5.3x higher matrix-pipe ceiling.
//
// Structure is that of mlp_vanilla.hip (tile = 64 points, 4 waves, 2 workgroups/CU, D[out][point]
// orientation, weights streamed from L2 in fragment order, activations in a swizzled LDS tile);
// what changes is the data format:
//   * weights are split ONCE at upload into [stage][n_tile][k_step(16)][hi|lo][lane][8 halves];
//   * activations are split ONCE per element in the layer epilogue and stored as two fp16 planes
//     (same 4 B/element LDS footprint as fp32), so the inner loop has no conversion work: per
//     16-deep k step a wave issues 4 weight loads (1 KiB each), 4 ds_read_b128 and 12 MFMAs.
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "kernels.h"
#include "mfma_tile.h"

namespace neo {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int ACT_LDH = 256;     // halves per activation row
constexpr int SIDE_LDH = 64;     // halves per side-buffer row (x0, later the view-direction encoding)
constexpr int NUM_STAGES = 10;   // L0..L7, bottleneck, view layer
constexpr int ST_N[NUM_STAGES] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 128};
constexpr int ST_KS[NUM_STAGES] = {4, 16, 16, 16, 16, 20, 16, 16, 16, 18};     // 16-deep k steps
constexpr int ST_SRC[NUM_STAGES] = {0, 1, 2, 3, 4, 5, 6, 7, 9, 8};
constexpr int ST_KIN[NUM_STAGES] = {63, 256, 256, 256, 256, 319, 256, 256, 256, 283};

constexpr int stage_w_off(int s) {   // in h8 units (16 B): per (n_tile, k_step) = 2 x 64 lanes
    int o = 0;
    for (int i = 0; i < s; ++i) o += (ST_N[i] / 32) * ST_KS[i] * 128;
    return o;
}
constexpr int stage_b_off(int s) {
    int o = 0;
    for (int i = 0; i < s; ++i) o += ST_N[i];
    return o;
}
constexpr int WPACK_H8 = stage_w_off(NUM_STAGES);
constexpr int HD_DW = 0, HD_DB = 256, HD_RW = 260, HD_RB = 644;

#define NEO_MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct HTile {          // a swizzled fp16 hi/lo tile in LDS
    _Float16* hi;
    _Float16* lo;
};

template <int LDH, int KM>
__device__ __forceinline__ int chunk_off(int row, int chunk) { return row * LDH + ((chunk ^ (row & KM)) << 3); }

__device__ __forceinline__ void split(float x, _Float16& hi, _Float16& lo) {
    asm("" : "+v"(x));          // opaque: no folding of a producing multiply into the conversion (see split_tile.h:split)
    hi = (_Float16)x;
    lo = (_Float16)__builtin_fmaf((float)hi, -1.0f, x);      // x - hi, exact; one mixed-precision fma
}

// two values at once: v_cvt_pk_f16_f32 + v_fma_mix_f32, the same bits as split() at 2 instructions per value instead of 5;
// max(x, 0) as v_max_i32 on the bit pattern, one instruction instead of canonicalise + max (see split_tile.h)
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float x0, float x1, h2v& hi, h2v& lo) {
    const f32x2v v = {x0, x1};
    hi = __builtin_convertvector(v, h2v);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
    const f32x2v r = {r0, r1};
    lo = __builtin_convertvector(r, h2v);
}
__device__ __forceinline__ float relu1(float x) {
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, xi > 0 ? xi : 0);
}

template <int LDH, int KM>
__device__ __forceinline__ void put_feat(const HTile& t, int p, int f, float v) {
    _Float16 h, l;
    split(v, h, l);
    const int o = chunk_off<LDH, KM>(p, f >> 3) + (f & 7);
    t.hi[o] = h;
    t.lo[o] = l;
}

// acc[nt][mt] += W[:, ks0*16 .. (ks0+n)*16) x tile^T over n k-steps: 3 fp16 MFMAs per product tile.
// N-tiles nt0..nt0+NTW-1, M-tiles mt0..mt0+MTW-1.
#ifndef NEO_VH_ABLATE
#define NEO_VH_ABLATE 0        // timing experiments: 1 no weight loads, 2 no LDS fragment reads, 4 a quarter of the epilogue stores, 8 no barriers
#endif
#define VH_SYNC() do { if (!(NEO_VH_ABLATE & 8)) __syncthreads(); } while (0)
#ifndef NEO_VH_SETPRIO
#define NEO_VH_SETPRIO 3      // s_setprio during the matrix phase: +0.5-0.7 % (tools/bench_kernel.py)
#endif
#ifndef NEO_VH_TILE_DEFAULT
#define NEO_VH_TILE_DEFAULT 64   // points per workgroup: 64 (two workgroups per CU) or 128 (one; $NEO_VANILLA_H_TILE)
#endif
#ifndef NEO_VH_PREFETCH
#define NEO_VH_PREFETCH 1      // weight fragments are requested this many k-steps ahead of their MFMAs
#endif
// Weight-fragment ring of a wave (NB slots x NTW N-tiles, hi + lo).  It lives in the kernel's scope so that the first
// D k-steps of the NEXT stage can be requested before this stage's barrier (NEO_VH_XLAYER): weights do not depend on the
// activations, and at a layer boundary nothing else is in flight.
template <int NTW>
struct WRing {
    static constexpr int D = NEO_VH_PREFETCH, NB = D + 1;
    h8 ah[NB][NTW], al[NB][NTW];
};

template <int NTW>
__device__ __forceinline__ void ring_load(WRing<NTW>& r, int slot, const char* wb, int KS, int nt0, int ks, const LaneCtx& L) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const uint32_t off = (uint32_t)(((nt0 + nt) * KS + ks) * 128 + L.lane) * 16u;
        r.ah[slot][nt] = *reinterpret_cast<const h8*>(wb + off);
        r.al[slot][nt] = *reinterpret_cast<const h8*>(wb + off + 1024u);
    }
}
// first D k-steps of a stage into slots 0..D-1 (what gemm_h expects when told `preloaded`)
template <int NTW>
__device__ __forceinline__ void ring_prime(WRing<NTW>& r, const h8* wp, int KS, int nt0, int ks0, const LaneCtx& L) {
#pragma unroll
    for (int d = 0; d < WRing<NTW>::D; ++d) ring_load(r, d, reinterpret_cast<const char*>(wp), KS, nt0, ks0 + d, L);
}

template <int NTW, int MTW, int LDH, int KM>
__device__ __forceinline__ void gemm_h(f32x16 (&acc)[NTW][MTW], const h8* __restrict__ wp, int KS, int nt0, int mt0,
                                       int ks0, int n, const HTile& tile, const LaneCtx& L, WRing<NTW>& ring,
                                       bool preloaded = false) {
    constexpr int D = WRing<NTW>::D, NB = WRing<NTW>::NB;
    auto& ah = ring.ah;
    auto& al = ring.al;
    // SGPR base + 32-bit VGPR byte offset; one k-step = 2 KB (hi 1 KB | lo 1 KB) further along an N-tile's stream
    const char* wb = reinterpret_cast<const char*>(wp);
    uint32_t off[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) off[nt] = (uint32_t)(((nt0 + nt) * KS + ks0) * 128 + L.lane) * 16u;
    auto load_w = [&](int slot, int s) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            ah[slot][nt] = *reinterpret_cast<const h8*>(wb + (off[nt] + 2048u * s));
            al[slot][nt] = *reinterpret_cast<const h8*>(wb + (off[nt] + 2048u * s + 1024u));
        }
    };
    if (!preloaded) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < n) load_w(d, d);
    }
#if NEO_VH_SETPRIO
    __builtin_amdgcn_s_setprio(NEO_VH_SETPRIO);      // matrix phase: issue ahead of the co-resident wave's epilogue VALU work
#endif
#pragma unroll 1
    for (int s = 0; s < n; s += NB) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (s + u < n) {
                if (s + u + D < n && !((NEO_VH_ABLATE & 1) && s + u + D >= NB)) load_w((u + D) % NB, s + u + D);
                h8 bh[MTW], bl[MTW];
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const int o = chunk_off<LDH, KM>((mt0 + mt) * 32 + L.l31, (((NEO_VH_ABLATE & 2) ? 0 : (s + u)) << 1) + L.half);
                    bh[mt] = *reinterpret_cast<const h8*>(tile.hi + o);
                    bl[mt] = *reinterpret_cast<const h8*>(tile.lo + o);
                }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = NEO_MFMA_H(al[u][nt], bh[mt], acc[nt][mt]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = NEO_MFMA_H(ah[u][nt], bl[mt], acc[nt][mt]);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = NEO_MFMA_H(ah[u][nt], bh[mt], acc[nt][mt]);
            }
        }
    }
#if NEO_VH_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

template <int NTW, int MTW>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NTW][MTW], const float* __restrict__ bias, int nt0,
                                          const LaneCtx& L) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        bias_tile(acc[nt][0], bias, nt0 + nt, L);
#pragma unroll
        for (int mt = 1; mt < MTW; ++mt) acc[nt][mt] = acc[nt][0];
    }
}

// epilogue: (ReLU) -> split -> two fp16 planes.  D holds outputs 8g+4*half+e (e<4) of point l31 in 4 consecutive
// registers, i.e. lanes l and l+32 hold the two 8-byte halves of one 16-byte chunk.  v_permlane32_swap exchanges them
// for a PAIR of chunks (g, g+1) so that lanes 0-31 own chunk g and lanes 32-63 chunk g+1 of their point: one
// ds_write_b128 per plane and chunk pair, conflict-free under the XOR swizzle (the 8-byte form was 2-way conflicted:
// rows r and r+16 of a 32-lane pass share banks).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void swap_halves(u32x2& x, u32x2& y) {      // x[lanes 32-63] <-> y[lanes 0-31]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const u32x2 r = __builtin_amdgcn_permlane32_swap(x[i], y[i], false, false);   // builtin: the compiler inserts the wait states
        x[i] = r[0];
        y[i] = r[1];
    }
}
template <bool RELU>
__device__ __forceinline__ void split_quad(const f32x16& acc, int g, h4& vh, h4& vl, const LaneCtx& L) {
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        float x0 = acc[4 * g + e], x1 = acc[4 * g + e + 1];
        if (RELU) { x0 = relu1(x0); x1 = relu1(x1); }
        L.amax = fmaxf(fmaxf(L.amax, fabsf(x0)), fabsf(x1));                              // range guard (split_tile.h)
        h2v h, l;
        split2(x0, x1, h, l);
        vh[e] = h[0]; vh[e + 1] = h[1];
        vl[e] = l[0]; vl[e + 1] = l[1];
    }
}
template <int NTW, int MTW, bool RELU>
__device__ __forceinline__ void store_act(const f32x16 (&acc)[NTW][MTW], const HTile& act, int nt0, int mt0,
                                          const LaneCtx& L) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
            for (int gp = 0; gp < ((NEO_VH_ABLATE & 4) ? 1 : 2); ++gp) {
                h4 h0, l0, h1, l1;
                split_quad<RELU>(acc[nt][mt], 2 * gp, h0, l0, L);
                split_quad<RELU>(acc[nt][mt], 2 * gp + 1, h1, l1, L);
                u32x2 xh = __builtin_bit_cast(u32x2, h0), yh = __builtin_bit_cast(u32x2, h1);
                u32x2 xl = __builtin_bit_cast(u32x2, l0), yl = __builtin_bit_cast(u32x2, l1);
                swap_halves(xh, yh);      // lanes < 32: (xh | yh) = chunk 2gp; lanes >= 32: chunk 2gp+1
                swap_halves(xl, yl);
                const int o = chunk_off<ACT_LDH, 15>((mt0 + mt) * 32 + L.l31, (nt0 + nt) * 4 + 2 * gp + L.half);
                *reinterpret_cast<u32x4*>(act.hi + o) = u32x4{xh[0], xh[1], yh[0], yh[1]};
                *reinterpret_cast<u32x4*>(act.lo + o) = u32x4{xl[0], xl[1], yl[0], yl[1]};
            }
        }
}

// NW = 4 or 8 waves per tile of TM = 32 * MT points.
//  * NW = 8 (MT = 2): each wave owns ONE 32-output N-tile of the 256-wide layers (for both M-tiles): twice the
//    resident waves per SIMD (4), unchanged weight traffic, twice the LDS fragment reads.
//  * MT = 4 (NW = 4): a 128-point tile, 2 N-tiles x 4 M-tiles per wave: every weight fragment feeds 24 instead of
//    12 MFMAs (half the L2 -> L1 weight stream per point) at the price of the whole LDS (160 KB) and one wave per SIMD.
template <int NW, int MT>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 4 : MT == 4 ? 1 : 2)) void k_vanilla_mlp_h(VanillaMlpHDev m,
                                                                                             const float* __restrict__ rays_o,
                                                                                             const float* __restrict__ dirs,
                                                                                             const float* __restrict__ t,
                                                                                             int t_row_stride, long P, int N,
                                                                                             float4* __restrict__ out) {
    constexpr int TM = 32 * MT;
    constexpr int NTW = 8 / NW;                 // N-tiles per wave in the 256-wide layers
    constexpr int LP = NW * 64 / TM;            // lanes per point in the VALU heads
    constexpr int PW = TM / 64;                 // waves that together cover the points once
    constexpr int NG = NW / PW;                 // such groups: they split the octaves of the encodings
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    HTile act{smem_h, smem_h + TM * ACT_LDH};
    HTile side{smem_h + 2 * TM * ACT_LDH, smem_h + 2 * TM * ACT_LDH + TM * SIDE_LDH};
    LaneCtx L;
    L.init();
    const int tid = threadIdx.x;
    const long tile0 = (long)blockIdx.x * TM;
    const h8* wp = reinterpret_cast<const h8*>(m.wpack);
    const int p_enc = (L.wv % PW) * 64 + L.lane, grp = L.wv / PW;      // this thread's point / octave group in the encodings

    // ---- pos_enc of the TM points into the side buffer (group q: octaves q, q+NG, ...) ----
    int my_ray;
    {
        const int p = p_enc;
        long g = tile0 + p;
        if (g >= P) g = P - 1;
        const int ray = (int)(g / N);
        const int s = (int)(g - (long)ray * N);
        my_ray = ray;
        const float tt = t[(long)ray * t_row_stride + s];
        float x[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            x[a] = rays_o[ray * 3 + a] + tt * dirs[ray * 3 + a];   // mul then add (helper.py:20-21)
            L.amax = fmaxf(L.amax, fabsf(x[a]));
        }
#pragma unroll 1
        for (int k = grp; k < 10; k += NG) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float sn, cs;
                enc_pair(x[a], k, sn, cs);
                put_feat<SIDE_LDH, 7>(side, p, 3 + k * 3 + a, sn);
                put_feat<SIDE_LDH, 7>(side, p, 33 + k * 3 + a, cs);
            }
        }
        if (grp == NG - 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) put_feat<SIDE_LDH, 7>(side, p, a, x[a]);
            put_feat<SIDE_LDH, 7>(side, p, 63, 0.0f);
        }
    }
    VH_SYNC();

    f32x16 acc[NTW][MT];
    const int nt0 = L.wv * NTW;
    WRing<NTW> ring;
    constexpr bool XL = (16 % WRing<NTW>::NB) == 0 && (4 % WRing<NTW>::NB) == 0;   // stages end on slot 0
    // ---- L0: 63 -> 256 ----
    init_bias<NTW, MT>(acc, m.bias + stage_b_off(0), nt0, L);
    gemm_h<NTW, MT, SIDE_LDH, 7>(acc, wp + stage_w_off(0), ST_KS[0], nt0, 0, 0, 4, side, L, ring);
    if (XL) ring_prime(ring, wp + stage_w_off(1), 16, nt0, 0, L);
    store_act<NTW, MT, true>(acc, act, nt0, 0, L);      // the activation planes are idle here
    VH_SYNC();
    // ---- L1..L7 (skip concat feeds L5) ----
#pragma unroll 1
    for (int s = 1; s <= 7; ++s) {
        const int woff = stage_w_off(1) + (s - 1) * (8 * 16 * 128) + (s > 5 ? 8 * 4 * 128 : 0);
        const int KS = s == 5 ? 20 : 16;
        init_bias<NTW, MT>(acc, m.bias + s * 256, nt0, L);
        gemm_h<NTW, MT, ACT_LDH, 15>(acc, wp + woff, KS, nt0, 0, 0, 16, act, L, ring, XL);
        if (s == 5) {
            if (XL) ring_prime(ring, wp + woff, KS, nt0, 16, L);
            gemm_h<NTW, MT, SIDE_LDH, 7>(acc, wp + woff, KS, nt0, 0, 16, 4, side, L, ring, XL);
        }
        if (XL && s < 7) {          // next stage: L(s+1) (its stream is 20 k-steps long for s + 1 == 5), or the bottleneck after L7
            const int nwoff = s < 7 ? stage_w_off(1) + s * (8 * 16 * 128) + (s + 1 > 5 ? 8 * 4 * 128 : 0) : stage_w_off(8);
            ring_prime(ring, wp + nwoff, s + 1 == 5 ? 20 : 16, nt0, 0, L);
        }
        VH_SYNC();
        store_act<NTW, MT, true>(acc, act, nt0, 0, L);
        if (s == 5) {
            // x0 is dead: the side buffer takes the view-direction encoding (group q: octaves q, q+NG, ... < 4)
            const int p = p_enc;
            float d[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) d[a] = dirs[my_ray * 3 + a];
#pragma unroll 1
            for (int k = grp; k < 4; k += NG) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float sn, cs;
                    enc_pair(d[a], k, sn, cs);
                    put_feat<SIDE_LDH, 7>(side, p, 3 + k * 3 + a, sn);
                    put_feat<SIDE_LDH, 7>(side, p, 15 + k * 3 + a, cs);
                }
            }
            if (grp == NG - 1) {
#pragma unroll
                for (int a = 0; a < 3; ++a) put_feat<SIDE_LDH, 7>(side, p, a, d[a]);
            }
            if (grp == NG - 2) {
#pragma unroll
                for (int f = 27; f < 32; ++f) put_feat<SIDE_LDH, 7>(side, p, f, 0.0f);
            }
        }
        VH_SYNC();
    }
    // ---- density head on h8 (VALU, LP lanes per point; x = hi + lo) ----
    float raw_sigma;
    {
        const int pt = tid / LP, part = tid % LP;
        constexpr int CH = 32 / LP;          // 16-B chunks (8 features) per lane
        const float* wd = m.heads + HD_DW;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int chunk = part * CH + ((c + part) % CH);
            const int o = chunk_off<ACT_LDH, 15>(pt, chunk);
            const h8 vh = *reinterpret_cast<const h8*>(act.hi + o);
            const h8 vl = *reinterpret_cast<const h8*>(act.lo + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += ((float)vh[e] + (float)vl[e]) * wd[chunk * 8 + e];
        }
#pragma unroll
        for (int o = 1; o < LP; o <<= 1) s += __shfl_xor(s, o, 64);
        raw_sigma = s + m.heads[HD_DB];
    }
    // ---- view layer: [bottleneck | dir enc] 283 -> 128, ReLU (4 N-tiles: split over M as well when NW = 8); with the bottleneck
    //      folded in (NEO_VH_FOLDB) its first 256 inputs are the trunk's last activations themselves ----
    {
        constexpr int MTV = NW == 8 ? 1 : MT;
        const int ntv = L.wv & 3, mtv = NW == 8 ? (L.wv >> 2) : 0;
        f32x16 accv[1][MTV];
        WRing<1> vring;
        init_bias<1, MTV>(accv, m.bias + stage_b_off(9), ntv, L);
        gemm_h<1, MTV, ACT_LDH, 15>(accv, wp + stage_w_off(9), 18, ntv, mtv, 0, 16, act, L, vring);
        gemm_h<1, MTV, SIDE_LDH, 7>(accv, wp + stage_w_off(9), 18, ntv, mtv, 16, 2, side, L, vring);
        VH_SYNC();
        store_act<1, MTV, true>(accv, act, ntv, mtv, L);
        VH_SYNC();
    }
    // ---- rgb head (VALU) + activations + store ----
    {
        const int pt = tid / LP, part = tid % LP;
        constexpr int CH = 16 / LP;
        const float* wr = m.heads + HD_RW;
        float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int chunk = part * CH + ((c + part) % CH);
            const int o = chunk_off<ACT_LDH, 15>(pt, chunk);
            const h8 vh = *reinterpret_cast<const h8*>(act.hi + o);
            const h8 vl = *reinterpret_cast<const h8*>(act.lo + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float h = (float)vh[e] + (float)vl[e];
                r += h * wr[chunk * 8 + e];
                g += h * wr[128 + chunk * 8 + e];
                b += h * wr[256 + chunk * 8 + e];
            }
        }
#pragma unroll
        for (int o = 1; o < LP; o <<= 1) {
            r += __shfl_xor(r, o, 64);
            g += __shfl_xor(g, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if (!(L.amax < 65504.0f)) atomicOr(m.flags, 2u);       // split range guard: an operand left the fp16 range
        const long gi = tile0 + pt;
        if (part == 0 && gi < P) {
            out[gi] = make_float4(colour_act(r + m.heads[HD_RB]), colour_act(g + m.heads[HD_RB + 1]),
                                  colour_act(b + m.heads[HD_RB + 2]), density_act(raw_sigma));
        }
    }
}

// Split one nn.Linear (out,in) into fp16 hi/lo fragments: dst[(nt*KS + ks)*2 + {0 hi,1 lo}][lane][8]
// = W[nt*32 + (lane&31)][ks*16 + 8*(lane>>5) + e], zero beyond k_in.
__global__ void k_pack_stage_h(const float* __restrict__ W, int n_out, int k_in, int KS, _Float16* __restrict__ dst) {
    const int total = (n_out / 32) * KS * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, blk = idx >> 9;
        const int ks = blk % KS, nt = blk / KS;
        const int n = nt * 32 + (lane & 31);
        const int k = ks * 16 + 8 * (lane >> 5) + e;
        const float w = (k < k_in) ? W[(long)n * k_in + k] : 0.0f;
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        const long base = ((long)(nt * KS + ks) * 2) * 512 + lane * 8 + e;
        dst[base] = hi;
        dst[base + 512] = lo;
    }
}

}  // namespace

size_t vanilla_wpack_h_bytes() { return (size_t)WPACK_H8 * 16; }
size_t vanilla_fold_floats() { return 128 * 283; }

// weights / biases order: pts_linears.0..7, views_linear.0, bottleneck_layer, density_layer, rgb_layer
void launch_vanilla_pack_h(const float* const* weights, const float* const* biases, void* wpack_h, float* fold_ws,
                           const float* bias_src, float* bias_h, hipStream_t s) {
    _Float16* base = reinterpret_cast<_Float16*>(wpack_h);
    (void)hipMemcpyAsync(bias_h, bias_src, (size_t)stage_b_off(NUM_STAGES) * sizeof(float), hipMemcpyDeviceToDevice, s);
    // bottleneck_layer (256 -> 256, NO activation) feeds views_linear.0 only (vanilla_nerf/model.py:113-121):
    // W_v [W_b h + b_b | d] + b_v = (W_v[:, :256] W_b) h + W_v[:, 256:] d + (W_v[:, :256] b_b + b_v).  The product is formed once
    // per upload (fp64 accumulation) and packed as the view layer; the kernel skips the bottleneck stage (65,536 of 593,408 MACs).
    launch_fold_bottleneck(weights[8], weights[9], biases[9], biases[8], 128, 256, 256, 27, fold_ws, bias_h + stage_b_off(9), s);
    for (int st = 0; st < NUM_STAGES; ++st) {
        const int total = (ST_N[st] / 32) * ST_KS[st] * 512;
        const float* src = st == 9 ? fold_ws : weights[ST_SRC[st]];
        hipLaunchKernelGGL(k_pack_stage_h, dim3((total + 255) / 256), dim3(256), 0, s, src, ST_N[st],
                           ST_KIN[st], ST_KS[st], base + (long)stage_w_off(st) * 8);
    }
}

void launch_vanilla_mlp_h(const VanillaMlpHDev& m, const float* rays_o, const float* dirs, const float* t,
                          int t_row_stride, int R, int N, float* out, hipStream_t s) {
    const long P = (long)R * N;
    if (P <= 0) return;
    static int nw = 0, tm = 0;
    if (nw == 0) {
        nw = 4;   // measured: 4 waves 396 TFLOP/s, 8 waves 381 (profiles/r01_vanilla_h_variants.log)
        tm = NEO_VH_TILE_DEFAULT;
        if (const char* e = getenv("NEO_VANILLA_H_WAVES")) nw = atoi(e) == 8 ? 8 : 4;
        if (const char* e = getenv("NEO_VANILLA_H_TILE")) tm = atoi(e) == 128 ? 128 : 64;
        if (nw == 8) tm = 64;
    }
    const size_t lds = (size_t)(2 * tm * ACT_LDH + 2 * tm * SIDE_LDH) * sizeof(_Float16);   // 80 KiB per 64 points
    const long tiles = (P + tm - 1) / tm;
    // per-device attribute: set on every launch (a host-side table write), not once per process
    auto go = [&](auto kernel, int threads) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3((unsigned)tiles), dim3(threads), lds, s, m, rays_o, dirs, t, t_row_stride, P, N,
                           reinterpret_cast<float4*>(out));
    };
    if (nw == 8) go(k_vanilla_mlp_h<8, 2>, 512);
    else if (tm == 128) go(k_vanilla_mlp_h<4, 4>, 256);
    else go(k_vanilla_mlp_h<4, 2>, 256);
}

}  // namespace neo

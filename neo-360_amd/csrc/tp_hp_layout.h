// Layout of the NeO-360 evaluator that gathers the PRE-PROJECTED latent (mlp_tp_hp.hip: k_tp_mlp_hp, k_tp_preproject):
// packed split-fp16 weight stream, bias / head offsets, channel order of the projected map.
#pragma once
#include <type_traits>

#include "split_tile.h"
#include "tp_common.h"

#ifndef NEO_TP_PRIO
#define NEO_TP_PRIO 0     // > 0: s_setprio around the L1..L3 k-steps of both evaluators (the wave that issues MFMAs wins the SIMD's arbitration)
#endif

namespace neo {
namespace hp {

using tp::TM;
using tp::blend4;
using tp::pe_feature;


// ---- packed weight layout (h8 units; one (n_tile, k_step) = hi 64 lanes + lo 64 lanes) -------------
// streamed stage: packed k = [world 128 | pos_enc 63/84 -> 64/96]
__host__ __device__ constexpr int pe_ksteps(int pe_c) { return pe_c == 3 ? 4 : 6; }
__host__ __device__ constexpr int ks_x(int pe_c) { return 8 + pe_ksteps(pe_c); }
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

constexpr int PROJ_TEXEL_BYTES = 1024;     // 256 fp32 channels per texel of the pre-projected map

// Bottleneck folded into view layer 0 (round 4).  NeRFPPMLP's bottleneck layer has NO activation and feeds views_linear.0 only
// (neo360/model.py:133-149), and the evaluators already apply it to the view MEAN of the trunk (linearity, mlp_tp_h.hip):
//     W_v0 [W_b h + b_b | d] + b_v0  =  (W_v0[:, :128] W_b) h + W_v0[:, 128:] d + (W_v0[:, :128] b_b + b_v0).
// The 64 x 128 product matrix and the bias are formed once per weight upload (k_fold_bottleneck, fp64 accumulation) and packed
// into the slot of view layer 0: the tail loses its 128 x 128 GEMM (48 of its 90 MFMAs per wave), an epilogue and two barriers.
#ifndef NEO_TP_FOLDB
#define NEO_TP_FOLDB 1
#endif

// ---- positional encoding in DOUBLING order (round 5) -------------------------------------------------------------------------
// The C-coordinate, 10-octave encoding is produced in HALF-CHUNKS of 4 packed features = (sin, shifted sin) of TWO consecutive
// octaves of ONE coordinate: the second octave comes from the first by angle doubling (common.h:sincos_pair2), ~14 instead of
// ~38 VALU instructions.  Half-chunk hc: hc < 4 C -> coordinate hc / 4, octaves 2 (hc % 4) and + 1 (octaves 0..7);
// 4 C <= hc < 5 C -> coordinate hc - 4 C, octaves 8, 9; hc == 5 C -> the C identity features, zero padded; beyond: zeros.
// The weight columns are packed in the same order (pe2_source_column).  0: the round-3 pair order (pair = octave * C + coordinate).
#ifndef NEO_PE_PAIR2
#define NEO_PE_PAIR2 0
#endif
__host__ __device__ constexpr int pe2_coord(int c, int hc) { return hc < 4 * c ? hc / 4 : hc - 4 * c; }
__host__ __device__ constexpr int pe2_oct0(int c, int hc) { return hc < 4 * c ? 2 * (hc % 4) : 8; }
// packed encoding position j (0 .. 8 * chunks) -> column of the reference's [x | sin, octave-major | shifted sin] encoding; -1 = zero
__host__ __device__ constexpr int pe2_source_column(int c, int j) {
    const int hc = j >> 2, u = (j >> 1) & 1, w = j & 1;
    if (hc < 5 * c) return c + (w ? 10 * c : 0) + (pe2_oct0(c, hc) + u) * c + pe2_coord(c, hc);
    if (hc == 5 * c) return (j & 3) < c ? (j & 3) : -1;
    return -1;
}
// chunk ch (wave-uniform) of the encoding of the point whose coordinates are xv[0 .. C-1], as the hi / lo fp16 planes' 16 bytes.
// The two half-chunks are computed and split ONE AFTER THE OTHER (a scheduling barrier between them): interleaved, their ~30
// temporaries push the evaluator over its 256-register budget (spills, measured with the first version of this function).
template <int C>
__device__ __forceinline__ void pe2_chunk(const f32x4 xv, int ch, h8& vh, h8& vl, const LaneCtx& L) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int hc = 2 * ch + h;                     // wave-uniform: scalar arithmetic and uniform branches
        float f[4];
        if (hc < 5 * C) {
            const int a = pe2_coord(C, hc);
            const float x = a == 0 ? xv[0] : a == 1 ? xv[1] : a == 2 ? xv[2] : xv[3];
            sincos_pair2(ldexpf(x, pe2_oct0(C, hc)), f[0], f[1], f[2], f[3]);
        } else if (hc == 5 * C) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[e] = e < C ? xv[e] : 0.0f;
                if (e < C) range_see(L, xv[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) f[e] = 0.0f;
        }
        h2 h0, l0, h1, l1;
        split2(f[0], f[1], h0, l0);
        split2(f[2], f[3], h1, l1);
        vh[4 * h] = h0[0]; vh[4 * h + 1] = h0[1]; vh[4 * h + 2] = h1[0]; vh[4 * h + 3] = h1[1];
        vl[4 * h] = l0[0]; vl[4 * h + 1] = l0[1]; vl[4 * h + 2] = l1[0]; vl[4 * h + 3] = l1[1];
        if (h == 0) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

// Channel order of the pre-projected map.  Output o in [0, 256) of [L0 | L3 skip]: N-tile nt = o / 32 (0..3 L0,
// 4..7 L3), r = o % 32.  Stored in 4 chunks of 64 channels; chunk c = (nt / 4) * 2 + r / 16 holds, for every
// wave w = nt % 4, the 16 outputs r % 16 = 8 gg + 4 half + e of its N-tile: position = w * 16 + r % 16.
// A wave's D fragment of N-tile nt (registers 4g+e <-> outputs 8g + 4 half + e) is then two 16-B pieces per chunk.
__host__ __device__ constexpr int proj_index(int o) {
    const int nt = o >> 5, r = o & 31;
    return ((nt >> 2) * 2 + (r >> 4)) * 64 + (nt & 3) * 16 + (r & 15);
}


}  // namespace hp
}  // namespace neo

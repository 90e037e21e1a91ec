// Layout of the NeO-360 evaluator that gathers the PRE-PROJECTED latent (mlp_tp_hp.hip: k_tp_mlp_hp, k_tp_preproject):
// packed split-fp16 weight stream, bias / head offsets, channel order of the projected map.
#pragma once
#include <type_traits>

#include "split_tile.h"
#include "tp_common.h"


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

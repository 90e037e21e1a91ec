// Layer-by-layer path of the Mip-NeRF 360 NeRF MLP (mlp_mip_h.hip + mip_gemm_h.h): workspace and launch entry.
#pragma once
#include "kernels.h"

namespace neo {

// Activations of one batch of intervals in MFMA fragment order (mip_gemm_h.h): the 504-d encoding (32 k-steps, 2 KB per
// interval) and two 1024-wide ping-pong buffers (64 k-steps, 4 KB per interval).  cap = intervals per batch, a multiple of
// 2048; 16384 keeps x0 + ya + yb (160 MB) + the 35 MB of weights inside the 256 MB Infinity Cache.
struct MipLayeredWs {
    char* x0;
    char* ya;
    char* yb;
    int cap;
};
constexpr int MIP_LAYERED_BATCH = 16384;
inline size_t mip_layered_x0_bytes(int cap) { return static_cast<size_t>(cap) * 2048; }
inline size_t mip_layered_y_bytes(int cap) { return static_cast<size_t>(cap) * 4096; }

// NeRF MLP (1024 x 8, rgb) over R x n intervals, split-fp16 arithmetic; same inputs / outputs as launch_mip_mlp_h
int launch_mip_mlp_h_layered(const MipMlpHDev& m, const MipLayeredWs& ws, const float* rays_o, const float* rays_d,
                             const float* viewdirs, const float* radii, const float* tdist, int R, int n, float* out,
                             hipStream_t s);

}  // namespace neo

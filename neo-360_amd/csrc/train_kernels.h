// Launchers of train_mlp.hip: the training-side MLPs (forward that keeps the activations + backward) on exact fp32 MFMA GEMMs.
// Kept apart from kernels.h so that work on the training path leaves the evaluators' sources (and the hash bench.py stamps the
// counter profiles with) untouched.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace neo {

// NeRFPPMLP (neo360/model.py:110-158): rows R = NV * P view-major; the input rows are the three tensors x_enc (R, pe),
// local (R, 512), world (R, 128) - never concatenated; cond (R, 27).  w / b order as neo_tp_upload_mlp.
size_t tp_train_tape_floats(int NV, long P);
size_t tp_train_scratch_floats(int NV, long P);
void launch_tp_train_forward(int pe, const float* const* w, const float* const* b, const float* x_enc, const float* local,
                             const float* world, const float* cond, int NV, long P, float* tape, float* raw_rgb,
                             float* raw_sigma, hipStream_t s, const float* pre = nullptr);
void launch_tp_train_backward(int pe, const float* const* w, const float* x_enc, const float* local, const float* world,
                              const float* cond, int NV, long P, const float* tape, float* scratch, const float* g_rgb,
                              const float* g_sigma, float* const* gw, float* const* gb, float* g_x_enc, float* g_local,
                              float* g_world, hipStream_t s, float* g_pre = nullptr);

// 1 (default): the per-row part of the projected-space NeRFPPMLP chain runs as one kernel each way (train_chain.h); 0: layer by layer.
// mode < 0 only queries.  Returns the previous mode.
int train_chain_mode(int mode);

// PixelNeRF's MLP (vanilla_nerf/model_pixel.py:96-131) on the projected latent: pre (R, 128), x_enc (R, 63), cond (R, 27).
// w / b order as neo_pix_upload_mlp.
size_t pix_train_tape_floats(int NV, long P);
size_t pix_train_scratch_floats(int NV, long P);
void launch_pix_train_forward(const float* const* w, const float* const* b, const float* x_enc, const float* pre, const float* cond,
                              int NV, long P, float* tape, float* raw_rgb, float* raw_sigma, hipStream_t s);
void launch_pix_train_backward(const float* const* w, const float* x_enc, const float* cond, int NV, long P, const float* tape,
                               float* scratch, const float* g_rgb, const float* g_sigma, float* const* gw, float* const* gb,
                               float* g_x_enc, float* g_pre, hipStream_t s);

// Mip-NeRF 360 MLP (mipnerf360/model.py:107-176): rows = R x n intervals, x0 (rows, 504), d_enc (R, 27); W / D = netwidth / netdepth, rgb = 1 with
// the colour branch.  w / b order as neo_mip_upload_mlp.  Outputs activated: rgbdens (rows, 4) = [rgb | density].
size_t mip_train_tape_floats(int W, int D, int rgb, long rows, long R);
size_t mip_train_scratch_floats(int W, int rgb, long rows, long R);
void launch_mip_train_forward(int W, int D, int rgb, const float* const* w, const float* const* b, const float* x0, const float* d_enc,
                              long R, int n, float* tape, float* rgbdens, hipStream_t s);
void launch_mip_train_backward(int W, int D, int rgb, const float* const* w, const float* x0, const float* d_enc, long R, int n,
                               const float* tape, float* scratch, const float* rgbdens, const float* g, float* const* gw,
                               float* const* gb, hipStream_t s);

// vanilla NeRFMLP (vanilla_nerf/model.py:100-125): rows R = rays x samples; x0 (R, 63), cond (R, 27).  w / b order as
// neo_vanilla_upload_mlp.
size_t vanilla_train_tape_floats(long R);
// Mip-NeRF 360 training operators (mip_train.hip): IPE rows (R n, 504) of the intervals of tdist (R, n + 1); backward of the
// exp(-cumsum) compositing: g_out (R, n, 4) = gradients of (rgb, density) from g_w (R, n, may be null) and g_c (R, 3, may be null)
void launch_mip_encode(const float* rays_o, const float* rays_d, const float* radii, const float* tdist, const float* basis, int R,
                       int n, float* out, hipStream_t s);
void launch_mip_composite_bwd(const float* rgbdens, const float* tdist, const float* rays_d, int R, int n, float bg, const float* g_w,
                              const float* g_c, float* g_out, hipStream_t s);
// one linear layer (exact fp32 MFMA GEMM, k_sgemm): y (+)= x W^T + b [ReLU];  gx (+)= gy W
void launch_linear_forward(long rows, int out_f, int in_f, const float* x, long ldx, const float* w, long ldw, const float* bias,
                           int relu, int accumulate, float* y, long ldy, hipStream_t s);
void launch_linear_input_grad(long rows, int in_f, int out_f, const float* gy, long ldy, const float* w, long ldw, int accumulate,
                              float* gx, long ldx, hipStream_t s);
// dW (M x N) += dY^T X over K rows, db (M, may be null) += column sums of dY; scratch: weight_grad_scratch_floats()
size_t weight_grad_scratch_floats();
void launch_weight_grad(int M, int N, int K, const float* dY, long ldy, const float* X, long ldx, float* dW, long ldw, float* db,
                        float* scratch, hipStream_t s);
size_t vanilla_train_scratch_floats(long R);
void launch_vanilla_train_forward(const float* const* w, const float* const* b, const float* x0, const float* cond, long R,
                                  float* tape, float* raw_rgb, float* raw_sigma, hipStream_t s);
void launch_vanilla_train_backward(const float* const* w, const float* x0, const float* cond, long R, const float* tape,
                                   float* scratch, const float* g_rgb, const float* g_sigma, float* const* gw, float* const* gb,
                                   float* g_x0, float* g_cond, hipStream_t s);

}  // namespace neo

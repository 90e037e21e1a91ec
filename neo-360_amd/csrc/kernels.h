// Host-side launch wrappers implemented by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace neo {

// rays.hip
// rays [ray0, ray0 + n) of the H x W frame -> rows [0, n) of the outputs
void launch_raygen(int H, int W, float focal, const float* c2w, int ray0, int n, float* rays_o, float* viewdirs,
                   float* rays_d, float* radii, hipStream_t s);
struct BoxFrame {        // one oriented box: world -> box frame (row-major 3x4 of a 4x4, float64) and its bounds
    double m[12];
    double lo[3], hi[3];
};
void launch_aabb_multi(const BoxFrame* boxes_dev, int n_boxes, const double* rays_o, const double* rays_d, int R,
                       uint8_t* hit_per_box, float* near, float* far, uint8_t* mask, hipStream_t s);
void launch_aabb(const double* bounds, const double* rays_o, const double* rays_d, int R, uint8_t* hit,
                 double* tmin, double* tmax, hipStream_t s);
void launch_sphere(const float* rays_o, const float* rays_d, int R, float* far, uint8_t* ok, uint32_t* flags,
                   hipStream_t s);

// sampling.hip
void launch_pos_enc(const float* x, int n, int C, int min_deg, int max_deg, float* out, hipStream_t s);
// mode 0 vanilla, 1 NeO-360 inside sphere, 2 NeO-360 outside sphere.
void launch_composite(int mode, const float* rgbsigma, const float* t, int t_row_stride, const float* rays_d,
                      const float* t_far, int R, int N, int white_bkgd, float* rgb, float* acc, float* depth, float* weights,
                      float* lambda, hipStream_t s);
// u: n_new quantiles (device), see Ctx::quantiles; u_row_stride 0 = one row shared by all rays (randomized=False),
// n_new = one row per ray (randomized=True: uniform draws)
int launch_resample(const float* t_prev, int t_prev_stride, const float* weights, const float* u, int u_row_stride, int R,
                    int n_prev, int n_new, int descending, float* t_out, hipStream_t s);

// training.hip — training-side operators
void launch_uniform(uint64_t seed, uint32_t stream, int rows, int cols, float* out, hipStream_t s);
// dst[b][c][r] = src[b][r][c] (batched 2-D transpose, tiled through LDS)
void launch_transpose(const float* src, long batch, int rows, int cols, float* dst, hipStream_t s);
void launch_tp_level0_rand(const float* far, const float* edges, int R, int N, float near, const float* u_fg,
                           const float* u_bg, float* fg_t, float* bg_s, hipStream_t s);
int launch_composite_bwd(int mode, const float* rgbsigma, const float* t, int t_row_stride, const float* rays_d,
                         const float* t_far, int R, int N, int white_bkgd, const float* g_rgb, const float* g_acc,
                         const float* g_depth, const float* g_w, const float* g_lam, float* g_rgbsigma, hipStream_t s);
void launch_distloss(const float* w, const float* m, int R, int N, float interval, float* loss_rays, float* grad_w,
                     hipStream_t s);
struct TpScene;
struct TpViews;
void launch_gather(const TpScene& sc, const TpViews& views, const float* pts, long P, float* world, float* local,
                   hipStream_t s);
// lookup in a caller-owned channels-last map (NV Hf Wf, C) at the latent's taps (C % 64 == 0) and its run-merged scatter backward;
// pitch > 0: the map's row pitch in floats (a C-column slice of a wider map: `map` / `g_map` point at the slice's first column)
void launch_map_gather(const TpScene& sc, const TpViews& views, const float* pts, long P, const float* map, int C, float* out,
                       hipStream_t s, long pitch = 0);
void launch_map_gather_bwd(const TpScene& sc, const TpViews& views, const float* pts, long P, const float* g_out, int C, float* g_map,
                           hipStream_t s, long pitch = 0);
// training call: lookup points (P,3) + per-view reference-order encodings (NV,P,21 C) of the samples tvals (R,N); input_ch 3 inside /
// 4 outside the sphere (far required); activations (raw -> (rgb, sigma) packed) with their backward
void launch_tp_train_points(int input_ch, const float* rays_o, const float* rays_d, const float* tvals, const float* far, int R, int N,
                            const TpViews& views, int nv, uint32_t* flags, float* look, float* x_enc, hipStream_t s);
void launch_tp_activate(const float* raw_rgb, const float* raw_sigma, const float* noise, float noise_scale, long P, float* rgbsigma,
                        hipStream_t s);
void launch_tp_activate_bwd(const float* raw_rgb, const float* raw_sigma, const float* noise, float noise_scale, long P,
                            const float* g_rgbsigma, float* g_rgb, float* g_sigma, hipStream_t s);
void launch_gather_bwd(const TpScene& sc, const TpViews& views, const float* pts, long P, const float* g_world,
                       const float* g_local, float* g_plane_xz, float* g_plane_xy, float* g_plane_yz, float* g_latent,
                       hipStream_t s);

// mlp_vanilla.hip
struct VanillaMlpDev {
    const float* wpack;   // packed GEMM stages (fragment order)
    const float* bias;    // concatenated per-stage biases
    const float* heads;   // density w[256], density b, rgb w[3][128], rgb b[3]
};
size_t vanilla_wpack_floats();
size_t vanilla_bias_floats();
size_t vanilla_heads_floats();
void launch_vanilla_pack(const float* const* weights, const float* const* biases, float* wpack, float* bias,
                         float* heads, hipStream_t s);
// mlp_vanilla_h.hip — the same MLP on the fp16 matrix cores with hi/lo-split operands (fp32-equivalent)
struct VanillaMlpHDev {
    const void* wpack;    // fp16 hi/lo fragments
    const float* bias;    // shared with the fp32 path
    const float* heads;
    uint32_t* flags;      // context assertion word (bit 1: split range guard)
};
size_t vanilla_wpack_h_bytes();
// pack_h.hip: [W_v[:, :nb] . W_b | W_v[:, nb:]] and b_v + W_v[:, :nb] . b_b - a linear bottleneck folded into the layer that consumes it
void launch_fold_bottleneck(const float* wv, const float* wb, const float* bb, const float* bv, int n_out, int nb, int nin,
                            int extra, float* wfold, float* bfold, hipStream_t s);
// biases / fold_ws: the split kernel folds the (linear) bottleneck into the view layer at pack time: bias_h = its own copy of the
// bias block with the view layer's entry replaced, fold_ws = vanilla_fold_floats() floats of scratch
size_t vanilla_fold_floats();
void launch_vanilla_pack_h(const float* const* weights, const float* const* biases, void* wpack_h, float* fold_ws,
                           const float* bias_src, float* bias_h, hipStream_t s);
void launch_vanilla_mlp_h(const VanillaMlpHDev& m, const float* rays_o, const float* dirs, const float* t,
                          int t_row_stride, int R, int N, float* out, hipStream_t s);
void launch_vanilla_mlp(const VanillaMlpDev& m, const float* rays_o, const float* dirs, const float* t,
                        int t_row_stride, int R, int N, float* out, hipStream_t s);

// mlp_tp.hip — NeO-360 decoder
constexpr int TP_MAX_VIEWS = 8;
struct TpMlpDev {
    const float* wpack;
    const float* bias;
    const float* heads;
};
struct TpScene {                     // channels-last feature maps owned by the context
    const float* latent;            // (NV, Hf, Wf, 512)
    const float* plane[3];          // xz, xy, yz: (NV, Hp, Wp, 128)
    int nv, Hf, Wf, Hp, Wp;
    float focal, cx, cy;            // source view 0's intrinsics (neo360/model.py:242-244)
    float sx, sy;                   // latent_scaling / image_size (encoder_pn.py:121-123, :204-206)
    float fy_sign = -1.0f;          // v = (-y/z) * (fy_sign * focal) + cy: -1 for NeRF_TP (model.py:243), +1 for the
                                    // PixelNeRF baseline, which passes (f, f) (vanilla_nerf/model_pixel.py:203-206)
    // Ray-patch tile order (round 6, neo_ctx_set_ray_grid): when the rays of a launch are pixels of a row-major image of width
    // grid_w (ray 0 of the launch = pixel grid_first of the frame), the evaluators walk the rays of every whole band of 2^grid_ph
    // image rows in 2^grid_pw x 2^grid_ph pixel PATCHES instead of row by row, so the ~64 tiles resident on an XCD cover a compact
    // piece of the image and share feature texels in that XCD's L2 in both image directions (shape chosen per region in
    // api_tp.hip:tp_launch).  0: rays in the caller's order.  Results do not depend on it.
    int grid_w = 0;
    long grid_first = 0;
    int grid_pw = 3, grid_ph = 3;   // log2 of the patch width / height in pixels (bands are 2^grid_ph image rows)
};
struct TpViews {                     // world -> camera per source view (neo360/util.py:52-70)
    float rot[TP_MAX_VIEWS][9];     // c2w[:3,:3]^T, row-major
    float trans[TP_MAX_VIEWS][3];   // -rot @ c2w[:3,3]
};
size_t tp_wpack_floats(int input_ch);
size_t tp_bias_floats();
size_t tp_heads_floats();
// fold_ws: tp_fold_floats() floats of scratch (stage V0F: view layer 0 with the bottleneck folded in)
void launch_tp_pack(int input_ch, const float* const* w, const float* const* b, float* wpack, float* bias,
                    float* heads, hipStream_t s, float* fold_ws);
void launch_channels_last(const float* src, int NV, int C, int H, int W, float* dst, hipStream_t s);
struct TpPlaneProj;
// proj != null: the latent pre-projected through [W0_loc | W3_loc] is gathered instead of the latent (k_tp_preproject);
// pp != null as well: the three tri-planes pre-projected through [W0_world | W3_world] likewise.  With proj the kernel reads the
// view-summed direction encodings from `dirsum` (rays, 32; launch_tp_dirsum below, on the same stream before this launch)
void launch_tp_mlp(int input_ch, const TpMlpDev& m, const TpScene& sc, const TpViews& views, const float* rays_o,
                   const float* rays_d, const float* viewdirs, const float* tvals, const float* far, int R, int N,
                   int chunk, uint32_t* flags, float* out, hipStream_t s, const float* proj = nullptr,
                   const TpPlaneProj* pp = nullptr, const float* dirsum = nullptr);

// weight re-packing into MFMA fragment order (defined in mlp_tp.hip)
struct PackSegs { int k0[3], len[3], col[3]; };
struct PackPerm { short col[256]; };      // packed k -> source column (-1: zero), passed by value (pack_h_perm)
void pack_block(const float* src, int ld, int rows, int KC, int nt0, PackSegs sg, float* dst, hipStream_t s);
void copy_floats(const float* src, int n, float* dst, hipStream_t s);

// mlp_mip.hip / mip_sampling.hip — Mip-NeRF 360
struct MipMlpDev {
    const float* wpack;
    const float* bias;
    const float* heads;
    const float* basis;   // (3,21) row-major
};
struct MipMlpHDev {       // split-fp16 fragments (mlp_mip_h.hip); bias / heads / basis shared with MipMlpDev
    const void* wpack;
    const float* bias;
    const float* heads;
    const float* basis;
    uint32_t* flags;      // context assertion word (bit 1: split range guard)
    const float* view_bias = nullptr;   // 128: views_linear.0's bias with the bottleneck folded in (b_v + W_v[:, :256] b_b); rgb MLP only
};
size_t mip_wpack_h_bytes(int width, int depth, int rgb);
size_t mip_fold_floats();     // scratch of launch_mip_pack_h (the folded view layer, fp32)
// rgb MLP: the (linear) bottleneck is folded into views_linear.0 at pack time (pack_h.hip:launch_fold_bottleneck): the fragments
// hold [W_v[:, :256] W_b | W_v[:, 256:]] (128 x (width + 27)), view_bias (128 floats) receives the folded bias
void launch_mip_pack_h(int width, int depth, int rgb, const float* const* w, const float* const* b, void* wpack_h, float* fold_ws,
                       float* view_bias, hipStream_t s);
int launch_mip_mlp_h(int width, int depth, int rgb, const MipMlpHDev& m, const float* rays_o, const float* rays_d,
                     const float* viewdirs, const float* radii, const float* tdist, int R, int n, float* out,
                     hipStream_t s);
size_t mip_wpack_floats(int width, int depth, int rgb);
size_t mip_bias_floats(int width, int depth, int rgb);
size_t mip_heads_floats(int width);
// w/b order: pts_linear.0..depth-1, density_layer[, bottleneck_layer, views_linear.0, rgb_layer]
void launch_mip_pack(int width, int depth, int rgb, const float* const* w, const float* const* b, float* wpack,
                     float* bias, float* heads, hipStream_t s);
int launch_mip_mlp(int width, int depth, int rgb, const MipMlpDev& m, const float* rays_o, const float* rays_d,
                   const float* viewdirs, const float* radii, const float* tdist, int R, int n, float* out,
                   hipStream_t s);
// one wave per ray: (optional max-dilate of the previous histogram) -> annealed logits -> softmax cdf ->
// n centre quantiles (u, device table) -> interval endpoints sdist (R,n+1) and metric distances tdist
int launch_mip_resample(const float* s_prev, const float* w_prev, int n_prev, int dilate, float dilation,
                        float anneal, const float* u, int R, int n, float s_near, float s_far, float* sdist,
                        float* tdist, hipStream_t s, const float* jitter = nullptr);
// weights = alpha * exp(-cumsum) with an opaque last interval; rgb = sum w c + max(0,1-acc) * bg
void launch_mip_composite(const float* rgbdens, const float* tdist, const float* rays_d, int R, int n, float bg,
                          float* weights, float* rgb, hipStream_t s);

// mlp_tp_h.hip — the NeO-360 evaluator on the fp16 matrix cores (hi/lo-split operands, fp32-equivalent)
struct TpMlpHDev {
    const void* wpack;    // fp16 hi/lo fragments
    const float* bias;    // shared with the fp32 path
    const float* heads;
    uint32_t* flags;      // context assertion word (bit 0: unit-sphere miss, bit 1: split range guard)
};
// Range checks backing the split-fp16 guard (pack_h.hip): any fp16 inf / NaN among n packed halves (a weight
// >= 65520 in magnitude, or non-finite), any fp32 value with |x| >= limit or non-finite -> flags |= 2.
void launch_half_range_check(const void* halves, size_t n, uint32_t* flags, hipStream_t s);
void launch_f32_range_check(const float* x, size_t n, float limit, uint32_t* flags, hipStream_t s);
size_t tp_wpack_h_bytes(int input_ch);
void launch_tp_pack_h(int input_ch, const float* const* w, void* wpack_h, hipStream_t s);
void launch_tp_mlp_h(int input_ch, const TpMlpHDev& m, const TpScene& sc, const TpViews& views, const float* rays_o,
                     const float* rays_d, const float* viewdirs, const float* tvals, const float* far, int R, int N,
                     int chunk, uint32_t* flags, float* out, hipStream_t s);

// mlp_tp_hp.hip — the split-fp16 evaluator on the latent pre-projected through [W0_loc | W3_loc] (default)
int tp_kc_x(int input_ch);            // k-chunks per N-tile of stage X in the fp32 fragment pack (mlp_tp.hip)
size_t tp_wpack_hp_bytes(int input_ch);
size_t tp_proj_bytes(long texels);    // pre-projected map: 256 fp32 channels per latent texel
// fold_ws: tp_fold_floats() floats of scratch (the folded view-layer-0 matrix, tp_hp_layout.h NEO_TP_FOLDB); bias_src: the shared
// 768-float bias block; bias_hp: the pre-projected evaluators' own copy of it (view layer 0's bias folded)
size_t tp_fold_floats();
void launch_tp_pack_hp(int input_ch, const float* const* w, const float* const* b, void* wpack_hp, float* fold_ws,
                       const float* bias_src, float* bias_hp, hipStream_t s);
// G (texels, 256) = latent_cl (texels, 512) . [W0_loc | W3_loc]^T from the fp32 fragment pack's stage X
// (in_ch = 128, first_chunk = 64: a tri-plane through the world columns [W0_world | W3_world], mlp_tp_hpp.hip)
void launch_tp_preproject(const float* latent_cl, long texels, const float* wpack_f32_stage_x, int kc_x, float* proj,
                          hipStream_t s, int channels = 256, int in_ch = 512, int first_chunk = 0);
// dirsum (R, 32): per ray the SUM over the source views of its view-direction encoding in each view's camera frame
// (27 features + 5 zeros), built by launch_tp_dirsum before every evaluator launch
void launch_tp_dirsum(const float* viewdirs, int R, const TpViews& views, int nv, float* dirsum, hipStream_t s);
void launch_tp_mlp_hp(int input_ch, const TpMlpHDev& m, const float* proj, const TpScene& sc, const TpViews& views,
                      const float* rays_o, const float* rays_d, const float* viewdirs, const float* tvals,
                      const float* far, int R, int N, int chunk, uint32_t* flags, float* out, const float* dirsum,
                      hipStream_t s);

// mlp_tp_hpp.hip — the same evaluator with the three TRI-PLANES pre-projected as well (through [W0_world | W3_world]):
// no world GEMM stage per point-view; 4 maps x 1 KB taps are blended and added to the L0 / L3-skip accumulators
struct TpPlaneProj { const float* p[3]; };      // xz, xy, yz: (NV * Hp * Wp, 256) fp32, channel order hp::proj_index
// proj_all: ONE buffer [projected latent | projected xz | xy | yz] (+ >= 4 KB of padding: the gather pipeline reads one
// chunk past its last item); plane_base_texels[j]: first texel (1 KB each) of plane j in it; total size < 4 GB.
size_t tp_proj_pad_bytes();
void launch_tp_mlp_hpp(int input_ch, const TpMlpHDev& m, const float* proj_all, const long plane_base_texels[3],
                       const TpScene& sc, const TpViews& views, const float* rays_o, const float* rays_d, const float* viewdirs,
                       const float* tvals, const float* far, int R, int N, int chunk, uint32_t* flags, float* out,
                       const float* dirsum, hipStream_t s);

// (train_mlp.hip's launchers are declared in train_kernels.h)

// mlp_pix_h.hip — PixelNeRF baseline decoder evaluator, split-fp16 arithmetic (the default)
size_t pix_wpack_h_bytes();
size_t pix_bias_floats();
size_t pix_heads_floats();
size_t pix_fold_floats();        // scratch of launch_pix_pack_h (the folded view-layer-0 matrix)
void launch_pix_pack_h(const float* const* w, const float* const* b, void* wpack_h, float* bias, float* heads,
                       float* fold_ws, hipStream_t s);
void launch_pix_mlp_h(const TpMlpHDev& m, const float* proj /* null: gather the latent itself */, const TpScene& sc, const TpViews& views, const float* rays_o,
                      const float* rays_d, const float* viewdirs, const float* tvals, int t_shared, int R, int N,
                      int chunk, float* out, hipStream_t s);

// mlp_pix.hip — the same decoder in EXACT fp32 MFMA arithmetic, latent gathered as the reference gathers it (round 5)
size_t pix_wpack_floats();
int pix_kc_x();                  // k-chunks per N-tile of its first stage (the first 64 = the latent columns k_tp_preproject reads)
void launch_pix_pack(const float* const* w, const float* const* b, float* fold_v0, const float* bias_src, float* bias_f32,
                     float* wpack, hipStream_t s);
void launch_pix_mlp(const TpMlpDev& m, const TpScene& sc, const TpViews& views, const float* rays_o, const float* rays_d,
                    const float* viewdirs, const float* tvals, int t_shared, int R, int N, int chunk, float* out, hipStream_t s);

// pillar.hip — pillar stage of the scene encoder
struct PillarGeom {
    int nv, G0, G1, G2;                // rows = nv * G0 * G1 * G2, x slowest (neo360/util.py:12-26)
    int Hf, Wf;
    float focal, cx, cy, sx, sy;       // view 0's intrinsics; latent_scaling / image size
    const float* axes;                 // device: torch.linspace values of the three axes, [3][256]
    float rot[TP_MAX_VIEWS][9], trans[TP_MAX_VIEWS][3], cpos[TP_MAX_VIEWS][3];
};
size_t pillar_wpack_bytes();
// w: depth_fc.common_branch.0 (512x518), .2, depth_fc.depth_encoder, pillar_aggregator_{xz,yz,xy}.0 (512x513)
void launch_pillar_pack(const float* const* w, void* wpack, hipStream_t s);
// bias: 6 x 512 in that order; head_w: pillar_aggregator_{xz,yz,xy}.2 weights (3 x 512); head_b_host: their biases;
// h1, h2, Lf: (M,512) workspaces, score (3,M); outputs channels-last floor-plans.  Returns -1 for unsupported grids.
int launch_pillar(const PillarGeom& gm, const float* latent_cl, const void* wpack, const float* bias, const float* head_w,
                  const float* head_b_host, float* h1, float* h2, float* Lf, float* score, uint32_t* flags, float* fp_yz,
                  float* fp_xz, float* fp_xy, hipStream_t s);

// sampling.hip — NeO-360 level-0 sample rows and fg/bg merge
void launch_tp_level0(const float* far, const float* edges, int R, int N, float near, float* fg_t, float* bg_s,
                      hipStream_t s);
void launch_tp_merge(const float* fg_rgb, const float* fg_depth, const float* lambda, const float* bg_rgb,
                     const float* bg_depth, int R, float* rgb, float* depth, hipStream_t s);

}  // namespace neo

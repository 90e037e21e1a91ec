/*
 * neo360_hip.h — C ABI of libneo360_hip.so, the MI355X (gfx950) implementation of
 * the NeO-360 ray-marching hot path.
 *
 * The reference (zubair-irshad/NeO-360) is pure Python/PyTorch and has no FFI of
 * its own; each entry point below replaces a reference *function* (cited as
 * file:line, paths relative to the reference root) at the granularity a
 * maintainer would bind with ctypes.
 *
 * Conventions
 *  - Plain C: pointers + sizes, no torch / C++ types.  `stream` is a hipStream_t
 *    passed as void* (NULL = default stream).
 *  - Unless marked [host], every pointer is a DEVICE pointer owned by the caller
 *    (e.g. a PyTorch-ROCm tensor's data_ptr()).  The library never frees caller
 *    memory and keeps no reference past the call, except neo_*_upload / set_*
 *    which copy / re-layout into context-owned buffers before returning
 *    (stream-ordered).
 *  - All tensors are dense, row-major fp32 unless stated.
 *  - Return value: 0 on success, negative neo_status otherwise (never throws).
 *    neo_last_error() gives a human-readable message for the calling thread.
 *  - No call synchronises the device; all work is enqueued on `stream`.
 *  - A context is bound to one device; calls on one context must not overlap
 *    from several host threads (same rule as the reference's nn.Module).
 */
#ifndef NEO360_HIP_H
#define NEO360_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct neo_ctx neo_ctx;

typedef enum {
    NEO_OK = 0,
    NEO_ERR_INVALID = -1,   /* bad argument (null pointer, unsupported size)  */
    NEO_ERR_HIP = -2,       /* a HIP runtime call failed (see neo_last_error) */
    NEO_ERR_STATE = -3,     /* weights / scene not uploaded yet               */
    NEO_ERR_NOMEM = -4
} neo_status;

/* ---- lifecycle -------------------------------------------------------- */
int neo_abi_version(void);                       /* bumps on any signature change */
const char* neo_last_error(void);
int neo_ctx_create(int device, neo_ctx** out);
int neo_ctx_destroy(neo_ctx* ctx);
/* Read-and-clear the device-side assertion word.  bit0 (NEO_FLAG_SPHERE_MISS): a ray missed the unit
 * sphere — the reference's AssertionError at models/neo360/helper.py:271,:426.  bit1 (NEO_FLAG_SPLIT_RANGE): an
 * operand of a split-fp16 kernel left the fp16 range (results of that call invalid; select precision 0).  bit2
 * (NEO_FLAG_SPLIT_STATIC, always together with bit1): the offending operand is a static one - packed weights or an
 * uploaded feature map - so every call on this (weights, scene) pair trips again.
 * Synchronises `stream`.  [flags: host out] */
#define NEO_FLAG_SPHERE_MISS 1u
#define NEO_FLAG_SPLIT_RANGE 2u
#define NEO_FLAG_SPLIT_STATIC 4u
int neo_ctx_poll_flags(neo_ctx* ctx, uint32_t* flags, void* stream);
/* The same read WITHOUT a synchronisation (what a caller's chunk loop wants: the reference's own loop makes 300
 * forward calls per frame, neo360/model.py:861-907).  post: enqueue on `stream` a copy of the word into a pinned host
 * slot + its clear + an event; returns at once.  take: OR of the posted reads that have completed (wait == 0), or of
 * all posted reads after waiting for their events (wait != 0); a read is reported once.  pending [host out, may be
 * NULL]: posted reads still in flight after the call.  Up to 64 reads may be outstanding; a post that finds all 64
 * still in flight posts nothing (the device word is sticky until a read clears it: a later read reports it, and
 * take(wait != 0) finishes with one synchronous read on `stream` in that case).  Neither post nor take(wait = 0)
 * ever blocks.  sync_count: how many blocking waits (stream / event synchronisations) the flag calls of this context
 * have issued so far - 0 for a loop of post + take(wait = 0). */
int neo_ctx_post_flags(neo_ctx* ctx, void* stream);
int neo_ctx_take_flags(neo_ctx* ctx, int wait, void* stream, uint32_t* flags, int* pending);
int neo_ctx_sync_count(neo_ctx* ctx, uint64_t* blocking_waits);
/* One context may be driven from several streams: its scratch (workspaces, per-launch tables, projected maps) is shared,
 * so every rendering / evaluator / backward call on a stream other than the previous call's first waits - on the
 * device, hipStreamWaitEvent, never on the host - for an event recorded behind that previous call.  stream_waits:
 * how many such cross-stream waits this context has inserted (0 for the usual one-stream caller). */
int neo_ctx_stream_waits(neo_ctx* ctx, uint64_t* cross_stream_waits);
/* Scratch lanes (round 6).  The reference drives the module as 300 calls of 1024 rays per frame (render_rays_test,
 * neo360/model.py:861-907); launched back to back on one stream every call ends on a partly filled machine (2,064 tiles
 * of a coarse launch on 512 workgroup slots).  A context therefore owns FOUR sets of render scratch (grown on first use): neo_ctx_set_lane
 * picks the set the next calls use (0, the default, .. 3).  neo_tp_render writes its lane's scratch only and reads the shared
 * weights / maps, so two renders on different lanes AND different streams are not ordered against each other and overlap
 * on the device; every other entry point (uploads, set_scene, evaluator / training calls) stays ordered against all
 * earlier calls of the context, on any lane.  A one-lane, one-stream caller sees no difference. */
int neo_ctx_set_lane(neo_ctx* ctx, int lane);
/* Pixel-grid hint for whole-frame renders (round 6).  When the rays handed to the next neo_tp_render calls are the pixels of a
 * row-major image `width` wide (ray 0 of a call = pixel `first_ray` of the frame: a rank's shard starts in the middle), the
 * evaluators visit the rays of every whole band of image rows in small pixel patches (2 x 2 inside the unit sphere, 8 x 8 outside)
 * instead of row by row: the workgroups resident on an XCD then cover a compact piece of the image and share feature texels in that
 * XCD's L2 in both image directions - fabric-side traffic of a full-frame launch 59.8 -> 41.7 GB inside, 11.6 -> 1.8 GB outside
 * (profiles/r06_ray_patch_order.log).  Pure scheduling: every output value is bitwise the one without the hint.  width = 0
 * (default) removes it; width must be a multiple of 8.  The reference has no counterpart (its chunk loop renders 1024 consecutive
 * rays at a time; datasets/ray_utils.py:96-104 defines the row-major order). */
int neo_ctx_set_ray_grid(neo_ctx* ctx, int width, long first_ray);

/* Arithmetic of the per-point MLP GEMMs of every renderer (vanilla, NeRF_TP, Mip-NeRF 360, PixelNeRF).
 * 1 (the context default, and the default of the Python modules, "f16x3"): fp16 MFMA with every fp32
 * operand split into hi+lo fp16 and three products per term (a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32
 * accumulate) - fp32-class error at a 5.3x higher matrix-pipe ceiling; operands must stay below the fp16
 * range (|x| < 65504: checked, see NEO_FLAG_SPLIT_RANGE).  0 ("f32"): exact fp32 MFMA
 * (v_mfma_f32_32x32x2_f32), no range limit.  Both meet the 1e-4 parity contract. */
int neo_ctx_set_precision(neo_ctx* ctx, int mode);

/* torch.linspace(start, end, steps) for fp32 on the host (symmetric fill,
 * step=(end-start)/(steps-1)) — the constant tables the samplers start from
 * (vanilla_nerf/helper.py:425, :589; neo360/helper.py:36, :197).  No GPU needed.
 * out [host]. */
void neo_linspace_host(float start, float end, int steps, float* out);

/* ---- ray generation ---------------------------------------------------- */
/* datasets/ray_utils.py:84-104 get_ray_directions + :133-176 get_rays
 * (output_view_dirs=True, output_radii=True).  c2w [host]: 12 floats, row-major
 * 3x4.  Outputs (H*W,3),(H*W,3),(H*W,3),(H*W); rays_d == viewdirs (the
 * reference aliases them); radii may be NULL. */
int neo_raygen(neo_ctx* ctx, int H, int W, float focal, const float* c2w,
               float* rays_o, float* viewdirs, float* rays_d, float* radii, void* stream);

/* datasets/ray_utils.py:17-68 == models/neo360/helper.py:275-323
 * bbox_intersection_batch: float64 slab test, rays already in the box frame.
 * bounds [host]: 6 doubles (min xyz, max xyz).  hit: uint8 0/1 (bit-exact
 * contract); tmin/tmax: doubles, 0 where missed; either may be NULL. */
int neo_aabb_intersect(neo_ctx* ctx, const double* bounds, const double* rays_o,
                       const double* rays_d, int R, uint8_t* hit, double* tmin,
                       double* tmax, void* stream);

/* The same for rays [ray0, ray0 + n_rays) of the row-major H x W frame, written to rows [0, n_rays) of the
 * outputs: a rank of a ray-sharded render (models/interface.py:30-50 gathers the pieces) generates only its range. */
int neo_raygen_range(neo_ctx* ctx, int H, int W, float focal, const float* c2w, int ray0, int n_rays,
                     float* rays_o, float* viewdirs, float* rays_d, float* radii, void* stream);

/* models/neo360/helper.py:359-373 sample_rays_in_bbox over n_boxes oriented boxes: per box the rays are moved to
 * the box frame in float64 (:325-331; world_to_box [host]: n_boxes x 16 doubles, the row-major 4x4
 * np.linalg.inv([R|T]) the caller computes as get_object_rays_in_bbox does, :348-357), slab-tested (:275-323;
 * bounds [host]: n_boxes x 6 doubles, min xyz then max xyz), tmin/tmax rounded to float32, and merged by the
 * running minimum that treats 0 as "no hit".  rays_o / rays_d: (R,3) float64.  Outputs (any may be NULL):
 * hit_per_box (n_boxes,R) uint8, near (R) / far (R) float32 merged, mask (R) uint8 = near != 0 && far != 0
 * (bit-exact contract). */
int neo_aabb_multi(neo_ctx* ctx, int n_boxes, const double* world_to_box, const double* bounds,
                   const double* rays_o, const double* rays_d, int R, uint8_t* hit_per_box, float* near,
                   float* far, uint8_t* mask, void* stream);

/* models/neo360/helper.py:253-273 intersect_sphere.  far (R); ok (R) uint8 =
 * (1-|p|^2 >= 0), may be NULL; a miss also raises flag bit0. */
int neo_intersect_sphere(neo_ctx* ctx, const float* rays_o, const float* rays_d, int R,
                         float* far, uint8_t* ok, void* stream);

/* ---- stage-level operators (stage-isolated parity) ---------- */
/* helper.py pos_enc (neo360/helper.py:121-125 == vanilla_nerf/helper.py:445-449):
 * x (n,C) -> out (n, C*(2*(max_deg-min_deg)+1)). */
int neo_pos_enc(neo_ctx* ctx, const float* x, int n, int C, int min_deg, int max_deg,
                float* out, void* stream);

/* sorted_piecewise_constant_pdf + sample_pdf's sort
 * (vanilla_nerf/helper.py:567-616, neo360/helper.py:174-231), randomized=False.
 * t_prev (R,n_prev), weights (R,n_prev): bins are the n_prev-1 midpoints of
 * t_prev, pdf weights are weights[:,1:-1] (the slicing the callers do at
 * vanilla_nerf/model.py:171-175, neo360/model.py:308-318).  t_out
 * (R, n_prev+n_new) sorted ascending, or descending when `descending` != 0
 * (the background branch's flip, neo360/helper.py:234-239). */
int neo_resample(neo_ctx* ctx, const float* t_prev, const float* weights, int R, int n_prev,
                 int n_new, int descending, float* t_out, void* stream);

/* volumetric_rendering.  mode 0: vanilla (vanilla_nerf/helper.py:521-559);
 * mode 1: NeO-360 inside sphere (neo360/helper.py:128-171, in_sphere=True,
 * needs t_far (R)); mode 2: NeO-360 outside sphere (t descending).
 * rgbsigma (R,N,4) = (r,g,b,sigma); t (R,N); rays_d (R,3).
 * Outputs: rgb (R,3), acc (R), depth (R), weights (R,N), lambda (R) [mode 1];
 * any output may be NULL. */
int neo_composite(neo_ctx* ctx, int mode, const float* rgbsigma, const float* t,
                  const float* rays_d, const float* t_far, int R, int N, int white_bkgd,
                  float* rgb, float* acc, float* depth, float* weights, float* lambda,
                  void* stream);

/* ---- vanilla NeRF (models/vanilla_nerf/model.py) -------------------------- */
/* Upload one NeRFMLP (vanilla_nerf/model.py:44-98).  slot 0 = coarse_mlp,
 * 1 = fine_mlp.  weights/biases [host arrays of 12 device pointers], order:
 * pts_linears.0..7, views_linear.0, bottleneck_layer, density_layer, rgb_layer;
 * torch.nn.Linear layout (out,in).  Re-packed on the device into the MFMA
 * fragment order; the caller's tensors are not referenced afterwards. */
int neo_vanilla_upload_mlp(neo_ctx* ctx, int slot, const float* const* weights,
                           const float* const* biases, void* stream);

/* pos_enc + NeRFMLP.forward + activations for R*N points
 * (vanilla_nerf/model.py:183-204): points o + t*dirs; t is (R,N) when
 * t_row_stride == N, or one shared row when t_row_stride == 0.
 * out (R,N,4) = (rgb after sigmoid+padding, sigma after softplus(raw-1)). */
int neo_vanilla_mlp(neo_ctx* ctx, int slot, const float* rays_o, const float* dirs,
                    const float* t, int t_row_stride, int R, int N, float* out, void* stream);

/* NeRF.forward (vanilla_nerf/model.py:154-216), randomized=False, both levels.
 * Samples are cast along `viewdirs`, compositing scales by |rays_d| (as the
 * reference).  Per level l: rgb_l (R,3), acc_l (R), depth_l (R); level-0
 * outputs may be NULL. */
int neo_vanilla_render(neo_ctx* ctx, const float* rays_o, const float* viewdirs,
                       const float* rays_d, int R, float near, float far, int n_coarse,
                       int n_fine, int white_bkgd, float* rgb0, float* acc0, float* depth0,
                       float* rgb1, float* acc1, float* depth1, void* stream);

/* ---- NeO-360 decoder (models/neo360/model.py NeRF_TP) ---------------------- */
/* Upload one NeRFPPMLP (neo360/model.py:37-108).  slot 0..3 = fg_coarse,
 * fg_fine, bg_coarse, bg_fine.  input_ch = 3 (fg) or 4 (bg).  weights/biases
 * [host arrays of 9 device pointers], order: pts_linears.0..3,
 * views_linear.0, views_linear.1, bottleneck_layer, density_layer, rgb_layer. */
int neo_tp_upload_mlp(neo_ctx* ctx, int slot, int input_ch, const float* const* weights,
                      const float* const* biases, void* stream);

/* Scene features = the encoder outputs the reference recomputes per chunk
 * (neo360/model.py:272-274).  planes: 3 x (NV,Cw,Hp,Wp) NCHW; latent
 * (NV,Cl,Hf,Wf) NCHW; re-laid out channels-last into context-owned buffers.
 * Cw must be 128 and Cl 512 (the reference's fixed widths). */
int neo_tp_set_scene(neo_ctx* ctx, const float* plane_xz, const float* plane_xy,
                     const float* plane_yz, int NV, int Cw, int Hp, int Wp,
                     const float* latent, int Cl, int Hf, int Wf, float image_w,
                     float image_h, void* stream);

/* Both arithmetic modes (the exact fp32 evaluator gathers the same projected maps since round 4; in it modes 2 and 3 both
 * project the planes for every slot).  A fresh context is in mode 3.  enable != 0: the 512-channel latent is pre-projected once per
 * (scene, MLP slot) through the local columns of pts_linears.0 and of pts_linears.3's skip half
 * (W . bilerp(F) = bilerp(W . F): neo360/model.py:110-158 is linear in the latent up to the first ReLU), and
 * the evaluator gathers the 256-channel result; costs 1 KB per latent texel and slot of context memory.
 * enable == 0: the latent itself is gathered and multiplied per point (the reference's operation order).
 * enable == 2: the three tri-planes are pre-projected the same way through the world columns (the 128-channel plane
 * sum of encoder_tp_fusion_conv.py:180-206 enters pts_linears.0 / .3 linearly too, model.py:123-137): no per-point world
 * GEMM stage, four 256-channel maps are gathered, blended and added; 1 KB per plane texel and slot on top.
 * enable == 3: as 2 for the outside-sphere slots 2, 3 (most of their samples lie outside every map: few taps, the GEMM stage
 * is pure saving), as 1 for the inside-sphere slots 0, 1 (every map carries weight: the larger taps cost more than the GEMM
 * stage they replace).  The Python modules default to 3. */
int neo_tp_set_preproject(neo_ctx* ctx, int enable);

/* `predict` + the feature lookups for one region at given sample positions
 * (neo360/model.py:343-464): slot 0/1 inside the sphere (tvals = t, ascending),
 * slot 2/3 outside (tvals = inverse radius, descending; needs far (R) from
 * neo_intersect_sphere).  tvals (R,N); out (R,N,4) = (rgb, sigma) after activations.
 * `chunk`, src_poses [host], focal/cx/cy as for neo_tp_render. */
int neo_tp_mlp(neo_ctx* ctx, int slot, const float* rays_o, const float* rays_d,
               const float* viewdirs, const float* tvals, const float* far, int R, int N,
               int chunk, const float* src_poses, int NV, float focal, float cx, float cy,
               float* out, void* stream);

/* NeRF_TP.forward decoder half (neo360/model.py:276-581), randomized=False,
 * out_depth=True semantics, for R rays processed in reference-sized chunks
 * (`chunk` rays per forward call: the view-direction tiling of
 * neo360/model.py:357-360 makes results depend on chunk membership).
 * src_poses [host]: NV*16 floats (c2w 4x4 row-major); focal/cx/cy: source view
 * 0's intrinsics (neo360/model.py:242-244).  Outputs per level l (any may be
 * NULL): rgb_l (R,3), fg_rgb_l (R,3), bg_rgb_l (R,3), fg_acc_l (R),
 * bg_lambda_l (R), depth_l (R). */
typedef struct {
    float* rgb; float* fg_rgb; float* bg_rgb; float* fg_acc; float* bg_lambda; float* depth;
} neo_tp_level_out;
int neo_tp_render(neo_ctx* ctx, const float* rays_o, const float* rays_d,
                  const float* viewdirs, int R, int chunk, const float* src_poses, int NV,
                  float focal, float cx, float cy, int n_coarse, int n_fine, int white_bkgd,
                  const neo_tp_level_out* level0, const neo_tp_level_out* level1, void* stream);

/* ---- scene encoder: pillar stage -------------------------------------------------------- */
/* Weights of the pillar stage of GridEncoder (models/neo360/encoder_tp_fusion_conv.py:263-279, :364-373).
 * weights/biases [host arrays of 9 device pointers], order: depth_fc.common_branch.0 (512x518), depth_fc.common_branch.2,
 * depth_fc.depth_encoder, pillar_aggregator_xz.0 (512x513), pillar_aggregator_xz.2 (1x512), pillar_aggregator_yz.0, .2,
 * pillar_aggregator_xy.0, .2. */
int neo_enc_upload(neo_ctx* ctx, const float* const* weights, const float* const* biases, void* stream);

/* GridEncoder.forward from the world grid to the inputs of its floor-plan conv nets (:472-578): for every cell of the
 * G0 x G1 x G2 world grid (x, y in [-1,1], z in [0,1]) and source view, [pixel-aligned latent | camera xyz | masked
 * direction] -> depth_fc -> three axis scorers -> softmax along x / y / z -> weighted sums.  latent (NV,512,Hf,Wf) NCHW =
 * SpatialEncoder's output; src_poses [host] NV*16; focal / cx / cy = view 0's intrinsics (:491-493).  Outputs
 * channels-last: fp_yz (NV,G1,G2,512), fp_xz (NV,G0,G2,512), fp_xy (NV,G0,G1,512) (the reference permutes them to NCHW
 * for its conv nets, :580-592).  Split-fp16 arithmetic (neo_ctx_set_precision mode 1). */
int neo_enc_floorplans(neo_ctx* ctx, const float* latent, int NV, int Hf, int Wf, float image_w, float image_h,
                       const float* src_poses, float focal, float cx, float cy, int G0, int G1, int G2,
                       float* fp_yz, float* fp_xz, float* fp_xy, void* stream);

/* ---- training-side operators ----------------------------------------------------------- */
/* Counter-based uniforms in [0,1): out[r][c] = (Philox4x32-10(key = seed, counter = (r, c, stream_id, 0))[0] >> 8) 2^-24
 * - the generator behind every randomized=True sampler here (the reference draws torch.rand: helper.py:49, :196). */
/* dst[b][c][r] = src[b][r][c]: batched 2-D transpose (fp32, tiled through LDS, out of place).  NCHW <-> channels-last of a feature map
 * under autograd: (NV, C, H W) -> (NV, H W, C) is batch NV, rows C, cols H W; the gradient goes back with rows H W, cols C
 * (the reference keeps NCHW and permutes inside grid_sample: encoder_tp_fusion_conv.py:180-206). */
int neo_transpose(neo_ctx* ctx, const float* src, long batch, int rows, int cols, float* dst, void* stream);
int neo_rand_uniform(neo_ctx* ctx, uint64_t seed, uint32_t stream_id, int rows, int cols, float* out, void* stream);

/* neo360/helper.py:24-75 sample_along_rays for both regions: far (R) from neo_intersect_sphere, near = 1e-4.
 * u_fg / u_bg (R, n_coarse+1) uniforms = randomized=True (stratified jitter, :44-51); both NULL = randomized=False.
 * fg_t (R, n_coarse+1) ascending t; bg_s (R, n_coarse+1) descending inverse radius. */
int neo_tp_sample_level0(neo_ctx* ctx, const float* far, int R, int n_coarse, const float* u_fg, const float* u_bg,
                         float* fg_t, float* bg_s, void* stream);

/* neo_resample with one row of quantiles per ray, u (R, n_new) in [0,1): sorted_piecewise_constant_pdf with
 * randomized=True (neo360/helper.py:195-196; the draws need not be sorted). */
int neo_resample_u(neo_ctx* ctx, const float* t_prev, const float* weights, const float* u, int R, int n_prev,
                   int n_new, int descending, float* t_out, void* stream);

/* Backward of neo_composite (same mode / inputs): upstream gradients g_rgb (R,3), g_acc (R), g_depth (R),
 * g_weights (R,N), g_lambda (R) (any may be NULL = zero) -> g_rgbsigma (R,N,4) = dL/d(rgb, sigma) per sample.
 * The sample positions carry no gradient (the reference detaches them, helper.py:224). */
int neo_composite_backward(neo_ctx* ctx, int mode, const float* rgbsigma, const float* t, const float* rays_d,
                           const float* t_far, int R, int N, int white_bkgd, const float* g_rgb,
                           const float* g_acc, const float* g_depth, const float* g_weights,
                           const float* g_lambda, float* g_rgbsigma, void* stream);

/* torch_efficient_distloss.eff_distloss(w, m, interval) (requirements.txt:29; call site neo360/model.py:1246-1260):
 * per ray interval/3 sum w^2 + 2 sum_{i>j} w_i w_j (m_i - m_j) evaluated with prefix sums.  loss_rays (R) = per-ray
 * loss (the reference returns their mean), grad_w (R,N) = d loss_ray / d w; either may be NULL. */
int neo_distloss(neo_ctx* ctx, const float* w, const float* m, int R, int N, float interval, float* loss_rays,
                 float* grad_w, void* stream);

/* NeRFPPMLP for TRAINING (the module the reference's training step differentiates, neo360/model.py:110-158 under
 * :697-820) on the rows the reference forms: [pos_enc | local 512 | world 128] per point-view, given as the three dense
 * tensors they are concatenated from (x_enc (NV*P, 21 input_ch), local_feat (NV*P, 512), world_feat (NV*P, 128): the
 * concatenation, 3.3 GB for a reference chunk level, is never materialised), and cond (NV*P, 27) = view-direction
 * encodings; view-major rows (row = v P + p).  w / b [host]: nine DEVICE
 * pointers each, order and nn.Linear (out, in) layout of neo_tp_upload_mlp (pts_linears.0..3, views_linear.0, .1,
 * bottleneck, density, rgb).  forward: raw_rgb (P,3), raw_sigma (P,1) (pre-activation, as the module returns them); the
 * activations the backward needs are written to `tape`, a CALLER-owned device buffer of neo_tp_mlp_train_tape_floats(NV, P)
 * floats (a training step runs the four MLPs of both levels forward before the first backward: one tape per call).
 * backward (with the tape, inputs, cond and weights of that forward; g_* = upstream gradients):
 * gw / gb [host]: nine device pointers each to gradient buffers of the weights' / biases' shapes, ZEROED by the caller
 * (partial sums are accumulated atomically); g_x_enc / g_local / g_world = gradients of the three input tensors
 * (overwritten; each may be NULL) - the latter two feed neo_tp_gather_backward.  Exact fp32 arithmetic (v_mfma_f32_32x32x2_f32).  At most
 * 4.19 M rows per call. */
long neo_tp_mlp_train_tape_floats(int NV, long P);
int neo_tp_mlp_train_forward(neo_ctx* ctx, int input_ch, const float* const* w, const float* const* b, const float* x_enc,
                             const float* local_feat, const float* world_feat, const float* cond, int NV, long P, float* tape,
                             float* raw_rgb, float* raw_sigma, void* stream);
int neo_tp_mlp_train_backward(neo_ctx* ctx, int input_ch, const float* const* w, const float* x_enc, const float* local_feat,
                              const float* world_feat, const float* cond, int NV, long P, const float* tape,
                              const float* g_rgb, const float* g_sigma, float* const* gw, float* const* gb, float* g_x_enc,
                              float* g_local, float* g_world, void* stream);

/* The vanilla NeRFMLP for training (vanilla_nerf/model.py:100-125 under the training step :255-283), same contract:
 * x0 (R, 63) encoded points, cond (R, 27) = each row's view-direction encoding (the ray's, tiled over its samples);
 * w / b [host]: twelve device pointers each in the order of neo_vanilla_upload_mlp; tape: caller-owned,
 * neo_vanilla_mlp_train_tape_floats(R) floats; backward: gw / gb zeroed by the caller, g_x0 (R, 63) / g_cond (R, 27) may
 * be NULL.  Exact fp32 arithmetic. */
long neo_vanilla_mlp_train_tape_floats(long R);
int neo_vanilla_mlp_train_forward(neo_ctx* ctx, const float* const* w, const float* const* b, const float* x0, const float* cond,
                                  long R, float* tape, float* raw_rgb, float* raw_sigma, void* stream);
int neo_vanilla_mlp_train_backward(neo_ctx* ctx, const float* const* w, const float* x0, const float* cond, long R,
                                   const float* tape, const float* g_rgb, const float* g_sigma, float* const* gw,
                                   float* const* gb, float* g_x0, float* g_cond, void* stream);

/* Sample points of the training call and their encodings (round 5; replaces torch restatements of neo360/helper.py:24-75,
 * :401-451, util.py:52-70 in the host code): for the samples tvals (R,N) of one region - input_ch 3: inside the sphere, t along
 * the ray; 4: outside, descending inverse radius s, `far` (R) required - look (R*N,3) = the world points the features are looked up
 * at (inside o + t d; outside o + (far (1-s) + 3 s) d), x_enc (NV, R*N, 21*input_ch) = pos_enc (reference feature order,
 * helper.py:121-125) of the camera-frame point per source view (outside: [R_v x' + t_v | s], x' the inverted-sphere point).
 * The device code is the evaluators' own per-point set-up: both paths see bitwise the same points. */
int neo_tp_train_points(neo_ctx* ctx, int input_ch, const float* rays_o, const float* rays_d, const float* tvals, const float* far,
                        int R, int N, const float* src_poses, int NV, float* look, float* x_enc, void* stream);
/* The reference's output activations (neo360/model.py:380-385) as one op: rgbsigma (P,4) = (sigmoid(raw_rgb) 1.002 - 0.001,
 * softplus(raw_sigma + noise * noise_scale - 1)); noise (P) may be NULL (model.py:381-384: density_noise).  backward: gradients
 * of raw_rgb (P,3) and raw_sigma (P) from g_rgbsigma (P,4). */
int neo_tp_activate(neo_ctx* ctx, const float* raw_rgb, const float* raw_sigma, const float* noise, float noise_scale, long P,
                    float* rgbsigma, void* stream);
int neo_tp_activate_backward(neo_ctx* ctx, const float* raw_rgb, const float* raw_sigma, const float* noise, float noise_scale, long P,
                             const float* g_rgbsigma, float* g_rgb, float* g_sigma, void* stream);

/* Projected-space training (round 5): the training analogue of the evaluators' pre-projection.  NeRFPPMLP reads the 512-channel
 * latent only through W0_loc and the skip half W3_loc (neo360/model.py:123-137), and bilinear interpolation is linear, so the caller
 * forms G = latent . [W0_loc | W3_loc]^T per TEXEL (a plain (texels, 512) x (512, 256) GEMM under autograd), gathers 256 instead of
 * 512 channels and hands them to the MLP as `pre` (NV*P, 256) = the local features' contribution to the pre-activations of layer 0
 * and of the skip half of layer 3.  neo_tp_gather_map / _backward: the lookup (and its scatter-add backward) in a CALLER-OWNED
 * channels-last map (NV Hf Wf, C), C % 64 == 0, at get_local_feats' taps of the scene geometry set with neo_tp_set_scene;
 * `texels` = rows of the caller's map (and of g_map): it MUST equal NV Hf Wf of the uploaded geometry, anything else is rejected
 * (the kernels index the map from the context's geometry - a map of another resolution would be read and scattered out of bounds).
 * neo_tp_mlp_train_forward_pre / _backward_pre: as neo_tp_mlp_train_forward / _backward with `pre` in place of the local feature
 * rows; the backward returns g_pre (NV*P, 256) = [dL/dz0 | dL/dz3] and leaves the local columns of gw[0] / gw[3] untouched (their
 * gradient is formed in texel space by the caller's GEMM). */
int neo_tp_gather_map(neo_ctx* ctx, const float* map, long texels, int C, const float* pts, long P, const float* src_poses, int NV, float focal,
                      float cx, float cy, float* out, void* stream);
int neo_tp_gather_map_backward(neo_ctx* ctx, long texels, int C, const float* pts, long P, const float* src_poses, int NV, float focal, float cx,
                               float cy, const float* g_out, float* g_map, void* stream);
/* The same lookup in a C-column SLICE of a wider channels-last map (round 6): row pitch `pitch` floats, `map` / `g_map` point at the
 * slice's first column (16-byte aligned).  One merged texel-space projection (texels, 4 x 256) then serves the four MLPs of NeRF_TP:
 * each gathers - and scatters its gradient into - its own 256 columns of ONE map / ONE gradient buffer. */
int neo_tp_gather_map_slice(neo_ctx* ctx, const float* map, long texels, long pitch, int C, const float* pts, long P, const float* src_poses,
                            int NV, float focal, float cx, float cy, float* out, void* stream);
int neo_tp_gather_map_slice_backward(neo_ctx* ctx, long texels, long pitch, int C, const float* pts, long P, const float* src_poses, int NV,
                                     float focal, float cx, float cy, const float* g_out, float* g_map, void* stream);
int neo_tp_mlp_train_forward_pre(neo_ctx* ctx, int input_ch, const float* const* w, const float* const* b, const float* x_enc,
                                 const float* pre, const float* world_feat, const float* cond, int NV, long P, float* tape,
                                 float* raw_rgb, float* raw_sigma, void* stream);
int neo_tp_mlp_train_backward_pre(neo_ctx* ctx, int input_ch, const float* const* w, const float* x_enc, const float* world_feat,
                                  const float* cond, int NV, long P, const float* tape, const float* g_rgb, const float* g_sigma,
                                  float* const* gw, float* const* gb, float* g_x_enc, float* g_pre, float* g_world, void* stream);

/* One linear layer of a training chain the caller composes (models whose MLP has no fused training kernel: PixelNeRF's decoder,
 * vanilla_nerf/model_pixel.py:96-131): y (rows, out_f) = x (rows, in_f) W^T + bias [then ReLU] (accumulate != 0: added onto y
 * first), and the input gradient gx (rows, in_f) (+)= gy (rows, out_f) W.  W (out_f, in_f) with row pitch ldw (a column block of
 * a wider matrix is addressed by its pitch), fp32, exact fp32 MFMA.  With neo_linear_weight_grad these are torch's addmm and its
 * two backward products. */
int neo_linear_forward(neo_ctx* ctx, long rows, int out_f, int in_f, const float* x, long ldx, const float* w, long ldw,
                       const float* bias, int relu, int accumulate, float* y, long ldy, void* stream);
int neo_linear_input_grad(neo_ctx* ctx, long rows, int in_f, int out_f, const float* gy, long ldy, const float* w, long ldw,
                          int accumulate, float* gx, long ldx, void* stream);
/* Schedule of the projected-space NeRFPPMLP training chain (neo_tp_mlp_train_forward_pre / _backward_pre), process-wide: 1 (default) =
 * everything per row - layers 0..3, bottleneck, view layer 0; backward: the input-gradient chain g_y0 -> g_z0 and g_world - as ONE
 * kernel each way with the activation tile in LDS and every layer written to HBM once (csrc/train_chain.h); 0 = one GEMM launch per
 * layer (rounds 3-5).  Same exact-fp32 products in both; mode < 0 only queries.  Returns the previous mode. */
int neo_train_chain_mode(int mode);
/* PixelNeRF's late-fusion MLP under autograd as ONE chain each way (round 6; vanilla_nerf/model_pixel.py:96-131 inside the training
 * step :255-300): rows R = NV * P view-major; x_enc (R, 63) camera-frame encodings, pre (R, 128) = the gathered PROJECTED latent
 * W0[:, 63:575] f (the 512-wide product is formed per texel by the caller, as for NeRFPPMLP), cond (R, 27) direction encodings.
 * w / b: the nine layers in neo_pix_upload_mlp's order.  tape: neo_pix_mlp_train_tape_floats(NV, P) floats kept between forward and
 * backward.  The backward returns the nine weight / bias gradients (gw / gb ZEROED by the caller; the latent columns of gw[0] stay
 * untouched: their gradient is the texel-space GEMM's), g_pre (R, 128) = dL/dz0 and, when g_x_enc is not null, dL/dx_enc. */
long neo_pix_mlp_train_tape_floats(int NV, long P);
int neo_pix_mlp_train_forward_pre(neo_ctx* ctx, const float* const* w, const float* const* b, const float* x_enc, const float* pre,
                                  const float* cond, int NV, long P, float* tape, float* raw_rgb, float* raw_sigma, void* stream);
int neo_pix_mlp_train_backward_pre(neo_ctx* ctx, const float* const* w, const float* x_enc, const float* cond, int NV, long P,
                                   const float* tape, const float* g_rgb, const float* g_sigma, float* const* gw, float* const* gb,
                                   float* g_x_enc, float* g_pre, void* stream);
/* Mip-NeRF 360's MLPs under autograd as ONE chain each way (round 6; mipnerf360/model.py:107-176 inside the training step :236-365):
 * rows = R rays x n intervals, x0 (rows, 504) integrated encodings (neo_mip_encode; data, no gradient), d_enc (R, 27) one direction
 * encoding per ray.  width / depth = netwidth / netdepth (256 x 4 proposal, 1024 x 8 NeRF; the layer after index 4 reads [h | x0]),
 * rgb = 0 for the proposal MLPs (disable_rgb: colour columns zero).  w / b in neo_mip_upload_mlp's order.  rgbdens (rows, 4) =
 * [sigmoid(raw) (1 + 2 pad) - pad | softplus(raw_density - 1)], activated as the reference returns them.  tape: caller-owned,
 * neo_mip_mlp_train_tape_floats floats, kept with rgbdens between forward and backward.  Backward: g_rgbdens (rows, 4) = dL/d rgbdens;
 * gw / gb ZEROED by the caller. */
long neo_mip_mlp_train_tape_floats(int width, int depth, int rgb, long R, int n);
int neo_mip_mlp_train_forward(neo_ctx* ctx, int width, int depth, int rgb, const float* const* w, const float* const* b, const float* x0,
                              const float* d_enc, long R, int n, float* tape, float* rgbdens, void* stream);
int neo_mip_mlp_train_backward(neo_ctx* ctx, int width, int depth, int rgb, const float* const* w, const float* x0, const float* d_enc,
                               long R, int n, const float* tape, const float* rgbdens, const float* g_rgbdens, float* const* gw,
                               float* const* gb, void* stream);
/* neo_tp_gather_map / _backward at the PixelNeRF decoder's taps (scene geometry of neo_pix_set_scene). */
int neo_pix_gather_map(neo_ctx* ctx, const float* map, long texels, int C, const float* pts, long P, const float* src_poses, int NV, float focal,
                       float cx, float cy, float* out, void* stream);
int neo_pix_gather_map_backward(neo_ctx* ctx, long texels, int C, const float* pts, long P, const float* src_poses, int NV, float focal, float cx,
                                float cy, const float* g_out, float* g_map, void* stream);

/* Weight (and bias) gradient of a linear layer y = x W^T + b over K rows - what autograd forms for every nn.Linear of the
 * reference's MLPs (torch's addmm backward), here for the texel-space projection of training.project_latent:
 * dW (M, N; row pitch ldw) += dY^T X, db (M) += column sums of dY (db may be NULL).  dY (K, M) row pitch ldy, X (K, N) row pitch
 * ldx, fp32, exact fp32 MFMA; M <= 1024, N <= 4096 (a (texels, 256)^T (texels, 512) product runs at ~110 TFLOP/s). */
int neo_linear_weight_grad(neo_ctx* ctx, int M, int N, long K, const float* dY, long ldy, const float* X, long ldx, float* dW,
                           long ldw, float* db, void* stream);

/* Stand-alone feature lookups of the scene set with neo_tp_set_scene, view-major rows (row = v P + p):
 * world (NV*P,128) = index_grid (encoder_tp_fusion_conv.py:122-209: three planes summed), local (NV*P,512) =
 * get_local_feats / SpatialEncoder.index (neo360/model.py:239-264).  pts (P,3) world points.  local (and, in the backward, g_local
 * together with g_latent) may be NULL: the tri-planes only. */
int neo_tp_gather(neo_ctx* ctx, const float* pts, long P, const float* src_poses, int NV, float focal, float cx,
                  float cy, float* world, float* local, void* stream);
/* Its backward: g_world / g_local scattered (atomic adds) into CHANNELS-LAST gradient maps the caller zeroed:
 * g_plane_* (NV,Hp,Wp,128), g_latent (NV,Hf,Wf,512).  Points carry no gradient. */
int neo_tp_gather_backward(neo_ctx* ctx, const float* pts, long P, const float* src_poses, int NV, float focal,
                           float cx, float cy, const float* g_world, const float* g_local, float* g_plane_xz,
                           float* g_plane_xy, float* g_plane_yz, float* g_latent, void* stream);

/* NeRF_TP.forward in its TRAINING form (neo360/model.py:531-579, out_depth=False): white_bkgd honoured, per level
 * rgb (R,3), fg_weights / bg_weights (R,N), the sample rows fg_tvals / bg_tvals (R,N) the caller derives sdist from
 * (:564-571), bg_acc (R), and optionally the per-sample (rgb, sigma) of both regions (inputs of
 * neo_composite_backward).  seed != 0 = randomized=True: stratified level-0 jitter and uniform level-1 quantiles
 * from neo_rand_uniform streams 0..3; seed == 0 = randomized=False.  Any output pointer may be NULL. */
typedef struct {
    float* rgb; float* fg_weights; float* bg_weights; float* fg_tvals; float* bg_tvals; float* bg_acc;
    float* fg_rgbsigma; float* bg_rgbsigma;
} neo_tp_train_out;
int neo_tp_render_train(neo_ctx* ctx, const float* rays_o, const float* rays_d, const float* viewdirs, int R,
                        int chunk, const float* src_poses, int NV, float focal, float cx, float cy, int n_coarse,
                        int n_fine, int white_bkgd, uint64_t seed, const neo_tp_train_out* level0,
                        const neo_tp_train_out* level1, void* stream);

/* ---- PixelNeRF baseline decoder (models/vanilla_nerf/model_pixel.py) ------------------------ */
/* Upload one NeRFMLP of model_pixel.py:35-94 (slot 0 = coarse_mlp, 1 = fine_mlp).  weights/biases
 * [host arrays of 9 device pointers], order: pts_linears.0..3 (128x575, 128x128 x3), views_linear.0
 * (128x155), views_linear.1 (128x128), bottleneck_layer, density_layer, rgb_layer (3x128).
 * This evaluator exists in the split-fp16 arithmetic only: neo_ctx_set_precision(ctx, 1). */
int neo_pix_upload_mlp(neo_ctx* ctx, int slot, const float* const* weights,
                       const float* const* biases, void* stream);

/* The image encoder's output the reference recomputes per chunk (model_pixel.py:176-178):
 * latent (NV,512,Hf,Wf) NCHW, re-laid out channels-last into a context-owned buffer. */
int neo_pix_set_scene(neo_ctx* ctx, const float* latent, int NV, int Cl, int Hf, int Wf,
                      float image_w, float image_h, void* stream);

/* enable != 0 (default): the 512-channel latent is pre-projected once per (scene, MLP slot) through the latent columns
 * of pts_linears.0 (model_pixel.py:96-131 is linear in the latent up to the first ReLU) and the evaluator gathers the
 * 128-channel result (512 B per latent texel and slot of context memory); 0: the latent itself is gathered and
 * multiplied per point, the reference's operation order. */
int neo_pix_set_preproject(neo_ctx* ctx, int enable);

/* Per-point outputs at given sample positions (model_pixel.py:198-237): tvals (R,N) along rays_d;
 * out (R,N,4) = (sigmoid rgb, relu sigma).  chunk / src_poses / focal / cx / cy as for neo_pix_render. */
int neo_pix_mlp(neo_ctx* ctx, int slot, const float* rays_o, const float* rays_d,
                const float* viewdirs, const float* tvals, int R, int N, int chunk,
                const float* src_poses, int NV, float focal, float cx, float cy, float* out,
                void* stream);

/* PixelNeRF.forward decoder half (model_pixel.py:180-256), randomized=False, for R rays processed in
 * reference-sized chunks (the view-direction tiling of :219-222 makes results depend on chunk
 * membership).  src_poses [host]: NV*16 floats; focal/cx/cy: source view 0's intrinsics (:203-204;
 * both image axes use +focal).  Outputs per level (any may be NULL): rgb (R,3), acc (R), depth (R). */
int neo_pix_render(neo_ctx* ctx, const float* rays_o, const float* rays_d, const float* viewdirs,
                   int R, int chunk, const float* src_poses, int NV, float focal, float cx,
                   float cy, float near, float far, int n_coarse, int n_fine, int white_bkgd,
                   float* rgb0, float* acc0, float* depth0, float* rgb1, float* acc1,
                   float* depth1, void* stream);

/* ---- Mip-NeRF 360 (models/mipnerf360/model.py) -------------------------------- */
/* Upload one MipNeRF360MLP (model.py:30-107).  slot 0..2 = mlps.0, mlps.1 (PropMLP: width 256,
 * depth 4, rgb 0) and mlps.2 (NeRFMLP: width 1024, depth 8, rgb 1) — the shapes the kernels are
 * specialised for.  weights/biases [host arrays of device pointers], order: pts_linear.0..depth-1,
 * density_layer, then (rgb) bottleneck_layer, views_linear.0, rgb_layer.  basis: the module's
 * pos_basis_t buffer, (3,21) row-major. */
int neo_mip_upload_mlp(neo_ctx* ctx, int slot, int width, int depth, int rgb,
                       const float* const* weights, const float* const* biases,
                       const float* basis, void* stream);

/* One proposal-resampling step (model.py:258-310; helper.py:154-243, :337-394), randomized=False:
 * s_prev (R,n_prev+1) / w_prev (R,n_prev) = previous level's normalised distances and weights
 * (level 0: s_prev = [0,1], w_prev = [1]); dilate != 0 applies max_dilate_weights(dilation,
 * domain (0,1), renormalize) and the [1:-1] trims first; logits = anneal*log(w) (-inf on empty
 * intervals).  Outputs sdist (R,n+1) and tdist = 1/(s/far + (1-s)/near) (R,n+1). */
int neo_mip_resample(neo_ctx* ctx, const float* s_prev, const float* w_prev, int R, int n_prev,
                     int dilate, float dilation, float anneal, int n, float near, float far,
                     float* sdist, float* tdist, void* stream);

/* cast_rays (cone, full covariance) + MipNeRF360MLP.forward for the R*n intervals of tdist (R,n+1)
 * (helper.py:278-334, model.py:109-176): out (R,n,4) = (rgb, density); rgb = 0 for a PropMLP. */
int neo_mip_mlp(neo_ctx* ctx, int slot, const float* rays_o, const float* rays_d,
                const float* viewdirs, const float* radii, const float* tdist, int R, int n,
                float* out, void* stream);

/* Schedule of the NeRF MLP (8 x 1024, models/mipnerf360/model.py:30-120) in split-fp16 arithmetic.  mode 1: layer by
 * layer - the encodings of a batch of 16384 intervals are written once as MFMA fragments, the eight trunk layers run as
 * 256 x 256-tile GEMMs whose activations stay in L2 / Infinity Cache between layers (one weight fragment feeds 4 interval
 * tiles), the fused evaluator adds the heads and the colour branch; 160 MB of context memory.  mode 0: the fused
 * evaluator end to end (one weight fragment per 32 intervals).  mode -1 (a fresh context): layer by layer for calls of
 * 8192 intervals or more.  Same arithmetic per product, a different fp32 summation order inside a layer only through
 * the k-step order, which is the same: results agree to the last bits (tests/test_gpu_mip360.py). */
int neo_mip_set_layered(neo_ctx* ctx, int mode);

/* compute_alpha_weights(opaque_background=True) + volumetric_rendering (helper.py:246-274):
 * rgbdens (R,n,4), tdist (R,n+1) -> weights (R,n), rgb (R,3) = sum w c + max(0,1-acc)*bg. */
int neo_mip_composite(neo_ctx* ctx, const float* rgbdens, const float* tdist, const float* rays_d,
                      int R, int n, float bg, float* weights, float* rgb, void* stream);

/* Training-side operators of the Mip-NeRF 360 renderer (mipnerf360/model.py:236-365 under training_step :436-470), for a caller
 * that composes the MLPs from neo_linear_* (training.mip_render_train):
 * neo_mip_resample_u = neo_mip_resample with the caller's quantile table u (n) and, when jitter != NULL, one offset per ray added
 * to it - helper.py:358-365 with single_jitter: u = linspace(0, 1 - u_max, n) + rand(R, 1) * max_jitter.
 * neo_mip_encode: the 504-d integrated positional encoding of the R x n intervals of tdist (R, n + 1) as fp32 rows (R n, 504),
 * reference feature order (helper.py:33-88, 278-334); pos_basis_t (3, 21) device pointer (the MLP's buffer).
 * neo_mip_composite_backward: backward of neo_mip_composite - g_weights (R, n) and g_rgb (R, 3) (either may be NULL) ->
 * g_rgbdens (R, n, 4) = gradients of (rgb, density); sample positions carry no gradient (stop_level_grad). */
int neo_mip_resample_u(neo_ctx* ctx, const float* s_prev, const float* w_prev, int R, int n_prev, int dilate, float dilation,
                       float anneal, int n, const float* u, const float* jitter, float near, float far, float* sdist, float* tdist,
                       void* stream);
int neo_mip_encode(neo_ctx* ctx, const float* rays_o, const float* rays_d, const float* radii, const float* tdist,
                   const float* pos_basis_t, int R, int n, float* out, void* stream);
int neo_mip_composite_backward(neo_ctx* ctx, const float* rgbdens, const float* tdist, const float* rays_d, int R, int n, float bg,
                               const float* g_weights, const float* g_rgb, float* g_rgbdens, void* stream);

/* MipNeRF360.forward(batch, train_frac, randomized=False, is_train=False, near, far)
 * (model.py:236-365), 3 levels (n_prop, n_prop, n_nerf samples).  Per level l (any pointer may be
 * NULL): rgb_l (R,3), sdist_l (R,n_l+1), weights_l (R,n_l), rgbdens_l (R,n_l,4) = per-interval
 * (rgb, density). */
typedef struct {
    float* rgb; float* sdist; float* weights; float* rgbdens;
} neo_mip_level_out;
int neo_mip_render(neo_ctx* ctx, const float* rays_o, const float* rays_d, const float* viewdirs,
                   const float* radii, int R, float train_frac, float near, float far, int n_prop,
                   int n_nerf, const neo_mip_level_out* levels /* [3] */, void* stream);

/* ---- profiling aid ---------------------------------------------------------- */
/* Total duration (ms) of the dominant (fused MLP) kernel launches, bracketed with HIP
 * events on the stream they were launched on, since timing was enabled; with the
 * launch count, the points evaluated and their ALGORITHMIC flops (reference
 * formulation, MAC x 2).  Reading synchronises the events. */
int neo_ctx_set_timing(neo_ctx* ctx, int enable);
int neo_ctx_read_timing(neo_ctx* ctx, double* total_ms, int* launches, double* total_points,
                        double* total_flops);
/* The same launches one by one, in launch order: duration (ms), which evaluator ran (kernel_id: 0 unspecified,
 * 1 k_tp_mlp_hp, 2 k_tp_mlp_hpp, 3 k_tp_mlp_h, 4 k_tp_mlp; Mip-NeRF 360: 5 proposal MLP (fused split evaluator), 6 NeRF MLP fused,
 * 7 NeRF MLP layer by layer - one span covers all batches of the call -, 8 exact fp32), points and algorithmic flops of each.  A NeO-360 frame is four
 * launches (inside / outside the sphere x coarse / fine) and, in pre-projection mode 3, two different kernels: the bench's
 * roofline object prices each kernel with ITS launches.  Arrays may be NULL; *count = launches recorded (may exceed capacity). */
int neo_ctx_read_spans(neo_ctx* ctx, int capacity, double* ms, int* kernel_id, double* points, double* flops, int* count);

#ifdef __cplusplus
}
#endif
#endif /* NEO360_HIP_H */

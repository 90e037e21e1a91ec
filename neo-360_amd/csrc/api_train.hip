// Training-side entry points of the C ABI: counter-based uniforms, stratified samplers, the
// backward of the compositing, stand-alone feature lookups with their backward, the distortion loss, and the
// training-mode forward of NeRF_TP (out_depth=False tuple, randomized sampling, white background honoured).
#include "ctx.h"
#include "train_kernels.h"

using namespace neo_host;

namespace {

void fill_views(const float* poses, int nv, neo::TpViews& v) {       // as api_tp.hip (neo360/util.py:64-66)
    for (int i = 0; i < nv; ++i) {
        const float* m = poses + i * 16;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) v.rot[i][r * 3 + c] = m[c * 4 + r];
        for (int r = 0; r < 3; ++r) {
            float acc = v.rot[i][r * 3 + 0] * m[0 * 4 + 3];
            acc = acc + v.rot[i][r * 3 + 1] * m[1 * 4 + 3];
            acc = acc + v.rot[i][r * 3 + 2] * m[2 * 4 + 3];
            v.trans[i][r] = -acc;
        }
    }
}

}  // namespace

int neo_tp_eval_region(neo_ctx* ctx, int slot, const neo::TpScene& sc, const neo::TpViews& views, const float* rays_o,
                       const float* rays_d, const float* viewdirs, const float* tvals, const float* far, int R, int N,
                       int chunk, float* out, hipStream_t s);     // api_tp.hip

extern "C" {

int neo_transpose(neo_ctx* ctx, const float* src, long batch, int rows, int cols, float* dst, void* stream) {
    ENTER(ctx);
    REQUIRE(batch >= 0 && batch <= 65535 && rows >= 0 && cols >= 0 && (rows + 63) / 64 <= 65535, "bad shape (batch <= 65535, rows <= 4.19 M)");
    if (batch == 0 || rows == 0 || cols == 0) return NEO_OK;
    REQUIRE(src && dst && src != dst, "null pointer / in-place transpose");
    neo::launch_transpose(src, batch, rows, cols, dst, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_rand_uniform(neo_ctx* ctx, uint64_t seed, uint32_t stream_id, int rows, int cols, float* out, void* stream) {
    ENTER(ctx);
    REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    if (rows == 0 || cols == 0) return NEO_OK;
    REQUIRE(out != nullptr, "null pointer");
    neo::launch_uniform(seed, stream_id, rows, cols, out, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_tp_train_points(neo_ctx* ctx, int input_ch, const float* rays_o, const float* rays_d, const float* tvals, const float* far,
                        int R, int N, const float* src_poses, int NV, float* look, float* x_enc, void* stream) {
    ENTER(ctx);
    REQUIRE(input_ch == 3 || input_ch == 4, "input_ch must be 3 (inside the sphere) or 4 (outside)");
    REQUIRE(R >= 0 && N >= 1 && NV >= 1 && NV <= neo::TP_MAX_VIEWS, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && rays_d && tvals && src_poses && look && x_enc, "null pointer");
    REQUIRE(input_ch == 3 || far, "far required outside the sphere");
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::launch_tp_train_points(input_ch, rays_o, rays_d, tvals, far, R, N, views, NV, ctx->flags, look, x_enc,
                                static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_tp_activate(neo_ctx* ctx, const float* raw_rgb, const float* raw_sigma, const float* noise, float noise_scale, long P,
                    float* rgbsigma, void* stream) {
    ENTER(ctx);
    REQUIRE(P >= 0, "negative count");
    if (P == 0) return NEO_OK;
    REQUIRE(raw_rgb && raw_sigma && rgbsigma, "null pointer");
    neo::launch_tp_activate(raw_rgb, raw_sigma, noise, noise_scale, P, rgbsigma, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_tp_activate_backward(neo_ctx* ctx, const float* raw_rgb, const float* raw_sigma, const float* noise, float noise_scale, long P,
                             const float* g_rgbsigma, float* g_rgb, float* g_sigma, void* stream) {
    ENTER(ctx);
    REQUIRE(P >= 0, "negative count");
    if (P == 0) return NEO_OK;
    REQUIRE(raw_rgb && raw_sigma && g_rgbsigma && g_rgb && g_sigma, "null pointer");
    neo::launch_tp_activate_bwd(raw_rgb, raw_sigma, noise, noise_scale, P, g_rgbsigma, g_rgb, g_sigma, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_tp_sample_level0(neo_ctx* ctx, const float* far, int R, int n_coarse, const float* u_fg, const float* u_bg,
                         float* fg_t, float* bg_s, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(R >= 0 && n_coarse >= 3 && n_coarse <= 1023, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(far && fg_t && bg_s, "null pointer");
    REQUIRE((u_fg == nullptr) == (u_bg == nullptr), "pass both uniform arrays (randomized) or neither");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float* edges = ctx->get_edges(n_coarse, 0.0f, 1.0f, s);
    if (!edges) return fail(NEO_ERR_HIP, "constant table upload failed");
    if (u_fg) neo::launch_tp_level0_rand(far, edges, R, n_coarse + 1, 1e-4f, u_fg, u_bg, fg_t, bg_s, s);
    else neo::launch_tp_level0(far, edges, R, n_coarse + 1, 1e-4f, fg_t, bg_s, s);
    return check_launch();
}

int neo_resample_u(neo_ctx* ctx, const float* t_prev, const float* weights, const float* u, int R, int n_prev, int n_new,
                   int descending, float* t_out, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(R >= 0 && n_new >= 1, "bad shape");
    REQUIRE(n_prev >= 4 && n_prev <= 257 && n_prev + n_new <= 1024, "unsupported sample counts");
    if (R == 0) return NEO_OK;
    REQUIRE(t_prev && weights && u && t_out, "null pointer");
    if (neo::launch_resample(t_prev, n_prev, weights, u, n_new, R, n_prev, n_new, descending, t_out, static_cast<hipStream_t>(stream)))
        return fail(NEO_ERR_INVALID, "unsupported sample counts");
    return check_launch();
}

int neo_composite_backward(neo_ctx* ctx, int mode, const float* rgbsigma, const float* t, const float* rays_d,
                           const float* t_far, int R, int N, int white_bkgd, const float* g_rgb, const float* g_acc,
                           const float* g_depth, const float* g_weights, const float* g_lambda, float* g_rgbsigma,
                           void* stream) {
    ENTER(ctx);
    REQUIRE(mode >= 0 && mode <= 2, "mode must be 0, 1 or 2");
    REQUIRE(R >= 0 && N >= 1 && N <= 1024, "bad shape (N <= 1024)");
    if (R == 0) return NEO_OK;
    REQUIRE(rgbsigma && t && g_rgbsigma, "null pointer");
    REQUIRE(mode == 2 || rays_d, "rays_d required");
    REQUIRE(mode != 1 || t_far, "t_far required for mode 1");
    if (neo::launch_composite_bwd(mode, rgbsigma, t, N, rays_d, t_far, R, N, white_bkgd, g_rgb, g_acc, g_depth, g_weights,
                                  g_lambda, g_rgbsigma, static_cast<hipStream_t>(stream)))
        return fail(NEO_ERR_INVALID, "unsupported sample count");
    return check_launch();
}

int neo_distloss(neo_ctx* ctx, const float* w, const float* m, int R, int N, float interval, float* loss_rays,
                 float* grad_w, void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0 && N >= 1, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(w && m && (loss_rays || grad_w), "null pointer");
    neo::launch_distloss(w, m, R, N, interval, loss_rays, grad_w, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_tp_gather(neo_ctx* ctx, const float* pts, long P, const float* src_poses, int NV, float focal, float cx, float cy,
                  float* world, float* local, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(P >= 0, "negative point count");
    if (P == 0) return NEO_OK;
    REQUIRE(pts && src_poses && world, "null pointer");            // local may be NULL: tri-planes only
    if (!ctx->scene_ready) return fail(NEO_ERR_STATE, "scene features not set (neo_tp_set_scene)");
    REQUIRE(NV == ctx->scene.nv, "NV differs from the uploaded scene");
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    neo::launch_gather(sc, views, pts, P, world, local, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_tp_gather_backward(neo_ctx* ctx, const float* pts, long P, const float* src_poses, int NV, float focal, float cx,
                           float cy, const float* g_world, const float* g_local, float* g_plane_xz, float* g_plane_xy,
                           float* g_plane_yz, float* g_latent, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(P >= 0, "negative point count");
    if (P == 0) return NEO_OK;
    REQUIRE(pts && src_poses && g_world && g_plane_xz && g_plane_xy && g_plane_yz, "null pointer");
    REQUIRE((g_local == nullptr) == (g_latent == nullptr), "g_local and g_latent go together (both NULL: tri-planes only)");
    if (!ctx->scene_ready) return fail(NEO_ERR_STATE, "scene features not set (neo_tp_set_scene)");
    REQUIRE(NV == ctx->scene.nv, "NV differs from the uploaded scene");
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    neo::launch_gather_bwd(sc, views, pts, P, g_world, g_local, g_plane_xz, g_plane_xy, g_plane_yz, g_latent,
                           static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_tp_gather_map(neo_ctx* ctx, const float* map, long texels, int C, const float* pts, long P, const float* src_poses, int NV, float focal,
                      float cx, float cy, float* out, void* stream) {
    ENTER(ctx);
    REQUIRE(P >= 0 && C >= 64 && C <= 1024 && C % 64 == 0, "bad shape (C a multiple of 64, <= 1024)");
    if (P == 0) return NEO_OK;
    REQUIRE(map && pts && src_poses && out, "null pointer");
    if (!ctx->scene_ready) return fail(NEO_ERR_STATE, "scene geometry not set (neo_tp_set_scene)");
    REQUIRE(NV == ctx->scene.nv, "NV differs from the uploaded scene");
    REQUIRE(texels == static_cast<long>(ctx->scene.nv) * ctx->scene.Hf * ctx->scene.Wf,
            "map rows differ from NV*Hf*Wf of the uploaded scene geometry (stale scene, or a map of another resolution)");
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    neo::launch_map_gather(sc, views, pts, P, map, C, out, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_tp_gather_map_backward(neo_ctx* ctx, long texels, int C, const float* pts, long P, const float* src_poses, int NV, float focal, float cx,
                               float cy, const float* g_out, float* g_map, void* stream) {
    ENTER(ctx);
    REQUIRE(P >= 0 && C >= 64 && C <= 1024 && C % 64 == 0, "bad shape (C a multiple of 64, <= 1024)");
    if (P == 0) return NEO_OK;
    REQUIRE(pts && src_poses && g_out && g_map, "null pointer");
    if (!ctx->scene_ready) return fail(NEO_ERR_STATE, "scene geometry not set (neo_tp_set_scene)");
    REQUIRE(NV == ctx->scene.nv, "NV differs from the uploaded scene");
    REQUIRE(texels == static_cast<long>(ctx->scene.nv) * ctx->scene.Hf * ctx->scene.Wf,
            "map rows differ from NV*Hf*Wf of the uploaded scene geometry (stale scene, or a map of another resolution)");
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    neo::launch_map_gather_bwd(sc, views, pts, P, g_out, C, g_map, static_cast<hipStream_t>(stream));
    return check_launch();
}

// a C-column slice of a wider map (row pitch `pitch` floats; `map` / `g_map` point at the slice's first column): one merged texel-space
// projection serves the four MLPs of NeRF_TP (round 6)
int neo_tp_gather_map_slice(neo_ctx* ctx, const float* map, long texels, long pitch, int C, const float* pts, long P, const float* src_poses,
                            int NV, float focal, float cx, float cy, float* out, void* stream) {
    ENTER(ctx);
    REQUIRE(P >= 0 && C >= 64 && C <= 1024 && C % 64 == 0 && pitch >= C && pitch % 4 == 0, "bad shape (C a multiple of 64, <= 1024; pitch >= C, a multiple of 4)");
    if (P == 0) return NEO_OK;
    REQUIRE(map && pts && src_poses && out, "null pointer");
    REQUIRE((reinterpret_cast<uintptr_t>(map) & 15) == 0, "the slice must start at a 16-byte boundary");
    if (!ctx->scene_ready) return fail(NEO_ERR_STATE, "scene geometry not set (neo_tp_set_scene)");
    REQUIRE(NV == ctx->scene.nv, "NV differs from the uploaded scene");
    REQUIRE(texels == static_cast<long>(ctx->scene.nv) * ctx->scene.Hf * ctx->scene.Wf,
            "map rows differ from NV*Hf*Wf of the uploaded scene geometry (stale scene, or a map of another resolution)");
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    neo::launch_map_gather(sc, views, pts, P, map, C, out, static_cast<hipStream_t>(stream), pitch);
    return check_launch();
}

int neo_tp_gather_map_slice_backward(neo_ctx* ctx, long texels, long pitch, int C, const float* pts, long P, const float* src_poses, int NV,
                                     float focal, float cx, float cy, const float* g_out, float* g_map, void* stream) {
    ENTER(ctx);
    REQUIRE(P >= 0 && C >= 64 && C <= 1024 && C % 64 == 0 && pitch >= C, "bad shape (C a multiple of 64, <= 1024; pitch >= C)");
    if (P == 0) return NEO_OK;
    REQUIRE(pts && src_poses && g_out && g_map, "null pointer");
    if (!ctx->scene_ready) return fail(NEO_ERR_STATE, "scene geometry not set (neo_tp_set_scene)");
    REQUIRE(NV == ctx->scene.nv, "NV differs from the uploaded scene");
    REQUIRE(texels == static_cast<long>(ctx->scene.nv) * ctx->scene.Hf * ctx->scene.Wf,
            "map rows differ from NV*Hf*Wf of the uploaded scene geometry (stale scene, or a map of another resolution)");
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    neo::launch_map_gather_bwd(sc, views, pts, P, g_out, C, g_map, static_cast<hipStream_t>(stream), pitch);
    return check_launch();
}

// the same lookup at the PixelNeRF decoder's taps (geometry of neo_pix_set_scene: (f, f) projection, model_pixel.py:198-206)
int neo_pix_gather_map(neo_ctx* ctx, const float* map, long texels, int C, const float* pts, long P, const float* src_poses, int NV, float focal,
                      float cx, float cy, float* out, void* stream) {
    ENTER(ctx);
    REQUIRE(P >= 0 && C >= 64 && C <= 1024 && C % 64 == 0, "bad shape (C a multiple of 64, <= 1024)");
    if (P == 0) return NEO_OK;
    REQUIRE(map && pts && src_poses && out, "null pointer");
    if (!ctx->pix_scene_ready) return fail(NEO_ERR_STATE, "scene geometry not set (neo_pix_set_scene)");
    REQUIRE(NV == ctx->pix_scene.nv, "NV differs from the uploaded scene");
    REQUIRE(texels == static_cast<long>(ctx->pix_scene.nv) * ctx->pix_scene.Hf * ctx->pix_scene.Wf,
            "map rows differ from NV*Hf*Wf of the uploaded scene geometry (stale scene, or a map of another resolution)");
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->pix_scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    neo::launch_map_gather(sc, views, pts, P, map, C, out, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_pix_gather_map_backward(neo_ctx* ctx, long texels, int C, const float* pts, long P, const float* src_poses, int NV, float focal, float cx,
                               float cy, const float* g_out, float* g_map, void* stream) {
    ENTER(ctx);
    REQUIRE(P >= 0 && C >= 64 && C <= 1024 && C % 64 == 0, "bad shape (C a multiple of 64, <= 1024)");
    if (P == 0) return NEO_OK;
    REQUIRE(pts && src_poses && g_out && g_map, "null pointer");
    if (!ctx->pix_scene_ready) return fail(NEO_ERR_STATE, "scene geometry not set (neo_pix_set_scene)");
    REQUIRE(NV == ctx->pix_scene.nv, "NV differs from the uploaded scene");
    REQUIRE(texels == static_cast<long>(ctx->pix_scene.nv) * ctx->pix_scene.Hf * ctx->pix_scene.Wf,
            "map rows differ from NV*Hf*Wf of the uploaded scene geometry (stale scene, or a map of another resolution)");
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->pix_scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    neo::launch_map_gather_bwd(sc, views, pts, P, g_out, C, g_map, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_tp_render_train(neo_ctx* ctx, const float* rays_o, const float* rays_d, const float* viewdirs, int R, int chunk,
                        const float* src_poses, int NV, float focal, float cx, float cy, int n_coarse, int n_fine,
                        int white_bkgd, uint64_t seed, const neo_tp_train_out* level0, const neo_tp_train_out* level1,
                        void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0 && chunk >= 1, "bad ray count / chunk");
    REQUIRE(n_coarse >= 3 && n_coarse <= 256 && n_fine >= 1 && n_coarse + 1 + n_fine <= 1024, "unsupported sample counts");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && rays_d && viewdirs && src_poses, "null pointer");
    if (!ctx->scene_ready) return fail(NEO_ERR_STATE, "scene features not set (neo_tp_set_scene)");
    REQUIRE(NV == ctx->scene.nv, "NV differs from the uploaded scene");
    hipStream_t s = static_cast<hipStream_t>(stream);
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    const int N0 = n_coarse + 1, N1 = N0 + n_fine;
    const float* edges = ctx->get_edges(n_coarse, 0.0f, 1.0f, s);
    const float* u_det = ctx->get_quantiles(n_fine, s);
    if (!edges || !u_det) return fail(NEO_ERR_HIP, "constant table upload failed");
    ORDERED(ctx, static_cast<hipStream_t>(stream));
    auto* W = ctx->ws;
    const size_t r = static_cast<size_t>(R);
    const size_t nu = seed ? r * (N0 > n_fine ? N0 : n_fine) * 4 : 4;
    if (W[0].reserve(r * 4) || W[1].reserve(r * N0 * 4) || W[2].reserve(r * N0 * 4) || W[3].reserve(r * N1 * 16) ||
        W[4].reserve(r * N1 * 16) || W[5].reserve(r * N1 * 4) || W[6].reserve(r * N1 * 4) || W[7].reserve(r * N1 * 4) ||
        W[8].reserve(r * N1 * 4) || W[9].reserve(r * 16 * 4) || W[10].reserve(nu) || W[11].reserve(nu))
        return NEO_ERR_NOMEM;
    float* far = W[0].as<float>();
    float* fg_t0 = W[1].as<float>();
    float* bg_s0 = W[2].as<float>();
    float* fg_out = W[3].as<float>();
    float* bg_out = W[4].as<float>();
    float* fg_w_ws = W[5].as<float>();
    float* bg_w_ws = W[6].as<float>();
    float* fg_t1 = W[7].as<float>();
    float* bg_s1 = W[8].as<float>();
    float* scratch = W[9].as<float>();
    float* ua = W[10].as<float>();
    float* ub = W[11].as<float>();
    neo::launch_sphere(rays_o, rays_d, R, far, nullptr, ctx->flags, s);
    // randomized=True (seed != 0): Philox streams 0/1 = level-0 jitter fg/bg, 2/3 = level-1 quantiles fg/bg
    if (seed) {
        neo::launch_uniform(seed, 0, R, N0, ua, s);
        neo::launch_uniform(seed, 1, R, N0, ub, s);
        neo::launch_tp_level0_rand(far, edges, R, N0, 1e-4f, ua, ub, fg_t0, bg_s0, s);
    } else {
        neo::launch_tp_level0(far, edges, R, N0, 1e-4f, fg_t0, bg_s0, s);
    }
    const float* fg_t = fg_t0;
    const float* bg_s = bg_s0;
    for (int level = 0; level < 2; ++level) {
        const int N = level == 0 ? N0 : N1;
        const neo_tp_train_out* lo = level == 0 ? level0 : level1;
        if (int rc = neo_tp_eval_region(ctx, level, sc, views, rays_o, rays_d, viewdirs, fg_t, nullptr, R, N, chunk, fg_out, s)) return rc;
        if (int rc = neo_tp_eval_region(ctx, 2 + level, sc, views, rays_o, rays_d, viewdirs, bg_s, far, R, N, chunk, bg_out, s)) return rc;
        float* fg_w = (lo && lo->fg_weights) ? lo->fg_weights : fg_w_ws;
        float* bg_w = (lo && lo->bg_weights) ? lo->bg_weights : bg_w_ws;
        float* fg_rgb = scratch;
        float* lam = scratch + r * 3;
        float* bg_rgb = scratch + r * 4;
        // training tuple (neo360/model.py:531-579): white_bkgd is honoured in both regions
        neo::launch_composite(1, fg_out, fg_t, N, rays_d, far, R, N, white_bkgd, fg_rgb, nullptr, nullptr, fg_w, lam, s);
        neo::launch_composite(2, bg_out, bg_s, N, nullptr, nullptr, R, N, white_bkgd, bg_rgb, lo ? lo->bg_acc : nullptr, nullptr, bg_w,
                              nullptr, s);
        if (lo && lo->rgb) neo::launch_tp_merge(fg_rgb, nullptr, lam, bg_rgb, nullptr, R, lo->rgb, nullptr, s);
        if (lo && lo->fg_tvals) HIP_TRY(hipMemcpyAsync(lo->fg_tvals, fg_t, r * N * 4, hipMemcpyDeviceToDevice, s));
        if (lo && lo->bg_tvals) HIP_TRY(hipMemcpyAsync(lo->bg_tvals, bg_s, r * N * 4, hipMemcpyDeviceToDevice, s));
        if (lo && lo->fg_rgbsigma) HIP_TRY(hipMemcpyAsync(lo->fg_rgbsigma, fg_out, r * N * 16, hipMemcpyDeviceToDevice, s));
        if (lo && lo->bg_rgbsigma) HIP_TRY(hipMemcpyAsync(lo->bg_rgbsigma, bg_out, r * N * 16, hipMemcpyDeviceToDevice, s));
        if (level == 0) {
            const float* u_fg = u_det;
            const float* u_bg = u_det;
            int ustride = 0;
            if (seed) {
                neo::launch_uniform(seed, 2, R, n_fine, ua, s);
                neo::launch_uniform(seed, 3, R, n_fine, ub, s);
                u_fg = ua; u_bg = ub; ustride = n_fine;
            }
            if (neo::launch_resample(fg_t0, N0, fg_w, u_fg, ustride, R, N0, n_fine, 0, fg_t1, s) ||
                neo::launch_resample(bg_s0, N0, bg_w, u_bg, ustride, R, N0, n_fine, 1, bg_s1, s))
                return fail(NEO_ERR_INVALID, "unsupported sample counts");
            fg_t = fg_t1;
            bg_s = bg_s1;
        }
    }
    return check_launch();
}


long neo_tp_mlp_train_tape_floats(int NV, long P) { return NV >= 1 && P >= 0 ? (long)neo::tp_train_tape_floats(NV, P) : 0; }

int neo_tp_mlp_train_forward(neo_ctx* ctx, int input_ch, const float* const* w, const float* const* b, const float* x_enc,
                             const float* local_feat, const float* world_feat, const float* cond, int NV, long P, float* tape,
                             float* raw_rgb, float* raw_sigma, void* stream) {
    ENTER(ctx);
    REQUIRE(input_ch == 3 || input_ch == 4, "input_ch must be 3 (inside the sphere) or 4 (outside)");
    REQUIRE(NV >= 1 && P >= 0, "bad shape");
    if (P == 0) return NEO_OK;
    REQUIRE((long)NV * P <= 4190000L, "at most 4.19 M rows (point-views) per call");
    REQUIRE(w && b && x_enc && local_feat && world_feat && cond && tape && raw_rgb && raw_sigma, "null pointer");
    for (int i = 0; i < 9; ++i) REQUIRE(w[i] && b[i], "null weight / bias pointer");
    neo::launch_tp_train_forward(input_ch * 21, w, b, x_enc, local_feat, world_feat, cond, NV, P, tape, raw_rgb, raw_sigma,
                                 static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_tp_mlp_train_backward(neo_ctx* ctx, int input_ch, const float* const* w, const float* x_enc, const float* local_feat,
                              const float* world_feat, const float* cond, int NV, long P, const float* tape,
                              const float* g_rgb, const float* g_sigma, float* const* gw, float* const* gb, float* g_x_enc,
                              float* g_local, float* g_world, void* stream) {
    ENTER(ctx);
    REQUIRE(NV >= 1 && P >= 0, "bad shape");
    if (P == 0) return NEO_OK;
    REQUIRE(input_ch == 3 || input_ch == 4, "input_ch must be 3 (inside the sphere) or 4 (outside)");
    REQUIRE((long)NV * P <= 4190000L, "at most 4.19 M rows (point-views) per call");
    REQUIRE(w && x_enc && local_feat && world_feat && cond && tape && g_rgb && g_sigma && gw && gb, "null pointer");
    for (int i = 0; i < 9; ++i) REQUIRE(w[i] && gw[i] && gb[i], "null weight / gradient pointer");
    ORDERED(ctx, static_cast<hipStream_t>(stream));
    if (ctx->train_scratch.reserve(neo::tp_train_scratch_floats(NV, P) * sizeof(float))) return NEO_ERR_NOMEM;
    neo::launch_tp_train_backward(input_ch * 21, w, x_enc, local_feat, world_feat, cond, NV, P, tape,
                                  ctx->train_scratch.as<float>(), g_rgb, g_sigma, gw, gb, g_x_enc, g_local, g_world,
                                  static_cast<hipStream_t>(stream));
    return check_launch();
}


int neo_tp_mlp_train_forward_pre(neo_ctx* ctx, int input_ch, const float* const* w, const float* const* b, const float* x_enc,
                                 const float* pre, const float* world_feat, const float* cond, int NV, long P, float* tape,
                                 float* raw_rgb, float* raw_sigma, void* stream) {
    ENTER(ctx);
    REQUIRE(input_ch == 3 || input_ch == 4, "input_ch must be 3 (inside the sphere) or 4 (outside)");
    REQUIRE(NV >= 1 && P >= 0, "bad shape");
    if (P == 0) return NEO_OK;
    REQUIRE((long)NV * P <= 4190000L, "at most 4.19 M rows (point-views) per call");
    REQUIRE(w && b && x_enc && pre && world_feat && cond && tape && raw_rgb && raw_sigma, "null pointer");
    for (int i = 0; i < 9; ++i) REQUIRE(w[i] && b[i], "null weight / bias pointer");
    neo::launch_tp_train_forward(input_ch * 21, w, b, x_enc, nullptr, world_feat, cond, NV, P, tape, raw_rgb, raw_sigma,
                                 static_cast<hipStream_t>(stream), pre);
    return check_launch();
}

int neo_tp_mlp_train_backward_pre(neo_ctx* ctx, int input_ch, const float* const* w, const float* x_enc, const float* world_feat,
                                  const float* cond, int NV, long P, const float* tape, const float* g_rgb, const float* g_sigma,
                                  float* const* gw, float* const* gb, float* g_x_enc, float* g_pre, float* g_world, void* stream) {
    ENTER(ctx);
    REQUIRE(NV >= 1 && P >= 0, "bad shape");
    if (P == 0) return NEO_OK;
    REQUIRE(input_ch == 3 || input_ch == 4, "input_ch must be 3 (inside the sphere) or 4 (outside)");
    REQUIRE((long)NV * P <= 4190000L, "at most 4.19 M rows (point-views) per call");
    REQUIRE(w && x_enc && world_feat && cond && tape && g_rgb && g_sigma && gw && gb && g_pre, "null pointer");
    for (int i = 0; i < 9; ++i) REQUIRE(w[i] && gw[i] && gb[i], "null weight / gradient pointer");
    ORDERED(ctx, static_cast<hipStream_t>(stream));
    if (ctx->train_scratch.reserve(neo::tp_train_scratch_floats(NV, P) * sizeof(float))) return NEO_ERR_NOMEM;
    neo::launch_tp_train_backward(input_ch * 21, w, x_enc, nullptr, world_feat, cond, NV, P, tape, ctx->train_scratch.as<float>(),
                                  g_rgb, g_sigma, gw, gb, g_x_enc, nullptr, g_world, static_cast<hipStream_t>(stream), g_pre);
    return check_launch();
}

int neo_train_chain_mode(int mode) { return neo::train_chain_mode(mode); }

long neo_pix_mlp_train_tape_floats(int NV, long P) { return (NV >= 1 && P >= 0) ? (long)neo::pix_train_tape_floats(NV, P) : 0; }

int neo_pix_mlp_train_forward_pre(neo_ctx* ctx, const float* const* w, const float* const* b, const float* x_enc, const float* pre,
                                  const float* cond, int NV, long P, float* tape, float* raw_rgb, float* raw_sigma, void* stream) {
    ENTER(ctx);
    REQUIRE(NV >= 1 && P >= 0, "bad shape");
    if (P == 0) return NEO_OK;
    REQUIRE((long)NV * P <= 4190000L, "at most 4.19 M rows (point-views) per call");
    REQUIRE(w && b && x_enc && pre && cond && tape && raw_rgb && raw_sigma, "null pointer");
    for (int i = 0; i < 9; ++i) REQUIRE(w[i] && b[i], "null weight / bias pointer");
    neo::launch_pix_train_forward(w, b, x_enc, pre, cond, NV, P, tape, raw_rgb, raw_sigma, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_pix_mlp_train_backward_pre(neo_ctx* ctx, const float* const* w, const float* x_enc, const float* cond, int NV, long P,
                                   const float* tape, const float* g_rgb, const float* g_sigma, float* const* gw, float* const* gb,
                                   float* g_x_enc, float* g_pre, void* stream) {
    ENTER(ctx);
    REQUIRE(NV >= 1 && P >= 0, "bad shape");
    if (P == 0) return NEO_OK;
    REQUIRE((long)NV * P <= 4190000L, "at most 4.19 M rows (point-views) per call");
    REQUIRE(w && x_enc && cond && tape && g_rgb && g_sigma && gw && gb && g_pre, "null pointer");
    for (int i = 0; i < 9; ++i) REQUIRE(w[i] && gw[i] && gb[i], "null weight / gradient pointer");
    ORDERED(ctx, static_cast<hipStream_t>(stream));
    if (ctx->train_scratch.reserve(neo::pix_train_scratch_floats(NV, P) * sizeof(float))) return NEO_ERR_NOMEM;
    neo::launch_pix_train_backward(w, x_enc, cond, NV, P, tape, ctx->train_scratch.as<float>(), g_rgb, g_sigma, gw, gb, g_x_enc, g_pre,
                                   static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_linear_forward(neo_ctx* ctx, long rows, int out_f, int in_f, const float* x, long ldx, const float* w, long ldw,
                       const float* bias, int relu, int accumulate, float* y, long ldy, void* stream) {
    ENTER(ctx);
    // the GEMM's grid.y is rows / 128 and HIP caps it at 65,535 (a larger call used to fail with a generic launch error)
    REQUIRE(rows >= 0 && rows <= 65535L * 128 && out_f >= 1 && out_f <= 4096 && in_f >= 1 && in_f <= 4096,
            "bad shape (rows <= 8,388,480 per call: split larger matrices by rows; features <= 4096)");
    REQUIRE(ldx >= in_f && ldw >= in_f && ldy >= out_f, "row pitch smaller than the row");
    if (rows == 0) return NEO_OK;
    REQUIRE(x && w && y, "null pointer");
    neo::launch_linear_forward(rows, out_f, in_f, x, ldx, w, ldw, bias, relu != 0, accumulate != 0, y, ldy, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_linear_input_grad(neo_ctx* ctx, long rows, int in_f, int out_f, const float* gy, long ldy, const float* w, long ldw,
                          int accumulate, float* gx, long ldx, void* stream) {
    ENTER(ctx);
    REQUIRE(rows >= 0 && rows <= 65535L * 128 && out_f >= 1 && out_f <= 4096 && in_f >= 1 && in_f <= 4096,
            "bad shape (rows <= 8,388,480 per call: split larger matrices by rows; features <= 4096)");
    REQUIRE(ldy >= out_f && ldw >= in_f && ldx >= in_f, "row pitch smaller than the row");
    if (rows == 0) return NEO_OK;
    REQUIRE(gy && w && gx, "null pointer");
    neo::launch_linear_input_grad(rows, in_f, out_f, gy, ldy, w, ldw, accumulate != 0, gx, ldx, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_linear_weight_grad(neo_ctx* ctx, int M, int N, long K, const float* dY, long ldy, const float* X, long ldx, float* dW,
                           long ldw, float* db, void* stream) {
    ENTER(ctx);
    REQUIRE(M >= 1 && M <= 1024 && N >= 1 && N <= 4096 && K >= 0 && K <= 2000000000L, "bad shape (M <= 1024, N <= 4096)");
    REQUIRE(ldy >= M && ldx >= N && ldw >= N, "row pitch smaller than the row");
    if (K == 0) return NEO_OK;
    REQUIRE(dY && X && dW, "null pointer");
    ORDERED(ctx, static_cast<hipStream_t>(stream));
    if (ctx->train_scratch.reserve(neo::weight_grad_scratch_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    neo::launch_weight_grad(M, N, (int)K, dY, ldy, X, ldx, dW, ldw, db, ctx->train_scratch.as<float>(), static_cast<hipStream_t>(stream));
    return check_launch();
}

long neo_vanilla_mlp_train_tape_floats(long R) { return R >= 0 ? (long)neo::vanilla_train_tape_floats(R) : 0; }

int neo_vanilla_mlp_train_forward(neo_ctx* ctx, const float* const* w, const float* const* b, const float* x0, const float* cond,
                                  long R, float* tape, float* raw_rgb, float* raw_sigma, void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0 && R <= 4190000L, "0 <= rows <= 4.19 M per call");
    if (R == 0) return NEO_OK;
    REQUIRE(w && b && x0 && cond && tape && raw_rgb && raw_sigma, "null pointer");
    for (int i = 0; i < 12; ++i) REQUIRE(w[i] && b[i], "null weight / bias pointer");
    neo::launch_vanilla_train_forward(w, b, x0, cond, R, tape, raw_rgb, raw_sigma, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_vanilla_mlp_train_backward(neo_ctx* ctx, const float* const* w, const float* x0, const float* cond, long R,
                                   const float* tape, const float* g_rgb, const float* g_sigma, float* const* gw,
                                   float* const* gb, float* g_x0, float* g_cond, void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0 && R <= 4190000L, "0 <= rows <= 4.19 M per call");
    if (R == 0) return NEO_OK;
    REQUIRE(w && x0 && cond && tape && g_rgb && g_sigma && gw && gb, "null pointer");
    for (int i = 0; i < 12; ++i) REQUIRE(w[i] && gw[i] && gb[i], "null weight / gradient pointer");
    ORDERED(ctx, static_cast<hipStream_t>(stream));
    if (ctx->train_scratch.reserve(neo::vanilla_train_scratch_floats(R) * sizeof(float))) return NEO_ERR_NOMEM;
    neo::launch_vanilla_train_backward(w, x0, cond, R, tape, ctx->train_scratch.as<float>(), g_rgb, g_sigma, gw, gb, g_x0, g_cond,
                                       static_cast<hipStream_t>(stream));
    return check_launch();
}

long neo_mip_mlp_train_tape_floats(int width, int depth, int rgb, long R, int n) {
    return (R >= 0 && n >= 1 && width >= 1 && depth >= 1) ? (long)neo::mip_train_tape_floats(width, depth, rgb, R * n, R) : 0;
}

static int mip_train_shape_ok(int width, int depth, long R, int n) {
    return width >= 64 && width <= 1024 && width % 64 == 0 && depth >= 1 && depth <= 8 && R >= 0 && n >= 1 && R * (long)n <= 4190000L;
}

int neo_mip_mlp_train_forward(neo_ctx* ctx, int width, int depth, int rgb, const float* const* w, const float* const* b, const float* x0,
                              const float* d_enc, long R, int n, float* tape, float* rgbdens, void* stream) {
    ENTER(ctx);
    REQUIRE(mip_train_shape_ok(width, depth, R, n), "width in [64, 1024] (multiple of 64), depth in [1, 8], at most 4.19 M rows (intervals) per call");
    if (R == 0) return NEO_OK;
    REQUIRE(w && b && x0 && tape && rgbdens && (!rgb || d_enc), "null pointer");
    for (int i = 0; i < depth + (rgb ? 4 : 1); ++i) REQUIRE(w[i] && b[i], "null weight / bias pointer");
    neo::launch_mip_train_forward(width, depth, rgb ? 1 : 0, w, b, x0, d_enc, R, n, tape, rgbdens, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_mip_mlp_train_backward(neo_ctx* ctx, int width, int depth, int rgb, const float* const* w, const float* x0, const float* d_enc,
                               long R, int n, const float* tape, const float* rgbdens, const float* g_rgbdens, float* const* gw,
                               float* const* gb, void* stream) {
    ENTER(ctx);
    REQUIRE(mip_train_shape_ok(width, depth, R, n), "width in [64, 1024] (multiple of 64), depth in [1, 8], at most 4.19 M rows (intervals) per call");
    if (R == 0) return NEO_OK;
    REQUIRE(w && x0 && tape && rgbdens && g_rgbdens && gw && gb && (!rgb || d_enc), "null pointer");
    for (int i = 0; i < depth + (rgb ? 4 : 1); ++i) REQUIRE(w[i] && gw[i] && gb[i], "null weight / gradient pointer");
    ORDERED(ctx, static_cast<hipStream_t>(stream));
    if (ctx->train_scratch.reserve(neo::mip_train_scratch_floats(width, rgb ? 1 : 0, R * n, R) * sizeof(float))) return NEO_ERR_NOMEM;
    neo::launch_mip_train_backward(width, depth, rgb ? 1 : 0, w, x0, d_enc, R, n, tape, ctx->train_scratch.as<float>(), rgbdens, g_rgbdens,
                                   gw, gb, static_cast<hipStream_t>(stream));
    return check_launch();
}

}  // extern "C"

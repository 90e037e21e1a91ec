// PixelNeRF baseline decoder entry points of the C ABI (models/vanilla_nerf/model_pixel.py:133-258).
#include "ctx.h"

using namespace neo_host;

namespace {

// algorithmic MACs (model_pixel.py:35-131): per point-view 575*128 + 3*128^2 + 128^2 (bottleneck) + 155*128 =
// 158,976; per point density 128 + 128*128 + rgb 384 = 16,896
double pix_flop_per_point(int nv) { return 2.0 * (nv * 158976.0 + 16896.0); }

// rot = c2w[:3,:3]^T ; trans = -rot @ c2w[:3,3]   (vanilla_nerf/util.py:20-34), fp32
void fill_views(const float* poses, int nv, neo::TpViews& v) {
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

int pix_launch(neo_ctx* ctx, int slot, const neo::TpScene& sc, const neo::TpViews& views, const float* rays_o,
               const float* rays_d, const float* viewdirs, const float* tvals, int t_shared, int R, int N, int chunk,
               float* out, hipStream_t s) {
    MlpSlot& sl = ctx->pix[slot];
    if (!sl.ready) return fail(NEO_ERR_STATE, "PixelNeRF MLP slot %d has no weights", slot);
    if (ctx->precision != 1) {
        // exact fp32 MFMA, the reference's operation order up to the folded view layer (mlp_pix.hip): what a range-guard retry
        // of the split evaluator lands on
        neo::TpMlpDev mf{sl.wpack.as<float>(), sl.bias_hp.as<float>(), sl.heads.as<float>()};
        ctx->span_kernel_next = 9;
        ctx->span_begin(s);
        neo::launch_pix_mlp(mf, sc, views, rays_o, rays_d, viewdirs, tvals, t_shared, R, N, chunk, out, s);
        ctx->span_end(s, static_cast<double>(R) * N, pix_flop_per_point(sc.nv));
        return NEO_OK;
    }
    guard_split_weights(sl, sl.wpack_h.p, neo::pix_wpack_h_bytes(), ctx->flags, s);
    if (!ctx->pix_latent_checked) {
        neo::launch_f32_range_check(sc.latent, static_cast<size_t>(sc.nv) * sc.Hf * sc.Wf * 512, 65504.0f, ctx->flags, s);
        ctx->pix_latent_checked = true;
    }
    const float* proj = nullptr;
    if (ctx->pix_preproject) {
        // latent pre-projected through this slot's pts_linears.0 latent columns (exact fp32 MFMA): rebuilt only when the
        // scene or the slot's weights changed since the last launch
        if (sl.proj_weights != sl.weights_epoch || sl.proj_scene != ctx->pix_scene_epoch) {
            if (int rc = ctx->touch_shared(s)) return rc;          // shared by both scratch lanes
            const long texels = static_cast<long>(sc.nv) * sc.Hf * sc.Wf;
            if (sl.proj.reserve(static_cast<size_t>(texels) * 512)) return NEO_ERR_NOMEM;
            neo::launch_tp_preproject(sc.latent, texels, sl.wpack.as<float>(), neo::pix_kc_x(), sl.proj.as<float>(), s, 128);
            sl.proj_weights = sl.weights_epoch;
            sl.proj_scene = ctx->pix_scene_epoch;
        }
        proj = sl.proj.as<float>();
    }
    neo::TpMlpHDev mh{sl.wpack_h.p, sl.bias.as<float>(), sl.heads.as<float>(), ctx->flags};
    ctx->span_begin(s);
    neo::launch_pix_mlp_h(mh, proj, sc, views, rays_o, rays_d, viewdirs, tvals, t_shared, R, N, chunk, out, s);
    ctx->span_end(s, static_cast<double>(R) * N, pix_flop_per_point(sc.nv));
    return NEO_OK;
}

}  // namespace

extern "C" {

int neo_pix_upload_mlp(neo_ctx* ctx, int slot, const float* const* weights, const float* const* biases,
                       void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(slot == 0 || slot == 1, "slot must be 0 (coarse_mlp) or 1 (fine_mlp)");
    REQUIRE(weights && biases, "null pointer table");
    for (int i = 0; i < 9; ++i) REQUIRE(weights[i] && biases[i], "null layer pointer");
    MlpSlot& sl = ctx->pix[slot];
    if (sl.wpack_h.reserve(neo::pix_wpack_h_bytes())) return NEO_ERR_NOMEM;
    if (sl.bias.reserve(neo::pix_bias_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.heads.reserve(neo::pix_heads_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.wpack.reserve(neo::pix_wpack_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.fold_ws.reserve(neo::pix_fold_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.bias_hp.reserve(neo::pix_bias_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    neo::launch_pix_pack_h(weights, biases, sl.wpack_h.p, sl.bias.as<float>(), sl.heads.as<float>(), sl.fold_ws.as<float>(),
                           static_cast<hipStream_t>(stream));
    // fp32 fragments of the exact evaluator (mlp_pix.hip); the first 64 k-chunks of its first stage are the latent columns of
    // pts_linears.0 that k_tp_preproject multiplies the scene latent with (pre-projection of the split evaluator)
    neo::launch_pix_pack(weights, biases, sl.fold_ws.as<float>(), sl.bias.as<float>(), sl.bias_hp.as<float>(), sl.wpack.as<float>(),
                         static_cast<hipStream_t>(stream));
    sl.input_ch = 3;
    sl.weights_epoch += 1;
    sl.ready = true;
    return check_launch();
}

int neo_pix_set_scene(neo_ctx* ctx, const float* latent, int NV, int Cl, int Hf, int Wf, float image_w,
                      float image_h, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(latent, "null pointer");
    REQUIRE(NV >= 1 && NV <= neo::TP_MAX_VIEWS, "1..8 source views supported");
    REQUIRE(Cl == 512, "the latent width is fixed by the reference MLP (512)");
    REQUIRE(Hf >= 2 && Wf >= 2, "feature maps must be at least 2x2");
    REQUIRE(static_cast<long>(NV) * Hf * Wf * 2048 <= 4294967295L, "latent too large for 32-bit byte offsets (NV*Hf*Wf < 2^21 texels)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ctx->pix_latent.reserve(static_cast<size_t>(NV) * Cl * Hf * Wf * 4)) return NEO_ERR_NOMEM;
    neo::launch_channels_last(latent, NV, Cl, Hf, Wf, ctx->pix_latent.as<float>(), s);
    ctx->pix_scene.latent = ctx->pix_latent.as<float>();
    for (int j = 0; j < 3; ++j) ctx->pix_scene.plane[j] = nullptr;     // no tri-planes in this decoder
    ctx->pix_scene.nv = NV;
    ctx->pix_scene.Hf = Hf; ctx->pix_scene.Wf = Wf; ctx->pix_scene.Hp = 2; ctx->pix_scene.Wp = 2;
    // latent_scaling = [Wf,Hf]/([Wf,Hf]-1)*2 ; scale = latent_scaling/image_size (vanilla_nerf/encoder.py index())
    const float wf = static_cast<float>(Wf), hf = static_cast<float>(Hf);
    ctx->pix_scene.sx = ((wf / (wf - 1.0f)) * 2.0f) / image_w;
    ctx->pix_scene.sy = ((hf / (hf - 1.0f)) * 2.0f) / image_h;
    ctx->pix_scene.fy_sign = 1.0f;                                     // model_pixel.py:203 passes (f, f)
    ctx->pix_scene_ready = true;
    ctx->pix_latent_checked = false;
    ctx->pix_scene_epoch += 1;
    return check_launch();
}

int neo_pix_set_preproject(neo_ctx* ctx, int enable) {
    ENTER(ctx);
    ctx->pix_preproject = enable != 0;
    if (!ctx->pix_preproject)
        for (auto& sl : ctx->pix) { sl.proj.release(); sl.proj_weights = sl.proj_scene = 0; }
    return NEO_OK;
}

int neo_pix_mlp(neo_ctx* ctx, int slot, const float* rays_o, const float* rays_d, const float* viewdirs,
                const float* tvals, int R, int N, int chunk, const float* src_poses, int NV, float focal, float cx,
                float cy, float* out, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(slot == 0 || slot == 1, "slot must be 0 or 1");
    REQUIRE(R >= 0 && N >= 1 && chunk >= 1, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && rays_d && viewdirs && tvals && src_poses && out, "null pointer");
    if (!ctx->pix_scene_ready) return fail(NEO_ERR_STATE, "scene latent not set (neo_pix_set_scene)");
    REQUIRE(NV == ctx->pix_scene.nv, "NV differs from the uploaded scene");
    hipStream_t s = static_cast<hipStream_t>(stream);
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->pix_scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    const int rc = pix_launch(ctx, slot, sc, views, rays_o, rays_d, viewdirs, tvals, 0, R, N, chunk, out, s);
    if (rc) return rc;
    return check_launch();
}

int neo_pix_render(neo_ctx* ctx, const float* rays_o, const float* rays_d, const float* viewdirs, int R, int chunk,
                   const float* src_poses, int NV, float focal, float cx, float cy, float near, float far,
                   int n_coarse, int n_fine, int white_bkgd, float* rgb0, float* acc0, float* depth0, float* rgb1,
                   float* acc1, float* depth1, void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0 && chunk >= 1, "bad ray count / chunk");
    REQUIRE(n_coarse >= 3 && n_coarse <= 256 && n_fine >= 1 && n_coarse + 1 + n_fine <= 1024, "unsupported sample counts");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && rays_d && viewdirs && src_poses, "null pointer");
    if (!ctx->pix_scene_ready) return fail(NEO_ERR_STATE, "scene latent not set (neo_pix_set_scene)");
    REQUIRE(NV == ctx->pix_scene.nv, "NV differs from the uploaded scene");
    hipStream_t s = static_cast<hipStream_t>(stream);
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->pix_scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    const int N0 = n_coarse + 1, N1 = N0 + n_fine;
    const float* t0 = ctx->get_edges(n_coarse, near, far, s);
    const float* u = ctx->get_quantiles(n_fine, s);
    if (!t0 || !u) return fail(NEO_ERR_HIP, "constant table upload failed");
    ORDERED_LANE(ctx, static_cast<hipStream_t>(stream));      // writes this lane's workspaces only (a lazy pre-projection: touch_shared)
    if (ctx->ws[0].reserve(static_cast<size_t>(R) * N0 * 16)) return NEO_ERR_NOMEM;
    if (ctx->ws[1].reserve(static_cast<size_t>(R) * N0 * 4)) return NEO_ERR_NOMEM;
    if (ctx->ws[2].reserve(static_cast<size_t>(R) * N1 * 4)) return NEO_ERR_NOMEM;
    if (ctx->ws[3].reserve(static_cast<size_t>(R) * N1 * 16)) return NEO_ERR_NOMEM;
    float* out0 = ctx->ws[0].as<float>();
    float* w0 = ctx->ws[1].as<float>();
    float* t1 = ctx->ws[2].as<float>();
    float* out1 = ctx->ws[3].as<float>();
    // level 0: one shared row of sample positions along rays_d (model_pixel.py:183-192; vanilla helper :415-442)
    int rc = pix_launch(ctx, 0, sc, views, rays_o, rays_d, viewdirs, t0, 1, R, N0, chunk, out0, s);
    if (rc) return rc;
    neo::launch_composite(0, out0, t0, 0, rays_d, nullptr, R, N0, white_bkgd, rgb0, acc0, depth0, w0, nullptr, s);
    // level 1: bins = mids(t0), weights[1:-1] (model_pixel.py:195-206); sort-merge
    if (neo::launch_resample(t0, 0, w0, u, 0, R, N0, n_fine, 0, t1, s)) return fail(NEO_ERR_INVALID, "unsupported sample counts");
    rc = pix_launch(ctx, 1, sc, views, rays_o, rays_d, viewdirs, t1, 0, R, N1, chunk, out1, s);
    if (rc) return rc;
    neo::launch_composite(0, out1, t1, N1, rays_d, nullptr, R, N1, white_bkgd, rgb1, acc1, depth1, nullptr, nullptr, s);
    return check_launch();
}

}  // extern "C"

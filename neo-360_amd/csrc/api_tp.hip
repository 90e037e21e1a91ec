// NeO-360 (NeRF_TP) entry points of the C ABI: weight / scene upload and the
// two-level, two-region render (neo360/model.py:266-581).
#include "ctx.h"

using namespace neo_host;

namespace {

// algorithmic MACs of NeRFPPMLP (neo360/model.py:37-158): per point-view 255,424 (fg) / 260,800 (bg),
// per point 4,416 (density 128 + 64x64 + rgb 192); SURVEY.md §8a row a13.
double tp_flop_per_point(int input_ch, int nv) {
    const double per_view = input_ch == 3 ? 255424.0 : 260800.0;
    return 2.0 * (nv * per_view + 4416.0);
}

// rot = c2w[:3,:3]^T ; trans = -rot @ c2w[:3,3]   (neo360/util.py:64-66), fp32
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

// The maps a slot gathers in pre-projection modes, recomputed (exact fp32 MFMA, k_tp_preproject: ~1 ms per map) only when the
// scene or the slot's weights changed since they were made: the latent through [W0_loc | W3_loc], and - planes = true - the three
// tri-planes through [W0_world | W3_world] (chunks 64..79 of the fp32 fragment stream).  Shared by both arithmetic modes.
int ensure_projections(neo_ctx* ctx, MlpSlot& sl, const neo::TpScene& sc, bool planes, hipStream_t s, long plane_base[3],
                       bool split_guard = false) {
    const long t_lat = static_cast<long>(sc.nv) * sc.Hf * sc.Wf, t_pl = static_cast<long>(sc.nv) * sc.Hp * sc.Wp;
    for (int j = 0; j < 3; ++j) plane_base[j] = t_lat + j * t_pl;
    const size_t need = neo::tp_proj_bytes(t_lat + (planes ? 3 * t_pl : 0)) + neo::tp_proj_pad_bytes();
    REQUIRE(need <= 4294967295UL, "projected maps too large for 32-bit byte offsets");
    if (need > sl.proj.cap) {          // growing re-allocates: everything in the buffer has to be made again
        if (int rc = ctx->touch_shared(s)) return rc;
        if (sl.proj.reserve(need)) return NEO_ERR_NOMEM;
        sl.proj_weights = sl.proj_scene = sl.projpl_weights = sl.projpl_scene = 0;
    }
    if (sl.proj_weights != sl.weights_epoch || sl.proj_scene != ctx->scene_epoch) {
        if (int rc = ctx->touch_shared(s)) return rc;      // shared by both scratch lanes: this call ends as an exclusive one
        neo::launch_tp_preproject(sc.latent, t_lat, sl.wpack.as<float>(), neo::tp_kc_x(sl.input_ch), sl.proj.as<float>(), s);
        // split arithmetic: a projected map is a STATIC operand of the evaluator (its blends are added to the layer-0 / skip
        // pre-activations, whose hi/lo split the epilogue guards) - checked once per (scene, weights) here, so that a latent the
        // fp16 range cannot hold is reported as a static trip (NEO_FLAG_SPLIT_STATIC: the frame API latches to the exact kernels
        // instead of paying a failed split attempt per frame)
        if (split_guard) neo::launch_f32_range_check(sl.proj.as<float>(), static_cast<size_t>(t_lat) * 256, 65504.0f, ctx->flags, s);
        sl.proj_weights = sl.weights_epoch;
        sl.proj_scene = ctx->scene_epoch;
    }
    if (planes && (sl.projpl_weights != sl.weights_epoch || sl.projpl_scene != ctx->scene_epoch)) {
        if (int rc = ctx->touch_shared(s)) return rc;
        for (int j = 0; j < 3; ++j)
            neo::launch_tp_preproject(sc.plane[j], t_pl, sl.wpack.as<float>(), neo::tp_kc_x(sl.input_ch),
                                      sl.proj.as<float>() + plane_base[j] * 256, s, 256, 128, 64);
        if (split_guard)         // the three projected planes are summed before they meet an accumulator
            neo::launch_f32_range_check(sl.proj.as<float>() + plane_base[0] * 256, static_cast<size_t>(3 * t_pl) * 256, 65504.0f / 3.0f, ctx->flags, s);
        sl.projpl_weights = sl.weights_epoch;
        sl.projpl_scene = ctx->scene_epoch;
    }
    return NEO_OK;
}

// one region's evaluator launch in the context's arithmetic mode
int tp_launch(neo_ctx* ctx, MlpSlot& sl, const neo::TpScene& sc, const neo::TpViews& views, const float* rays_o,
              const float* rays_d, const float* viewdirs, const float* tvals, const float* far, int R, int N, int chunk,
              float* out, hipStream_t s) {
    ORDERED(ctx, s);      // context scratch (tp_dirsum, projected maps, ws[]) is shared by all streams
    const int slot_index = static_cast<int>(&sl - ctx->tp);
    // Patch shape of the ray-patch tile order (neo_ctx_set_ray_grid), per region - measured fabric-side bytes per full-frame launch
    // (profiles/r06_ray_patch_order.log): inside the sphere a ray's footprint is ~2 MB of projected texels, so only the nearest
    // neighbours of a ray are still in the XCD's 4 MB L2: 2 x 2 patches (59.8 -> 41.7 GB; 4 x 4: 45.3, 8 x 8: 51.5); outside, where
    // most taps are the zero-weight placeholder, 8 x 8 (11.6 -> 1.8 GB).  $NEO_TP_PATCH = "<log2 w>,<log2 h>" overrides both.
    neo::TpScene scp = sc;
    if (scp.grid_w > 0) {
        static int epw = -2, eph = -2;
        if (epw == -2) {
            epw = eph = -1;
            if (const char* e = getenv("NEO_TP_PATCH")) (void)sscanf(e, "%d,%d", &epw, &eph);
        }
        scp.grid_pw = epw >= 0 ? epw : (slot_index < 2 ? 1 : 3);
        scp.grid_ph = eph >= 0 ? eph : (slot_index < 2 ? 1 : 3);
        if (scp.grid_w % (1 << scp.grid_pw) != 0) scp.grid_w = 0;
    }
    long plane_base[3] = {0, 0, 0};          // first texel of each projected tri-plane inside sl.proj
    if (ctx->precision == 1) {
        // range guard of the split arithmetic: tri-planes are summed over three maps before they are split
        if (ctx->planes_checked != ctx->scene_epoch) {
            const size_t n = static_cast<size_t>(sc.nv) * sc.Hp * sc.Wp * 128;
            for (int j = 0; j < 3; ++j) neo::launch_f32_range_check(sc.plane[j], n, 65504.0f / 3.0f, ctx->flags, s);
            ctx->planes_checked = ctx->scene_epoch;
        }
        if (ctx->preproject) {
            // mode 2: planes projected for every slot; mode 3 (default): for the two OUTSIDE-sphere slots only.  Outside, most
            // samples leave the tri-plane volume and every source image: the work list of mlp_tp_hpp.hip gathers 6 of 16
            // (row group, map) pairs on average and the world GEMM stage is pure saving (-8 % / -6 % on the fine / coarse
            // launch).  Inside, every map carries weight on almost every row: 24 more 1 KB gather items per tile-view cost
            // more than the GEMM stage they replace (+1..7 %): profiles/r04_tp_hp_experiments.log.
            const bool planes = ctx->preproject == 2 || (ctx->preproject == 3 && slot_index >= 2);
            if (int rc = ensure_projections(ctx, sl, sc, planes, s, plane_base, true)) return rc;
            guard_split_weights(sl, sl.wpack_hp.p, neo::tp_wpack_hp_bytes(sl.input_ch), ctx->flags, s);
            neo::TpMlpHDev mh{sl.wpack_hp.p, sl.bias_hp.as<float>(), sl.heads.as<float>(), ctx->flags};
            // the view-direction encodings enter the MLP only through their mean over the views, and they depend on the
            // ray alone: summed once per ray here instead of once per sample and view inside the evaluator
            if (ctx->tp_dirsum->reserve(static_cast<size_t>(R) * 32 * sizeof(float))) return NEO_ERR_NOMEM;
            neo::launch_tp_dirsum(viewdirs, R, views, sc.nv, ctx->tp_dirsum->as<float>(), s);
            ctx->span_kernel_next = planes ? 2 : 1;
            ctx->span_begin(s);
            if (planes)
                neo::launch_tp_mlp_hpp(sl.input_ch, mh, sl.proj.as<float>(), plane_base, scp, views, rays_o, rays_d, viewdirs, tvals, far,
                                       R, N, chunk, ctx->flags, out, ctx->tp_dirsum->as<float>(), s);
            else
                neo::launch_tp_mlp_hp(sl.input_ch, mh, sl.proj.as<float>(), scp, views, rays_o, rays_d, viewdirs, tvals, far, R, N,
                                      chunk, ctx->flags, out, ctx->tp_dirsum->as<float>(), s);
        } else {
            guard_split_weights(sl, sl.wpack_h.p, neo::tp_wpack_h_bytes(sl.input_ch), ctx->flags, s);
            if (ctx->latent_checked != ctx->scene_epoch) {
                neo::launch_f32_range_check(sc.latent, static_cast<size_t>(sc.nv) * sc.Hf * sc.Wf * 512, 65504.0f, ctx->flags, s);
                ctx->latent_checked = ctx->scene_epoch;
            }
            neo::TpMlpHDev mh{sl.wpack_h.p, sl.bias.as<float>(), sl.heads.as<float>(), ctx->flags};
            ctx->span_kernel_next = 3;
            ctx->span_begin(s);
            neo::launch_tp_mlp_h(sl.input_ch, mh, scp, views, rays_o, rays_d, viewdirs, tvals, far, R, N, chunk, ctx->flags, out, s);
        }
    } else {
        // exact fp32 MFMA; with pre-projection on, the same algorithm as the split default in the reference's arithmetic: the
        // projected maps are gathered and added, their GEMM stages are not executed per point (the matrix work dominates this
        // kernel at 1/16 of the fp16 rate, so the planes are projected in modes 2 AND 3, for every slot)
        neo::TpMlpDev m{sl.wpack.as<float>(), sl.bias.as<float>(), sl.heads.as<float>()};
        const bool planes = ctx->preproject >= 2;
        if (ctx->preproject)
            if (int rc = ensure_projections(ctx, sl, sc, planes, s, plane_base)) return rc;
        const float* pbase = sl.proj.as<float>();
        const neo::TpPlaneProj pp{{pbase + plane_base[0] * 256, pbase + plane_base[1] * 256, pbase + plane_base[2] * 256}};
        if (ctx->preproject) {      // as on the split path: the direction encodings summed over the views once per ray
            if (ctx->tp_dirsum->reserve(static_cast<size_t>(R) * 32 * sizeof(float))) return NEO_ERR_NOMEM;
            neo::launch_tp_dirsum(viewdirs, R, views, sc.nv, ctx->tp_dirsum->as<float>(), s);
        }
        ctx->span_kernel_next = 4;
        ctx->span_begin(s);
        neo::launch_tp_mlp(sl.input_ch, m, scp, views, rays_o, rays_d, viewdirs, tvals, far, R, N, chunk, ctx->flags, out, s,
                           ctx->preproject ? sl.proj.as<float>() : nullptr, planes ? &pp : nullptr,
                           ctx->preproject ? ctx->tp_dirsum->as<float>() : nullptr);
    }
    ctx->span_end(s, static_cast<double>(R) * N, tp_flop_per_point(sl.input_ch, sc.nv));
    return NEO_OK;
}

}  // namespace

// one region's evaluator launch for other translation units (api_train.hip)
int neo_tp_eval_region(neo_ctx* ctx, int slot, const neo::TpScene& sc, const neo::TpViews& views, const float* rays_o,
                       const float* rays_d, const float* viewdirs, const float* tvals, const float* far, int R, int N,
                       int chunk, float* out, hipStream_t s) {
    MlpSlot& sl = ctx->tp[slot];
    if (!sl.ready) return fail(NEO_ERR_STATE, "NeRF_TP MLP slot %d has no weights", slot);
    return tp_launch(ctx, sl, sc, views, rays_o, rays_d, viewdirs, tvals, far, R, N, chunk, out, s);
}

extern "C" {

int neo_tp_upload_mlp(neo_ctx* ctx, int slot, int input_ch, const float* const* weights,
                      const float* const* biases, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(slot >= 0 && slot < 4, "slot must be 0..3 (fg_coarse, fg_fine, bg_coarse, bg_fine)");
    REQUIRE(input_ch == 3 || input_ch == 4, "input_ch must be 3 (fg) or 4 (bg)");
    REQUIRE(weights && biases, "null pointer table");
    for (int i = 0; i < 9; ++i) REQUIRE(weights[i] && biases[i], "null layer pointer");
    MlpSlot& sl = ctx->tp[slot];
    if (sl.wpack.reserve(neo::tp_wpack_floats(input_ch) * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.bias.reserve(neo::tp_bias_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.heads.reserve(neo::tp_heads_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.wpack_h.reserve(neo::tp_wpack_h_bytes(input_ch))) return NEO_ERR_NOMEM;
    if (sl.wpack_hp.reserve(neo::tp_wpack_hp_bytes(input_ch))) return NEO_ERR_NOMEM;
    if (sl.fold_ws.reserve(neo::tp_fold_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    neo::launch_tp_pack(input_ch, weights, biases, sl.wpack.as<float>(), sl.bias.as<float>(), sl.heads.as<float>(),
                        static_cast<hipStream_t>(stream), sl.fold_ws.as<float>());
    neo::launch_tp_pack_h(input_ch, weights, sl.wpack_h.p, static_cast<hipStream_t>(stream));
    if (sl.bias_hp.reserve(neo::tp_bias_floats() * sizeof(float)) || sl.fold_ws.reserve(neo::tp_fold_floats() * sizeof(float)))
        return NEO_ERR_NOMEM;
    neo::launch_tp_pack_hp(input_ch, weights, biases, sl.wpack_hp.p, sl.fold_ws.as<float>(), sl.bias.as<float>(),
                           sl.bias_hp.as<float>(), static_cast<hipStream_t>(stream));
    sl.weights_epoch += 1;
    sl.input_ch = input_ch;
    sl.ready = true;
    return check_launch();
}

int neo_tp_set_scene(neo_ctx* ctx, const float* plane_xz, const float* plane_xy, const float* plane_yz, int NV,
                     int Cw, int Hp, int Wp, const float* latent, int Cl, int Hf, int Wf, float image_w,
                     float image_h, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(plane_xz && plane_xy && plane_yz && latent, "null pointer");
    REQUIRE(NV >= 1 && NV <= neo::TP_MAX_VIEWS, "1..8 source views supported");
    REQUIRE(Cw == 128 && Cl == 512, "feature widths are fixed by the reference MLP (128 world, 512 local)");
    REQUIRE(Hp >= 2 && Wp >= 2 && Hf >= 2 && Wf >= 2, "feature maps must be at least 2x2");
    REQUIRE(static_cast<long>(NV) * Hf * Wf * 2048 <= 4294967295L && static_cast<long>(NV) * Hp * Wp * 512 <= 4294967295L,
            "feature maps too large for 32-bit byte offsets (NV*Hf*Wf < 2^21 texels)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float* planes[3] = {plane_xz, plane_xy, plane_yz};
    for (int j = 0; j < 3; ++j) {
        if (ctx->plane[j].reserve(static_cast<size_t>(NV) * Cw * Hp * Wp * 4)) return NEO_ERR_NOMEM;
        neo::launch_channels_last(planes[j], NV, Cw, Hp, Wp, ctx->plane[j].as<float>(), s);
        ctx->scene.plane[j] = ctx->plane[j].as<float>();
    }
    if (ctx->latent.reserve(static_cast<size_t>(NV) * Cl * Hf * Wf * 4)) return NEO_ERR_NOMEM;
    neo::launch_channels_last(latent, NV, Cl, Hf, Wf, ctx->latent.as<float>(), s);
    ctx->scene.latent = ctx->latent.as<float>();
    ctx->scene.nv = NV;
    ctx->scene.Hf = Hf; ctx->scene.Wf = Wf; ctx->scene.Hp = Hp; ctx->scene.Wp = Wp;
    // latent_scaling = [Wf,Hf]/([Wf,Hf]-1)*2 (encoder_pn.py:204-206); scale = latent_scaling/image_size (:121-123)
    const float wf = static_cast<float>(Wf), hf = static_cast<float>(Hf);
    const float lsx = (wf / (wf - 1.0f)) * 2.0f, lsy = (hf / (hf - 1.0f)) * 2.0f;
    ctx->scene.sx = lsx / image_w;
    ctx->scene.sy = lsy / image_h;
    ctx->scene.fy_sign = -1.0f;                                       // NeRF_TP projects with (f, -f) (model.py:243)
    ctx->scene_epoch += 1;
    ctx->scene_ready = true;
    return check_launch();
}

int neo_tp_set_preproject(neo_ctx* ctx, int enable) {
    ENTER(ctx);
    (void)hipDeviceSynchronize();        // projected maps may be released below: nothing of this context may still read them
    REQUIRE(enable >= 0 && enable <= 3, "preproject mode must be 0 (off), 1 (latent), 2 (latent + tri-planes) or 3 (2 except slot 0)");
    ctx->preproject = enable;
    for (auto& sl : ctx->tp) sl.range_checked = 0;       // the other fragment set is checked at its first launch
    if (!ctx->preproject)
        for (auto& sl : ctx->tp) { sl.proj.release(); sl.proj_weights = sl.proj_scene = 0; }
    if (ctx->preproject < 2)
        for (auto& sl : ctx->tp) {           // the plane part goes with the next (smaller) allocation; mark it stale now
            if (ctx->preproject) sl.proj.release();
            sl.proj_weights = sl.proj_scene = sl.projpl_weights = sl.projpl_scene = 0;
        }
    return NEO_OK;
}

int neo_tp_mlp(neo_ctx* ctx, int slot, const float* rays_o, const float* rays_d, const float* viewdirs,
               const float* tvals, const float* far, int R, int N, int chunk, const float* src_poses, int NV,
               float focal, float cx, float cy, float* out, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(slot >= 0 && slot < 4, "slot must be 0..3");
    REQUIRE(R >= 0 && N >= 1 && chunk >= 1, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && rays_d && viewdirs && tvals && src_poses && out, "null pointer");
    if (!ctx->scene_ready) return fail(NEO_ERR_STATE, "scene features not set (neo_tp_set_scene)");
    REQUIRE(NV == ctx->scene.nv, "NV differs from the uploaded scene");
    MlpSlot& sl = ctx->tp[slot];
    if (!sl.ready) return fail(NEO_ERR_STATE, "NeRF_TP MLP slot %d has no weights", slot);
    REQUIRE(sl.input_ch == (slot < 2 ? 3 : 4), "slots 0,1 must hold fg weights, 2,3 bg weights");
    REQUIRE(slot < 2 || far, "far required for the outside-sphere slots");
    hipStream_t s = static_cast<hipStream_t>(stream);
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    if (int rc = tp_launch(ctx, sl, sc, views, rays_o, rays_d, viewdirs, tvals, far, R, N, chunk, out, s)) return rc;
    return check_launch();
}

int neo_tp_render(neo_ctx* ctx, const float* rays_o, const float* rays_d, const float* viewdirs, int R, int chunk,
                  const float* src_poses, int NV, float focal, float cx, float cy, int n_coarse, int n_fine,
                  int white_bkgd, const neo_tp_level_out* level0, const neo_tp_level_out* level1, void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0 && chunk >= 1, "bad ray count / chunk");
    REQUIRE(n_coarse >= 3 && n_coarse <= 256 && n_fine >= 1 && n_coarse + 1 + n_fine <= 1024, "unsupported sample counts");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && rays_d && viewdirs && src_poses, "null pointer");
    if (!ctx->scene_ready) return fail(NEO_ERR_STATE, "scene features not set (neo_tp_set_scene)");
    REQUIRE(NV == ctx->scene.nv, "NV differs from the uploaded scene");
    for (int i = 0; i < 4; ++i)
        if (!ctx->tp[i].ready) return fail(NEO_ERR_STATE, "NeRF_TP MLP slot %d has no weights", i);
    REQUIRE(ctx->tp[0].input_ch == 3 && ctx->tp[1].input_ch == 3 && ctx->tp[2].input_ch == 4 && ctx->tp[3].input_ch == 4,
            "slots 0,1 must be fg (input_ch 3), slots 2,3 bg (input_ch 4)");
    (void)white_bkgd;  // out_depth=True semantics: the reference composites with white_bkgd=False (model.py:477-501)
    hipStream_t s = static_cast<hipStream_t>(stream);
    neo::TpViews views{};
    fill_views(src_poses, NV, views);
    neo::TpScene sc = ctx->scene;
    sc.focal = focal; sc.cx = cx; sc.cy = cy;
    sc.grid_w = ctx->ray_grid_w;              // pixel-grid hint of the frame API: the evaluators walk the rays in 8 x 8 patches
    sc.grid_first = ctx->ray_grid_first;

    const int N0 = n_coarse + 1, N1 = N0 + n_fine;
    const float near = 1e-4f;                                         // model.py:277
    const float* edges = ctx->get_edges(n_coarse, 0.0f, 1.0f, s);     // linspace(0,1,n+1)
    const float* u = ctx->get_quantiles(n_fine, s);
    if (!edges || !u) return fail(NEO_ERR_HIP, "constant table upload failed");

    // workspaces: this lane's set (neo_ctx_set_lane); the call reads shared weights / maps and writes lane scratch only
    ORDERED_LANE(ctx, s);
    auto* W = ctx->ws;
    const size_t r = static_cast<size_t>(R);
    if (W[0].reserve(r * 4) || W[1].reserve(r * N0 * 4) || W[2].reserve(r * N0 * 4) || W[3].reserve(r * N1 * 16) ||
        W[4].reserve(r * N1 * 16) || W[5].reserve(r * N0 * 4) || W[6].reserve(r * N0 * 4) || W[7].reserve(r * N1 * 4) ||
        W[8].reserve(r * N1 * 4) || W[9].reserve(r * 16 * 4))
        return NEO_ERR_NOMEM;
    float* far = W[0].as<float>();
    float* fg_t0 = W[1].as<float>();
    float* bg_s0 = W[2].as<float>();
    float* fg_out = W[3].as<float>();
    float* bg_out = W[4].as<float>();
    float* fg_w0 = W[5].as<float>();
    float* bg_w0 = W[6].as<float>();
    float* fg_t1 = W[7].as<float>();
    float* bg_s1 = W[8].as<float>();
    float* scratch = W[9].as<float>();   // per-ray: fg_rgb(3) fg_depth fg_acc lambda bg_rgb(3) bg_depth
    float* s_fg_rgb = scratch;
    float* s_fg_depth = scratch + r * 3;
    float* s_fg_acc = scratch + r * 4;
    float* s_lambda = scratch + r * 5;
    float* s_bg_rgb = scratch + r * 6;
    float* s_bg_depth = scratch + r * 9;

    neo::launch_sphere(rays_o, rays_d, R, far, nullptr, ctx->flags, s);
    neo::launch_tp_level0(far, edges, R, N0, near, fg_t0, bg_s0, s);

    const float* fg_t = fg_t0;
    const float* bg_s = bg_s0;
    for (int level = 0; level < 2; ++level) {
        const int N = level == 0 ? N0 : N1;
        const neo_tp_level_out* lo = level == 0 ? level0 : level1;
        MlpSlot& fg = ctx->tp[level];
        MlpSlot& bg = ctx->tp[2 + level];
        if (int rc = tp_launch(ctx, fg, sc, views, rays_o, rays_d, viewdirs, fg_t, nullptr, R, N, chunk, fg_out, s)) return rc;
        if (int rc = tp_launch(ctx, bg, sc, views, rays_o, rays_d, viewdirs, bg_s, far, R, N, chunk, bg_out, s)) return rc;
        float* fg_rgb = (lo && lo->fg_rgb) ? lo->fg_rgb : s_fg_rgb;
        float* bg_rgb = (lo && lo->bg_rgb) ? lo->bg_rgb : s_bg_rgb;
        float* fg_acc = (lo && lo->fg_acc) ? lo->fg_acc : s_fg_acc;
        float* lam = (lo && lo->bg_lambda) ? lo->bg_lambda : s_lambda;
        neo::launch_composite(1, fg_out, fg_t, N, rays_d, far, R, N, 0, fg_rgb, fg_acc, s_fg_depth,
                              level == 0 ? fg_w0 : nullptr, lam, s);
        neo::launch_composite(2, bg_out, bg_s, N, nullptr, nullptr, R, N, 0, bg_rgb, nullptr, s_bg_depth,
                              level == 0 ? bg_w0 : nullptr, nullptr, s);
        if (lo && (lo->rgb || lo->depth))
            neo::launch_tp_merge(fg_rgb, s_fg_depth, lam, bg_rgb, s_bg_depth, R, lo->rgb, lo->depth, s);
        if (level == 0) {
            // hierarchical resampling (model.py:306-332): fg ascending; bg on the descending inverse radius
            if (neo::launch_resample(fg_t0, N0, fg_w0, u, 0, R, N0, n_fine, 0, fg_t1, s) ||
                neo::launch_resample(bg_s0, N0, bg_w0, u, 0, R, N0, n_fine, 1, bg_s1, s))
                return fail(NEO_ERR_INVALID, "unsupported sample counts");
            fg_t = fg_t1;
            bg_s = bg_s1;
        }
    }
    return check_launch();
}

}  // extern "C"

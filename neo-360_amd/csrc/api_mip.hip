// Mip-NeRF 360 entry points of the C ABI (models/mipnerf360/model.py:236-365).
#include "ctx.h"
#include "mip_layered.h"
#include "train_kernels.h"

using namespace neo_host;

namespace {

// algorithmic MACs per interval: PropMLP 504*256 + 3*256^2 + 256 = 325,888;
// NeRFMLP 504*1024 + 6*1024^2 + 1528*1024 - ... = 8,672,000 (trunk 8,372,224 + density 1,024 +
// bottleneck 262,144 + view 36,224 + rgb 384)
double mip_flop_per_point(int width, int rgb) { return 2.0 * (width == 1024 && rgb ? 8672000.0 : 325888.0); }

int mip_mlp_launch(neo_ctx* ctx, int slot, const float* rays_o, const float* rays_d, const float* viewdirs,
                   const float* radii, const float* tdist, int R, int n, float* out, hipStream_t s) {
    MlpSlot& sl = ctx->mip[slot];
    if (!sl.ready) return fail(NEO_ERR_STATE, "MipNeRF360 MLP slot %d has no weights", slot);
    const int* sh = ctx->mip_shape[slot];
    if (ctx->precision == 1) guard_split_weights(sl, sl.wpack_h.p, neo::mip_wpack_h_bytes(sh[0], sh[1], sh[2]), ctx->flags, s);
    // NeRF MLP in split arithmetic: layer-by-layer trunk (mip_gemm_h.h) from 8192 intervals on (neo_mip_set_layered); a
    // batch's activations live in context-owned buffers (grow-only, 160 MB)
    const bool layered = ctx->precision == 1 && sh[0] == 1024 && sh[1] == 8 && sh[2] &&
                         (ctx->mip_layered == 1 || (ctx->mip_layered < 0 && static_cast<long>(R) * n >= 8192));
    const int cap = neo::MIP_LAYERED_BATCH;
    if (layered && (ctx->mip_lws[0].reserve(neo::mip_layered_x0_bytes(cap)) || ctx->mip_lws[1].reserve(neo::mip_layered_y_bytes(cap)) ||
                    ctx->mip_lws[2].reserve(neo::mip_layered_y_bytes(cap))))
        return NEO_ERR_NOMEM;
    // span ids (neo_ctx_read_spans): 5 fused split evaluator of a proposal MLP, 6 of the NeRF MLP, 7 the NeRF MLP layer by layer
    // (k_mip_ipe_h + 8 x k_mip_gemm_h + the fused evaluator's tail per batch), 8 exact fp32 evaluator
    ctx->span_kernel_next = ctx->precision != 1 ? 8 : layered ? 7 : (sh[0] == 1024 ? 6 : 5);
    ctx->span_begin(s);
    int rc;
    if (ctx->precision == 1) {      // split-fp16 matrix cores (fp32-equivalent), neo_ctx_set_precision
        neo::MipMlpHDev mh{sl.wpack_h.p, sl.bias.as<float>(), sl.heads.as<float>(), ctx->mip_basis.as<float>(), ctx->flags,
                           sl.bias_hp.as<float>()};
        if (layered) {
            const neo::MipLayeredWs ws{static_cast<char*>(ctx->mip_lws[0].p), static_cast<char*>(ctx->mip_lws[1].p),
                                       static_cast<char*>(ctx->mip_lws[2].p), cap};
            rc = neo::launch_mip_mlp_h_layered(mh, ws, rays_o, rays_d, viewdirs, radii, tdist, R, n, out, s);
        } else {
            rc = neo::launch_mip_mlp_h(sh[0], sh[1], sh[2], mh, rays_o, rays_d, viewdirs, radii, tdist, R, n, out, s);
        }
    } else {
        neo::MipMlpDev m{sl.wpack.as<float>(), sl.bias.as<float>(), sl.heads.as<float>(), ctx->mip_basis.as<float>()};
        rc = neo::launch_mip_mlp(sh[0], sh[1], sh[2], m, rays_o, rays_d, viewdirs, radii, tdist, R, n, out, s);
    }
    ctx->span_end(s, static_cast<double>(R) * n, mip_flop_per_point(sh[0], sh[2]));
    if (rc) return fail(NEO_ERR_INVALID, "unsupported MipNeRF360 MLP shape");
    return check_launch();
}

}  // namespace

extern "C" {

int neo_mip_upload_mlp(neo_ctx* ctx, int slot, int width, int depth, int rgb, const float* const* weights,
                       const float* const* biases, const float* basis, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(slot >= 0 && slot < 3, "slot must be 0..2");
    REQUIRE((width == 256 && depth == 4 && !rgb) || (width == 1024 && depth == 8 && rgb),
            "supported shapes: PropMLP (256, 4, no rgb) and NeRFMLP (1024, 8, rgb)");
    REQUIRE(weights && biases && basis, "null pointer");
    const int nl = depth + 1 + (rgb ? 3 : 0);
    for (int i = 0; i < nl; ++i) REQUIRE(weights[i] && biases[i], "null layer pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MlpSlot& sl = ctx->mip[slot];
    if (sl.wpack.reserve(neo::mip_wpack_floats(width, depth, rgb) * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.bias.reserve(neo::mip_bias_floats(width, depth, rgb) * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.heads.reserve(neo::mip_heads_floats(width) * sizeof(float))) return NEO_ERR_NOMEM;
    if (ctx->mip_basis.reserve(63 * sizeof(float))) return NEO_ERR_NOMEM;
    neo::copy_floats(basis, 63, ctx->mip_basis.as<float>(), s);
    if (sl.wpack_h.reserve(neo::mip_wpack_h_bytes(width, depth, rgb))) return NEO_ERR_NOMEM;
    neo::launch_mip_pack(width, depth, rgb, weights, biases, sl.wpack.as<float>(), sl.bias.as<float>(),
                         sl.heads.as<float>(), s);
    if (rgb && (sl.fold_ws.reserve(neo::mip_fold_floats() * sizeof(float)) || sl.bias_hp.reserve(128 * sizeof(float)))) return NEO_ERR_NOMEM;
    neo::launch_mip_pack_h(width, depth, rgb, weights, biases, sl.wpack_h.p, sl.fold_ws.as<float>(), sl.bias_hp.as<float>(), s);
    ctx->mip_shape[slot][0] = width;
    ctx->mip_shape[slot][1] = depth;
    ctx->mip_shape[slot][2] = rgb;
    sl.weights_epoch += 1;
    sl.ready = true;
    return check_launch();
}

int neo_mip_resample(neo_ctx* ctx, const float* s_prev, const float* w_prev, int R, int n_prev, int dilate,
                     float dilation, float anneal, int n, float near, float far, float* sdist, float* tdist,
                     void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(R >= 0 && n_prev >= 1 && n >= 2, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(s_prev && w_prev && sdist && tdist, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float* u = ctx->get_centre_quantiles(n, s);
    if (!u) return fail(NEO_ERR_HIP, "quantile table upload failed");
    // construct_ray_warps (helper.py:171-175): s_near = 1/near, s_far = 1/far as python floats -> fp32 scalars
    const float s_near = static_cast<float>(1.0 / static_cast<double>(near));
    const float s_far = static_cast<float>(1.0 / static_cast<double>(far));
    if (neo::launch_mip_resample(s_prev, w_prev, n_prev, dilate, dilation, anneal, u, R, n, s_near, s_far, sdist, tdist, s))
        return fail(NEO_ERR_INVALID, "unsupported sample counts (3*n_prev+1 and n must be <= 256)");
    return check_launch();
}

int neo_mip_resample_u(neo_ctx* ctx, const float* s_prev, const float* w_prev, int R, int n_prev, int dilate, float dilation,
                       float anneal, int n, const float* u, const float* jitter, float near, float far, float* sdist, float* tdist,
                       void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0 && n_prev >= 1 && n >= 2, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(s_prev && w_prev && u && sdist && tdist, "null pointer");
    const float s_near = static_cast<float>(1.0 / static_cast<double>(near));
    const float s_far = static_cast<float>(1.0 / static_cast<double>(far));
    if (neo::launch_mip_resample(s_prev, w_prev, n_prev, dilate, dilation, anneal, u, R, n, s_near, s_far, sdist, tdist,
                                 static_cast<hipStream_t>(stream), jitter))
        return fail(NEO_ERR_INVALID, "unsupported sample counts (3*n_prev+1 and n must be <= 256)");
    return check_launch();
}

int neo_mip_encode(neo_ctx* ctx, const float* rays_o, const float* rays_d, const float* radii, const float* tdist,
                   const float* pos_basis_t, int R, int n, float* out, void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0 && n >= 1, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && rays_d && radii && tdist && pos_basis_t && out, "null pointer");
    neo::launch_mip_encode(rays_o, rays_d, radii, tdist, pos_basis_t, R, n, out, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_mip_composite_backward(neo_ctx* ctx, const float* rgbdens, const float* tdist, const float* rays_d, int R, int n, float bg,
                               const float* g_weights, const float* g_rgb, float* g_rgbdens, void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0 && n >= 1, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(rgbdens && tdist && rays_d && g_rgbdens, "null pointer");
    neo::launch_mip_composite_bwd(rgbdens, tdist, rays_d, R, n, bg, g_weights, g_rgb, g_rgbdens, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_mip_mlp(neo_ctx* ctx, int slot, const float* rays_o, const float* rays_d, const float* viewdirs,
                const float* radii, const float* tdist, int R, int n, float* out, void* stream) {
    ENTER(ctx);
    REQUIRE(slot >= 0 && slot < 3, "slot must be 0..2");
    REQUIRE(R >= 0 && n >= 1, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && rays_d && viewdirs && radii && tdist && out, "null pointer");
    ORDERED(ctx, static_cast<hipStream_t>(stream));          // the layer-by-layer path writes context-owned activation buffers
    return mip_mlp_launch(ctx, slot, rays_o, rays_d, viewdirs, radii, tdist, R, n, out, static_cast<hipStream_t>(stream));
}

int neo_mip_set_layered(neo_ctx* ctx, int mode) {
    ENTER(ctx);
    REQUIRE(mode >= -1 && mode <= 1, "mode must be -1 (auto), 0 (fused evaluator) or 1 (layer by layer)");
    ctx->mip_layered = mode;
    return NEO_OK;
}

int neo_mip_composite(neo_ctx* ctx, const float* rgbdens, const float* tdist, const float* rays_d, int R, int n,
                      float bg, float* weights, float* rgb, void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0 && n >= 1, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(rgbdens && tdist && rays_d, "null pointer");
    neo::launch_mip_composite(rgbdens, tdist, rays_d, R, n, bg, weights, rgb, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_mip_render(neo_ctx* ctx, const float* rays_o, const float* rays_d, const float* viewdirs, const float* radii,
                   int R, float train_frac, float near, float far, int n_prop, int n_nerf,
                   const neo_mip_level_out* levels, void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0, "negative ray count");
    REQUIRE(n_prop >= 2 && 3 * n_prop + 1 <= 256 && n_nerf >= 2 && n_nerf <= 256, "unsupported sample counts");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && rays_d && viewdirs && radii, "null pointer");
    for (int i = 0; i < 3; ++i)
        if (!ctx->mip[i].ready) return fail(NEO_ERR_STATE, "MipNeRF360 MLP slot %d has no weights", i);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float s_near = static_cast<float>(1.0 / static_cast<double>(near));
    const float s_far = static_cast<float>(1.0 / static_cast<double>(far));
    // anneal = bias(train_frac, slope 10) = 10 tf / (9 tf + 1)   (model.py:286-290), python double -> fp32 scalar
    const float anneal = static_cast<float>((10.0 * train_frac) / (9.0 * train_frac + 1.0));
    const int nmax = n_prop > n_nerf ? n_prop : n_nerf;
    // level-0 histogram: sdist = [0, 1], weights = [1]  (model.py:240-256; near_anneal_rate=None) - a constant table per ray
    // count, made once (round 6: it used to be uploaded, with a stream synchronisation, by every call)
    const size_t r = static_cast<size_t>(R);
    auto seed_it = ctx->mip_seed.find(R);
    if (seed_it == ctx->mip_seed.end()) {
        DevBuf& sb = ctx->mip_seed[R];
        if (sb.reserve(r * 3 * 4)) return NEO_ERR_NOMEM;
        std::vector<float> h(r * 3);
        for (size_t i = 0; i < r; ++i) { h[i * 2] = 0.0f; h[i * 2 + 1] = 1.0f; h[2 * r + i] = 1.0f; }
        HIP_TRY(hipMemcpyAsync(sb.p, h.data(), r * 3 * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));                     // h goes out of scope; first use of this ray count only
        seed_it = ctx->mip_seed.find(R);
    }
    ORDERED_LANE(ctx, static_cast<hipStream_t>(stream));      // writes this lane's workspaces / activation buffers only
    auto* W = ctx->ws;
    // two ping-pong sets of (sdist, tdist, weights, rgbdens) + the level-0 seed histogram
    for (int k = 0; k < 2; ++k)
        if (W[k * 4 + 0].reserve(r * (nmax + 1) * 4) || W[k * 4 + 1].reserve(r * (nmax + 1) * 4) ||
            W[k * 4 + 2].reserve(r * nmax * 4) || W[k * 4 + 3].reserve(r * nmax * 16))
            return NEO_ERR_NOMEM;
    const float* seed = seed_it->second.as<float>();
    const float* s_prev = seed;
    const float* w_prev = seed + 2 * r;
    int n_prev = 1;
    int prod = 1;
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int n = lvl < 2 ? n_prop : n_nerf;
        // dilation = bias + multiplier * (s_far - s_near) / prod  (model.py:267-272), python double -> fp32
        const float dilation = static_cast<float>(0.0025 + 0.5 * (1.0 - 0.0) / static_cast<double>(prod));
        prod *= n;
        const int k = lvl & 1;
        const neo_mip_level_out* lo = levels ? &levels[lvl] : nullptr;
        float* sdist = (lo && lo->sdist) ? lo->sdist : W[k * 4 + 0].as<float>();
        float* tdist = W[k * 4 + 1].as<float>();
        float* wts = (lo && lo->weights) ? lo->weights : W[k * 4 + 2].as<float>();
        float* rd = (lo && lo->rgbdens) ? lo->rgbdens : W[k * 4 + 3].as<float>();
        const float* u = ctx->get_centre_quantiles(n, s);
        if (!u) return fail(NEO_ERR_HIP, "quantile table upload failed");
        if (neo::launch_mip_resample(s_prev, w_prev, n_prev, lvl > 0, dilation, anneal, u, R, n, s_near, s_far, sdist, tdist, s))
            return fail(NEO_ERR_INVALID, "unsupported sample counts");
        const int rc = mip_mlp_launch(ctx, lvl, rays_o, rays_d, viewdirs, radii, tdist, R, n, rd, s);
        if (rc) return rc;
        neo::launch_mip_composite(rd, tdist, rays_d, R, n, 1.0f, wts, lo ? lo->rgb : nullptr, s);
        s_prev = sdist;
        w_prev = wts;
        n_prev = n;
    }
    return check_launch();
}

}  // extern "C"

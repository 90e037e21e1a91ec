// Scene-encoder entry points of the C ABI: the pillar stage of GridEncoder (SURVEY.md §8f row 1;
// models/neo360/encoder_tp_fusion_conv.py:472-578) between the ResNet latent and the floor-plan conv nets.
#include "ctx.h"

using namespace neo_host;

extern "C" {

int neo_enc_upload(neo_ctx* ctx, const float* const* weights, const float* const* biases, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(weights && biases, "null pointer table");
    for (int i = 0; i < 9; ++i) REQUIRE(weights[i] && biases[i], "null layer pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MlpSlot& sl = ctx->enc;
    if (sl.wpack_h.reserve(neo::pillar_wpack_bytes())) return NEO_ERR_NOMEM;
    if (sl.bias.reserve(6 * 512 * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.heads.reserve(3 * 512 * sizeof(float))) return NEO_ERR_NOMEM;
    // order: depth_fc.common_branch.0, .2, depth_fc.depth_encoder, then per axis (xz, yz, xy): aggregator .0, .2
    const float* hidden[6] = {weights[0], weights[1], weights[2], weights[3], weights[5], weights[7]};
    const float* hidden_b[6] = {biases[0], biases[1], biases[2], biases[3], biases[5], biases[7]};
    neo::launch_pillar_pack(hidden, sl.wpack_h.p, s);
    for (int i = 0; i < 6; ++i) neo::copy_floats(hidden_b[i], 512, sl.bias.as<float>() + 512 * i, s);
    for (int a = 0; a < 3; ++a) {
        neo::copy_floats(weights[4 + 2 * a], 512, sl.heads.as<float>() + 512 * a, s);
        HIP_TRY(hipMemcpyAsync(&ctx->enc_head_b[a], biases[4 + 2 * a], sizeof(float), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));        // the three scalar head biases are kernel arguments
    sl.weights_epoch += 1;
    sl.ready = true;
    return check_launch();
}

int neo_enc_floorplans(neo_ctx* ctx, const float* latent, int NV, int Hf, int Wf, float image_w, float image_h,
                       const float* src_poses, float focal, float cx, float cy, int G0, int G1, int G2, float* fp_yz,
                       float* fp_xz, float* fp_xy, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(latent && src_poses && fp_yz && fp_xz && fp_xy, "null pointer");
    REQUIRE(NV >= 1 && NV <= neo::TP_MAX_VIEWS, "1..8 source views supported");
    REQUIRE(Hf >= 2 && Wf >= 2, "latent must be at least 2x2");
    REQUIRE(G0 >= 1 && G1 >= 1 && G2 >= 1 && G0 <= 256 && G1 <= 256 && G2 <= 256, "grid sizes must be 1..256");
    REQUIRE(static_cast<long>(NV) * Hf * Wf * 2048 <= 4294967295L, "latent too large for 32-bit byte offsets");
    if (!ctx->enc.ready) return fail(NEO_ERR_STATE, "encoder weights not uploaded (neo_enc_upload)");
    if (ctx->precision != 1) return fail(NEO_ERR_STATE, "the pillar stage exists in the split-fp16 arithmetic only (neo_ctx_set_precision(ctx, 1))");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long M = static_cast<long>(NV) * G0 * G1 * G2;
    if (ctx->enc_latent.reserve(static_cast<size_t>(NV) * 512 * Hf * Wf * 4)) return NEO_ERR_NOMEM;
    neo::launch_channels_last(latent, NV, 512, Hf, Wf, ctx->enc_latent.as<float>(), s);
    for (int i = 0; i < 3; ++i)
        if (ctx->enc_ws[i].reserve(static_cast<size_t>(M) * 512 * 4)) return NEO_ERR_NOMEM;
    if (ctx->enc_ws[3].reserve(static_cast<size_t>(M) * 3 * 4)) return NEO_ERR_NOMEM;
    // world grid axes: torch.linspace values, x / y in [-1, 1], z in [0, 1] (side_lengths = [1,1,1], :481-489)
    float axes[3 * 256] = {};
    neo_linspace_host(-1.0f, 1.0f, G0, axes);
    neo_linspace_host(-1.0f, 1.0f, G1, axes + 256);
    neo_linspace_host(0.0f, 1.0f, G2, axes + 512);
    if (ctx->enc_axes.reserve(sizeof axes)) return NEO_ERR_NOMEM;
    HIP_TRY(hipMemcpyAsync(ctx->enc_axes.p, axes, sizeof axes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));        // `axes` is on the host stack
    neo::PillarGeom gm{};
    gm.nv = NV; gm.G0 = G0; gm.G1 = G1; gm.G2 = G2; gm.Hf = Hf; gm.Wf = Wf;
    gm.focal = focal; gm.cx = cx; gm.cy = cy;
    const float wf = static_cast<float>(Wf), hf = static_cast<float>(Hf);
    gm.sx = ((wf / (wf - 1.0f)) * 2.0f) / image_w;
    gm.sy = ((hf / (hf - 1.0f)) * 2.0f) / image_h;
    gm.axes = ctx->enc_axes.as<float>();
    for (int i = 0; i < NV; ++i) {           // rot = c2w[:3,:3]^T, trans = -rot c2w[:3,3] (neo360/util.py:64-66)
        const float* m = src_poses + i * 16;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) gm.rot[i][r * 3 + c] = m[c * 4 + r];
        for (int r = 0; r < 3; ++r) {
            float acc = gm.rot[i][r * 3 + 0] * m[0 * 4 + 3];
            acc = acc + gm.rot[i][r * 3 + 1] * m[1 * 4 + 3];
            acc = acc + gm.rot[i][r * 3 + 2] * m[2 * 4 + 3];
            gm.trans[i][r] = -acc;
            gm.cpos[i][r] = m[r * 4 + 3];
        }
    }
    MlpSlot& sl = ctx->enc;
    guard_split_weights(sl, sl.wpack_h.p, neo::pillar_wpack_bytes(), ctx->flags, s);
    neo::launch_f32_range_check(ctx->enc_latent.as<float>(), static_cast<size_t>(NV) * 512 * Hf * Wf, 65504.0f, ctx->flags, s);
    ctx->span_begin(s);
    const int rc = neo::launch_pillar(gm, ctx->enc_latent.as<float>(), sl.wpack_h.p, sl.bias.as<float>(), sl.heads.as<float>(),
                                      ctx->enc_head_b, ctx->enc_ws[0].as<float>(), ctx->enc_ws[1].as<float>(),
                                      ctx->enc_ws[2].as<float>(), ctx->enc_ws[3].as<float>(), ctx->flags, fp_yz, fp_xz, fp_xy, s);
    // algorithmic MACs per cell-view: 518*512 + 2*512^2 + 3*(513*512 + 512) (encoder_tp_fusion_conv.py:263-279, :364-373)
    ctx->span_end(s, static_cast<double>(M), 2.0 * (518.0 * 512 + 2.0 * 512 * 512 + 3.0 * (513.0 * 512 + 512)));
    if (rc) return fail(NEO_ERR_INVALID, "unsupported grid");
    return check_launch();
}

}  // extern "C"

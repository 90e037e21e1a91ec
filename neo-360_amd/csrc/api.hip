// extern "C" surface of libneo360_hip.so (see include/neo360_hip.h): lifecycle, ray
// generation, stage-level operators and the vanilla NeRF path.
#include "ctx.h"

using namespace neo_host;

namespace neo_host {

static thread_local std::string g_err;
std::string& last_error() { return g_err; }

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace neo_host

// torch.linspace(start, end, steps) in fp32, CPU algorithm (symmetric fill from both
// ends, step = (end-start)/(steps-1), fused multiply-add per element), restated so the library needs no torch.
extern "C" void neo_linspace_host(float start, float end, int steps, float* out) {
    if (steps <= 0) return;
    if (steps == 1) { out[0] = start; return; }
    const float step = (end - start) / static_cast<float>(steps - 1);
    const int half = steps / 2;
    for (int i = 0; i < steps; ++i) {
        // torch's CPU kernel evaluates both branches with a fused multiply-add
        if (i < half) out[i] = fmaf(step, static_cast<float>(i), start);
        else out[i] = fmaf(-step, static_cast<float>(steps - i - 1), end);
    }
}

const float* neo_ctx::get_quantiles(int n_new, hipStream_t s) {
    auto it = quantiles.find(n_new);
    if (it != quantiles.end()) return it->second.as<float>();
    std::vector<float> h(n_new);
    // linspace(0, 1 - 2^-32, n): the end point rounds to exactly 1.0f in fp32
    neo_linspace_host(0.0f, static_cast<float>(1.0 - 1.0 / 4294967296.0), n_new, h.data());
    DevBuf& b = quantiles[n_new];
    if (b.reserve(n_new * sizeof(float))) return nullptr;
    if (hipMemcpyAsync(b.p, h.data(), n_new * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) return nullptr;
    (void)hipStreamSynchronize(s);  // h goes out of scope; first use only
    return b.as<float>();
}

// helper.py:349-351 (deterministic_center): linspace(pad, 1 - pad - eps, n), pad = 1/(2n), in fp32
const float* neo_ctx::get_centre_quantiles(int n, hipStream_t s) {
    auto it = centre_quantiles.find(n);
    if (it != centre_quantiles.end()) return it->second.as<float>();
    std::vector<float> h(n);
    const double pad = 1.0 / (2.0 * n);
    neo_linspace_host(static_cast<float>(pad), static_cast<float>(1.0 - pad - 1.1920929e-07), n, h.data());
    DevBuf& b = centre_quantiles[n];
    if (b.reserve(n * sizeof(float))) return nullptr;
    if (hipMemcpyAsync(b.p, h.data(), n * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) return nullptr;
    (void)hipStreamSynchronize(s);
    return b.as<float>();
}

// near*(1-s) + far*s over linspace(0,1,n+1), separate fp32 ops (vanilla_nerf/helper.py:425-429)
const float* neo_ctx::get_edges(int n, float near, float far, hipStream_t s) {
    uint32_t a, b2;
    memcpy(&a, &near, 4);
    memcpy(&b2, &far, 4);
    const auto key = std::make_pair(n, (static_cast<uint64_t>(a) << 32) | b2);
    auto it = edges.find(key);
    if (it != edges.end()) return it->second.as<float>();
    std::vector<float> h(n + 1);
    neo_linspace_host(0.0f, 1.0f, n + 1, h.data());
    for (int i = 0; i <= n; ++i) {
        const float om = 1.0f - h[i];
        const float lo = near * om;
        const float hi = far * h[i];
        h[i] = lo + hi;
    }
    DevBuf& buf = edges[key];
    if (buf.reserve((n + 1) * sizeof(float))) return nullptr;
    if (hipMemcpyAsync(buf.p, h.data(), (n + 1) * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) return nullptr;
    (void)hipStreamSynchronize(s);
    return buf.as<float>();
}

static int order_wait(neo_ctx* c, hipStream_t s, neo_ctx::OrderPoint& p) {
    if (!p.valid || p.stream == s) return NEO_OK;      // same stream: already in order
    if (hipStreamWaitEvent(s, p.ev, 0) != hipSuccess) return neo_host::fail(NEO_ERR_HIP, "hipStreamWaitEvent failed");
    c->order_waits += 1;
    return NEO_OK;
}
int neo_ctx::order_begin(hipStream_t s, bool lane_call) {
    // nested scopes (neo_tp_render -> tp_launch, an upload inside a render call): only the OUTERMOST one waits and records
    if (order_depth == 0) {
        order_is_lane = lane_call;
        shared_dirty = false;
        if (int rc = order_wait(this, s, order_excl)) return rc;
        if (lane_call) {
            if (int rc = order_wait(this, s, order_lane[lane])) return rc;
        } else {
            for (auto& p : order_lane)
                if (int rc = order_wait(this, s, p)) return rc;
        }
    }
    order_depth += 1;
    return NEO_OK;
}
int neo_ctx::touch_shared(hipStream_t s) {
    if (order_depth == 0 || !order_is_lane || shared_dirty) return NEO_OK;
    shared_dirty = true;
    for (auto& p : order_lane)
        if (int rc = order_wait(this, s, p)) return rc;
    return NEO_OK;
}
void neo_ctx::order_end(hipStream_t s) {
    const bool excl = !order_is_lane || shared_dirty;
    OrderPoint& p = excl ? order_excl : order_lane[lane];
    if (!p.ev && hipEventCreateWithFlags(&p.ev, hipEventDisableTiming) != hipSuccess) { p.ev = nullptr; return; }
    if (hipEventRecord(p.ev, s) == hipSuccess) { p.stream = s; p.valid = true; }
    if (excl)                              // everything earlier is behind this point now
        for (auto& l : order_lane) l.valid = false;
}

neo_order_scope::~neo_order_scope() {
    if (--c->order_depth == 0) c->order_end(s);
}

void neo_ctx::span_begin(hipStream_t s) {
    if (!timing) return;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
    spans.emplace_back(a, b);
    span_kernel.push_back(span_kernel_next);
    span_kernel_next = 0;
}
void neo_ctx::span_end(hipStream_t s, double points, double flop_per_point) {
    if (!timing) return;
    (void)hipEventRecord(spans.back().second, s);
    timed_points += points;
    timed_flops += points * flop_per_point;
    span_points.push_back(points);
    span_flops.push_back(points * flop_per_point);
}

extern "C" {

int neo_abi_version(void) { return 1; }

const char* neo_last_error(void) { return neo_host::last_error().c_str(); }

int neo_ctx_create(int device, neo_ctx** out) {
    REQUIRE(out != nullptr, "null out pointer");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(NEO_ERR_INVALID, "device %d out of range (%d visible)", device, count);
    DeviceGuard g(device);
    if (!g.ok) return fail(NEO_ERR_HIP, "hipSetDevice(%d) failed", device);
    neo_ctx* c = new neo_ctx();
    c->device = device;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->flags), sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(c->flags, 0, sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->flag_host), neo_ctx::FLAG_RING * sizeof(uint32_t), hipHostMallocDefault);
    for (int i = 0; e == hipSuccess && i < neo_ctx::FLAG_RING; ++i) e = hipEventCreateWithFlags(&c->flag_ev[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        for (auto& ev : c->flag_ev) if (ev) (void)hipEventDestroy(ev);
        if (c->flag_host) (void)hipHostFree(c->flag_host);
        if (c->flags) (void)hipFree(c->flags);
        delete c;
        return fail(NEO_ERR_HIP, "context allocation: %s", hipGetErrorString(e));
    }
    *out = c;
    return NEO_OK;
}

int neo_ctx_destroy(neo_ctx* ctx) {
    if (!ctx) return NEO_OK;
    DeviceGuard g(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& sl : ctx->vanilla) sl.release();
    for (auto& kv : ctx->quantiles) kv.second.release();
    for (auto& kv : ctx->centre_quantiles) kv.second.release();
    for (auto& sl : ctx->mip) sl.release();
    ctx->mip_basis.release();
    for (auto& set : ctx->mip_lws_sets) for (auto& b : set) b.release();
    for (auto& kv : ctx->mip_seed) kv.second.release();
    for (auto& kv : ctx->edges) kv.second.release();
    for (auto& set : ctx->ws_sets) for (auto& b : set) b.release();
    for (auto& sl : ctx->tp) sl.release();
    for (auto& sl : ctx->pix) sl.release();
    ctx->pix_latent.release();
    ctx->boxes.release();
    ctx->enc.release();
    ctx->enc_latent.release();
    ctx->enc_axes.release();
    for (auto& b : ctx->enc_ws) b.release();
    ctx->latent.release();
    for (auto& b : ctx->tp_dirsum_sets) b.release();
    ctx->train_scratch.release();
    for (auto& b : ctx->plane) b.release();
    for (auto& sp : ctx->spans) { (void)hipEventDestroy(sp.first); (void)hipEventDestroy(sp.second); }
    for (auto& ev : ctx->flag_ev) if (ev) (void)hipEventDestroy(ev);
    if (ctx->order_excl.ev) (void)hipEventDestroy(ctx->order_excl.ev);
    for (auto& l : ctx->order_lane) if (l.ev) (void)hipEventDestroy(l.ev);
    if (ctx->flag_host) (void)hipHostFree(ctx->flag_host);
    if (ctx->flags) (void)hipFree(ctx->flags);
    delete ctx;
    return NEO_OK;
}

int neo_ctx_poll_flags(neo_ctx* ctx, uint32_t* flags, void* stream) {
    ENTER(ctx);
    REQUIRE(flags != nullptr, "null flags");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(flags, ctx->flags, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(ctx->flags, 0, sizeof(uint32_t), s));
    HIP_TRY(hipStreamSynchronize(s));
    ctx->blocking_waits += 1;
    return NEO_OK;
}

// Deferred form of the same read: nothing here waits for the device.
int neo_ctx_post_flags(neo_ctx* ctx, void* stream) {
    ENTER(ctx);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // ring full (the host runs far ahead of the device and nobody looked): retire what has completed; if every slot
    // is still in flight, post nothing - the device word is sticky until a read clears it, so the next posted read
    // (or the synchronous read of take(wait)) reports whatever this call raised
    while (ctx->flag_posted - ctx->flag_taken == neo_ctx::FLAG_RING) {
        const int old = static_cast<int>(ctx->flag_taken % neo_ctx::FLAG_RING);
        if (hipEventQuery(ctx->flag_ev[old]) != hipSuccess) {
            ctx->flag_unposted = true;
            return NEO_OK;
        }
        ctx->flag_carry |= ctx->flag_host[old];
        ctx->flag_taken += 1;
    }
    ctx->flag_unposted = false;
    const int slot = static_cast<int>(ctx->flag_posted % neo_ctx::FLAG_RING);
    HIP_TRY(hipMemcpyAsync(ctx->flag_host + slot, ctx->flags, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(ctx->flags, 0, sizeof(uint32_t), s));
    HIP_TRY(hipEventRecord(ctx->flag_ev[slot], s));
    ctx->flag_posted += 1;
    return NEO_OK;
}

int neo_ctx_take_flags(neo_ctx* ctx, int wait, void* stream, uint32_t* flags, int* pending) {
    ENTER(ctx);
    REQUIRE(flags != nullptr, "null flags");
    uint32_t acc = ctx->flag_carry;
    ctx->flag_carry = 0;
    while (ctx->flag_taken < ctx->flag_posted) {          // posted in stream order: retire oldest first
        const int slot = static_cast<int>(ctx->flag_taken % neo_ctx::FLAG_RING);
        hipError_t q = hipEventQuery(ctx->flag_ev[slot]);
        if (q == hipErrorNotReady) {
            if (!wait) break;
            HIP_TRY(hipEventSynchronize(ctx->flag_ev[slot]));
            ctx->blocking_waits += 1;
        } else if (q != hipSuccess) {
            return fail(NEO_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q));
        }
        acc |= ctx->flag_host[slot];
        ctx->flag_taken += 1;
    }
    if (wait && ctx->flag_unposted) {
        // calls after the last posted read (their post found the ring full): one synchronous read covers them
        uint32_t now = 0;
        hipStream_t s = static_cast<hipStream_t>(stream);
        HIP_TRY(hipMemcpyAsync(&now, ctx->flags, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemsetAsync(ctx->flags, 0, sizeof(uint32_t), s));
        HIP_TRY(hipStreamSynchronize(s));
        ctx->blocking_waits += 1;
        ctx->flag_unposted = false;
        acc |= now;
    }
    *flags = acc;
    if (pending) *pending = static_cast<int>(ctx->flag_posted - ctx->flag_taken);
    return NEO_OK;
}

int neo_ctx_sync_count(neo_ctx* ctx, uint64_t* blocking_waits) {
    ENTER(ctx);
    REQUIRE(blocking_waits != nullptr, "null out pointer");
    *blocking_waits = ctx->blocking_waits;
    return NEO_OK;
}

int neo_ctx_set_lane(neo_ctx* ctx, int lane) {
    ENTER(ctx);
    REQUIRE(lane >= 0 && lane < neo_ctx::LANES, "lane out of range (0 .. 3)");
    ctx->lane = lane;
    ctx->ws = ctx->ws_sets[lane];
    ctx->tp_dirsum = &ctx->tp_dirsum_sets[lane];
    ctx->mip_lws = ctx->mip_lws_sets[lane];
    return NEO_OK;
}

int neo_ctx_set_ray_grid(neo_ctx* ctx, int width, long first_ray) {
    ENTER(ctx);
    REQUIRE(width >= 0 && width % 8 == 0 && first_ray >= 0, "width must be a multiple of 8 (0: no hint), first_ray >= 0");
    ctx->ray_grid_w = width;
    ctx->ray_grid_first = width ? first_ray : 0;
    return NEO_OK;
}

int neo_ctx_stream_waits(neo_ctx* ctx, uint64_t* cross_stream_waits) {
    ENTER(ctx);
    REQUIRE(cross_stream_waits != nullptr, "null out pointer");
    *cross_stream_waits = ctx->order_waits;
    return NEO_OK;
}

int neo_ctx_set_precision(neo_ctx* ctx, int mode) {
    ENTER(ctx);
    REQUIRE(mode == 0 || mode == 1, "precision mode must be 0 (fp32 MFMA) or 1 (fp16 MFMA, hi/lo-split operands)");
    if (mode == 1 && ctx->precision != 1) {
        // entering the split arithmetic: the one-time range checks of packed weights and feature maps run again at the
        // next split launch (a frame that tripped the guard and was re-rendered exactly must trip it again, not pass
        // because "already checked")
        for (auto& sl : ctx->vanilla) sl.range_checked = 0;
        for (auto& sl : ctx->tp) sl.range_checked = 0;
        for (auto& sl : ctx->mip) sl.range_checked = 0;
        for (auto& sl : ctx->pix) sl.range_checked = 0;
        ctx->enc.range_checked = 0;
        ctx->planes_checked = ctx->latent_checked = 0;
        ctx->pix_latent_checked = false;
    }
    ctx->precision = mode;
    return NEO_OK;
}

int neo_ctx_set_timing(neo_ctx* ctx, int enable) {
    ENTER(ctx);
    for (auto& sp : ctx->spans) { (void)hipEventDestroy(sp.first); (void)hipEventDestroy(sp.second); }
    ctx->spans.clear();
    ctx->span_kernel.clear();
    ctx->span_points.clear();
    ctx->span_flops.clear();
    ctx->timed_points = 0.0;
    ctx->timed_flops = 0.0;
    ctx->timing = enable != 0;
    return NEO_OK;
}

int neo_ctx_read_timing(neo_ctx* ctx, double* total_ms, int* launches, double* total_points, double* total_flops) {
    ENTER(ctx);
    double ms = 0.0;
    for (auto& sp : ctx->spans) {
        HIP_TRY(hipEventSynchronize(sp.second));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, sp.first, sp.second));
        ms += t;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = static_cast<int>(ctx->spans.size());
    if (total_points) *total_points = ctx->timed_points;
    if (total_flops) *total_flops = ctx->timed_flops;
    return NEO_OK;
}

int neo_ctx_read_spans(neo_ctx* ctx, int capacity, double* ms, int* kernel_id, double* points, double* flops, int* count) {
    ENTER(ctx);
    REQUIRE(count != nullptr && capacity >= 0, "null count / negative capacity");
    const int n = static_cast<int>(ctx->spans.size());
    *count = n;
    for (int i = 0; i < n && i < capacity; ++i) {
        HIP_TRY(hipEventSynchronize(ctx->spans[i].second));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, ctx->spans[i].first, ctx->spans[i].second));
        if (ms) ms[i] = t;
        if (kernel_id) kernel_id[i] = ctx->span_kernel[i];
        if (points) points[i] = i < static_cast<int>(ctx->span_points.size()) ? ctx->span_points[i] : 0.0;
        if (flops) flops[i] = i < static_cast<int>(ctx->span_flops.size()) ? ctx->span_flops[i] : 0.0;
    }
    return NEO_OK;
}

int neo_raygen(neo_ctx* ctx, int H, int W, float focal, const float* c2w, float* rays_o, float* viewdirs,
               float* rays_d, float* radii, void* stream) {
    ENTER(ctx);
    REQUIRE(H >= 3 && W >= 1, "image must be at least 3 rows (the radii rule reads row H-3)");
    REQUIRE(c2w && rays_o && viewdirs && rays_d, "null pointer");
    neo::launch_raygen(H, W, focal, c2w, 0, H * W, rays_o, viewdirs, rays_d, radii, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_raygen_range(neo_ctx* ctx, int H, int W, float focal, const float* c2w, int ray0, int n_rays, float* rays_o,
                     float* viewdirs, float* rays_d, float* radii, void* stream) {
    ENTER(ctx);
    REQUIRE(H >= 3 && W >= 1, "image must be at least 3 rows (the radii rule reads row H-3)");
    REQUIRE(ray0 >= 0 && n_rays >= 0 && static_cast<long>(ray0) + n_rays <= static_cast<long>(H) * W, "ray range outside the frame");
    if (n_rays == 0) return NEO_OK;
    REQUIRE(c2w && rays_o && viewdirs && rays_d, "null pointer");
    neo::launch_raygen(H, W, focal, c2w, ray0, n_rays, rays_o, viewdirs, rays_d, radii, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_aabb_multi(neo_ctx* ctx, int n_boxes, const double* world_to_box, const double* bounds, const double* rays_o,
                   const double* rays_d, int R, uint8_t* hit_per_box, float* near, float* far, uint8_t* mask,
                   void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0 && n_boxes >= 1 && n_boxes <= 4096, "bad ray / box count");
    if (R == 0) return NEO_OK;
    REQUIRE(world_to_box && bounds && rays_o && rays_d, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<neo::BoxFrame> h(n_boxes);
    for (int b = 0; b < n_boxes; ++b) {
        for (int i = 0; i < 12; ++i) h[b].m[i] = world_to_box[b * 16 + i];
        for (int a = 0; a < 3; ++a) { h[b].lo[a] = bounds[b * 6 + a]; h[b].hi[a] = bounds[b * 6 + 3 + a]; }
    }
    if (ctx->boxes.reserve(h.size() * sizeof(neo::BoxFrame))) return NEO_ERR_NOMEM;
    HIP_TRY(hipMemcpyAsync(ctx->boxes.p, h.data(), h.size() * sizeof(neo::BoxFrame), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));          // `h` is pageable host memory about to go out of scope
    neo::launch_aabb_multi(ctx->boxes.as<neo::BoxFrame>(), n_boxes, rays_o, rays_d, R, hit_per_box, near, far, mask, s);
    return check_launch();
}

int neo_aabb_intersect(neo_ctx* ctx, const double* bounds, const double* rays_o, const double* rays_d, int R,
                       uint8_t* hit, double* tmin, double* tmax, void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0, "negative ray count");
    if (R == 0) return NEO_OK;
    REQUIRE(bounds && rays_o && rays_d, "null pointer");
    neo::launch_aabb(bounds, rays_o, rays_d, R, hit, tmin, tmax, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_intersect_sphere(neo_ctx* ctx, const float* rays_o, const float* rays_d, int R, float* far, uint8_t* ok,
                         void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0, "negative ray count");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && rays_d && far, "null pointer");
    neo::launch_sphere(rays_o, rays_d, R, far, ok, ctx->flags, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_pos_enc(neo_ctx* ctx, const float* x, int n, int C, int min_deg, int max_deg, float* out, void* stream) {
    ENTER(ctx);
    REQUIRE(n >= 0 && C >= 1 && max_deg >= min_deg, "bad shape");
    if (n == 0) return NEO_OK;
    REQUIRE(x && out, "null pointer");
    neo::launch_pos_enc(x, n, C, min_deg, max_deg, out, static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_resample(neo_ctx* ctx, const float* t_prev, const float* weights, int R, int n_prev, int n_new,
                 int descending, float* t_out, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(R >= 0 && n_new >= 1, "bad shape");
    REQUIRE(n_prev >= 4 && n_prev <= 257 && n_prev + n_new <= 1024, "unsupported sample counts");
    if (R == 0) return NEO_OK;
    REQUIRE(t_prev && weights && t_out, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float* u = ctx->get_quantiles(n_new, s);
    if (!u) return fail(NEO_ERR_HIP, "quantile table upload failed");
    if (neo::launch_resample(t_prev, n_prev, weights, u, 0, R, n_prev, n_new, descending, t_out, s))
        return fail(NEO_ERR_INVALID, "unsupported sample counts");
    return check_launch();
}

int neo_composite(neo_ctx* ctx, int mode, const float* rgbsigma, const float* t, const float* rays_d,
                  const float* t_far, int R, int N, int white_bkgd, float* rgb, float* acc, float* depth,
                  float* weights, float* lambda, void* stream) {
    ENTER(ctx);
    REQUIRE(mode >= 0 && mode <= 2, "mode must be 0, 1 or 2");
    REQUIRE(R >= 0 && N >= 1, "bad shape");
    if (R == 0) return NEO_OK;
    REQUIRE(rgbsigma && t, "null pointer");
    REQUIRE(mode == 2 || rays_d, "rays_d required");
    REQUIRE(mode != 1 || t_far, "t_far required for mode 1");
    neo::launch_composite(mode, rgbsigma, t, N, rays_d, t_far, R, N, white_bkgd, rgb, acc, depth, weights, lambda,
                          static_cast<hipStream_t>(stream));
    return check_launch();
}

int neo_vanilla_upload_mlp(neo_ctx* ctx, int slot, const float* const* weights, const float* const* biases,
                           void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(slot == 0 || slot == 1, "slot must be 0 (coarse) or 1 (fine)");
    REQUIRE(weights && biases, "null pointer table");
    for (int i = 0; i < 12; ++i) REQUIRE(weights[i] && biases[i], "null layer pointer");
    MlpSlot& sl = ctx->vanilla[slot];
    if (sl.wpack.reserve(neo::vanilla_wpack_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.bias.reserve(neo::vanilla_bias_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.heads.reserve(neo::vanilla_heads_floats() * sizeof(float))) return NEO_ERR_NOMEM;
    if (sl.wpack_h.reserve(neo::vanilla_wpack_h_bytes())) return NEO_ERR_NOMEM;
    neo::launch_vanilla_pack(weights, biases, sl.wpack.as<float>(), sl.bias.as<float>(), sl.heads.as<float>(),
                             static_cast<hipStream_t>(stream));
    if (sl.bias_hp.reserve(neo::vanilla_bias_floats() * sizeof(float)) || sl.fold_ws.reserve(neo::vanilla_fold_floats() * sizeof(float)))
        return NEO_ERR_NOMEM;
    neo::launch_vanilla_pack_h(weights, biases, sl.wpack_h.p, sl.fold_ws.as<float>(), sl.bias.as<float>(), sl.bias_hp.as<float>(),
                               static_cast<hipStream_t>(stream));
    sl.weights_epoch += 1;
    sl.ready = true;
    return check_launch();
}

static int vanilla_mlp_launch(neo_ctx* ctx, int slot, const float* rays_o, const float* dirs, const float* t,
                              int t_row_stride, int R, int N, float* out, hipStream_t s) {
    MlpSlot& sl = ctx->vanilla[slot];
    if (!sl.ready) return fail(NEO_ERR_STATE, "vanilla MLP slot %d has no weights", slot);
    if (ctx->precision == 1) guard_split_weights(sl, sl.wpack_h.p, neo::vanilla_wpack_h_bytes(), ctx->flags, s);
    ctx->span_begin(s);
    if (ctx->precision == 1) {
        neo::VanillaMlpHDev mh{sl.wpack_h.p, sl.bias_hp.as<float>(), sl.heads.as<float>(), ctx->flags};
        neo::launch_vanilla_mlp_h(mh, rays_o, dirs, t, t_row_stride, R, N, out, s);
    } else {
        neo::VanillaMlpDev m{sl.wpack.as<float>(), sl.bias.as<float>(), sl.heads.as<float>()};
        neo::launch_vanilla_mlp(m, rays_o, dirs, t, t_row_stride, R, N, out, s);
    }
    ctx->span_end(s, static_cast<double>(R) * N, 2.0 * 593408.0);  // NeRFMLP MACs/point (vanilla_nerf/model.py:44-125)
    return check_launch();
}

int neo_vanilla_mlp(neo_ctx* ctx, int slot, const float* rays_o, const float* dirs, const float* t,
                    int t_row_stride, int R, int N, float* out, void* stream) {
    ENTER(ctx);
    ORDERED(ctx, static_cast<hipStream_t>(stream));      // touches context-owned memory: ordered across streams
    REQUIRE(slot == 0 || slot == 1, "slot must be 0 or 1");
    REQUIRE(R >= 0 && N >= 1, "bad shape");
    REQUIRE(t_row_stride == 0 || t_row_stride == N, "t_row_stride must be 0 or N");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && dirs && t && out, "null pointer");
    return vanilla_mlp_launch(ctx, slot, rays_o, dirs, t, t_row_stride, R, N, out, static_cast<hipStream_t>(stream));
}

int neo_vanilla_render(neo_ctx* ctx, const float* rays_o, const float* viewdirs, const float* rays_d, int R,
                       float near, float far, int n_coarse, int n_fine, int white_bkgd, float* rgb0, float* acc0,
                       float* depth0, float* rgb1, float* acc1, float* depth1, void* stream) {
    ENTER(ctx);
    REQUIRE(R >= 0, "negative ray count");
    REQUIRE(n_coarse >= 3 && n_coarse <= 256 && n_fine >= 1 && n_coarse + 1 + n_fine <= 1024,
            "unsupported sample counts");
    if (R == 0) return NEO_OK;
    REQUIRE(rays_o && viewdirs && rays_d, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int N0 = n_coarse + 1, N1 = N0 + n_fine;
    const float* t0 = ctx->get_edges(n_coarse, near, far, s);
    const float* u = ctx->get_quantiles(n_fine, s);
    if (!t0 || !u) return fail(NEO_ERR_HIP, "constant table upload failed");
    ORDERED_LANE(ctx, static_cast<hipStream_t>(stream));      // writes this lane's workspaces only, reads the packed weights
    if (ctx->ws[0].reserve(static_cast<size_t>(R) * N0 * 16)) return NEO_ERR_NOMEM;
    if (ctx->ws[1].reserve(static_cast<size_t>(R) * N0 * 4)) return NEO_ERR_NOMEM;
    if (ctx->ws[2].reserve(static_cast<size_t>(R) * N1 * 4)) return NEO_ERR_NOMEM;
    if (ctx->ws[3].reserve(static_cast<size_t>(R) * N1 * 16)) return NEO_ERR_NOMEM;
    float* out0 = ctx->ws[0].as<float>();
    float* w0 = ctx->ws[1].as<float>();
    float* t1 = ctx->ws[2].as<float>();
    float* out1 = ctx->ws[3].as<float>();
    // level 0: shared t row (stride 0); samples along viewdirs (vanilla_nerf/model.py:158-167)
    int rc = vanilla_mlp_launch(ctx, 0, rays_o, viewdirs, t0, 0, R, N0, out0, s);
    if (rc) return rc;
    neo::launch_composite(0, out0, t0, 0, rays_d, nullptr, R, N0, white_bkgd, rgb0, acc0, depth0, w0, nullptr, s);
    // level 1: bins = mids(t0), weights[1:-1] (model.py:171-181); sort-merge
    if (neo::launch_resample(t0, 0, w0, u, 0, R, N0, n_fine, 0, t1, s)) return fail(NEO_ERR_INVALID, "unsupported sample counts");
    rc = vanilla_mlp_launch(ctx, 1, rays_o, viewdirs, t1, N1, R, N1, out1, s);
    if (rc) return rc;
    neo::launch_composite(0, out1, t1, N1, rays_d, nullptr, R, N1, white_bkgd, rgb1, acc1, depth1, nullptr, nullptr, s);
    return check_launch();
}

}  // extern "C"


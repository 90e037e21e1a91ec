// Library context and host-side helpers shared by the C-ABI translation units.
#pragma once
#include "../../include/neo360_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "kernels.h"

struct neo_ctx;

namespace neo_host {

std::string& last_error();
int fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return neo_host::fail(NEO_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define REQUIRE(cond, msg)                                                \
    do {                                                                  \
        if (!(cond)) return neo_host::fail(NEO_ERR_INVALID, "%s", msg);   \
    } while (0)

// A grow-only device buffer: the steady state of a render loop allocates nothing.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail(NEO_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        cap = bytes;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct MlpSlot {
    DevBuf wpack, bias, heads;
    DevBuf wpack_h;      // fp16 hi/lo split fragments (precision mode 1)
    DevBuf wpack_hp;     // NeRF_TP: split fragments of the pre-projected evaluator (no local-latent k-steps)
    DevBuf bias_hp, fold_ws;   // NeRF_TP pre-projected evaluators: their bias block (view layer 0 with the bottleneck folded in) + pack scratch
    DevBuf proj;         // NeRF_TP: ONE buffer [latent through this slot's [W0_loc | W3_loc] | the three tri-planes through
                         // [W0_world | W3_world] (preproject modes 2, 3) | padding], 256 fp32 channels = 1 KB per texel
    uint64_t projpl_weights = 0, projpl_scene = 0;   // (weights_epoch, scene_epoch) the plane part was computed for; 0 = never
    uint64_t weights_epoch = 0;             // bumped by every upload
    uint64_t range_checked = 0;             // weights_epoch whose split fragments passed through the range check
    uint64_t proj_weights = 0, proj_scene = 0;   // (weights_epoch, scene_epoch) `proj` was computed for; 0 = never
    int input_ch = 0;
    bool ready = false;
    void release() {
        wpack.release(); bias.release(); heads.release(); wpack_h.release(); wpack_hp.release(); proj.release();
        bias_hp.release(); fold_ws.release();
        proj_weights = proj_scene = projpl_weights = projpl_scene = 0;
        ready = false;
    }
};

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != prev && prev >= 0) (void)hipSetDevice(prev);
    }
};

#define ENTER(ctx)                                                                  \
    REQUIRE((ctx) != nullptr, "null context");                                      \
    neo_host::DeviceGuard guard_((ctx)->device);                                    \
    if (!guard_.ok) return neo_host::fail(NEO_ERR_HIP, "hipSetDevice failed")

// Split-fp16 range guard, weight side: the first split launch after an upload scans the packed hi/lo fragments for
// fp16 inf / NaN (|w| >= 65520 or a non-finite weight) and raises FLAG_SPLIT_RANGE in the context's flag word.
inline void guard_split_weights(MlpSlot& sl, const void* halves, size_t bytes, uint32_t* flags, hipStream_t s) {
    if (sl.range_checked == sl.weights_epoch) return;
    neo::launch_half_range_check(halves, bytes / 2, flags, s);
    sl.range_checked = sl.weights_epoch;
}

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NEO_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return 0;
}

}  // namespace neo_host

// Orders this call behind the previous call of the same context when the stream differs (neo_ctx::order_begin) and
// records the ordering event when the enclosing scope ends, on every exit path.  EVERY entry point that reads or writes
// context-owned device memory opens one - the evaluators and renders (workspaces, direction table, projected maps) and,
// since round 5, the uploads / set_scene / gather calls as well (packed weights, folded biases, channels-last maps: a repack
// on stream B must not overtake a frame still reading the old fragments on stream A).
struct neo_order_scope {
    neo_ctx* c;
    hipStream_t s;
    ~neo_order_scope();
};
#define ORDERED(ctx, s)                                   \
    if (int rc_ = (ctx)->order_begin(s)) return rc_;      \
    neo_order_scope order_scope_{(ctx), (s)}
#define ORDERED_LANE(ctx, s)                              \
    if (int rc_ = (ctx)->order_begin(s, true)) return rc_; \
    neo_order_scope order_scope_{(ctx), (s)}

extern "C" void neo_linspace_host(float start, float end, int steps, float* out);

struct neo_ctx {
    int device = 0;
    uint32_t* flags = nullptr;  // device word, bit0 = ray missed the unit sphere
    neo_host::MlpSlot vanilla[2];
    neo_host::MlpSlot tp[4];
    neo_host::MlpSlot mip[3];
    neo_host::MlpSlot pix[2];          // PixelNeRF coarse / fine
    int mip_shape[3][3] = {};          // width, depth, rgb per slot
    neo_host::DevBuf mip_basis;
    neo_host::DevBuf mip_lws_sets[4][3];   // layer-by-layer NeRF MLP (mip_layered.h): encoding fragments + two activation buffers, per scratch lane
    neo_host::DevBuf* mip_lws = mip_lws_sets[0];
    std::map<int, neo_host::DevBuf> mip_seed;   // R -> the level-0 histogram of neo_mip_render (sdist = [0, 1], weights = [1]) for R rays
    int mip_layered = -1;              // 1: NeRF MLP layer by layer, 0: fused evaluator, -1 (default): layer by layer from 8192 intervals
    std::map<int, neo_host::DevBuf> centre_quantiles;             // n -> linspace(1/2n, 1-1/2n-eps, n)
    // NeO-360 scene features, channels-last, context-owned
    neo_host::DevBuf latent, plane[3];
    neo::TpScene scene{};
    bool scene_ready = false;
    uint64_t scene_epoch = 0;          // bumped by every neo_tp_set_scene
    uint64_t planes_checked = 0, latent_checked = 0;   // scene_epoch whose maps passed the split range check
    // Scratch LANES (round 6): four sets (grown on first use; the Python host keeps two calls in flight) of the render workspaces and of the per-launch direction table.  A caller that drives the
    // module chunk by chunk (the reference's render_rays_test loop, neo360/model.py:861-907) alternates lanes - and streams -
    // so that chunk i + 1's evaluators start while chunk i's last workgroups drain (models.py: NeRF_TP overlap).  `lane`
    // selects the set the NEXT calls use (neo_ctx_set_lane); everything else in the context is shared by both lanes.
    static constexpr int LANES = 4;
    int lane = 0;
    neo_host::DevBuf tp_dirsum_sets[LANES];
    neo_host::DevBuf* tp_dirsum = &tp_dirsum_sets[0];   // (rays, 32): view-summed direction encodings of the current launch (k_tp_mlp_hp)
    int ray_grid_w = 0;                // neo_ctx_set_ray_grid: the next renders' rays are row-major pixels of an image this wide ...
    long ray_grid_first = 0;           // ... ray 0 of a render = pixel ray_grid_first of the frame (0 = no hint: caller's ray order)
    int preproject = 3;                // 0 off; 1 gather the latent pre-projected through the first-layer weights; 2 the tri-planes too; 3 (default): planes for the outside-sphere slots only
    // PixelNeRF scene latent: its own buffer / descriptor / ready flag (a context may hold both decoders)
    neo_host::DevBuf pix_latent;
    neo::TpScene pix_scene{};
    bool pix_scene_ready = false;
    bool pix_latent_checked = false;
    uint64_t pix_scene_epoch = 0;
    int pix_preproject = 1;            // PixelNeRF: gather the latent pre-projected through pts_linears.0 (mlp_pix_h.hip)
    std::map<int, neo_host::DevBuf> quantiles;                    // n_new -> linspace(0, fl32(1-2^-32), n_new)
    std::map<std::pair<int, uint64_t>, neo_host::DevBuf> edges;   // (n, near/far bits) -> level-0 t row
    neo_host::DevBuf ws_sets[LANES][12];                          // render workspaces (grow-only), one set per lane
    neo_host::DevBuf* ws = ws_sets[0];                            // the current lane's set
    neo_host::DevBuf boxes;                                       // neo_aabb_multi: box frames + bounds
    // pillar stage of the scene encoder (neo_enc_*): packed weights, biases (6x512), scorer heads (3x512), workspaces
    neo_host::MlpSlot enc;
    float enc_head_b[3] = {0.f, 0.f, 0.f};
    neo_host::DevBuf enc_latent, enc_axes, enc_ws[4];
    // training-side NeRFPPMLP (neo_tp_mlp_train_backward): gradient scratch (the activation tape is the caller's: a
    // training step runs several MLP forwards before the first backward)
    neo_host::DevBuf train_scratch;
    int precision = 1;   // 1 (default): fp16 MFMA with hi/lo-split operands (fp32-equivalent); 0: exact fp32 MFMA
    // deferred reads of the flag word (neo_ctx_post_flags / neo_ctx_take_flags): pinned host words + one event each
    static constexpr int FLAG_RING = 64;
    uint32_t* flag_host = nullptr;
    hipEvent_t flag_ev[FLAG_RING] = {};
    uint64_t flag_posted = 0, flag_taken = 0;    // monotone; slot = index % FLAG_RING
    uint32_t flag_carry = 0;                     // reads retired by post() while making room
    bool flag_unposted = false;                  // the last post() found the ring full and posted nothing
    uint64_t blocking_waits = 0;                 // stream / event synchronisations issued by the flag calls
    // Context-owned scratch (tp_dirsum, train_scratch, ws[]) is rewritten by every launch on whatever stream the caller
    // passes: launches of one context are therefore ORDERED across streams - a call on a stream other than the previous
    // call's first waits (device-side, hipStreamWaitEvent) for the event recorded behind that call.
    // Round 6: two ordering domains.  An EXCLUSIVE call (ORDERED: uploads, set_scene, evaluator / training / gather calls -
    // anything that may write shared context memory) is ordered behind every earlier call of the context and every later
    // call behind it.  A LANE call (ORDERED_LANE: the whole-chunk renders, which write only their lane's scratch and READ the
    // shared weights / maps) is ordered behind the last exclusive call and the previous call of ITS lane only - two lane calls
    // on different lanes and streams run concurrently.  A lane call that turns out to write shared memory after all (lazy
    // pre-projection, a first-use range check: `touch_shared()`) ends as an exclusive one.
    struct OrderPoint { hipStream_t stream = nullptr; hipEvent_t ev = nullptr; bool valid = false; };
    OrderPoint order_excl, order_lane[LANES];
    uint64_t order_waits = 0;                    // cross-stream waits inserted (tests)
    int order_depth = 0;                         // live ORDERED scopes of the current call (only the outermost waits / records)
    bool order_is_lane = false, shared_dirty = false;
    int touch_shared(hipStream_t s);            // a lane call is about to WRITE shared context memory: wait for the other lanes, end as exclusive
    int order_begin(hipStream_t s, bool lane_call = false);
    void order_end(hipStream_t s);
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    std::vector<int> span_kernel;                // which evaluator each span bracketed (neo_ctx_read_spans)
    std::vector<double> span_points, span_flops;
    int span_kernel_next = 0;                    // set by the launch site just before span_begin
    double timed_points = 0.0, timed_flops = 0.0;

    const float* get_quantiles(int n_new, hipStream_t s);
    const float* get_centre_quantiles(int n, hipStream_t s);
    const float* get_edges(int n, float near, float far, hipStream_t s);
    void span_begin(hipStream_t s);
    void span_end(hipStream_t s, double points, double flop_per_point);
};

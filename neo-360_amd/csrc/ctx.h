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
    DevBuf wpack_h;      // fp16 hi/lo split fragments (vanilla path, precision mode 1)
    int input_ch = 0;
    bool ready = false;
    void release() { wpack.release(); bias.release(); heads.release(); wpack_h.release(); ready = false; }
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

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NEO_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return 0;
}

}  // namespace neo_host

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
    std::map<int, neo_host::DevBuf> centre_quantiles;             // n -> linspace(1/2n, 1-1/2n-eps, n)
    // NeO-360 scene features, channels-last, context-owned
    neo_host::DevBuf latent, plane[3];
    neo::TpScene scene{};
    bool scene_ready = false;
    std::map<int, neo_host::DevBuf> quantiles;                    // n_new -> linspace(0, fl32(1-2^-32), n_new)
    std::map<std::pair<int, uint64_t>, neo_host::DevBuf> edges;   // (n, near/far bits) -> level-0 t row
    neo_host::DevBuf ws[12];                                      // render workspaces (grow-only)
    int precision = 0;   // 0: fp32 MFMA, 1: fp16 MFMA with hi/lo-split operands (fp32-equivalent)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    double timed_points = 0.0, timed_flops = 0.0;

    const float* get_quantiles(int n_new, hipStream_t s);
    const float* get_centre_quantiles(int n, hipStream_t s);
    const float* get_edges(int n, float near, float far, hipStream_t s);
    void span_begin(hipStream_t s);
    void span_end(hipStream_t s, double points, double flop_per_point);
};

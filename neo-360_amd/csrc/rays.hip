// Ray generation, float64 ray/AABB slab test, unit-sphere exit depth.
// All three are elementwise, HBM-bound kernels: one lane per ray, coalesced
// stores, no shared state.
#include "common.h"
#include "kernels.h"

namespace neo {

struct Pose34 { float m[12]; };

// datasets/ray_utils.py:84-104 + :133-176.
// Rays [ray0, ray0 + n) of the row-major frame are written to rows [0, n) of the outputs: a rank of a ray-sharded
// render generates only its own range.
__global__ void k_raygen(int H, int W, float focal, Pose34 c2w, int ray0, int n, float* __restrict__ rays_o,
                         float* __restrict__ viewdirs, float* __restrict__ rays_d,
                         float* __restrict__ radii) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // output row
    if (idx >= n) return;
    const int pix = ray0 + idx;
    const int row = pix / W, col = pix - row * W;
    const float half_w = (float)W / 2, half_h = (float)H / 2;
    auto world_dir = [&](int r, float* out) {
        const float cx = ((float)col - half_w) / focal;
        const float cy = -((float)r - half_h) / focal;
        const float cz = -1.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            out[a] = cx * c2w.m[a * 4 + 0] + cy * c2w.m[a * 4 + 1] + cz * c2w.m[a * 4 + 2];
    };
    float d[3];
    world_dir(row, d);
    if (radii) {
        // |d[row] - d[row+1]| * 2/sqrt(12); the last row repeats dx[-2] of an
        // (H-1)-row difference array, i.e. the difference of rows H-3 and H-2.
        const int r0 = row < H - 1 ? row : H - 3;
        float a[3], b[3];
        world_dir(r0, a);
        world_dir(r0 + 1, b);
        const float dx = sqrtf((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) +
                               (a[2] - b[2]) * (a[2] - b[2]));
        radii[idx] = dx * 2.0f / 3.46410155296325683594f;  // torch.sqrt(tensor(12, int8)) in fp32
    }
    const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float u = d[a] / nrm;
        rays_o[idx * 3 + a] = c2w.m[a * 4 + 3];
        viewdirs[idx * 3 + a] = u;
        rays_d[idx * 3 + a] = u;
    }
}

struct Box6 { double lo[3], hi[3]; };

// datasets/ray_utils.py:34-68: float64 slab test, x -> y -> z with early-outs,
// zero direction components replaced by 1e-14, origin-inside rejected.
__global__ void k_aabb(Box6 box, const double* __restrict__ rays_o, const double* __restrict__ rays_d,
                       int R, uint8_t* __restrict__ hit, double* __restrict__ tmin_out,
                       double* __restrict__ tmax_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double o[3], inv[3];
    int neg[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[a] = rays_o[r * 3 + a];
        double d = rays_d[r * 3 + a];
        if (d == 0.0) d = 1.0e-14;
        inv[a] = 1.0 / d;
        neg[a] = inv[a] < 0.0;
    }
    auto lo_of = [&](int a) { return ((neg[a] ? box.hi[a] : box.lo[a]) - o[a]) * inv[a]; };
    auto hi_of = [&](int a) { return ((neg[a] ? box.lo[a] : box.hi[a]) - o[a]) * inv[a]; };
    double tmin = lo_of(0), tmax = hi_of(0);
    bool ok = true;
#pragma unroll
    for (int a = 1; a < 3; ++a) {
        const double l = lo_of(a), h = hi_of(a);
        if (ok && (tmin > h || l > tmax)) ok = false;
        if (ok) {
            if (l > tmin) tmin = l;
            if (h < tmax) tmax = h;
        }
    }
    if (ok && (tmin < 0.0 || tmax < 0.0)) ok = false;
    if (hit) hit[r] = ok ? 1 : 0;
    if (tmin_out) tmin_out[r] = ok ? tmin : 0.0;
    if (tmax_out) tmax_out[r] = ok ? tmax : 0.0;
}

// models/neo360/helper.py:325-373: sample_rays_in_bbox = for every box { rays -> box frame (float64:
// o' = R o + t, d' = R d with [R|t] = inverse([R_box|T_box]), :325-331); slab test (:275-323); float64 -> float32
// (torch.Tensor(...), :340-344) } merged by element-wise min with 0 as "no hit" (:359-373).
// The reference evaluates R o with numpy's matmul = OpenBLAS dgemm, whose FMA kernels accumulate k-ascending as
// fma(R2, o2, fma(R1, o1, R0 * o0)) on every FMA-capable x86 host; the translation is a separate add.  Reproduced
// literally, so that masks agree bit for bit with the reference run on such a host (tests/golden/g2_aabb.npz).
__device__ __forceinline__ bool slab_test(const double* lo_b, const double* hi_b, const double* o, const double* dir,
                                          double& tmin, double& tmax) {
    double inv[3];
    int neg[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double d = dir[a];
        if (d == 0.0) d = 1.0e-14;
        inv[a] = 1.0 / d;
        neg[a] = inv[a] < 0.0;
    }
    auto lo_of = [&](int a) { return ((neg[a] ? hi_b[a] : lo_b[a]) - o[a]) * inv[a]; };
    auto hi_of = [&](int a) { return ((neg[a] ? lo_b[a] : hi_b[a]) - o[a]) * inv[a]; };
    tmin = lo_of(0);
    tmax = hi_of(0);
    bool ok = true;
#pragma unroll
    for (int a = 1; a < 3; ++a) {
        const double l = lo_of(a), h = hi_of(a);
        if (ok && (tmin > h || l > tmax)) ok = false;
        if (ok) {
            if (l > tmin) tmin = l;
            if (h < tmax) tmax = h;
        }
    }
    if (ok && (tmin < 0.0 || tmax < 0.0)) ok = false;
    return ok;
}

__global__ void k_aabb_multi(const BoxFrame* __restrict__ boxes, int n_boxes, const double* __restrict__ rays_o,
                             const double* __restrict__ rays_d, int R, uint8_t* __restrict__ hit_per_box,
                             float* __restrict__ near_out, float* __restrict__ far_out, uint8_t* __restrict__ mask) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const double o[3] = {rays_o[r * 3], rays_o[r * 3 + 1], rays_o[r * 3 + 2]};
    const double d[3] = {rays_d[r * 3], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
    float all_near = 0.0f, all_far = 0.0f;
    for (int b = 0; b < n_boxes; ++b) {
        const BoxFrame& B = boxes[b];
        double ob[3], db[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double* row = B.m + a * 4;
            ob[a] = __builtin_fma(row[2], o[2], __builtin_fma(row[1], o[1], row[0] * o[0])) + row[3];
            db[a] = __builtin_fma(row[2], d[2], __builtin_fma(row[1], d[1], row[0] * d[0]));
        }
        double tmin, tmax;
        const bool ok = slab_test(B.lo, B.hi, ob, db, tmin, tmax);
        if (hit_per_box) hit_per_box[(long)b * R + r] = ok ? 1 : 0;
        const float near = ok ? (float)tmin : 0.0f, far = ok ? (float)tmax : 0.0f;     // torch.Tensor(float64 array)
        all_near = (all_near == 0.0f || near == 0.0f) ? fmaxf(near, all_near) : fminf(near, all_near);
        all_far = (all_far == 0.0f || far == 0.0f) ? fmaxf(far, all_far) : fminf(far, all_far);
    }
    if (near_out) near_out[r] = all_near;
    if (far_out) far_out[r] = all_far;
    if (mask) mask[r] = (all_near != 0.0f && all_far != 0.0f) ? 1 : 0;
}

// models/neo360/helper.py:253-273.
__global__ void k_sphere(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int R,
                         float* __restrict__ far, uint8_t* __restrict__ ok_out, uint32_t* flags) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float f;
    const bool ok = sphere_exit(rays_o + r * 3, rays_d + r * 3, f);
    far[r] = f;
    if (ok_out) ok_out[r] = ok ? 1 : 0;
    if (!ok) atomicOr(flags, 1u);
}

void launch_raygen(int H, int W, float focal, const float* c2w, int ray0, int n, float* rays_o, float* viewdirs,
                   float* rays_d, float* radii, hipStream_t s) {
    Pose34 p;
    for (int i = 0; i < 12; ++i) p.m[i] = c2w[i];
    if (n <= 0) return;
    hipLaunchKernelGGL(k_raygen, dim3((n + 255) / 256), dim3(256), 0, s, H, W, focal, p, ray0, n, rays_o, viewdirs,
                       rays_d, radii);
}

void launch_aabb_multi(const BoxFrame* boxes, int n_boxes, const double* rays_o, const double* rays_d, int R,
                       uint8_t* hit_per_box, float* near, float* far, uint8_t* mask, hipStream_t s) {
    hipLaunchKernelGGL(k_aabb_multi, dim3((R + 255) / 256), dim3(256), 0, s, boxes, n_boxes, rays_o, rays_d, R,
                       hit_per_box, near, far, mask);
}

void launch_aabb(const double* bounds, const double* rays_o, const double* rays_d, int R, uint8_t* hit,
                 double* tmin, double* tmax, hipStream_t s) {
    Box6 b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = bounds[a]; b.hi[a] = bounds[3 + a]; }
    hipLaunchKernelGGL(k_aabb, dim3((R + 255) / 256), dim3(256), 0, s, b, rays_o, rays_d, R, hit, tmin, tmax);
}

void launch_sphere(const float* rays_o, const float* rays_d, int R, float* far, uint8_t* ok, uint32_t* flags,
                   hipStream_t s) {
    hipLaunchKernelGGL(k_sphere, dim3((R + 255) / 256), dim3(256), 0, s, rays_o, rays_d, R, far, ok, flags);
}

}  // namespace neo

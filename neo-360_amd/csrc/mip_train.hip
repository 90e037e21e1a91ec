// Training-side operators of the Mip-NeRF 360 renderer (mipnerf360/model.py:236-365 under LitMipNeRF360.training_step :436-470):
//   k_mip_encode          the 504-d integrated positional encoding of every interval as fp32 rows - conical frustum -> Gaussian ->
//                         contraction -> lift onto the 21-direction basis -> IPE (helper.py:33-88, 278-334) - the input of the
//                         MLPs when the caller composes them from linear-layer operators (api_train.hip: neo_linear_*);
//                         the same arithmetic, statement by statement, as the point set-up of the fused evaluators (mlp_mip.hip)
//   k_mip_composite_bwd   backward of compute_alpha_weights(opaque_background=True) + volumetric_rendering (helper.py:246-275):
//                         gradients of the interval weights and of the composited colour -> gradients of rgb and density
// The sample positions carry no gradient (model.py:308-309: stop_level_grad), so the encoding needs no backward.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"
#include "train_kernels.h"

namespace neo {

namespace {

constexpr int NB = 21;                 // basis directions
constexpr int ER = 8;                  // rows (intervals) per block of k_mip_encode
constexpr float EPS32 = 1.1920929e-07f;

// Gaussian of interval i of a ray, contracted: z (3) and J cov J^T (3 x 3).  conical_frustum_to_gaussian (helper.py:293-306),
// lift_gaussian with diag=False (:320-334), contract with the closed form of the reference's autograd Jacobian (:33-66).
__device__ __forceinline__ void mip_row_gaussian(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                 const float* __restrict__ radii, const float* __restrict__ tdist, int ray, int i,
                                                 int n, float (&z)[3], float (&cc)[3][3]) {
    const float t0 = tdist[(long)ray * (n + 1) + i], t1 = tdist[(long)ray * (n + 1) + i + 1];
    float o[3], d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { o[a] = rays_o[ray * 3 + a]; d[a] = rays_d[ray * 3 + a]; }
    const float rad = radii[ray];
    const float mu = (t0 + t1) / 2.0f, hw = (t1 - t0) / 2.0f;
    const float mu2 = mu * mu, hw2 = hw * hw;
    const float denom = fmaxf(3.0f * mu2 + hw2, EPS32);
    const float t_mean = mu + (2.0f * mu * hw2) / denom;
    const float hw4 = hw2 * hw2;
    const float t_var = hw2 / 3.0f - (4.0f / 15.0f) * hw4 * (12.0f * mu2 - hw2) / (denom * denom);
    float r_var = mu2 / 4.0f + (5.0f / 12.0f) * hw2 - (4.0f / 15.0f) * hw4 / denom;
    r_var = r_var * (rad * rad);
    float x[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) x[a] = d[a] * t_mean + o[a];
    const float dm = fmaxf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], 1e-10f);
    float cov[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float outer = d[a] * d[b];
            const float null_o = (a == b ? 1.0f : 0.0f) - d[a] * (d[b] / dm);
            cov[a][b] = t_var * outer + r_var * null_o;
        }
    const float msq = fmaxf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2], 1e-32f);
    float J[3][3];
    if (msq <= 1.0f) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            z[a] = x[a];
#pragma unroll
            for (int b = 0; b < 3; ++b) J[a][b] = a == b ? 1.0f : 0.0f;
        }
    } else {
        const float rt = sqrtf(msq);
        const float sc = (2.0f * rt - 1.0f) / msq;
        const float coef = 2.0f / (msq * rt) - 2.0f * sc / msq;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            z[a] = sc * x[a];
#pragma unroll
            for (int b = 0; b < 3; ++b) J[a][b] = (a == b ? sc : 0.0f) + coef * x[a] * x[b];
        }
    }
    float tmp[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) tmp[a][b] = J[a][0] * cov[0][b] + J[a][1] * cov[1][b] + J[a][2] * cov[2][b];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) cc[a][b] = tmp[a][0] * J[b][0] + tmp[a][1] * J[b][1] + tmp[a][2] * J[b][2];
}

// out (R n, 504): feature f < 252: exp(-var_j 4^k / 2) sin(mean_j 2^k), f = 21 k + j; f >= 252: the same with the phase fl32(pi/2)
__global__ __launch_bounds__(256) void k_mip_encode(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    const float* __restrict__ radii, const float* __restrict__ tdist,
                                                    const float* __restrict__ basis, int R, int n, float* __restrict__ out) {
    __shared__ float rowz[ER][12];
    __shared__ float lift[ER][2 * NB];
    const int tid = threadIdx.x;
    const long P = (long)R * n;
    const long row0 = (long)blockIdx.x * ER;
    if (tid < ER) {
        long g = row0 + tid;
        if (g >= P) g = P - 1;
        const int ray = (int)(g / n), i = (int)(g - (long)ray * n);
        float z[3], cc[3][3];
        mip_row_gaussian(rays_o, rays_d, radii, tdist, ray, i, n, z, cc);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            rowz[tid][a] = z[a];
#pragma unroll
            for (int b = 0; b < 3; ++b) rowz[tid][3 + a * 3 + b] = cc[a][b];
        }
    }
    __syncthreads();
    // lift_and_diagonalize (helper.py:70-73): mean_j = z . b_j ; var_j = sum_a b_aj (cov b_j)_a
    if (tid < ER * NB) {
        const int row = tid / NB, j = tid - row * NB;
        const float b0 = basis[j], b1 = basis[NB + j], b2 = basis[2 * NB + j];
        const float* rz = rowz[row];
        const float mj = rz[0] * b0 + rz[1] * b1 + rz[2] * b2;
        float vj = 0.f;
        const float bb[3] = {b0, b1, b2};
#pragma unroll
        for (int a = 0; a < 3; ++a) vj += bb[a] * (rz[3 + a * 3] * b0 + rz[3 + a * 3 + 1] * b1 + rz[3 + a * 3 + 2] * b2);
        lift[row][j] = mj;
        lift[row][NB + j] = vj;
    }
    __syncthreads();
    for (int idx = tid; idx < ER * 504; idx += 256) {
        const int row = idx / 504, f = idx - row * 504;
        if (row0 + row >= P) break;
        const bool shifted = f >= 252;
        const int g = shifted ? f - 252 : f;
        const int k = g / NB, j = g - k * NB;
        const float mean = lift[row][j], var = lift[row][NB + j];
        const float arg = ldexpf(mean, k);
        out[(row0 + row) * 504 + f] = expf(-0.5f * ldexpf(var, 2 * k)) * sin_cw(shifted ? arg + HALF_PI_F32 : arg);
    }
}

// One thread per ray.  w_i = alpha_i T_i, alpha_i = 1 - exp(-dd_i) (i < n - 1), alpha_{n-1} = 1, T_i = exp(-sum_{j<i} dd_j),
// dd_i = density_i (t_{i+1} - t_i) |d|;  colour = sum_i w_i rgb_i + max(1 - sum_i w_i, 0) bg.
//   G_i      = g_w_i + g_c . rgb_i - [1 - acc > 0] bg (g_c . 1)        (total gradient of w_i)
//   g_dd_j   = [j < n - 1] G_j exp(-dd_j) T_j - sum_{i > j} G_i w_i     (the last interval's dd is the constant inf)
//   g_rgb_i  = w_i g_c ;  g_density_i = g_dd_i (t_{i+1} - t_i) |d|
// Two forward sweeps: the first sums G_i w_i, the second subtracts its running prefix (double accumulators, as the forward's carry).
__global__ void k_mip_composite_bwd(const float4* __restrict__ rgbdens, const float* __restrict__ tdist,
                                    const float* __restrict__ rays_d, int R, int n, float bg, const float* __restrict__ g_w,
                                    const float* __restrict__ g_c, float4* __restrict__ g_out) {
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= R) return;
    const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
    const float* tr = tdist + (long)ray * (n + 1);
    const float4* cs = rgbdens + (long)ray * n;
    const float gc0 = g_c ? g_c[ray * 3] : 0.0f, gc1 = g_c ? g_c[ray * 3 + 1] : 0.0f, gc2 = g_c ? g_c[ray * 3 + 2] : 0.0f;
    double before = 0.0, acc = 0.0;
    for (int i = 0; i < n; ++i) {                       // acc = sum of the weights (decides the background term)
        const float dd = cs[i].w * ((tr[i + 1] - tr[i]) * dn);
        const float alpha = (i == n - 1) ? 1.0f : 1.0f - expf(-dd);
        acc += (double)(alpha * expf(-(float)before));
        before += (double)dd;
    }
    const float gbg = (1.0f - (float)acc > 0.0f) ? bg * (gc0 + gc1 + gc2) : 0.0f;
    double total = 0.0;
    before = 0.0;
    for (int i = 0; i < n; ++i) {
        const float4 c = cs[i];
        const float dd = c.w * ((tr[i + 1] - tr[i]) * dn);
        const float alpha = (i == n - 1) ? 1.0f : 1.0f - expf(-dd);
        const float w = alpha * expf(-(float)before);
        const float G = (g_w ? g_w[(long)ray * n + i] : 0.0f) + gc0 * c.x + gc1 * c.y + gc2 * c.z - gbg;
        total += (double)G * (double)w;
        before += (double)dd;
    }
    double prefix = 0.0;
    before = 0.0;
    for (int i = 0; i < n; ++i) {
        const float4 c = cs[i];
        const float delta = (tr[i + 1] - tr[i]) * dn;
        const float dd = c.w * delta;
        const float T = expf(-(float)before);
        const float alpha = (i == n - 1) ? 1.0f : 1.0f - expf(-dd);
        const float w = alpha * T;
        const float G = (g_w ? g_w[(long)ray * n + i] : 0.0f) + gc0 * c.x + gc1 * c.y + gc2 * c.z - gbg;
        prefix += (double)G * (double)w;
        const float own = (i == n - 1) ? 0.0f : G * expf(-dd) * T;
        const float g_dd = own - (float)(total - prefix);
        g_out[(long)ray * n + i] = make_float4(w * gc0, w * gc1, w * gc2, g_dd * delta);
        before += (double)dd;
    }
}

}  // namespace

void launch_mip_encode(const float* rays_o, const float* rays_d, const float* radii, const float* tdist, const float* basis, int R,
                       int n, float* out, hipStream_t s) {
    const long P = (long)R * n;
    hipLaunchKernelGGL(k_mip_encode, dim3((unsigned)((P + ER - 1) / ER)), dim3(256), 0, s, rays_o, rays_d, radii, tdist, basis, R, n, out);
}

void launch_mip_composite_bwd(const float* rgbdens, const float* tdist, const float* rays_d, int R, int n, float bg, const float* g_w,
                              const float* g_c, float* g_out, hipStream_t s) {
    hipLaunchKernelGGL(k_mip_composite_bwd, dim3((R + 63) / 64), dim3(64), 0, s, reinterpret_cast<const float4*>(rgbdens), tdist, rays_d,
                       R, n, bg, g_w, g_c, reinterpret_cast<float4*>(g_out));
}

}  // namespace neo

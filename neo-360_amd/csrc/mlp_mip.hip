// Fused Mip-NeRF 360 point evaluator (models/mipnerf360/model.py:30-176 + helper.py:33-88, 278-370):
//   conical frustum -> Gaussian (mean, full 3x3 cov) -> contraction (closed-form Jacobian)
//   -> projection onto the 21-direction geodesic basis -> integrated positional encoding (504)
//   -> MLP (PropMLP 4x256, or NeRFMLP 8x1024 with skip + bottleneck/view/rgb branch) -> activations.
//
// tile = 32 intervals; workgroup = 512 threads = 8 waves (2 per SIMD).  The 32 x W fp32
// activation tile lives in LDS (128 KB at W = 1024: one workgroup per CU), every wave owns W/8
// outputs (4 accumulator tiles at W = 1024).  The 504-wide encoding is never stored: it is
// recomputed 64 features at a time into a double-buffered 8 KB LDS tile for layer 0 and again
// for the skip layer (1008 transcendentals per point are <0.1 % of the 17.3 MFLOP per point).
// Weights (34.7 MB for the NeRF MLP) stream from L2 in MFMA fragment order; workgroups of an XCD
// run the same schedule, so each fragment is fetched from the Infinity Cache about once per XCD.
//
// Algorithmic work: PropMLP 325,888 MAC, NeRFMLP 8,672,000 MAC per interval.
#include "kernels.h"
#include "mfma_tile.h"

namespace neo {

namespace {

constexpr int TMR = 32;          // rows (intervals) per tile
constexpr int XB_LD = 64;        // streamed-encoding tile [32][64]
constexpr int NB = 21;           // basis directions
constexpr float EPS32 = 1.1920929e-07f;

// ---- packed layout (floats) --------------------------------------------------------------
struct MipLayout {
    int w_layer[8];   // offset of trunk layer i
    int kc_layer[8];
    int w_bott, w_view;
    int b_layer[8], b_bott, b_view;
    int total_w, total_b;
};

__host__ __device__ inline MipLayout mip_layout(int W, int depth, int rgb) {
    MipLayout L{};
    int ow = 0, ob = 0;
    for (int i = 0; i < depth; ++i) {
        const int kc = i == 0 ? 64 : (W / 8 + ((i == 5) ? 64 : 0));
        L.w_layer[i] = ow;
        L.kc_layer[i] = kc;
        L.b_layer[i] = ob;
        ow += (W / 32) * kc * 256;
        ob += W;
    }
    if (rgb) {
        L.w_bott = ow; ow += 8 * (W / 8) * 256;     // 256 outputs
        L.b_bott = ob; ob += 256;
        L.w_view = ow; ow += 4 * 36 * 256;          // 128 outputs, K = 256 + 27 -> 288
        L.b_view = ob; ob += 128;
    }
    L.total_w = ow;
    L.total_b = ob;
    return L;
}
// heads: density w[W] | density b (4) | rgb w[3][128] | rgb b (4)
__host__ __device__ inline int hd_db(int W) { return W; }
__host__ __device__ inline int hd_rw(int W) { return W + 4; }
__host__ __device__ inline int hd_rb(int W) { return W + 4 + 384; }
__host__ __device__ inline int hd_total(int W) { return W + 4 + 384 + 4; }

template <int NTW>
__device__ __forceinline__ void mma1(const f32x4 (&a)[NTW], const f32x4 b, f32x16 (&acc)[NTW]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[nt] = NEO_MFMA(a[nt][e], b[e], acc[nt]);
}

// acc[nt] += W-stage chunks [kc0, kc0+n) x tile chunks [0, n); this wave's N-tiles nt0..nt0+NTW-1, one M-tile.
template <int NTW, int LD, int KM>
__device__ __forceinline__ void gemm_m1(f32x16 (&acc)[NTW], const f32x4* __restrict__ wp, int KC, int nt0, int kc0,
                                        int n, const float* __restrict__ tile, const LaneCtx& L) {
    f32x4 a[2][NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) a[0][nt] = load_a(wp, KC, nt0 + nt, kc0, L.lane);
#pragma unroll 1
    for (int c = 0; c < n; c += 2) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) a[1][nt] = load_a(wp, KC, nt0 + nt, kc0 + (c + 1 < n ? c + 1 : c), L.lane);
        f32x4 b = load_b<LD, KM>(tile, 0, c, L);
        mma1<NTW>(a[0], b, acc);
        if (c + 1 < n) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) a[0][nt] = load_a(wp, KC, nt0 + nt, kc0 + (c + 2 < n ? c + 2 : c + 1), L.lane);
            b = load_b<LD, KM>(tile, 0, c + 1, L);
            mma1<NTW>(a[1], b, acc);
        }
    }
}

template <int W, int DEPTH, bool RGB>
__global__ __launch_bounds__(512, (W == 1024 ? 2 : 4)) void k_mip_mlp(MipMlpDev m, const float* __restrict__ rays_o,
                                                                      const float* __restrict__ rays_d,
                                                                      const float* __restrict__ viewdirs,
                                                                      const float* __restrict__ radii,
                                                                      const float* __restrict__ tdist, int R, int n,
                                                                      float4* __restrict__ out) {
    constexpr int NTW = W / 256;             // N-tiles per wave for W-wide layers (8 waves x NTW x 32 = W)
    constexpr int KCW = W / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem;                        // [32][W] swizzled
    float* xbuf = act + TMR * W;              // 2 x [32][64] streamed encoding
    float* lift = xbuf + 2 * TMR * XB_LD;     // [32][44]: 21 lifted means | 21 lifted variances
    float* dsm = lift + TMR * 44;             // [32][32] view-direction encoding (rgb branch)
    float* rowz = dsm + TMR * 32;             // [32][12]: contracted mean (3) | contracted covariance (9)
    LaneCtx L;
    L.init();
    const int tid = threadIdx.x;
    const long P = (long)R * n;
    const long tile0 = (long)blockIdx.x * TMR;
    const MipLayout lay = mip_layout(W, DEPTH, RGB ? 1 : 0);
    const f32x4* wp = reinterpret_cast<const f32x4*>(m.wpack);

    // ---- per-row Gaussian, contraction ------------------------------------------------------------
    if (tid < TMR) {
        long g = tile0 + tid;
        if (g >= P) g = P - 1;
        const int ray = (int)(g / n), i = (int)(g - (long)ray * n);
        const float t0 = tdist[(long)ray * (n + 1) + i], t1 = tdist[(long)ray * (n + 1) + i + 1];
        float o[3], d[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { o[a] = rays_o[ray * 3 + a]; d[a] = rays_d[ray * 3 + a]; }
        const float rad = radii[ray];
        // conical_frustum_to_gaussian (helper.py:293-306)
        const float mu = (t0 + t1) / 2.0f, hw = (t1 - t0) / 2.0f;
        const float mu2 = mu * mu, hw2 = hw * hw;
        const float denom = fmaxf(3.0f * mu2 + hw2, EPS32);
        const float t_mean = mu + (2.0f * mu * hw2) / denom;
        const float hw4 = hw2 * hw2;
        const float t_var = hw2 / 3.0f - (4.0f / 15.0f) * hw4 * (12.0f * mu2 - hw2) / (denom * denom);
        float r_var = mu2 / 4.0f + (5.0f / 12.0f) * hw2 - (4.0f / 15.0f) * hw4 / denom;
        r_var = r_var * (rad * rad);
        // lift_gaussian, diag=False (helper.py:320-334)
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
        // contract (helper.py:33-66): z, J = dz/dx (closed form of the reference's autograd Jacobian)
        const float msq = fmaxf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2], 1e-32f);
        float z[3], J[3][3];
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
        float tmp[3][3], cc[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) tmp[a][b] = J[a][0] * cov[0][b] + J[a][1] * cov[1][b] + J[a][2] * cov[2][b];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) cc[a][b] = tmp[a][0] * J[b][0] + tmp[a][1] * J[b][1] + tmp[a][2] * J[b][2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            rowz[tid * 12 + a] = z[a];
#pragma unroll
            for (int b = 0; b < 3; ++b) rowz[tid * 12 + 3 + a * 3 + b] = cc[a][b];
        }
        if (RGB) {
            // view-direction encoding, append_identity=True (helper.py:92-99): [d | sin(d 2^k) | sin(d 2^k + pi/2)], k<4
            float vd[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { vd[a] = viewdirs[ray * 3 + a]; dsm[swz_index<32, 7>(tid, a)] = vd[a]; }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float sn, cs;
                    enc_pair(vd[a], k, sn, cs);
                    dsm[swz_index<32, 7>(tid, 3 + k * 3 + a)] = sn;
                    dsm[swz_index<32, 7>(tid, 15 + k * 3 + a)] = cs;
                }
#pragma unroll
            for (int f = 27; f < 32; ++f) dsm[swz_index<32, 7>(tid, f)] = 0.0f;
        }
    }
    __syncthreads();
    // ---- lift_and_diagonalize (helper.py:70-73): mean_j = z . b_j ; var_j = sum_i b_ij (cov b_j)_i ----
    for (int idx = tid; idx < TMR * NB; idx += 512) {
        const int row = idx / NB, j = idx - row * NB;
        const float b0 = m.basis[j], b1 = m.basis[NB + j], b2 = m.basis[2 * NB + j];
        const float* rz = rowz + row * 12;
        const float mj = rz[0] * b0 + rz[1] * b1 + rz[2] * b2;
        float vj = 0.f;
        const float bb[3] = {b0, b1, b2};
#pragma unroll
        for (int a = 0; a < 3; ++a) vj += bb[a] * (rz[3 + a * 3] * b0 + rz[3 + a * 3 + 1] * b1 + rz[3 + a * 3 + 2] * b2);
        lift[row * 44 + j] = mj;
        lift[row * 44 + NB + j] = vj;
    }
    __syncthreads();

    // integrated_pos_enc (helper.py:77-88): 64 features of stage s for (row, 4 consecutive features)
    auto produce = [&](int s, float* buf) {
        const int row = tid & 31, q = tid >> 5;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = s * 64 + q * 4 + e;
            float val = 0.0f;
            if (f < 504) {
                const bool shifted = f >= 252;
                const int g = shifted ? f - 252 : f;
                const int k = g / NB, j = g - k * NB;
                const float mean = lift[row * 44 + j], var = lift[row * 44 + NB + j];
                const float arg = ldexpf(mean, k);
                val = expf(-0.5f * ldexpf(var, 2 * k)) * sin_cw(shifted ? arg + HALF_PI_F32 : arg);
            }
            v[e] = val;
        }
        *reinterpret_cast<f32x4*>(buf + row * XB_LD + ((q ^ (row & 15)) << 2)) = v;
    };
    // acc += W_layer[:, kcbase .. kcbase+64 chunks] * ipe^T, streamed through the double buffer
    auto stream_ipe = [&](f32x16 (&acc)[NTW], const f32x4* wl, int KC, int kcbase) {
        produce(0, xbuf);
        __syncthreads();
#pragma unroll 1
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) produce(s + 1, xbuf + ((s + 1) & 1) * (TMR * XB_LD));
            gemm_m1<NTW, XB_LD, 15>(acc, wl, KC, L.wv * NTW, kcbase + s * 8, 8, xbuf + (s & 1) * (TMR * XB_LD), L);
            __syncthreads();
        }
    };

    f32x16 acc[NTW];
    // ---- trunk ----
#pragma unroll 1
    for (int layer = 0; layer < DEPTH; ++layer) {
        const f32x4* wl = wp + lay.w_layer[layer] / 4;
        const int KC = lay.kc_layer[layer];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) bias_tile(acc[nt], m.bias + lay.b_layer[layer], L.wv * NTW + nt, L);
        if (layer == 0) {
            stream_ipe(acc, wl, KC, 0);
        } else {
            gemm_m1<NTW, W, 15>(acc, wl, KC, L.wv * NTW, 0, KCW, act, L);
            if (layer == 5) stream_ipe(acc, wl, KC, KCW);    // skip concat: [h | ipe]
            __syncthreads();
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) store_tile<W, 15, true>(acc[nt], act, L.wv * NTW + nt, 0, L);
        __syncthreads();
    }
    // ---- density head (VALU): 16 lanes per row ----
    float raw_density;
    {
        const int row = tid >> 4, part = tid & 15;
        constexpr int CH = W / 4 / 16;      // 16-B chunks per lane
        float s = 0.f;
#pragma unroll 4
        for (int c = 0; c < CH; ++c) {
            const int chunk = part * CH + ((c + part) % CH);
            const f32x4 h = *reinterpret_cast<const f32x4*>(act + row * W + ((chunk ^ (row & 15)) << 2));
            const f32x4 w = *reinterpret_cast<const f32x4*>(m.heads + chunk * 4);
            s += h[0] * w[0] + h[1] * w[1] + h[2] * w[2] + h[3] * w[3];
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
        raw_density = s + m.heads[hd_db(W)];
    }
    float r = 0.f, g = 0.f, b = 0.f;
    if (RGB) {
        // ---- bottleneck W -> 256 (no activation): one N-tile per wave ----
        f32x16 ab[1];
        bias_tile(ab[0], m.bias + lay.b_bott, L.wv, L);
        gemm_m1<1, W, 15>(ab, wp + lay.w_bott / 4, KCW, L.wv, 0, KCW, act, L);
        __syncthreads();
        store_tile<W, 15, false>(ab[0], act, L.wv, 0, L);
        __syncthreads();
        // ---- view layer [bottleneck 256 | dir enc 27] -> 128, ReLU: waves 0..3 ----
        if (L.wv < 4) {
            bias_tile(ab[0], m.bias + lay.b_view, L.wv, L);
            gemm_m1<1, W, 15>(ab, wp + lay.w_view / 4, 36, L.wv, 0, 32, act, L);
            gemm_m1<1, 32, 7>(ab, wp + lay.w_view / 4, 36, L.wv, 32, 4, dsm, L);
        }
        __syncthreads();
        if (L.wv < 4) store_tile<W, 15, true>(ab[0], act, L.wv, 0, L);
        __syncthreads();
        // ---- rgb head: 16 lanes per row, 8 features each ----
        const int row = tid >> 4, part = tid & 15;
        const float* wr = m.heads + hd_rw(W);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int chunk = part * 2 + c;
            const f32x4 h = *reinterpret_cast<const f32x4*>(act + row * W + ((chunk ^ (row & 15)) << 2));
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + chunk * 4);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + 128 + chunk * 4);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 256 + chunk * 4);
            r += h[0] * w0[0] + h[1] * w0[1] + h[2] * w0[2] + h[3] * w0[3];
            g += h[0] * w1[0] + h[1] * w1[1] + h[2] * w1[2] + h[3] * w1[3];
            b += h[0] * w2[0] + h[1] * w2[1] + h[2] * w2[2] + h[3] * w2[3];
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            r += __shfl_xor(r, o, 64);
            g += __shfl_xor(g, o, 64);
            b += __shfl_xor(b, o, 64);
        }
    }
    {
        const int row = tid >> 4, part = tid & 15;
        const long gi = tile0 + row;
        if (part == 0 && gi < P) {
            float4 o4;
            if (RGB) {
                o4.x = colour_act(r + m.heads[hd_rb(W)]);
                o4.y = colour_act(g + m.heads[hd_rb(W) + 1]);
                o4.z = colour_act(b + m.heads[hd_rb(W) + 2]);
            } else {
                o4.x = o4.y = o4.z = 0.0f;       // disable_rgb: zeros_like(means) (model.py:131-136)
            }
            o4.w = density_act(raw_density);     // softplus(raw + density_bias), density_bias = -1
            out[gi] = o4;
        }
    }
}

template <int W>
size_t lds_bytes() { return (size_t)(TMR * W + 2 * TMR * XB_LD + TMR * 44 + TMR * 32 + TMR * 12) * sizeof(float); }

}  // namespace

size_t mip_wpack_floats(int width, int depth, int rgb) { return mip_layout(width, depth, rgb).total_w; }
size_t mip_bias_floats(int width, int depth, int rgb) { return mip_layout(width, depth, rgb).total_b; }
size_t mip_heads_floats(int width) { return hd_total(width); }

void launch_mip_pack(int width, int depth, int rgb, const float* const* w, const float* const* b, float* wpack,
                     float* bias, float* heads, hipStream_t s) {
    const MipLayout lay = mip_layout(width, depth, rgb);
    const PackSegs none = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < depth; ++i) {
        PackSegs sg = none;
        const int kin = i == 0 ? 504 : (i == 5 ? width + 504 : width);
        sg.len[0] = kin;                      // [h | ipe] are contiguous source columns; zero padded to 8*KC
        pack_block(w[i], kin, width, lay.kc_layer[i], 0, sg, wpack + lay.w_layer[i], s);
        copy_floats(b[i], width, bias + lay.b_layer[i], s);
    }
    (void)hipMemsetAsync(heads, 0, hd_total(width) * sizeof(float), s);
    copy_floats(w[depth], width, heads, s);
    copy_floats(b[depth], 1, heads + hd_db(width), s);
    if (rgb) {
        PackSegs sb = none;
        sb.len[0] = width;
        pack_block(w[depth + 1], width, 256, width / 8, 0, sb, wpack + lay.w_bott, s);
        copy_floats(b[depth + 1], 256, bias + lay.b_bott, s);
        PackSegs sv = none;
        sv.len[0] = 283;
        pack_block(w[depth + 2], 283, 128, 36, 0, sv, wpack + lay.w_view, s);
        copy_floats(b[depth + 2], 128, bias + lay.b_view, s);
        copy_floats(w[depth + 3], 384, heads + hd_rw(width), s);
        copy_floats(b[depth + 3], 3, heads + hd_rb(width), s);
    }
}

int launch_mip_mlp(int width, int depth, int rgb, const MipMlpDev& m, const float* rays_o, const float* rays_d,
                   const float* viewdirs, const float* radii, const float* tdist, int R, int n, float* out,
                   hipStream_t s) {
    const long P = (long)R * n;
    if (P <= 0) return 0;
    const long tiles = (P + TMR - 1) / TMR;
    // per-device attribute: set on every launch (a host-side table write), not once per process
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mip_mlp<1024, 8, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<1024>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mip_mlp<256, 4, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<256>());
    if (width == 1024 && depth == 8 && rgb)
        hipLaunchKernelGGL((k_mip_mlp<1024, 8, true>), dim3((unsigned)tiles), dim3(512), lds_bytes<1024>(), s, m, rays_o,
                           rays_d, viewdirs, radii, tdist, R, n, reinterpret_cast<float4*>(out));
    else if (width == 256 && depth == 4 && !rgb)
        hipLaunchKernelGGL((k_mip_mlp<256, 4, false>), dim3((unsigned)tiles), dim3(512), lds_bytes<256>(), s, m, rays_o,
                           rays_d, viewdirs, radii, tdist, R, n, reinterpret_cast<float4*>(out));
    else
        return -1;
    return 0;
}

}  // namespace neo

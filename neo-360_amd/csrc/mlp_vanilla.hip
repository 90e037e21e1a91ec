// Fused  pos_enc -> NeRFMLP -> activations  for the vanilla NeRF path
// (vanilla_nerf/model.py:100-125 + :183-204), fp32 MFMA on gfx950.
//
// Work decomposition
//   tile      = 64 consecutive points of the flattened (ray, sample) array
//   workgroup = 256 threads = 4 wavefronts (one per SIMD); 2 workgroups per CU
//               (72 KB LDS each) so one group's epilogue / encoding VALU work
//               overlaps the other's MFMA stream.
//   GEMM      = out^T[n][point] = W[n][k] * act^T[k][point] with
//               v_mfma_f32_32x32x2_f32: A operand = weights (32 outputs x 2 k),
//               B operand = activations (2 k x 32 points), D = 32 outputs x 32 points.
//               A wave owns 64 of a layer's 256 outputs (2 N-tiles) for all 64
//               points (2 M-tiles): 4 accumulator tiles = 64 VGPR.
//   weights   : re-packed once (neo_vanilla_upload_mlp) into fragment order
//               [stage][n_tile][k_chunk(8)][lane][4], so one wave-wide 16-B load
//               (1 KiB contiguous, L2-resident: the whole MLP is 2.4 MB) feeds 4
//               MFMAs per accumulator tile with no LDS staging and no conflicts.
//               Within a chunk, MFMA m contracts k = {8c+m, 8c+4+m}.
//   acts      : the 64x256 fp32 activation tile lives in LDS (64 KB), 16-B chunks
//               XOR-swizzled by (point & 15) so both the ds_read_b128 of the B
//               fragments and the ds_write_b128 of the epilogue are conflict-free.
//               The D layout puts 4 consecutive outputs of one point in 4
//               consecutive accumulator registers -> one ds_write_b128 each.
//   x0 (the 63-d encoding, re-used by the skip layer) is kept as B fragments in
//   registers (64 VGPR); the 27-d view encoding lives in an 8 KB LDS side buffer.
//   density / rgb heads (1 and 3 outputs) run on the VALU with 4 lanes per point.
//
// Roofline: 593,408 MAC per point (algorithmic), fp32 MFMA peak 157.3 TFLOP/s.
#include "common.h"
#include "kernels.h"

namespace neo {

namespace {

constexpr int TM = 64;
constexpr int ACT_LD = 256;
constexpr int DIR_LD = 32;
constexpr int NUM_STAGES = 10;  // L0..L7, bottleneck, view layer
// stage s: outputs, 8-wide k chunks, source Linear (index into the 12 uploaded layers), true fan-in
constexpr int ST_N[NUM_STAGES] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 128};
constexpr int ST_KC[NUM_STAGES] = {8, 32, 32, 32, 32, 40, 32, 32, 32, 36};
constexpr int ST_SRC[NUM_STAGES] = {0, 1, 2, 3, 4, 5, 6, 7, 9, 8};
constexpr int ST_KIN[NUM_STAGES] = {63, 256, 256, 256, 256, 319, 256, 256, 256, 283};

constexpr int stage_w_off(int s) {  // in floats
    int o = 0;
    for (int i = 0; i < s; ++i) o += (ST_N[i] / 32) * ST_KC[i] * 256;
    return o;
}
constexpr int stage_b_off(int s) {
    int o = 0;
    for (int i = 0; i < s; ++i) o += ST_N[i];
    return o;
}
constexpr int WPACK_FLOATS = stage_w_off(NUM_STAGES);
constexpr int BIAS_FLOATS = stage_b_off(NUM_STAGES);
// heads: density w[256] | density b (4) | rgb w[3][128] | rgb b (4)
constexpr int HD_DW = 0, HD_DB = 256, HD_RW = 260, HD_RB = 644, HEADS_FLOATS = 648;

#define NEO_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct LaneCtx {
    int lane, wv, half, l31, key;
};

template <int NTW>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NTW][2], const float* __restrict__ bias, int nt0,
                                          const LaneCtx& L) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + (nt0 + nt) * 32 + 8 * g + 4 * L.half);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[nt][0][4 * g + e] = b[e];
                acc[nt][1][4 * g + e] = b[e];
            }
        }
    }
}

template <int NTW>
__device__ __forceinline__ void load_a(f32x4 (&a)[NTW], const f32x4* __restrict__ wp, int KC, int nt0, int kc,
                                       int lane) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) a[nt] = wp[((nt0 + nt) * KC + kc) * 64 + lane];
}

template <int NTW>
__device__ __forceinline__ void mma_chunk(const f32x4 (&a)[NTW], const f32x4 (&b)[2], f32x16 (&acc)[NTW][2]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = NEO_MFMA(a[nt][m], b[mt][m], acc[nt][mt]);
}

// B fragments of chunk c from a swizzled LDS tile (row stride LD floats, swizzle mask KM).
template <int LD, int KM>
__device__ __forceinline__ void load_b(f32x4 (&b)[2], const float* __restrict__ tile, int c, const LaneCtx& L) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
        b[mt] = *reinterpret_cast<const f32x4*>(tile + (mt * 32 + L.l31) * LD + ((((c << 1) + L.half) ^ (L.key & KM)) << 2));
}

// acc += W[:, kc0*8 : (kc0+n)*8] * tile^T, chunks streamed two at a time with the
// next pair's weights in flight (n is even for every stage).
template <int NTW, int LD, int KM>
__device__ __forceinline__ void gemm_lds(f32x16 (&acc)[NTW][2], const f32x4* __restrict__ wp, int KC, int nt0,
                                         int kc0, int n, const float* __restrict__ tile, const LaneCtx& L) {
    f32x4 a0[2][NTW], a1[2][NTW];
    load_a<NTW>(a0[0], wp, KC, nt0, kc0, L.lane);
    load_a<NTW>(a0[1], wp, KC, nt0, kc0 + 1, L.lane);
    for (int c = 0; c < n; c += 4) {
        f32x4 b[2];
        if (c + 2 < n) {
            load_a<NTW>(a1[0], wp, KC, nt0, kc0 + c + 2, L.lane);
            load_a<NTW>(a1[1], wp, KC, nt0, kc0 + c + 3, L.lane);
        }
        load_b<LD, KM>(b, tile, c, L);
        mma_chunk<NTW>(a0[0], b, acc);
        load_b<LD, KM>(b, tile, c + 1, L);
        mma_chunk<NTW>(a0[1], b, acc);
        if (c + 2 < n) {
            if (c + 4 < n) {
                load_a<NTW>(a0[0], wp, KC, nt0, kc0 + c + 4, L.lane);
                load_a<NTW>(a0[1], wp, KC, nt0, kc0 + c + 5, L.lane);
            }
            load_b<LD, KM>(b, tile, c + 2, L);
            mma_chunk<NTW>(a1[0], b, acc);
            load_b<LD, KM>(b, tile, c + 3, L);
            mma_chunk<NTW>(a1[1], b, acc);
        }
    }
}

// acc += W[:, kc0*8 : (kc0+8)*8] * x0^T with x0 held as register fragments.
template <int NTW>
__device__ __forceinline__ void gemm_regs(f32x16 (&acc)[NTW][2], const f32x4* __restrict__ wp, int KC, int nt0,
                                          int kc0, const f32x4 (&xf)[8][2], const LaneCtx& L) {
    f32x4 a[2][NTW];
    load_a<NTW>(a[0], wp, KC, nt0, kc0, L.lane);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c + 1 < 8) load_a<NTW>(a[(c + 1) & 1], wp, KC, nt0, kc0 + c + 1, L.lane);
        mma_chunk<NTW>(a[c & 1], xf[c], acc);
    }
}

template <int NTW, bool RELU>
__device__ __forceinline__ void store_act(const f32x16 (&acc)[NTW][2], float* __restrict__ act, int nt0,
                                          const LaneCtx& L) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[nt][mt][4 * g + e];
                    v[e] = RELU ? fmaxf(x, 0.0f) : x;
                }
                const int chunk = (nt0 + nt) * 8 + 2 * g + L.half;
                *reinterpret_cast<f32x4*>(act + (mt * 32 + L.l31) * ACT_LD + ((chunk ^ L.key) << 2)) = v;
            }
}

__device__ __forceinline__ void put_x0(float* act, int p, int f, float v) {
    act[p * ACT_LD + ((((f >> 2) ^ (p & 15))) << 2) + (f & 3)] = v;
}
__device__ __forceinline__ void put_dir(float* dsm, int p, int f, float v) {
    dsm[p * DIR_LD + ((((f >> 2) ^ (p & 7))) << 2) + (f & 3)] = v;
}

__global__ __launch_bounds__(256, 2) void k_vanilla_mlp(VanillaMlpDev m, const float* __restrict__ rays_o,
                                                         const float* __restrict__ dirs,
                                                         const float* __restrict__ t, int t_row_stride, long P,
                                                         int N, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem;                   // [64][256] swizzled
    float* dsm = smem + TM * ACT_LD;     // [64][32]  swizzled view-direction encoding
    LaneCtx L;
    L.lane = threadIdx.x & 63;
    L.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    L.half = L.lane >> 5;
    L.l31 = L.lane & 31;
    L.key = L.lane & 15;
    const long tile0 = (long)blockIdx.x * TM;
    const f32x4* wp = reinterpret_cast<const f32x4*>(m.wpack);

    // ---- encodings: wave q handles a quarter of the octaves for all 64 points ----
    {
        const int p = L.lane;
        long g = tile0 + p;
        if (g >= P) g = P - 1;
        const int ray = (int)(g / N);
        const int s = (int)(g - (long)ray * N);
        const float tt = t[(long)ray * t_row_stride + s];
        float x[3], d[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            d[a] = dirs[ray * 3 + a];
            x[a] = rays_o[ray * 3 + a] + tt * d[a];  // mul then add, as cast_rays (helper.py:20-21)
        }
        const int q = L.wv;
        const int k_lo = q == 0 ? 0 : q == 1 ? 3 : q == 2 ? 6 : 8;
        const int k_hi = q == 0 ? 3 : q == 1 ? 6 : q == 2 ? 8 : 10;
        for (int k = k_lo; k < k_hi; ++k) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float sn, cs;
                enc_pair(x[a], k, sn, cs);
                put_x0(act, p, 3 + k * 3 + a, sn);
                put_x0(act, p, 33 + k * 3 + a, cs);
            }
        }
        if (q == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) put_x0(act, p, a, x[a]);
            put_x0(act, p, 63, 0.0f);
        }
        if (q == 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) put_dir(dsm, p, a, d[a]);
#pragma unroll
            for (int f = 27; f < 32; ++f) put_dir(dsm, p, f, 0.0f);
        }
        if (q >= 2) {
            for (int k = (q - 2) * 2; k < (q - 2) * 2 + 2; ++k) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float sn, cs;
                    enc_pair(d[a], k, sn, cs);
                    put_dir(dsm, p, 3 + k * 3 + a, sn);
                    put_dir(dsm, p, 15 + k * 3 + a, cs);
                }
            }
        }
    }
    __syncthreads();

    // x0 as B fragments (8 chunks x 2 M-tiles), kept until the skip layer
    f32x4 xf[8][2];
#pragma unroll
    for (int c = 0; c < 8; ++c) load_b<ACT_LD, 15>(xf[c], act, c, L);

    f32x16 acc[2][2];
    const int nt0 = L.wv * 2;

    // ---- L0: 63 -> 256 ----
    init_bias<2>(acc, m.bias + stage_b_off(0), nt0, L);
    gemm_regs<2>(acc, wp + stage_w_off(0) / 4, ST_KC[0], nt0, 0, xf, L);
    __syncthreads();
    store_act<2, true>(acc, act, nt0, L);
    __syncthreads();

    // ---- L1..L7 (skip concat feeds L5) ----
#pragma unroll 1
    for (int s = 1; s <= 7; ++s) {
        const int woff = stage_w_off(1) + (s - 1) * 65536 + (s > 5 ? 81920 - 65536 : 0);
        const int KC = s == 5 ? 40 : 32;
        init_bias<2>(acc, m.bias + s * 256, nt0, L);
        gemm_lds<2, ACT_LD, 15>(acc, wp + woff / 4, KC, nt0, 0, 32, act, L);
        if (s == 5) gemm_regs<2>(acc, wp + woff / 4, KC, nt0, 32, xf, L);
        __syncthreads();
        store_act<2, true>(acc, act, nt0, L);
        __syncthreads();
    }

    // ---- density head on h8 (VALU, 4 lanes per point) ----
    float raw_sigma;
    {
        const int pt = L.wv * 16 + (L.lane >> 2), part = L.lane & 3;
        const float* wd = m.heads + HD_DW;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int chunk = part * 16 + ((c + 4 * part) & 15);
            const f32x4 h = *reinterpret_cast<const f32x4*>(act + pt * ACT_LD + ((chunk ^ (pt & 15)) << 2));
            const f32x4 w = *reinterpret_cast<const f32x4*>(wd + chunk * 4);
            s += h[0] * w[0] + h[1] * w[1] + h[2] * w[2] + h[3] * w[3];
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        raw_sigma = s + m.heads[HD_DB];
    }

    // ---- bottleneck: 256 -> 256, no activation ----
    init_bias<2>(acc, m.bias + stage_b_off(8), nt0, L);
    gemm_lds<2, ACT_LD, 15>(acc, wp + stage_w_off(8) / 4, 32, nt0, 0, 32, act, L);
    __syncthreads();
    store_act<2, false>(acc, act, nt0, L);
    __syncthreads();

    // ---- view layer: [bottleneck | dir enc] 283 -> 128, ReLU ----
    {
        f32x16 accv[1][2];
        const int ntv = L.wv;
        init_bias<1>(accv, m.bias + stage_b_off(9), ntv, L);
        gemm_lds<1, ACT_LD, 15>(accv, wp + stage_w_off(9) / 4, 36, ntv, 0, 32, act, L);
        gemm_lds<1, DIR_LD, 7>(accv, wp + stage_w_off(9) / 4, 36, ntv, 32, 4, dsm, L);
        __syncthreads();
        store_act<1, true>(accv, act, ntv, L);
        __syncthreads();
    }

    // ---- rgb head (VALU) + activations + store ----
    {
        const int pt = L.wv * 16 + (L.lane >> 2), part = L.lane & 3;
        const float* wr = m.heads + HD_RW;
        float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int chunk = part * 8 + ((c + 2 * part) & 7);
            const f32x4 h = *reinterpret_cast<const f32x4*>(act + pt * ACT_LD + ((chunk ^ (pt & 15)) << 2));
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + chunk * 4);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + 128 + chunk * 4);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 256 + chunk * 4);
            r += h[0] * w0[0] + h[1] * w0[1] + h[2] * w0[2] + h[3] * w0[3];
            g += h[0] * w1[0] + h[1] * w1[1] + h[2] * w1[2] + h[3] * w1[3];
            b += h[0] * w2[0] + h[1] * w2[1] + h[2] * w2[2] + h[3] * w2[3];
        }
        r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64);
        g += __shfl_xor(g, 1, 64); g += __shfl_xor(g, 2, 64);
        b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64);
        const long gi = tile0 + pt;
        if (part == 0 && gi < P) {
            out[gi] = make_float4(colour_act(r + m.heads[HD_RB]), colour_act(g + m.heads[HD_RB + 1]),
                                  colour_act(b + m.heads[HD_RB + 2]), density_act(raw_sigma));
        }
    }
}

// Re-pack one nn.Linear (out,in) into MFMA fragment order, zero-padding k >= K_in.
__global__ void k_pack_stage(const float* __restrict__ W, int n_out, int k_in, int KC, float* __restrict__ dst) {
    const int total = (n_out / 32) * KC * 256;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 3, lane = (idx >> 2) & 63, blk = idx >> 8;
        const int kc = blk % KC, nt = blk / KC;
        const int n = nt * 32 + (lane & 31);
        const int k = kc * 8 + 4 * (lane >> 5) + e;
        dst[idx] = (k < k_in) ? W[(long)n * k_in + k] : 0.0f;
    }
}

__global__ void k_copy(const float* __restrict__ src, int n, float* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

}  // namespace

size_t vanilla_wpack_floats() { return WPACK_FLOATS; }
size_t vanilla_bias_floats() { return BIAS_FLOATS; }
size_t vanilla_heads_floats() { return HEADS_FLOATS; }

void launch_vanilla_pack(const float* const* weights, const float* const* biases, float* wpack, float* bias,
                         float* heads, hipStream_t s) {
    for (int st = 0; st < NUM_STAGES; ++st) {
        const int src = ST_SRC[st];
        const int total = (ST_N[st] / 32) * ST_KC[st] * 256;
        hipLaunchKernelGGL(k_pack_stage, dim3((total + 255) / 256), dim3(256), 0, s, weights[src], ST_N[st],
                           ST_KIN[st], ST_KC[st], wpack + stage_w_off(st));
        hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, biases[src], ST_N[st], bias + stage_b_off(st));
    }
    (void)hipMemsetAsync(heads, 0, HEADS_FLOATS * sizeof(float), s);
    hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, weights[10], 256, heads + HD_DW);
    hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, biases[10], 1, heads + HD_DB);
    hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, weights[11], 384, heads + HD_RW);
    hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, biases[11], 3, heads + HD_RB);
}

void launch_vanilla_mlp(const VanillaMlpDev& m, const float* rays_o, const float* dirs, const float* t,
                        int t_row_stride, int R, int N, float* out, hipStream_t s) {
    const long P = (long)R * N;
    if (P <= 0) return;
    const size_t lds = (TM * ACT_LD + TM * DIR_LD) * sizeof(float);
    // per-device attribute: set on every launch (a host-side table write), not once per process
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_vanilla_mlp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long tiles = (P + TM - 1) / TM;
    hipLaunchKernelGGL(k_vanilla_mlp, dim3((unsigned)tiles), dim3(256), lds, s, m, rays_o, dirs, t, t_row_stride, P,
                       N, reinterpret_cast<float4*>(out));
}

}  // namespace neo

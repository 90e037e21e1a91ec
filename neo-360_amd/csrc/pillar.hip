// Pillar stage of the NeO-360 scene encoder (models/neo360/encoder_tp_fusion_conv.py:472-578):
// for every cell of the G0 x G1 x G2 world grid and every source view
//     x = [pixel-aligned ResNet latent (512, bilinear) | camera-frame xyz | masked unit direction to the camera]  (518)
//     L = depth_fc(x): 518 -> 512 -> 512 -> 512 (ReLU, ReLU, linear)                                          (:263-279)
//     s_a = Linear(512 -> 1)(ReLU(Linear(513 -> 512)([L | world coordinate a])))  for the three axes          (:364-373)
// and the three floor-plans  sum_a softmax_a(s_a) L  along x / y / z (:562-578), emitted channels-last.
// 1.58 MMAC per cell-view; the reference evaluates it once per 1024-ray chunk (300 x per frame), here it runs once per
// scene.  Dense row-wise layers over 786,432 rows: GEMM-shaped work on the fp16 matrix cores with hi/lo-split fp32
// operands (split_tile.h), one kernel per layer with the activations resident in HBM (3.2 GB of traffic per layer
// against 0.41 TMAC: matrix-bound), the 512 -> 1 scorer head fused into its hidden layer's epilogue.
//
// k_pillar_dense: workgroup = 64 rows x all 512 outputs, 8 waves (wave w: N-tiles 2w, 2w+1 x both 32-row M-tiles); the
// input is streamed 64 features at a time into a double-buffered hi/lo LDS tile (gathers / loads of stage s+1 issued
// before the MFMAs of stage s), weights come from L2 in fragment order (pack_h.hip).
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "split_tile.h"
#include "tp_common.h"

namespace neo {

namespace {

constexpr int PT = 64;                 // rows per tile
constexpr int PW = 512;                // layer width
constexpr int KS_MAIN = 32;            // k-steps of the 512 main features
constexpr int KS_ALL = 33;             // + one k-step of extras (camera xyz + direction, or the axis coordinate)

// IN: 0 = gathered latent + [cam xyz | dir] extras (first layer), 1 = rows of X (512), 2 = rows of X + axis coordinate
// EPI: 0 = bias + ReLU -> Y, 1 = bias -> Y, 2 = bias + ReLU -> dot with the 512 -> 1 head -> score
template <int IN, int EPI>
__global__ __launch_bounds__(512, 2) void k_pillar_dense(PillarGeom gm, const float* __restrict__ latent, const float* __restrict__ X,
                                                        const h8* __restrict__ wp, const float* __restrict__ bias,
                                                        const float* __restrict__ head_w, float head_b, int coord_axis,
                                                        long M, uint32_t* __restrict__ flags, float* __restrict__ Y,
                                                        float* __restrict__ score) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* hbase = reinterpret_cast<_Float16*>(smem);
    auto xbuf = [&](int b) { return HT{hbase + b * (2 * PT * 64), hbase + b * (2 * PT * 64) + PT * 64}; };   // 2 x 16 KB
    int* loc_off = reinterpret_cast<int*>(smem + 8192);          // [64][4]
    float* loc_w = smem + 8192 + 256;                            // [64][4]
    float* extra = smem + 8192 + 512;                            // [64][8]: extras of the 33rd k-step
    float* sred = smem + 8192 + 1024;                            // [8][64] scorer partial sums
    LaneCtx L;
    L.init();
    const int tid = threadIdx.x;
    const long row0 = (long)blockIdx.x * PT;
    const int KS = IN == 1 ? KS_MAIN : KS_ALL;

    // ---- per-row set-up (threads 0..63): grid cell -> world point -> camera frame, taps, extras ----
    if (IN != 1 && tid < PT) {
        long m = row0 + tid;
        if (m >= M) m = M - 1;
        const long NC = (long)gm.G0 * gm.G1 * gm.G2;
        const int v = (int)(m / NC);
        const long cell = m - (long)v * NC;
        const int i = (int)(cell / ((long)gm.G1 * gm.G2)), j = (int)((cell / gm.G2) % gm.G1), k = (int)(cell % gm.G2);
        const float w3[3] = {gm.axes[i], gm.axes[256 + j], gm.axes[512 + k]};
        if (IN == 0) {
            const float* rot = gm.rot[v];
            const float* trn = gm.trans[v];
            const float cxp = (rot[0] * w3[0] + rot[1] * w3[1] + rot[2] * w3[2]) + trn[0];
            const float cyp = (rot[3] * w3[0] + rot[4] * w3[1] + rot[5] * w3[2]) + trn[1];
            const float czp = (rot[6] * w3[0] + rot[7] * w3[1] + rot[8] * w3[2]) + trn[2];
            const float mask = czp < 1e-3f ? 1.0f : 0.0f;                       // :509
            float d[3], n2 = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) { d[a] = w3[a] - gm.cpos[v][a]; const float e = d[a] + 1e-9f; n2 += e * e; }
            const float nrm = sqrtf(n2);
            const float den = czp + 1e-9f;
            const float u = (-cxp / den) * gm.focal + gm.cx;
            const float w_ = (-cyp / den) * (-gm.focal) + gm.cy;
            const tp::TapSet t = tp::bilinear_taps(u * gm.sx - 1.0f, w_ * gm.sy - 1.0f, gm.Wf, gm.Hf);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                loc_off[tid * 4 + q] = (int)((uint32_t)(v * gm.Hf * gm.Wf + t.off[q]) * 2048u);
                loc_w[tid * 4 + q] = t.w[q];
            }
            extra[tid * 8 + 0] = cxp; extra[tid * 8 + 1] = cyp; extra[tid * 8 + 2] = czp;
#pragma unroll
            for (int a = 0; a < 3; ++a) extra[tid * 8 + 3 + a] = (d[a] / nrm) * mask;
            extra[tid * 8 + 6] = 0.f; extra[tid * 8 + 7] = 0.f;
        } else {
            extra[tid * 8] = w3[coord_axis];
#pragma unroll
            for (int a = 1; a < 8; ++a) extra[tid * 8 + a] = 0.f;
        }
    }
    __syncthreads();

    f32x16 acc[2][2];
    const int nts[2] = {2 * L.wv, 2 * L.wv + 1};
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        bias_tile(acc[nt][0], bias, nts[nt], L);
        acc[nt][1] = acc[nt][0];
    }
    // producer: 16 lanes per row, 32 rows per pass, 2 passes per 64-feature stage
    const int col4 = tid & 15, rg = tid >> 4;
    f32x4 tap[2][4];
    auto issue = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int row = rg + 32 * ps;
            if (IN == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    tap[ps][q] = tp::load_tap(latent, (uint32_t)loc_off[row * 4 + q] + 16u * col4 + 256u * s);
            } else {
                long m = row0 + row;
                if (m >= M) m = M - 1;
                tap[ps][0] = *reinterpret_cast<const f32x4*>(X + m * PW + s * 64 + col4 * 4);
            }
        }
    };
    auto finish = [&](int s) __attribute__((always_inline)) {
        const HT buf = xbuf(s & 1);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int row = rg + 32 * ps;
            f32x4 val;
            if (IN == 0) val = tp::blend4(tap[ps], *reinterpret_cast<const f32x4*>(loc_w + row * 4));
            else val = tap[ps][0];
            range_see4(L, val);
            h4 vh, vl;
            split4(val, vh, vl);
            const int o = chunk_off<64>(row, col4 >> 1) + 4 * (col4 & 1);
            *reinterpret_cast<h4*>(buf.hi + o) = vh;
            *reinterpret_cast<h4*>(buf.lo + o) = vl;
        }
    };
    // the extras k-step: features 0..7 of chunk 0 / 1 of a row, zero beyond
    auto finish_extra = [&](const HT& buf) __attribute__((always_inline)) {
        if (tid < 2 * PT) {
            const int row = tid >> 1, ch = tid & 1;
            h8 vh, vl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float val = ch == 0 ? extra[row * 8 + e] : 0.0f;
                range_see(L, val);
                _Float16 h, l;
                split(val, h, l);
                vh[e] = h;
                vl[e] = l;
            }
            const int o = chunk_off<64>(row, ch);
            *reinterpret_cast<h8*>(buf.hi + o) = vh;
            *reinterpret_cast<h8*>(buf.lo + o) = vl;
        }
    };
    h8 wh[2], wl[2];
    const char* wb = reinterpret_cast<const char*>(wp);
    uint32_t w_off[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) w_off[nt] = (uint32_t)(nts[nt] * KS * 2 * 64 + L.lane) * 16u;
    auto load_w = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            wh[nt] = *reinterpret_cast<const h8*>(wb + (w_off[nt] + 2048u * ks));
            wl[nt] = *reinterpret_cast<const h8*>(wb + (w_off[nt] + 2048u * ks + 1024u));
        }
    };
    auto mma = [&](const HT& tile, int tks) __attribute__((always_inline)) {
        h8 bh[2], bl[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int o = chunk_off<64>(mt * 32 + L.l31, (tks << 1) + L.half);
            bh[mt] = *reinterpret_cast<const h8*>(tile.hi + o);
            bl[mt] = *reinterpret_cast<const h8*>(tile.lo + o);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[nt][mt] = NEO_MFMA_H(wl[nt], bh[mt], acc[nt][mt]);
                acc[nt][mt] = NEO_MFMA_H(wh[nt], bl[mt], acc[nt][mt]);
                acc[nt][mt] = NEO_MFMA_H(wh[nt], bh[mt], acc[nt][mt]);
            }
    };
    load_w(0);
    issue(0);
    finish(0);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < 8; ++s) {
        if (s < 7) issue(s + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mma(xbuf(s & 1), q);
            if (4 * s + q + 1 < KS) load_w(4 * s + q + 1);
        }
        if (s < 7) finish(s + 1);
        else if (IN != 1) finish_extra(xbuf(0));          // stage 8 lands in buffer 0 (stage 6's, long consumed)
        __syncthreads();
    }
    if (IN != 1) {
        mma(xbuf(0), 0);
        __syncthreads();
    }
    range_commit(L, flags);

    // ---- epilogue ----
    if (EPI != 2) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const long m = row0 + mt * 32 + L.l31;
                if (m >= M) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 val;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = acc[nt][mt][4 * g + e];
                        val[e] = EPI == 0 ? fmaxf(x, 0.0f) : x;
                    }
                    *reinterpret_cast<f32x4*>(Y + m * PW + nts[nt] * 32 + 8 * g + 4 * L.half) = val;
                }
            }
    } else {
        // score = head_w . relu(hidden) + head_b: this wave's 64 outputs of each row, then across the 8 waves
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float part = 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 hw = *reinterpret_cast<const f32x4*>(head_w + nts[nt] * 32 + 8 * g + 4 * L.half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) part = __builtin_fmaf(fmaxf(acc[nt][mt][4 * g + e], 0.0f), hw[e], part);
                }
            part += __shfl_xor(part, 32, 64);
            if (L.half == 0) sred[L.wv * 64 + mt * 32 + L.l31] = part;
        }
        __syncthreads();
        if (tid < PT && row0 + tid < M) {
            float s = head_b;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += sred[w * 64 + tid];
            score[row0 + tid] = s;
        }
    }
}

// floor-plan[v][a][b][:] = sum_p softmax_p(score[v][cell(p; a, b)]) L[v][cell][:]; AXIS = the axis summed over
// (0: x -> yz plan (G1,G2); 1: y -> xz plan (G0,G2); 2: z -> xy plan (G0,G1)).  One workgroup of 128 threads per cell,
// each thread 4 channels; L rows are 2 KB contiguous.
template <int AXIS>
__global__ __launch_bounds__(128) void k_pillar_aggregate(int G0, int G1, int G2, const float* __restrict__ Lf,
                                                          const float* __restrict__ score, float* __restrict__ out) {
    const int GA = AXIS == 0 ? G0 : AXIS == 1 ? G1 : G2;
    const int A = AXIS == 0 ? G1 : G0, B = AXIS == 2 ? G1 : G2;
    const long cellab = blockIdx.x;                    // v * A * B + a * B + b
    const int v = (int)(cellab / ((long)A * B));
    const int a = (int)((cellab / B) % A), b = (int)(cellab % B);
    const long NC = (long)G0 * G1 * G2;
    auto cell = [&](int p) -> long {
        const int i = AXIS == 0 ? p : a, j = AXIS == 1 ? p : (AXIS == 0 ? a : b), k = AXIS == 2 ? p : b;
        return (long)v * NC + ((long)i * G1 + j) * G2 + k;
    };
    float mx = -__builtin_inff();
    for (int p = 0; p < GA; ++p) mx = fmaxf(mx, score[cell(p)]);
    float den = 0.f;
    for (int p = 0; p < GA; ++p) den += expf(score[cell(p)] - mx);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < GA; ++p) {
        const long c = cell(p);
        const float w = expf(score[c] - mx) / den;
        const f32x4 val = *reinterpret_cast<const f32x4*>(Lf + c * PW + threadIdx.x * 4);
        sum = sum + val * w;
    }
    *reinterpret_cast<f32x4*>(out + cellab * PW + threadIdx.x * 4) = sum;
}

}  // namespace

// fragments: stage 0 = depth_fc.0 (33 k-steps: [latent 512 | cam 3 | dir 3 | pad]), 1 = depth_fc.2 (32), 2 = depth_encoder
// (32), 3..5 = scorer hidden layers xz, yz, xy (33: [L 512 | coordinate | pad]); h8 units per stage = 16 N-tiles x KS x 128
size_t pillar_wpack_bytes() { return (size_t)(16 * 128) * (3 * KS_ALL + 2 * KS_MAIN + KS_ALL) * 16; }
static size_t stage_off_h8(int st) {
    const int ks[6] = {KS_ALL, KS_MAIN, KS_MAIN, KS_ALL, KS_ALL, KS_ALL};
    size_t o = 0;
    for (int i = 0; i < st; ++i) o += (size_t)16 * ks[i] * 128;
    return o;
}

void launch_pillar_pack(const float* const* w, void* wpack, hipStream_t s) {
    // w: depth_fc.0 (512x518), depth_fc.2, depth_encoder, agg_xz.0 (512x513), agg_yz.0, agg_xy.0
    _Float16* base = reinterpret_cast<_Float16*>(wpack);
    const PackSegs none = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    PackSegs s518 = none; s518.len[0] = 518;
    PackSegs s512 = none; s512.len[0] = 512;
    PackSegs s513 = none; s513.len[0] = 513;
    pack_h(w[0], 518, 512, KS_ALL, 0, s518, base + stage_off_h8(0) * 8, s);
    pack_h(w[1], 512, 512, KS_MAIN, 0, s512, base + stage_off_h8(1) * 8, s);
    pack_h(w[2], 512, 512, KS_MAIN, 0, s512, base + stage_off_h8(2) * 8, s);
    for (int a = 0; a < 3; ++a) pack_h(w[3 + a], 513, 512, KS_ALL, 0, s513, base + stage_off_h8(3 + a) * 8, s);
}

int launch_pillar(const PillarGeom& gm, const float* latent_cl, const void* wpack, const float* bias /* 6 x 512 */,
                  const float* head_w /* 3 x 512 */, const float* head_b_host /* 3 */, float* h1, float* h2, float* Lf,
                  float* score /* 3 x M */, uint32_t* flags, float* fp_yz, float* fp_xz, float* fp_xy, hipStream_t s) {
    if (gm.G0 > 256 || gm.G1 > 256 || gm.G2 > 256) return -1;
    const long M = (long)gm.nv * gm.G0 * gm.G1 * gm.G2;
    const unsigned tiles = (unsigned)((M + PT - 1) / PT);
    static size_t lds_pad = ~size_t(0);
    if (lds_pad == ~size_t(0)) {          // occupancy experiments: $NEO_PILLAR_LDS_PAD extra bytes of dynamic LDS
        const char* e = getenv("NEO_PILLAR_LDS_PAD");
        lds_pad = e ? (size_t)atol(e) : 0;
    }
    const size_t lds = (8192 + 1024 + 512) * sizeof(float) + lds_pad;
    if (lds > 65536) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pillar_dense<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pillar_dense<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pillar_dense<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pillar_dense<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const h8* wp = reinterpret_cast<const h8*>(wpack);
    hipLaunchKernelGGL((k_pillar_dense<0, 0>), dim3(tiles), dim3(512), lds, s, gm, latent_cl, nullptr, wp + stage_off_h8(0), bias,
                       nullptr, 0.f, 0, M, flags, h1, nullptr);
    hipLaunchKernelGGL((k_pillar_dense<1, 0>), dim3(tiles), dim3(512), lds, s, gm, nullptr, h1, wp + stage_off_h8(1), bias + 512,
                       nullptr, 0.f, 0, M, flags, h2, nullptr);
    hipLaunchKernelGGL((k_pillar_dense<1, 1>), dim3(tiles), dim3(512), lds, s, gm, nullptr, h2, wp + stage_off_h8(2), bias + 1024,
                       nullptr, 0.f, 0, M, flags, Lf, nullptr);
    // scorers: xz uses the y coordinate, yz the x coordinate, xy the z coordinate (:556-574)
    const int coord[3] = {1, 0, 2};
    for (int a = 0; a < 3; ++a)
        hipLaunchKernelGGL((k_pillar_dense<2, 2>), dim3(tiles), dim3(512), lds, s, gm, nullptr, Lf, wp + stage_off_h8(3 + a),
                           bias + 1536 + 512 * a, head_w + 512 * a, head_b_host[a], coord[a], M, flags, nullptr, score + (long)a * M);
    // softmax along y -> xz plan, along x -> yz plan, along z -> xy plan
    hipLaunchKernelGGL((k_pillar_aggregate<1>), dim3((unsigned)(gm.nv * gm.G0 * gm.G2)), dim3(128), 0, s, gm.G0, gm.G1, gm.G2, Lf, score, fp_xz);
    hipLaunchKernelGGL((k_pillar_aggregate<0>), dim3((unsigned)(gm.nv * gm.G1 * gm.G2)), dim3(128), 0, s, gm.G0, gm.G1, gm.G2, Lf, score + M, fp_yz);
    hipLaunchKernelGGL((k_pillar_aggregate<2>), dim3((unsigned)(gm.nv * gm.G0 * gm.G1)), dim3(128), 0, s, gm.G0, gm.G1, gm.G2, Lf, score + 2 * M, fp_xy);
    return 0;
}

}  // namespace neo

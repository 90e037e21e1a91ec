// PixelNeRF baseline decoder point evaluator (models/vanilla_nerf/model_pixel.py:35-131, :195-237) in the reference's
// own arithmetic: EXACT fp32 MFMA (v_mfma_f32_32x32x2_f32), the latent gathered as the reference gathers it (no
// pre-projection).  It is the kernel a range-guard retry of the split-fp16 evaluator (mlp_pix_h.hip) lands on, and what
// `precision = "f32"` selects (round 5; until then PixelNeRF existed in the split arithmetic only).
//
// Per 64-point tile, per source view: tap descriptors + direction encodings (tp_common.h) -> the 575-wide input
// [512 pixel-aligned latent | 63 pos_enc of the camera-frame point | 0] streamed 64 features at a time (9 stages) through
// a double-buffered fp32 LDS tile into ONE 128-wide GEMM (wave w = N-tile w, both M-tiles) -> ReLU -> L1, L2, L3 (ReLU).
// Everything behind relu(L3_v) is linear up to the view mean (model_pixel.py:110-131: bottleneck without activation ->
// views_linear.0 on [bottleneck | direction encoding] -> mean over the views -> ReLU), so the loop accumulates
// sum_v relu(L3_v) (registers) and sum_v dir_enc_v (LDS) and the tail runs once per tile on the view means: density head
// (ReLU), view layer 0 WITH THE BOTTLENECK FOLDED IN ([W_v0[:, :128] W_b | W_v0[:, 128:]], formed once per upload in fp64:
// pack_h.hip:launch_fold_bottleneck - the same matrix the split evaluator uses), 128 x 128 (ReLU), rgb head (sigmoid).
// The view direction of row (view, ray b, sample s) is that of ray (b*N+s) mod B of the reference chunk (:219-222).
//
// Algorithmic work per point-view 158,976 MAC + 16,896 per point (api_pix.hip).  Correctness first: one barrier pair per
// stage, no software pipelining of the gathers - this path exists so that no checkpoint can turn a frame into an exception.
#include "tp_common.h"

namespace neo {

namespace {

using tp::TM;
using tp::blend4;
using tp::pe_feature;

// fp32 fragment pack (mfma_tile.h order), k-chunks of 8 per N-tile
constexpr int KC_X = 72;                                  // 512 latent + 64 (63 pos_enc + pad)
constexpr int FX_X = 0;
constexpr int FX_1 = FX_X + 4 * KC_X * 256;
constexpr int FX_2 = FX_1 + 4 * 16 * 256;
constexpr int FX_3 = FX_2 + 4 * 16 * 256;
constexpr int FX_V0 = FX_3 + 4 * 16 * 256;                // folded view layer 0: 128 x (128 + 27 -> 32) = 20 chunks
constexpr int FX_V1 = FX_V0 + 4 * 20 * 256;
constexpr int FX_TOTAL = FX_V1 + 4 * 16 * 256;
// bias / heads blocks: the split evaluator's (mlp_pix_h.hip), B_V0 holding the FOLDED bias
constexpr int B_0 = 0, B_1 = 128, B_2 = 256, B_3 = 384, B_V0 = 640, B_V1 = 768;
constexpr int HD_DW = 0, HD_DB = 128, HD_RW = 132, HD_RB = 516;

__device__ __forceinline__ void mma8(const f32x4 a, const f32x4 (&b)[2], f32x16 (&acc)[2]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = NEO_MFMA(a[e], b[mt][e], acc[mt]);
}

// acc[mt] += W-stage chunks [kc0, kc0 + n) of N-tile nt x tile chunks [tc0, tc0 + n)
template <int LD, int KM>
__device__ __forceinline__ void gemm_rows(f32x16 (&acc)[2], const f32x4* __restrict__ wp, int KC, int nt, int kc0, int tc0,
                                          int n, const float* __restrict__ tile, const LaneCtx& L) {
    f32x4 a = load_a(wp, KC, nt, kc0, L.lane);
#pragma unroll 1
    for (int c = 0; c < n; ++c) {
        const f32x4 an = load_a(wp, KC, nt, kc0 + (c + 1 < n ? c + 1 : c), L.lane);
        f32x4 b[2];
        b[0] = load_b<LD, KM>(tile, 0, tc0 + c, L);
        b[1] = load_b<LD, KM>(tile, 1, tc0 + c, L);
        mma8(a, b, acc);
        a = an;
    }
}

__global__ __launch_bounds__(256, 2) void k_pix_mlp(TpMlpDev m, TpScene sc, TpViews views, const float* __restrict__ rays_o,
                                                   const float* __restrict__ rays_d, const float* __restrict__ viewdirs,
                                                   const float* __restrict__ tvals, int t_shared, int R, int N, int chunk,
                                                   float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem + tp::OFF_ACT;                               // [64][128] fp32, 16-B chunks XOR-swizzled with row & 15
    auto xb = [&](int b) { return act + b * (TM * 64); };          // two [64][64] stage tiles alias it
    float* dsum = smem + tp::OFF_DIR;                              // [64][32]: sum over the views of the direction encodings
    const tp::Scratch S = tp::carve(smem);
    const int* loc_off = S.loc_off;
    const float* loc_w = S.loc_w;
    const float* cam_enc = S.cam_enc;

    LaneCtx L;
    L.init();
    const int tid = threadIdx.x;
    const long P = (long)R * N;
    const long tile0 = tp::xcd_tile(blockIdx.x, (P + TM - 1) / TM) * TM;
    if (tile0 >= P) return;
    const f32x4* wp = reinterpret_cast<const f32x4*>(m.wpack);

    tp::point_setup<3>(S, tid, tile0, P, N, R, chunk, rays_o, rays_d, viewdirs, tvals, nullptr, nullptr, t_shared != 0);
    __syncthreads();

    f32x16 hsum[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = 0.f; hsum[1][r] = 0.f; }

#pragma unroll 1
    for (int v = 0; v < sc.nv; ++v) {
        tp::view_descriptors<2048>(S, L, sc, views.rot[v], views.trans[v], v, [&](int p, int f, float val) {
            const int di = swz_index<32, 7>(p, f);
            dsum[di] = v == 0 ? val : dsum[di] + val;               // (p, f) is owned by one thread in every view
        });
        __syncthreads();

        // ---- L0 over the streamed 575-wide input ----
        f32x16 acc[2];
        bias_tile(acc[0], m.bias + B_0, L.wv, L);
        acc[1] = acc[0];
        const int col4 = tid & 15, rg = tid >> 4;
        const uint32_t lane_b = 16u * col4;
#pragma unroll 1
        for (int s = 0; s < 9; ++s) {
            float* buf = xb(s & 1);
            if (s < 8) {
                // latent channels 64 s .. 64 s + 63 of rows rg, rg + 16, rg + 32, rg + 48: 4 taps of 16 B per thread and row
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = rg + 16 * q;
                    const int4 off = *reinterpret_cast<const int4*>(loc_off + row * 4);
                    f32x4 tap[4];
                    tap[0] = tp::load_tap(sc.latent, (uint32_t)off.x + lane_b + 256u * s);
                    tap[1] = tp::load_tap(sc.latent, (uint32_t)off.y + lane_b + 256u * s);
                    tap[2] = tp::load_tap(sc.latent, (uint32_t)off.z + lane_b + 256u * s);
                    tap[3] = tp::load_tap(sc.latent, (uint32_t)off.w + lane_b + 256u * s);
                    const f32x4 val = blend4(tap, *reinterpret_cast<const f32x4*>(loc_w + row * 4));
                    *reinterpret_cast<f32x4*>(buf + row * 64 + ((col4 ^ (row & 15)) << 2)) = val;
                }
            } else {
                // the 63-d encoding of the camera-frame point in the reference's feature order (helper.py:445-449) + one zero
                const int row = tid & 63, q = tid >> 6;
                const float xc[3] = {cam_enc[row * 4], cam_enc[row * 4 + 1], cam_enc[row * 4 + 2]};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 val;
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = pe_feature<3>(xc, q * 16 + c * 4 + e);
                    *reinterpret_cast<f32x4*>(buf + row * 64 + (((q * 4 + c) ^ (row & 15)) << 2)) = val;
                }
            }
            __syncthreads();       // stage s is complete (and every wave has finished reading stage s - 1's buffer)
            gemm_rows<64, 15>(acc, wp + FX_X / 4, KC_X, L.wv, s * 8, 0, 8, buf, L);
            // the buffer written NEXT (s + 1) is the one read at stage s - 1; all waves finished those reads before the barrier above
        }
        __syncthreads();           // the stage tiles alias `act`
        store_tile<128, 15, true>(acc[0], act, L.wv, 0, L);
        store_tile<128, 15, true>(acc[1], act, L.wv, 1, L);
        __syncthreads();
        // ---- L1, L2, L3 ----
#pragma unroll 1
        for (int layer = 1; layer <= 3; ++layer) {
            bias_tile(acc[0], m.bias + (layer == 1 ? B_1 : layer == 2 ? B_2 : B_3), L.wv, L);
            acc[1] = acc[0];
            gemm_rows<128, 15>(acc, wp + (layer == 1 ? FX_1 : layer == 2 ? FX_2 : FX_3) / 4, 16, L.wv, 0, 0, 16, act, L);
            __syncthreads();
            if (layer < 3) {
                store_tile<128, 15, true>(acc[0], act, L.wv, 0, L);
                store_tile<128, 15, true>(acc[1], act, L.wv, 1, L);
                __syncthreads();
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            hsum[0][r] += fmaxf(acc[0][r], 0.0f);
            hsum[1][r] += fmaxf(acc[1][r], 0.0f);
        }
    }

    // ---- view means -> density head, folded view layer 0, 128 x 128, rgb head ----
    const float nvf = (float)sc.nv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = hsum[0][r] / nvf; hsum[1][r] = hsum[1][r] / nvf; }
    store_tile<128, 15, false>(hsum[0], act, L.wv, 0, L);
    store_tile<128, 15, false>(hsum[1], act, L.wv, 1, L);
    {
        const int p = tid >> 2, f0 = (tid & 3) << 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int di = swz_index<32, 7>(p, f0 + j);
            dsum[di] = f0 + j < 27 ? dsum[di] / nvf : 0.0f;          // features 27..31 are padding (view_descriptors wrote zeros)
        }
    }
    __syncthreads();
    float sigma = 0.0f;
    if (tid < TM) {
        float sg = m.heads[HD_DB];
#pragma unroll 4
        for (int k = 0; k < 128; ++k) sg = fmaf(act[swz_index<128, 15>(tid, k)], m.heads[HD_DW + k], sg);
        sigma = fmaxf(sg, 0.0f);                                      // model_pixel.py:165: ReLU density
    }
    f32x16 y[2];
    bias_tile(y[0], m.bias + B_V0, L.wv, L);
    y[1] = y[0];
    gemm_rows<128, 15>(y, wp + FX_V0 / 4, 20, L.wv, 0, 0, 16, act, L);
    gemm_rows<32, 7>(y, wp + FX_V0 / 4, 20, L.wv, 16, 0, 4, dsum, L);
    __syncthreads();
    store_tile<128, 15, true>(y[0], act, L.wv, 0, L);
    store_tile<128, 15, true>(y[1], act, L.wv, 1, L);
    __syncthreads();
    bias_tile(y[0], m.bias + B_V1, L.wv, L);
    y[1] = y[0];
    gemm_rows<128, 15>(y, wp + FX_V1 / 4, 16, L.wv, 0, 0, 16, act, L);
    __syncthreads();
    store_tile<128, 15, true>(y[0], act, L.wv, 0, L);
    store_tile<128, 15, true>(y[1], act, L.wv, 1, L);
    __syncthreads();
    if (tid < TM) {
        const float* wr = m.heads + HD_RW;
        float r = m.heads[HD_RB], g = m.heads[HD_RB + 1], b = m.heads[HD_RB + 2];
#pragma unroll 4
        for (int k = 0; k < 128; ++k) {
            const float h = act[swz_index<128, 15>(tid, k)];
            r = fmaf(h, wr[k], r);
            g = fmaf(h, wr[128 + k], g);
            b = fmaf(h, wr[256 + k], b);
        }
        const long gi = tile0 + tid;
        if (gi < P) out[gi] = make_float4(sigmoid_act(r), sigmoid_act(g), sigmoid_act(b), sigma);
    }
}

}  // namespace

size_t pix_wpack_floats() { return (size_t)FX_TOTAL; }

int pix_kc_x() { return KC_X; }

void launch_pix_pack(const float* const* w, const float* const* b, float* fold_v0, const float* bias_src, float* bias_f32,
                     float* wpack, hipStream_t s) {
    // w / b order: pts_linears.0..3, views_linear.0, views_linear.1, bottleneck, density, rgb.  bias_f32 = this evaluator's own
    // copy of the bias block with view layer 0's entry folded; fold_v0 = 128 x 155 floats of scratch for the folded matrix
    // [W_v0[:, :128] W_b | W_v0[:, 128:]].  Stage X: packed k = [latent 512 | pos_enc 63 | 0] <- source columns
    // [pos_enc 63 | latent 512]; its first 64 chunks per N-tile are the latent columns k_tp_preproject reads (pix_kc_x()).
    (void)hipMemcpyAsync(bias_f32, bias_src, 896 * sizeof(float), hipMemcpyDeviceToDevice, s);
    launch_fold_bottleneck(w[4], w[6], b[6], b[4], 128, 128, 128, 27, fold_v0, bias_f32 + B_V0, s);
    const PackSegs x = {{0, 512, 0}, {512, 63, 0}, {63, 0, 0}};
    pack_block(w[0], 575, 128, KC_X, 0, x, wpack + FX_X, s);
    const PackSegs p128 = {{0, 0, 0}, {128, 0, 0}, {0, 0, 0}};
    pack_block(w[1], 128, 128, 16, 0, p128, wpack + FX_1, s);
    pack_block(w[2], 128, 128, 16, 0, p128, wpack + FX_2, s);
    pack_block(w[3], 128, 128, 16, 0, p128, wpack + FX_3, s);
    const PackSegs v0 = {{0, 0, 0}, {155, 0, 0}, {0, 0, 0}};
    pack_block(fold_v0, 155, 128, 20, 0, v0, wpack + FX_V0, s);
    pack_block(w[5], 128, 128, 16, 0, p128, wpack + FX_V1, s);
}

void launch_pix_mlp(const TpMlpDev& m, const TpScene& sc, const TpViews& views, const float* rays_o, const float* rays_d,
                    const float* viewdirs, const float* tvals, int t_shared, int R, int N, int chunk, float* out, hipStream_t s) {
    const long P = (long)R * N;
    if (P <= 0) return;
    const size_t lds = tp::LDS_WORDS * sizeof(float);
    const long tiles = tp::xcd_grid((P + TM - 1) / TM);
    hipLaunchKernelGGL(k_pix_mlp, dim3((unsigned)tiles), dim3(256), lds, s, m, sc, views, rays_o, rays_d, viewdirs, tvals,
                       t_shared, R, N, chunk, reinterpret_cast<float4*>(out));
}

}  // namespace neo

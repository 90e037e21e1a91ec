// Device code shared by the NeO-360 point evaluators (fp32-MFMA kernel mlp_tp.hip and the
// split-fp16 kernel mlp_tp_h.hip): LDS carve of the per-tile scratch, bilinear tap descriptors,
// per-point world-space setup (incl. the inverted-sphere parameterisation) and the per-view
// descriptor pass (world2camera, projection, 4-tap offsets / weights, view-direction encoding).
#pragma once
#include "kernels.h"
#include "mfma_tile.h"

namespace neo {
namespace tp {

constexpr int TM = 64;

// LDS carve (in 4-byte words).  The first 10240 words hold the kernel-specific activation /
// streamed-input tiles (32 KB) and the view-direction encoding (8 KB); the rest is shared scratch.
constexpr int OFF_ACT = 0;
constexpr int OFF_DIR = OFF_ACT + TM * 128;
constexpr int OFF_LOC_OFF = OFF_DIR + TM * 32;           // int[64][4]
constexpr int OFF_LOC_W = OFF_LOC_OFF + TM * 4;
constexpr int OFF_PL_OFF = OFF_LOC_W + TM * 4;           // int[3][64][4]
constexpr int OFF_PL_W = OFF_PL_OFF + 3 * TM * 4;
constexpr int OFF_CAM = OFF_PL_W + 3 * TM * 4;           // float[64][4]: camera-frame point (+1/r) for pos_enc
constexpr int OFF_PE = OFF_CAM + TM * 4;                 // float[64][4]: world point to encode (+1/r)
constexpr int OFF_FEAT = OFF_PE + TM * 4;                // float[64][4]: world point for feature lookups
constexpr int OFF_VDIR = OFF_FEAT + TM * 4;              // float[64][4]: world view direction (Q1-indexed ray)
constexpr int OFF_DENSW = OFF_VDIR + TM * 4;             // float[128]: density-head weights (split-fp16 kernel)
constexpr int LDS_WORDS = OFF_DENSW + 128;

// Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2).  Consecutive tiles hold neighbouring
// samples that share feature texels, so each XCD is given a CONTIGUOUS range of tiles: workgroup b works on tile
// (b % 8) * ceil(T / 8) + b / 8.  The launch rounds the grid up to a multiple of 8; surplus workgroups exit.
constexpr int XCDS = 8;
__device__ __forceinline__ long xcd_tile(unsigned bid, long tiles) {
#ifdef NEO_NO_XCD_REMAP
    return bid;
#else
    const long per = (tiles + XCDS - 1) / XCDS;
    return (long)(bid % XCDS) * per + bid / XCDS;
#endif
}
inline long xcd_grid(long tiles) { return ((tiles + XCDS - 1) / XCDS) * XCDS; }

struct Scratch {
    int* loc_off; float* loc_w; int* pl_off; float* pl_w;
    float* cam_enc; float* pe_world; float* feat_world; float* vdir_world;
};
__device__ __forceinline__ Scratch carve(float* smem) {
    Scratch S;
    S.loc_off = reinterpret_cast<int*>(smem + OFF_LOC_OFF);
    S.loc_w = smem + OFF_LOC_W;
    S.pl_off = reinterpret_cast<int*>(smem + OFF_PL_OFF);
    S.pl_w = smem + OFF_PL_W;
    S.cam_enc = smem + OFF_CAM;
    S.pe_world = smem + OFF_PE;
    S.feat_world = smem + OFF_FEAT;
    S.vdir_world = smem + OFF_VDIR;
    return S;
}

struct TapSet {
    int off[4];
    float w[4];
};

// Bilinear taps of F.grid_sample(align_corners=True, padding zeros) at normalised (gx, gy)
// on a Wd x Hd map; offsets in texels, invalid taps get weight 0 / offset 0.  (view_descriptors turns
// the texel indices into BYTE offsets of the channels-last maps: x2048 for the 512-channel latent,
// x512 for the 128-channel planes; the largest, 3x240x320x2048 B, fits 32 bits.)
// Tap order nw, ne, sw, se; weights (x1-x)(y1-y), (x-x0)(y1-y), (x1-x)(y-y0), (x-x0)(y-y0).
__device__ __forceinline__ TapSet bilinear_taps(float gx, float gy, int Wd, int Hd) {
    const float x = ((gx + 1.0f) / 2.0f) * (float)(Wd - 1);
    const float y = ((gy + 1.0f) / 2.0f) * (float)(Hd - 1);
    const float x0 = floorf(x), y0 = floorf(y);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    const float wx0 = x1 - x, wx1 = x - x0, wy0 = y1 - y, wy1 = y - y0;
    const float xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
    const float ws[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
    TapSet t;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool ok = xs[k] >= 0.0f && xs[k] <= (float)(Wd - 1) && ys[k] >= 0.0f && ys[k] <= (float)(Hd - 1);
        t.off[k] = ok ? (int)ys[k] * Wd + (int)xs[k] : 0;
        t.w[k] = ok ? ws[k] : 0.0f;
    }
    return t;
}

__device__ __forceinline__ f32x4 blend4(const f32x4 (&tap)[4], const f32x4 w) {
    // nw*w0 + ne*w1 + sw*w2 + se*w3, accumulated in that order; fused multiply-adds (one rounding per
    // tap instead of torch's two: the blends differ from grid_sample's by <= 1 ulp of the feature)
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = tap[0][e] * w[0];
        a = __builtin_fmaf(tap[1][e], w[1], a);
        a = __builtin_fmaf(tap[2][e], w[2], a);
        a = __builtin_fmaf(tap[3][e], w[3], a);
        v[e] = a;
    }
    return v;
}

// the same blend ADDED to a running sum (tri-planes: the three maps of a row group are summed; chaining the fused multiply-adds
// through the sum saves the separate add per value - one rounding per tap instead of per tap and per map)
__device__ __forceinline__ f32x4 blend4_acc(const f32x4 (&tap)[4], const f32x4 w, const f32x4 acc) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = __builtin_fmaf(tap[0][e], w[0], acc[e]);
        a = __builtin_fmaf(tap[1][e], w[1], a);
        a = __builtin_fmaf(tap[2][e], w[2], a);
        a = __builtin_fmaf(tap[3][e], w[3], a);
        v[e] = a;
    }
    return v;
}

// 16 B of gathered features at byte offset `off` (32-bit) from a uniform base: the address stays
// SGPR base + one VGPR offset instead of 64-bit per-lane pointer arithmetic
__device__ __forceinline__ f32x4 load_tap(const float* __restrict__ base, uint32_t off) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
}

// feature f of the positional encoding of a C-vector x (C = 3 or 4, 10 octaves): pad -> 0
template <int C>
__device__ __forceinline__ float pe_feature(const float* x, int f) {
    if (f < C) return x[f];
    const int g = f - C;
    if (g < 10 * C) return sin_cw(ldexpf(x[g % C], g / C));
    const int h = g - 10 * C;
    if (h < 10 * C) return sin_cw(ldexpf(x[h % C], h / C) + HALF_PI_F32);
    return 0.0f;
}


// Virtual point index (tile order) -> the point it stands for: identity unless the launch carries a pixel-grid hint
// (TpScene::grid_w), then the rays of every WHOLE band of 2^ph image rows inside the launch are visited patch by patch
// (2^pw x 2^ph pixels, row-major inside a patch, patches left to right).  A bijection on the launch's points; rays outside whole
// bands (a shard's ragged ends) keep their place.
__device__ __forceinline__ long patch_point(long gv, int N, int R, int grid_w, long grid_first, int pw = 3, int ph = 3) {
    if (grid_w <= 0) return gv;
    const long rayv = gv / N;
    const int s = (int)(gv - rayv * N);
    const long band = (long)grid_w << ph;
    const long G = grid_first + rayv;
    const long b = G / band;
    if (b * band < grid_first || (b + 1) * band > grid_first + R) return gv;
    const int k = (int)(G - b * band), r = k & ((1 << (pw + ph)) - 1);
    const long Gt = b * band + (long)(r >> pw) * grid_w + ((long)(k >> (pw + ph)) << pw) + (r & ((1 << pw) - 1));
    return (Gt - grid_first) * N + s;
}

// ---- per-point world-space quantities (once per tile; threads 0..63) ---------------------------
// row `tid` of the tile (any thread may compute any row)
template <int PE_C>
__device__ __forceinline__ void point_setup_row(const Scratch& S, int tid, long tile0, long P, int N, int R, int chunk,
                                                const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                const float* __restrict__ viewdirs, const float* __restrict__ tvals,
                                                const float* __restrict__ far_arr, uint32_t* __restrict__ flags,
                                                bool t_shared = false, int grid_w = 0, long grid_first = 0, int pw = 3, int ph = 3) {
    float* pe_world = S.pe_world;
    float* feat_world = S.feat_world;
    float* vdir_world = S.vdir_world;
    {
        long g = tile0 + tid;
        if (g >= P) g = P - 1;
        g = patch_point(g, N, R, grid_w, grid_first, pw, ph);
        const int ray = (int)(g / N);
        const int s = (int)(g - (long)ray * N);
        const int c0 = (ray / chunk) * chunk;                    // first ray of this ray's reference chunk
        const int bc = min(chunk, R - c0);                       // rays in that chunk (last one may be short)
        const int gl = (ray - c0) * N + s;                       // flattened (ray, sample) index inside the chunk
        const int dray = c0 + gl % bc;                           // neo360/model.py:357-360 tiling: direction of ray (b*N+s) mod B
        const float tv = tvals[t_shared ? (long)s : g];        // t_shared: one row of sample positions for all rays
        float o[3], d[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { o[a] = rays_o[ray * 3 + a]; d[a] = rays_d[ray * 3 + a]; }
        if (PE_C == 3) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float x = o[a] + tv * d[a];
                pe_world[tid * 4 + a] = x;
                feat_world[tid * 4 + a] = x;
            }
            pe_world[tid * 4 + 3] = 0.0f;
        } else {
            // inverted-sphere point (neo360/helper.py:401-451) and the linear lookup point
            // o + (far(1-s) + 3 s) d (helper.py:59-73, :232-246)
            const float far = far_arr[ray];
            const float dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            const float d1 = -(d[0] * o[0] + d[1] * o[1] + d[2] * o[2]) / dd;
            float pm[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) pm[a] = o[a] + d1 * d[a];
            const float rmid = sqrtf(pm[0] * pm[0] + pm[1] * pm[1] + pm[2] * pm[2]);
            const float inv_len = 1.0f / sqrtf(dd);
            const float margin = 1.0f - rmid * rmid;
            if (!(margin >= 0.0f)) atomicOr(flags, 1u);
            const float d2 = sqrtf(margin) * inv_len;
            float ps[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) ps[a] = o[a] + (d1 + d2) * d[a];
            float ax[3] = {o[1] * ps[2] - o[2] * ps[1], o[2] * ps[0] - o[0] * ps[2], o[0] * ps[1] - o[1] * ps[0]};
            const float an = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
#pragma unroll
            for (int a = 0; a < 3; ++a) ax[a] = ax[a] / an;
            const float ang = asinf(rmid) - asinf(rmid * tv);
            const float ca = cosf(ang), sa = sinf(ang);
            const float cr[3] = {ax[1] * ps[2] - ax[2] * ps[1], ax[2] * ps[0] - ax[0] * ps[2], ax[0] * ps[1] - ax[1] * ps[0]};
            const float dotp = ax[0] * ps[0] + ax[1] * ps[1] + ax[2] * ps[2];
            float tn[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) tn[a] = ps[a] * ca + cr[a] * sa + ax[a] * dotp * (1.0f - ca);
            const float nn = sqrtf(tn[0] * tn[0] + tn[1] * tn[1] + tn[2] * tn[2]) + 1e-10f;
            const float tl = far * (1.0f - tv) + 3.0f * tv;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                pe_world[tid * 4 + a] = tn[a] / nn;
                feat_world[tid * 4 + a] = o[a] + tl * d[a];
            }
            pe_world[tid * 4 + 3] = tv;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) vdir_world[tid * 4 + a] = viewdirs[dray * 3 + a];
        vdir_world[tid * 4 + 3] = __int_as_float(dray);          // the ray whose direction this point carries (quirk Q1)
    }
}

template <int PE_C>
__device__ __forceinline__ void point_setup(const Scratch& S, int tid, long tile0, long P, int N, int R, int chunk,
                                            const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                            const float* __restrict__ viewdirs, const float* __restrict__ tvals,
                                            const float* __restrict__ far_arr, uint32_t* __restrict__ flags,
                                            bool t_shared = false, int grid_w = 0, long grid_first = 0, int pw = 3, int ph = 3) {
    if (tid < TM)
        point_setup_row<PE_C>(S, tid, tile0, P, N, R, chunk, rays_o, rays_d, viewdirs, tvals, far_arr, flags, t_shared, grid_w, grid_first, pw, ph);
}

// ---- per-view descriptors (all 4 waves; lane = row) ---------------------------------------------------
// put_dir(row, feature, value) stores one feature of the 27(+5 pad)-wide view-direction encoding.
// LOC_TEXEL_BYTES: bytes per latent texel the local taps address (512 fp32 channels, or 256 for the
// pre-projected map of mlp_tp_hp.hip).
// DIRS = false: the view-direction encoding is not computed here (mlp_tp_hp.hip takes the per-ray sum over the views
// from a table built once per launch, k_tp_dirsum).
// PL_TEXEL_BYTES: bytes per tri-plane texel the plane taps address (128 fp32 channels, or 256 for the pre-projected
// planes of mlp_tp_hpp.hip).
template <int LOC_TEXEL_BYTES = 2048, bool DIRS = true, int PL_TEXEL_BYTES = 128 * 4, class PutDir>
// pl_base_texels: added to this wave's tri-plane texel indices (mlp_tp_hpp.hip keeps the four projected maps in ONE buffer
// and addresses them from one base pointer; 0 for every other caller).
__device__ __forceinline__ void view_descriptors(const Scratch& S, const LaneCtx& L, const TpScene& sc,
                                                 const float* rot, const float* trn, int v, PutDir put_dir,
                                                 int pl_base_texels = 0) {
    int* loc_off = S.loc_off; float* loc_w = S.loc_w; int* pl_off = S.pl_off; float* pl_w = S.pl_w;
    float* cam_enc = S.cam_enc; const float* pe_world = S.pe_world; const float* feat_world = S.feat_world;
    const float* vdir_world = S.vdir_world;
        {
            const int p = L.lane;
            const float fx = feat_world[p * 4], fy = feat_world[p * 4 + 1], fz = feat_world[p * 4 + 2];
            // world2camera (neo360/util.py:52-70): R^T x then + (-R^T t)
            const float cx_ = (rot[0] * fx + rot[1] * fy + rot[2] * fz) + trn[0];
            const float cy_ = (rot[3] * fx + rot[4] * fy + rot[5] * fz) + trn[1];
            const float cz_ = (rot[6] * fx + rot[7] * fy + rot[8] * fz) + trn[2];
            TapSet t;
            int* dst_off;
            float* dst_w;
            int base, texel_bytes;
            if (L.wv == 0) {
                // pixel-aligned latent (neo360/model.py:239-264, encoder_pn.py:116-150), view 0's intrinsics
                const float den = cz_ + 1e-9f;
                const float u = (-cx_ / den) * sc.focal + sc.cx;
                const float w_ = (-cy_ / den) * (sc.fy_sign * sc.focal) + sc.cy;
                t = bilinear_taps(u * sc.sx - 1.0f, w_ * sc.sy - 1.0f, sc.Wf, sc.Hf);
                dst_off = loc_off; dst_w = loc_w;
                base = v * sc.Hf * sc.Wf;
                texel_bytes = LOC_TEXEL_BYTES;
                // camera-frame point that gets encoded (fg: same point; bg: the unit-sphere point)
                const float ex = pe_world[p * 4], ey = pe_world[p * 4 + 1], ez = pe_world[p * 4 + 2];
                cam_enc[p * 4 + 0] = (rot[0] * ex + rot[1] * ey + rot[2] * ez) + trn[0];
                cam_enc[p * 4 + 1] = (rot[3] * ex + rot[4] * ey + rot[5] * ez) + trn[1];
                cam_enc[p * 4 + 2] = (rot[6] * ex + rot[7] * ey + rot[8] * ez) + trn[2];
                cam_enc[p * 4 + 3] = pe_world[p * 4 + 3];
            } else {
                // tri-planes (encoder_tp_fusion_conv.py:122-209): camera coordinates used directly as
                // grid coordinates; xz -> (x,z), xy -> (x,y), yz -> (y,z)
                const float ga = L.wv == 3 ? cy_ : cx_;
                const float gb = L.wv == 2 ? cy_ : cz_;
                t = bilinear_taps(ga, gb, sc.Wp, sc.Hp);
                dst_off = pl_off + (L.wv - 1) * TM * 4; dst_w = pl_w + (L.wv - 1) * TM * 4;
                base = v * sc.Hp * sc.Wp + pl_base_texels;
                texel_bytes = PL_TEXEL_BYTES;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dst_off[p * 4 + k] = (int)((uint32_t)(base + t.off[k]) * (uint32_t)texel_bytes);   // byte offset mod 2^32
                dst_w[p * 4 + k] = t.w[k];
            }
            if constexpr (!DIRS) return;
            // view-direction encoding in this view's camera frame; wave q takes octave q
            const float dx = vdir_world[p * 4], dy = vdir_world[p * 4 + 1], dz = vdir_world[p * 4 + 2];
            const float dc[3] = {rot[0] * dx + rot[1] * dy + rot[2] * dz, rot[3] * dx + rot[4] * dy + rot[5] * dz,
                                 rot[6] * dx + rot[7] * dy + rot[8] * dz};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float sn, cs;
                enc_pair(dc[a], L.wv, sn, cs);
                put_dir(p, 3 + L.wv * 3 + a, sn);
                put_dir(p, 15 + L.wv * 3 + a, cs);
            }
            if (L.wv == 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) put_dir(p, a, dc[a]);
            }
            if (L.wv == 1) {
#pragma unroll
                for (int f = 27; f < 32; ++f) put_dir(p, f, 0.0f);
            }
        }
}

}  // namespace tp
}  // namespace neo

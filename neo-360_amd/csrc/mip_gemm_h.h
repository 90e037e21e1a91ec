// One 1024-output layer of the Mip-NeRF 360 NeRF MLP (models/mipnerf360/model.py:30-120: 8 x Linear(1024) + ReLU, the
// 504-d encoding re-concatenated in front of layer 5) as a layer-by-layer GEMM on the fp16 matrix cores with hi/lo-split fp32
// operands - the tiling DESIGN.md 4.4 sets against the fused evaluator's M = 32 rows per weight fragment:
//
//     Y[outputs][intervals] = act( W[outputs][K] * X[K][intervals] + b ),   K = K0 (+ K1: a second source, the skip concat)
//
// Activations live in global memory (L2 / Infinity Cache between two layers) in MFMA FRAGMENT ORDER, the same order the
// weights are packed in (split_tile.h:pack_h):
//     byte ((tile * KS + ks) * 2 + plane) * 1024 + lane * 16,   tile = 32 intervals (or 32 outputs), ks = 16-deep k-step,
//     plane 0 = hi, 1 = lo; lane (l31, half) holds features 16 ks + 8 half + 0..7 of interval (output) l31
// so a wave-wide 16-byte load IS an operand fragment: no layout change anywhere, LDS is a plain staging copy.
//
// Workgroup = 8 waves = 256 outputs x 256 intervals; wave (wo, wi) owns 64 outputs x 128 intervals = 2 x 4 accumulator
// tiles (128 VGPRs).  Weight fragments come straight from L2 (each is used by 4 interval tiles = 12 MFMAs; the two
// waves with the same wo read the same fragments), the intervals' fragments are staged through a double-buffered
// 2 x 32 KB LDS buffer (2 k-steps per stage, ONE barrier per stage = per 48 MFMAs of a wave; S must be even).  Per 16-deep k-step a
// workgroup moves 32 KB of weights + 16 KB of activations through the CU's vector-memory path for 8 x 24 MFMAs
// (>= 1536 matrix-pipe cycles for a SIMD's two waves): 31 B/clk of 64, against 128 KB per row-layer = 85 B/clk asked
// for by the fused kernel.
//
// Workgroup id -> tile: the 4 output slabs of one interval slab run on the SAME XCD at the same time (ids id, id + 8,
// id + 16, id + 24 land on one XCD), so an interval slab is pulled into that XCD's L2 once; every XCD streams all of W.
#pragma once
#include "split_tile.h"

namespace neo {

struct MipGemmArgs {
    const char* w;        // this layer's fragments: 32 output tiles x (ks0 + ks1) k-steps x 2 KB
    const float* bias;    // 1024
    const char* x0;       // source 0: [interval tile][ks0][2 KB]
    const char* x1;       // source 1 (k-steps ks0 .. ks0 + ks1 - 1 of the layer) or nullptr
    char* y;              // [interval tile][64][2 KB]
    int ks0, ks1;         // both even
    int n_it;             // interval tiles of the batch, a multiple of 64
    uint32_t* flags;
};

constexpr int MG_THREADS = 512;
constexpr int MG_STAGE_BYTES = 8 * 2 * 2048;      // 8 interval tiles x 2 k-steps x (hi | lo)
constexpr int MG_LDS_BYTES = 2 * MG_STAGE_BYTES;

// A_LDS (experiment, tools/gemm_h_bench.hip): the weight fragments are staged through LDS as well ([8 output tiles | 8 interval
// tiles] x 2 k-steps = 64 KB per stage, 128 KB double-buffered): half the weight bytes through the vector-memory path (16 KB
// instead of 32 KB per k-step and workgroup), 50 % more LDS traffic.
template <bool RELU, bool A_LDS = false>
__device__ __forceinline__ void mip_gemm_tile(const MipGemmArgs& a, const int id, char* __restrict__ mg_lds, const LaneCtx& L) {
    const int tid = threadIdx.x;
    const int slab_o = (id >> 3) & 3, slab_i = ((id >> 5) << 3) | (id & 7);
    const int KS = a.ks0 + a.ks1, S = KS >> 1, S0 = a.ks0 >> 1;
    const int wo = L.wv & 3, wi = L.wv >> 2;
    const int nt0 = slab_o * 8 + wo * 2;
    const int it0 = slab_i * 8;

    // ---- staging copy: pass p moves interval tiles 2p, 2p + 1 of the stage (4 KB each), 16 B per thread ----
    const int it_l = tid >> 8;
    const uint32_t rem = (uint32_t)(tid & 255) * 16u;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int STAGE = A_LDS ? 2 * MG_STAGE_BYTES : MG_STAGE_BYTES;      // A_LDS: [8 output tiles | 8 interval tiles]
    constexpr int B_AT = A_LDS ? MG_STAGE_BYTES : 0;
    constexpr int NP = A_LDS ? 8 : 4;
    u32x4 g[NP];
    auto load_stage = [&](int s) {
        const bool second = s >= S0;
        const char* xs = second ? a.x1 : a.x0;
        const int kss = second ? a.ks1 : a.ks0;
        const int sl = second ? s - S0 : s;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (A_LDS && p < 4) {
                const size_t off = ((size_t)(slab_o * 8 + 2 * p + it_l) * KS + 2 * s) * 2048 + rem;
                g[p] = *reinterpret_cast<const u32x4*>(a.w + off);
            } else {
                const size_t off = ((size_t)(it0 + 2 * (p - (A_LDS ? 4 : 0)) + it_l) * kss + 2 * sl) * 2048 + rem;
                g[p] = *reinterpret_cast<const u32x4*>(xs + off);
            }
        }
    };
    auto put_stage = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            *reinterpret_cast<u32x4*>(mg_lds + buf * STAGE + p * 8192 + tid * 16) = g[p];
    };
    // ---- weight fragments: four slots, k-step k + 2 is requested when k-step k starts.  The vector-memory counter retires
    // loads in order, so a wait for the weights of k + 1 also waits for every older request: the depth of the weight
    // prefetch IS the slack the (older) interval-stage requests get - 2 k-steps = 48 MFMAs per wave.
    h8 ah[4][2], al[4][2];
    const char* wl = a.w + (size_t)L.lane * 16;
    auto load_a = [&](int slot, int ks) {
        ks = ks < KS ? ks : KS - 1;                    // past the end: repeat the last k-step (no branch around a load)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const char* p = wl + ((size_t)(nt0 + nt) * KS + ks) * 2048;
            ah[slot][nt] = *reinterpret_cast<const h8*>(p);
            al[slot][nt] = *reinterpret_cast<const h8*>(p + 1024);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int it = 0; it < 4; ++it) bias_tile(acc[nt][it], a.bias, nt0 + nt, L);
    // interval fragments: one tile (hi, lo) per sub-step, read from the stage buffer one sub-step (6 MFMAs) ahead
    h8 bh[2], bl[2];
    const int boff = wi * (4 * 4096) + L.lane * 16;
    auto load_a_lds = [&](int slot, int buf, int u) {      // A_LDS: this wave's two output tiles of k-step u of the stage
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const char* p = mg_lds + buf * STAGE + (wo * 2 + nt) * 4096 + u * 2048 + L.lane * 16;
            ah[slot][nt] = *reinterpret_cast<const h8*>(p);
            al[slot][nt] = *reinterpret_cast<const h8*>(p + 1024);
        }
    };
    auto load_b = [&](int slot, int buf, int u, int it) {
        const char* bb = mg_lds + buf * STAGE + B_AT + boff + u * 2048 + it * 4096;
        bh[slot] = *reinterpret_cast<const h8*>(bb);
        bl[slot] = *reinterpret_cast<const h8*>(bb + 1024);
    };
    // Stage s = k-steps 2s, 2s + 1 = 8 sub-steps (k-step, interval tile) of 6 MFMAs.  The loop body is two stages so that
    // every register slot index is a compile-time constant.  Per stage:
    //   sub-step (u, it): [it == 0: weights of k-step 2s + u + 2 requested] | fragment of the NEXT sub-step read from LDS | 6 MFMAs
    //   after (1, 2):     stage s + 1: registers -> LDS | barrier | stage s + 2: global -> registers requested
    //   (1, 3) reads its "next" fragment from the buffer the barrier has just published.
    load_stage(0);
    if (!A_LDS) {
        load_a(0, 0);
        load_a(1, 1);
    }
    put_stage(0);
    __syncthreads();
    load_stage(S > 1 ? 1 : 0);
    load_b(0, 0, 0, 0);
    if (A_LDS) load_a_lds(0, 0, 0);
#pragma unroll 1
    for (int sp = 0; sp < S; sp += 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int s = sp + q;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int slot = 2 * q + u;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int sub = u * 4 + it;
                    if (!A_LDS && it == 0) load_a((slot + 2) & 3, 2 * s + u + 2);
                    if (A_LDS && sub == 3) load_a_lds(1, q, 1);                                // the stage's second k-step
                    if (sub < 7) load_b((sub + 1) & 1, q, (sub + 1) >> 2, (sub + 1) & 3);      // buffer s & 1 == q (sp is even)
                    else {
                        load_b(0, q ^ 1, 0, 0);                                                // first fragment of stage s + 1
                        if (A_LDS) load_a_lds(0, q ^ 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int as = A_LDS ? u : slot;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc[nt][it] = NEO_MFMA_H(al[as][nt], bh[sub & 1], acc[nt][it]);
                        acc[nt][it] = NEO_MFMA_H(ah[as][nt], bl[sub & 1], acc[nt][it]);
                        acc[nt][it] = NEO_MFMA_H(ah[as][nt], bh[sub & 1], acc[nt][it]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (sub == 6) {
                        put_stage(q ^ 1);
                        __syncthreads();
                        load_stage(s + 2 < S ? s + 2 : S - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
    // ---- epilogue: (ReLU) -> split -> the next layer's fragments.  D holds outputs 32 nt + 8 g + 4 half + e of interval
    // l31 in registers 4g..4g+3: k-step 2 nt + g/2 of the destination, fragment half g & 1, bytes 8 half .. 8 half + 7 of the
    // lane's 16-byte chunk - the 64 lanes of one store cover 512 consecutive bytes.
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        char* yt = a.y + (size_t)(it0 + wi * 4 + it) * (64 * 2048) + L.l31 * 16 + L.half * 8;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[nt][it][4 * gq + e];
                    v[e] = RELU ? relu1(x) : x;
                }
                range_see4(L, v);
                h4 vh, vl;
                split4(v, vh, vl);
                char* p = yt + (size_t)(2 * (nt0 + nt) + (gq >> 1)) * 2048 + (gq & 1) * 512;
                *reinterpret_cast<h4*>(p) = vh;
                *reinterpret_cast<h4*>(p + 1024) = vl;
            }
    }
}

template <bool RELU, bool A_LDS = false>
__global__ __launch_bounds__(MG_THREADS, 2) void k_mip_gemm_h(MipGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char mg_lds[];
    LaneCtx L;
    L.init();
    mip_gemm_tile<RELU, A_LDS>(a, (int)blockIdx.x, mg_lds, L);
    range_commit(L, a.flags);
}

// grid for a batch of n_it interval tiles (n_it % 64 == 0): 4 output slabs x n_it / 8 interval slabs
inline unsigned mip_gemm_grid(int n_it) { return (unsigned)(n_it / 8 * 4); }

}  // namespace neo

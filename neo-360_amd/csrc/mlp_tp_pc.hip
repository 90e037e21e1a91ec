// NeO-360 decoder point evaluator, split-fp16 arithmetic on the pre-projected latent (see mlp_tp_hp.hip), scheduled as
// producer / consumer wave groups.  Opt-in alternative to k_tp_mlp_hp (neo_tp_set_preproject(ctx, 2)); same packed
// weights, same projected map, same results up to fp32 summation order.
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "tp_hp_layout.h"

namespace neo {

namespace {

using namespace hp;

// =====================================================================================================================
// k_tp_mlp_pc: the same evaluator as k_tp_mlp_hp with the work of a tile split between two GROUPS of waves of one
// 8-wave workgroup (one workgroup per CU, one wave of each group per SIMD):
//   producers (waves 0-3): tap descriptors, gathers, bilinear blends, hi/lo splits, positional encodings.  Each wave
//       owns 16 rows of the tile and depends on no other producer wave: no producer-side barriers at all.
//   consumers (waves 4-7): every MFMA, the layer epilogues, the view-mean tail.
// A view's streamed input travels as 7 (fg) / 8 (bg) MESSAGES of 16 KB through a ring of NEO_PC_NB LDS buffers:
// 4 chunks of the pre-projected latent (fp32 [64][64], added to the accumulators), 2 tri-plane stages and 1-2 pos_enc
// stages (hi/lo planes [64][64], multiplied on the matrix cores).  Hand-off by progress counters in LDS (one per wave,
// monotone, polled; LDS requests of a wave are performed in order, so data written before a counter is visible to
// whoever sees the counter): pflag[w] = messages wave w finished, cflag[w] = messages consumer w is done reading.
// Producers run up to NB messages ahead - across view boundaries - so gathers of view v+1 overlap L1..L3 of view v:
// in k_tp_mlp_hp a workgroup's gather / blend phases and its matrix phases are serial (producer-only 4.4 ms +
// consumer-only 5.7 ms = 10.1 ms vs 9.4 ms for the whole kernel: profiles/r02_tp_hp_experiments.log).
#ifndef NEO_PC_NB
#define NEO_PC_NB 5
#endif
#ifndef NEO_PC_ABLATE
#define NEO_PC_ABLATE 0       // timing experiments: 1 producers only signal, 2 consumers only wait / release, 4 no consumer barriers,
                              // 8 no L1..L3, 16 no streamed-stage MFMAs, 32 no latent-chunk adds
#endif
#ifndef NEO_PC_RING
#define NEO_PC_RING 4         // tap register sets of a producer wave; must divide 40
#endif
#ifndef NEO_PC_LD
#define NEO_PC_LD 4           // the same for the 128 x 128 layers (one N-tile per wave: 8 VGPRs per k-step)
#endif
#ifndef NEO_PC_TD
#define NEO_PC_TD 6           // and for the tail (bottleneck, view layers: 3-6 MFMAs per k-step)
#endif
#ifndef NEO_PC_CPRIO
#define NEO_PC_CPRIO 2
#endif
#ifndef NEO_PC_WD
#define NEO_PC_WD 2           // weight fragments are requested this many k-steps ahead in the consumer GEMMs
#endif
namespace pc {
constexpr int NB = NEO_PC_NB;
constexpr int OFF_ACT = 0;                                             // [64][128] hi/lo: 8192 words
constexpr int OFF_MSG = OFF_ACT + TM * 128;                            // NB x 4096 words
constexpr int OFF_DIR = OFF_MSG + NB * 4096;                           // [64][32] fp32 sums, later hi/lo planes
constexpr int OFF_SCR = OFF_DIR + 2 * TM * 32;                         // (two buffers: tile parity); then tp::Scratch arrays, same relative layout
constexpr int SCR_WORDS = tp::LDS_WORDS - tp::OFF_PE;                  // only pe_world / feat_world / vdir_world / dens_w of tp::Scratch are used
constexpr int OFF_FLAGS = OFF_SCR + SCR_WORDS;                         // pflag[4] cflag[4] cbar[4] pad[4]
constexpr int DESC_WORDS = 2304;                                       // one view's tap descriptors + camera-frame points
constexpr int OFF_DESC = OFF_FLAGS + 16;                               // two copies: the next view's are written while this one's are read
constexpr int OFF_BIAS = OFF_DESC + 2 * DESC_WORDS;                               // all biases (768) and head weights (328): read once per kernel
constexpr int OFF_HEADS = OFF_BIAS + 768;
constexpr int LDS_WORDS = OFF_HEADS + 336;

// The counters are read and written with explicit LDS instructions: a C++ volatile access would make the compiler
// wait for EVERY outstanding memory operation (s_waitcnt vmcnt(0)) and throw away the weight / tap prefetch.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_addr(const volatile void* p) { return (uint32_t)(uintptr_t)p; }   // generic -> LDS offset
__device__ __forceinline__ void spin_until(volatile const int* f4, int need) {
    const uint32_t a = lds_addr(f4);
    for (;;) {
        i32x4 f;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(a) : "memory");
        if (min(min(f[0], f[1]), min(f[2], f[3])) >= need) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void post(volatile int* slot, int value, int lane) {
    // this wave's LDS traffic so far is done (reads returned, writes performed) before the counter moves
    const uint32_t a = lds_addr(slot);
    if (lane == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" : : "v"(a), "v"(value) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
}  // namespace pc

template <int PE_C>
__global__ __launch_bounds__(512, 2) void k_tp_mlp_pc(TpMlpHDev m, const float* __restrict__ proj, TpScene sc, TpViews views,
                                                       const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ viewdirs, const float* __restrict__ tvals,
                                                       const float* __restrict__ far_arr, int R, int N, int chunk,
                                                       uint32_t* __restrict__ flags, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KSX = ks_x(PE_C);
    constexpr int NPE = PE_C == 3 ? 1 : 2;
    constexpr int MPV = 6 + NPE;                       // messages per view
    _Float16* abase = reinterpret_cast<_Float16*>(smem + pc::OFF_ACT);
    const HT act{abase, abase + TM * 128};
    auto mbuf = [&](int msg) { return smem + pc::OFF_MSG + (msg % pc::NB) * 4096; };
    auto xbuf = [&](int msg) { _Float16* h = reinterpret_cast<_Float16*>(mbuf(msg)); return HT{h, h + TM * 64}; };
    const tp::Scratch S = tp::carve(smem + (pc::OFF_SCR - tp::OFF_PE));      // (its descriptor arrays are not used here)
    float* dens_w = smem + (pc::OFF_SCR - tp::OFF_PE) + tp::OFF_DENSW;
    volatile int* pflag = reinterpret_cast<volatile int*>(smem + pc::OFF_FLAGS);
    volatile int* cflag = pflag + 4;
    volatile int* cbarf = pflag + 8;

    const int tid = threadIdx.x;
    const long P = (long)R * N;
    // persistent workgroups: gridDim.x = 8 j workgroups; XCD x = blockIdx % 8 owns the contiguous tile range
    // [x per, (x+1) per) and its workgroups walk it with stride gridDim / 8 (neighbouring tiles share texels in that L2)
    const long ntiles = (P + TM - 1) / TM;
    const long per = (ntiles + tp::XCDS - 1) / tp::XCDS;
    const long t_lo = (long)(blockIdx.x % tp::XCDS) * per + blockIdx.x / tp::XCDS;
    const long t_hi = min(ntiles, (long)(blockIdx.x % tp::XCDS + 1) * per);
    const long t_step = gridDim.x / tp::XCDS;
    const h8* wp = reinterpret_cast<const h8*>(m.wpack);

    if (tid < 128) dens_w[tid] = m.heads[HD_DW + tid];
    if (tid < 16) reinterpret_cast<int*>(smem + pc::OFF_FLAGS)[tid] = 0;
    for (int i = tid; i < 768; i += 512) smem[pc::OFF_BIAS + i] = m.bias[i];
    if (tid < HD_RB + 3) smem[pc::OFF_HEADS + tid] = m.heads[tid];
    const float* lbias = smem + pc::OFF_BIAS;
    const float* lheads = smem + pc::OFF_HEADS;
    __syncthreads();               // the only workgroup-wide barrier

    LaneCtx L;
    L.init();

    if (L.wv < 4) {
        // =============================== producer ===============================
        // One continuous gather pipeline over all views of all tiles of this workgroup: the descriptors of the NEXT view
        // (and the world-space set-up of the next tile) are computed while the current view's gathers are in flight, the
        // last PRING-1 issue slots of a view already request the first items of the next one, and the positional
        // encodings are computed between gathers and only written after the view's last tri-plane stage (messages are
        // produced in the order they are consumed).
        const int pw = L.wv, lane = L.lane;
        const int col4 = lane & 15, rl = lane >> 4;
        const uint32_t lane_b = 16u * col4;
        const int r16 = lane & 15, grp = lane >> 4;
        const int drow = 4 * pw + (r16 & 3) + 16 * (r16 >> 2);        // the row this lane describes / encodes
        constexpr int PRING = NEO_PC_RING;
        constexpr int NI = 40;                                         // items 0..23: tri-planes (stage, row group, plane); 24..39: latent chunks
        constexpr int MSG_G = 2 + NPE;                                 // message order of a view: W0, W1, E0, (E1), G0..G3
        static_assert(NI % PRING == 0, "the tap ring must wrap consistently from one view to the next");
        float* dbuf = smem + pc::OFF_DESC;                             // [2][loc_off 256 | loc_w 256 | pl_off 768 | pl_w 768 | cam 256]
        auto d_loc_off = [&](int par) { return reinterpret_cast<int*>(dbuf + par * pc::DESC_WORDS); };
        auto d_loc_w = [&](int par) { return dbuf + par * pc::DESC_WORDS + 256; };
        auto d_pl_off = [&](int par) { return reinterpret_cast<int*>(dbuf + par * pc::DESC_WORDS + 512); };
        auto d_pl_w = [&](int par) { return dbuf + par * pc::DESC_WORDS + 1280; };
        auto d_cam = [&](int par) { return dbuf + par * pc::DESC_WORDS + 2048; };

        // ---- descriptors of this wave's 16 rows for view v -> buffer par: lane = (row, map), map 0 = latent, 1..3 = planes ----
        auto describe = [&](int v, int par, float* dsum) __attribute__((always_inline)) {
            const float* rot = views.rot[v];
            const float* trn = views.trans[v];
            const float fx = S.feat_world[drow * 4], fy = S.feat_world[drow * 4 + 1], fz = S.feat_world[drow * 4 + 2];
            const float cx_ = (rot[0] * fx + rot[1] * fy + rot[2] * fz) + trn[0];
            const float cy_ = (rot[3] * fx + rot[4] * fy + rot[5] * fz) + trn[1];
            const float cz_ = (rot[6] * fx + rot[7] * fy + rot[8] * fz) + trn[2];
            tp::TapSet t;
            int* dst_off;
            float* dst_w;
            int base, texel_bytes;
            if (grp == 0) {
                const float den = cz_ + 1e-9f;
                const float u = (-cx_ / den) * sc.focal + sc.cx;
                const float w_ = (-cy_ / den) * (sc.fy_sign * sc.focal) + sc.cy;
                t = tp::bilinear_taps(u * sc.sx - 1.0f, w_ * sc.sy - 1.0f, sc.Wf, sc.Hf);
                dst_off = d_loc_off(par); dst_w = d_loc_w(par);
                base = v * sc.Hf * sc.Wf;
                texel_bytes = PROJ_TEXEL_BYTES;
                const float ex = S.pe_world[drow * 4], ey = S.pe_world[drow * 4 + 1], ez = S.pe_world[drow * 4 + 2];
                float* cam = d_cam(par);
                cam[drow * 4 + 0] = (rot[0] * ex + rot[1] * ey + rot[2] * ez) + trn[0];
                cam[drow * 4 + 1] = (rot[3] * ex + rot[4] * ey + rot[5] * ez) + trn[1];
                cam[drow * 4 + 2] = (rot[6] * ex + rot[7] * ey + rot[8] * ez) + trn[2];
                cam[drow * 4 + 3] = S.pe_world[drow * 4 + 3];
            } else {
                const float ga = grp == 3 ? cy_ : cx_;
                const float gb = grp == 2 ? cy_ : cz_;
                t = tp::bilinear_taps(ga, gb, sc.Wp, sc.Hp);
                dst_off = d_pl_off(par) + (grp - 1) * TM * 4; dst_w = d_pl_w(par) + (grp - 1) * TM * 4;
                base = v * sc.Hp * sc.Wp;
                texel_bytes = 128 * 4;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dst_off[drow * 4 + k] = (int)((uint32_t)(base + t.off[k]) * (uint32_t)texel_bytes);
                dst_w[drow * 4 + k] = t.w[k];
            }
            // view-direction encoding in this view's camera frame, octave grp; (row, feature) is owned by this lane in every
            // view; view 0 starts the sums of a tile (the buffer still holds the hi / lo planes of two tiles ago)
            const float dx = S.vdir_world[drow * 4], dy = S.vdir_world[drow * 4 + 1], dz = S.vdir_world[drow * 4 + 2];
            const float dc[3] = {rot[0] * dx + rot[1] * dy + rot[2] * dz, rot[3] * dx + rot[4] * dy + rot[5] * dz,
                                 rot[6] * dx + rot[7] * dy + rot[8] * dz};
            auto put_dir = [&](int f, float val) {
                float* d = dsum + drow * 32 + (f ^ (drow & 31));
                *d = v == 0 ? val : *d + val;
            };
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float sn, cs;
                enc_pair(dc[a], grp, sn, cs);
                put_dir(3 + grp * 3 + a, sn);
                put_dir(15 + grp * 3 + a, cs);
            }
            if (grp == 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) put_dir(a, dc[a]);
            }
            if (grp == 1 && v == 0) {
#pragma unroll
                for (int f = 27; f < 32; ++f) put_dir(f, 0.0f);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // descriptors are read back by other lanes of this wave
            __builtin_amdgcn_wave_barrier();
        };
        auto setup = [&](long tile) __attribute__((always_inline)) {
            // world-space quantities of this wave's rows (the four lanes of a row compute the same values; one writes)
            if (grp == 0) tp::point_setup_row<PE_C>(S, drow, tile * TM, P, N, R, chunk, rays_o, rays_d, viewdirs, tvals, far_arr, flags);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        };
        auto read_off = [&](auto ic, int par) __attribute__((always_inline)) -> int4 {
            constexpr int i = decltype(ic)::value;
            if constexpr (i >= 24) {
                constexpr int q = i % 4;
                return *reinterpret_cast<const int4*>(d_loc_off(par) + (4 * pw + rl + 16 * q) * 4);
            } else {
                constexpr int w = i, q = (w % 12) / 3, j = w % 3;
                return *reinterpret_cast<const int4*>(d_pl_off(par) + (j * TM + 4 * pw + rl + 16 * q) * 4);
            }
        };
        auto read_w = [&](auto ic, int par) __attribute__((always_inline)) -> f32x4 {
            constexpr int i = decltype(ic)::value;
            if constexpr (i >= 24) {
                constexpr int q = i % 4;
                return *reinterpret_cast<const f32x4*>(d_loc_w(par) + (4 * pw + rl + 16 * q) * 4);
            } else {
                constexpr int w = i, q = (w % 12) / 3, j = w % 3;
                return *reinterpret_cast<const f32x4*>(d_pl_w(par) + (j * TM + 4 * pw + rl + 16 * q) * 4);
            }
        };
        f32x4 taps[PRING][4];
        auto issue = [&](auto ic, const int4 off) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if constexpr ((NEO_PC_ABLATE & 64) != 0) {
                for (int k = 0; k < 4; ++k) taps[i % PRING][k] = f32x4{(float)off.x, 0.f, 0.f, 0.f};
            } else if constexpr (i >= 24) {
                constexpr int c = (i - 24) / 4;
                taps[i % PRING][0] = tp::load_tap(proj, (uint32_t)off.x + lane_b + 256u * c);
                taps[i % PRING][1] = tp::load_tap(proj, (uint32_t)off.y + lane_b + 256u * c);
                taps[i % PRING][2] = tp::load_tap(proj, (uint32_t)off.z + lane_b + 256u * c);
                taps[i % PRING][3] = tp::load_tap(proj, (uint32_t)off.w + lane_b + 256u * c);
            } else {
                constexpr int w = i, s2 = w / 12, j = w % 3;
                taps[i % PRING][0] = tp::load_tap(sc.plane[j], (uint32_t)off.x + lane_b + 256u * s2);
                taps[i % PRING][1] = tp::load_tap(sc.plane[j], (uint32_t)off.y + lane_b + 256u * s2);
                taps[i % PRING][2] = tp::load_tap(sc.plane[j], (uint32_t)off.z + lane_b + 256u * s2);
                taps[i % PRING][3] = tp::load_tap(sc.plane[j], (uint32_t)off.w + lane_b + 256u * s2);
            }
        };

        f32x4 wsum;
        int4 offq[2];
        f32x4 wq[2];
        int gv = 0;                                   // views produced so far (all tiles): parity = descriptor buffer, x MPV = message base
        if (t_lo < t_hi) {
            setup(t_lo);
            describe(0, 0, smem + pc::OFF_DIR);
            static_for<0, PRING - 1>([&](auto ic) { issue(ic, read_off(ic, 0)); });
            offq[(PRING - 1) & 1] = read_off(std::integral_constant<int, PRING - 1>(), 0);
            wq[0] = read_w(std::integral_constant<int, 0>(), 0);
        }
        int tile_it = 0;
#pragma unroll 1
        for (long tile = t_lo; tile < t_hi; tile += t_step, ++tile_it) {
#pragma unroll 1
        for (int v = 0; v < sc.nv; ++v, ++gv) {
            const int mbase = gv * MPV;
            const int par = gv & 1;
            const bool last_view = v + 1 == sc.nv;
            const bool have_next = !last_view || tile + t_step < t_hi;
            const float* cam = d_cam(par);
            // message k of this view may be written once message (mbase + k - NB) has been read by every consumer
            auto wait_free = [&](int k) __attribute__((always_inline)) {
                const int msg = mbase + k;
                if (msg >= pc::NB) pc::spin_until(cflag, msg - pc::NB + 1);
            };
            auto done = [&](int k) __attribute__((always_inline)) { pc::post(pflag + pw, mbase + k + 1, lane); };
            h8 pe_h[3], pe_l[3];
            auto write_pe = [&]() __attribute__((always_inline)) {
                wait_free(2);
                {
                    const HT buf = xbuf(mbase + 2);
                    if constexpr ((NEO_PC_ABLATE & 128) == 0) {
                        int o = chunk_off<64>(drow, grp);
                        *reinterpret_cast<h8*>(buf.hi + o) = pe_h[0];
                        *reinterpret_cast<h8*>(buf.lo + o) = pe_l[0];
                        o = chunk_off<64>(drow, grp + 4);
                        *reinterpret_cast<h8*>(buf.hi + o) = pe_h[1];
                        *reinterpret_cast<h8*>(buf.lo + o) = pe_l[1];
                    }
                }
                done(2);
                if constexpr (NPE == 2) {
                    wait_free(3);
                    if constexpr ((NEO_PC_ABLATE & 128) == 0) {
                        const HT buf = xbuf(mbase + 3);
                        const int o = chunk_off<64>(drow, grp);
                        *reinterpret_cast<h8*>(buf.hi + o) = pe_h[2];
                        *reinterpret_cast<h8*>(buf.lo + o) = pe_l[2];
                    }
                    done(3);
                }
            };
            auto finish = [&](auto ic, const f32x4 wgt) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i >= 24) {
                    constexpr int c = (i - 24) / 4, q = i % 4;
                    if constexpr (q == 0) wait_free(MSG_G + c);
                    const int row = 4 * pw + rl + 16 * q;
                    const f32x4 val = blend4(taps[i % PRING], wgt);
                    *reinterpret_cast<f32x4*>(mbuf(mbase + MSG_G + c) + row * 64 + ((col4 ^ (row & 15)) << 2)) = val;
                    if constexpr (q == 3) done(MSG_G + c);
                } else {
                    constexpr int w = i, s2 = w / 12, q = (w % 12) / 3, j = w % 3;
                    if constexpr (q == 0 && j == 0) wait_free(s2);
                    const int row = 4 * pw + rl + 16 * q;
                    const f32x4 val = blend4(taps[i % PRING], wgt);
                    if constexpr (j == 0) wsum = val; else wsum = wsum + val;
                    if constexpr (j == 2) {
                        range_see4(L, wsum);
                        h4 vh, vl;
                        split4(wsum, vh, vl);
                        const HT buf = xbuf(mbase + s2);
                        const int o = chunk_off<64>(row, col4 >> 1) + 4 * (col4 & 1);
                        *reinterpret_cast<h4*>(buf.hi + o) = vh;
                        *reinterpret_cast<h4*>(buf.lo + o) = vl;
                    }
                    if constexpr (q == 3 && j == 2) done(s2);
                    if constexpr (i == 23) write_pe();          // both tri-plane stages are out: the pos_enc stage(s) follow
                }
            };
            if constexpr ((NEO_PC_ABLATE & 1) != 0) {
                for (int k = 0; k < MPV; ++k) { wait_free(k); done(k); }
                continue;
            }
            // pos_enc of this wave's rows: chunks grp and grp + 4 of stage 0 (and chunk grp of stage 1, bg), kept in
            // registers until the tri-plane stages are out
            auto encode = [&](int slot, int ps, int ch) __attribute__((always_inline)) {
                const float xc[4] = {cam[drow * 4], cam[drow * 4 + 1], cam[drow * 4 + 2], cam[drow * 4 + 3]};
                if (slot == 0) { range_see(L, xc[0]); range_see(L, xc[1]); range_see(L, xc[2]); }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 h, l;
                    split(pe_feature<PE_C>(xc, ps * 64 + ch * 8 + e), h, l);
                    pe_h[slot][e] = h;
                    pe_l[slot][e] = l;
                }
            };
            static_for<0, NI>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i == 24) {
                    if (last_view && have_next) setup(tile + t_step);
                }
                if constexpr (i == 28) {
                    if (have_next) describe(last_view ? 0 : v + 1, par ^ 1, smem + pc::OFF_DIR + ((tile_it + (last_view ? 1 : 0)) & 1) * (TM * 32));
                }
                if constexpr ((NEO_PC_ABLATE & 128) == 0) {
                    if constexpr (i == 4) encode(0, 0, grp);
                    if constexpr (i == 10) encode(1, 0, grp + 4);
                    if constexpr (i == 16 && NPE == 2) encode(2, 1, grp);
                }
                // descriptor reads one item ahead of their use; items >= NI are the next view's (other buffer)
                constexpr int jo = i + PRING, jw = i + 1, ji = i + PRING - 1;
                if constexpr (jo < NI) offq[jo & 1] = read_off(std::integral_constant<int, jo>(), par);
                else if (have_next) offq[jo & 1] = read_off(std::integral_constant<int, jo - NI>(), par ^ 1);
                if constexpr (jw < NI) wq[jw & 1] = read_w(std::integral_constant<int, jw>(), par);
                else if (have_next) wq[jw & 1] = read_w(std::integral_constant<int, 0>(), par ^ 1);
                if constexpr (ji < NI) issue(std::integral_constant<int, ji>(), offq[ji & 1]);
                else if (have_next) issue(std::integral_constant<int, ji - NI>(), offq[ji & 1]);
                finish(ic, wq[i & 1]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        }
        range_commit(L, m.flags);
        return;
    }

    // =============================== consumer ===============================
    const int cw = L.wv - 4;
    L.wv = cw;
#if NEO_PC_CPRIO
    __builtin_amdgcn_s_setprio(NEO_PC_CPRIO);                      // the matrix stream issues ahead of the producers' VALU work
#endif
    const int ctid = tid - 256;                                    // 0..255 within the consumer group
    int cepoch = 0;
    auto cbar = [&]() __attribute__((always_inline)) {             // barrier among the four consumer waves
        if constexpr ((NEO_PC_ABLATE & 4) != 0) return;
        ++cepoch;
        pc::post(cbarf + cw, cepoch, L.lane);
        pc::spin_until(cbarf, cepoch);
    };
    auto wait_full = [&](int msg) __attribute__((always_inline)) { pc::spin_until(pflag, msg + 1); };
    auto release = [&](int msg) __attribute__((always_inline)) { pc::post(cflag + cw, msg + 1, L.lane); };

    const int vnt = cw & 1, vmt = cw >> 1;
    const char* wb = reinterpret_cast<const char*>(wp);
    constexpr int WD = NEO_PC_WD, WS = WD + 1;                     // weight ring: WS slots
    int tile_it = 0;
#pragma unroll 1
    for (long tile = t_lo; tile < t_hi; tile += t_step, ++tile_it) {
    const long tile0 = tile * TM;
    float* dsum = smem + pc::OFF_DIR + (tile_it & 1) * (TM * 32);
    _Float16* dbase = reinterpret_cast<_Float16*>(dsum);
    const HT dsm{dbase, dbase + TM * 32};
    f32x16 hsum[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = 0.f; hsum[1][r] = 0.f; }

#pragma unroll 1
    for (int v = 0; v < sc.nv; ++v) {
        const int mbase = (tile_it * sc.nv + v) * MPV;
        if constexpr ((NEO_PC_ABLATE & 2) != 0) {
            for (int k = 0; k < MPV; ++k) { wait_full(mbase + k); release(mbase + k); }
            continue;
        }
        asm volatile("" : "+v"(L.lane));                           // as in k_tp_mlp_hp: no hoisting of swizzled addresses
        L.half = L.lane >> 5;
        L.l31 = L.lane & 31;
        L.key = L.lane & 15;
        f32x16 accx[2][2];
        bias_tile(accx[0][0], lbias + B_0, cw, L);
        accx[0][1] = accx[0][0];
        bias_tile(accx[1][0], lbias + B_3, cw, L);
        accx[1][1] = accx[1][0];
        // ---- streamed stage weights: k-steps 0..KSX-1 of N-tiles cw (L0) and 4 + cw (L3 skip), WD steps ahead ----
        h8 wh[WS][2], wl[WS][2];
        uint32_t wx_off[2];
        wx_off[0] = (uint32_t)((hoff_x() + cw * KSX * 128) + L.lane) * 16u;
        wx_off[1] = (uint32_t)((hoff_x() + (4 + cw) * KSX * 128) + L.lane) * 16u;
        auto load_wx = [&](auto kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(kc)::value;
            if constexpr (ks < KSX) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    wh[ks % WS][nt] = *reinterpret_cast<const h8*>(wb + (wx_off[nt] + 2048u * ks));
                    wl[ks % WS][nt] = *reinterpret_cast<const h8*>(wb + (wx_off[nt] + 2048u * ks + 1024u));
                }
            }
        };
        static_for<0, WD>([&](auto kc) { load_wx(kc); });
        // ---- world + pos_enc stages on the matrix cores ----
        static_for<0, KSX>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            constexpr int msg_k = ks / 4, tks = ks % 4;
            if constexpr (tks == 0) wait_full(mbase + msg_k);
            load_wx(std::integral_constant<int, ks + WD>());
            const HT tile = xbuf(mbase + msg_k);
            h8 bh[2], bl[2];
            if constexpr ((NEO_PC_ABLATE & 16) == 0) {
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
                    accx[nt][mt] = NEO_MFMA_H(wl[ks % WS][nt], bh[mt], accx[nt][mt]);
                    accx[nt][mt] = NEO_MFMA_H(wh[ks % WS][nt], bl[mt], accx[nt][mt]);
                    accx[nt][mt] = NEO_MFMA_H(wh[ks % WS][nt], bh[mt], accx[nt][mt]);
                }
            }
            if constexpr (tks == 3 || ks == KSX - 1) release(mbase + msg_k);
            __builtin_amdgcn_sched_barrier(0);          // keep the weight ring WD k-steps deep
        });
        // ---- pre-projected latent chunks: add this wave's pieces ----
        static_for<0, 4>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            wait_full(mbase + 2 + NPE + c);
            const float* buf = mbuf(mbase + 2 + NPE + c);
            if constexpr ((NEO_PC_ABLATE & 32) == 0)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int row = mt * 32 + L.l31;
                    const int piece = cw * 4 + gg * 2 + L.half;
                    const f32x4 val = *reinterpret_cast<const f32x4*>(buf + row * 64 + ((piece ^ (row & 15)) << 2));
                    constexpr int g0 = 2 * (c & 1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) accx[c >> 1][mt][4 * (g0 + gg) + e] += val[e];
                }
            release(mbase + 2 + NPE + c);
        });
        // ---- L1, L2, L3: one weight stream of 24 k-steps (N-tile cw), requested LD k-steps ahead ACROSS the layer
        //      boundaries (weights do not wait for the barriers), B fragments of the next k-step read before the MFMAs ----
        constexpr int LD = NEO_PC_LD, LS = LD + 1;
        h8 lwh[LS], lwl[LS];
        const uint32_t lw_off = (uint32_t)(cw * 8 * 128 + L.lane) * 16u;          // + layer base + 2048 ks
        auto load_l = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < 24) {
                constexpr int layer = g / 8, ks = g % 8;
                constexpr uint32_t base = (uint32_t)(layer == 0 ? hoff_1(PE_C) : layer == 1 ? hoff_2(PE_C) : hoff_3a(PE_C)) * 16u;
                lwh[g % LS] = *reinterpret_cast<const h8*>(wb + (base + lw_off + 2048u * ks));
                lwl[g % LS] = *reinterpret_cast<const h8*>(wb + (base + lw_off + 2048u * ks + 1024u));
            }
        };
        static_for<0, LD>([&](auto gc) { load_l(gc); });
        f32x16 acc[1][2];
        store_tile_h<true>(accx[0][0], act, cw, 0, L);
        store_tile_h<true>(accx[0][1], act, cw, 1, L);
        cbar();
        h8 bh[2][2], bl[2][2];
        auto read_b = [&](int slot, int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int o = chunk_off<128>(mt * 32 + L.l31, (ks << 1) + L.half);
                bh[slot][mt] = *reinterpret_cast<const h8*>(act.hi + o);
                bl[slot][mt] = *reinterpret_cast<const h8*>(act.lo + o);
            }
        };
        if constexpr ((NEO_PC_ABLATE & 8) == 0)
        static_for<0, 24>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int layer = g / 8, ks = g % 8;
            if constexpr (ks == 0) {
                if constexpr (layer < 2) {
                    bias_tile(acc[0][0], lbias + (layer == 0 ? B_1 : B_2), cw, L);
                    acc[0][1] = acc[0][0];
                } else {
                    acc[0][0] = accx[1][0];
                    acc[0][1] = accx[1][1];
                }
                read_b(0, 0);
            }
            load_l(std::integral_constant<int, g + LD>());
            if constexpr (ks < 7) read_b((ks + 1) & 1, ks + 1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[0][mt] = NEO_MFMA_H(lwl[g % LS], bh[ks & 1][mt], acc[0][mt]);
                acc[0][mt] = NEO_MFMA_H(lwh[g % LS], bl[ks & 1][mt], acc[0][mt]);
                acc[0][mt] = NEO_MFMA_H(lwh[g % LS], bh[ks & 1][mt], acc[0][mt]);
            }
            if constexpr (ks == 7) {
                if constexpr (layer < 2) {
                    cbar();
                    store_tile_h<true>(acc[0][0], act, cw, 0, L);
                    store_tile_h<true>(acc[0][1], act, cw, 1, L);
                    cbar();
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        hsum[0][r] += fmaxf(acc[0][0][r], 0.0f);
                        hsum[1][r] += fmaxf(acc[0][1][r], 0.0f);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the weight ring LD k-steps deep
        });
        cbar();           // every consumer is done reading this view's activations
    }

    // ---- view mean of the trunk -> density head (consumer group only from here on) ----
    wait_full((tile_it + 1) * sc.nv * MPV - 1);          // the producers' last view of this tile: its direction-encoding sums are complete
    const float nvf = (float)sc.nv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { hsum[0][r] = hsum[0][r] / nvf; hsum[1][r] = hsum[1][r] / nvf; }
    store_tile_h<false>(hsum[0], act, cw, 0, L);
    store_tile_h<false>(hsum[1], act, cw, 1, L);
    float dmean[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dmean[j] = dsum[(ctid >> 2) * 32 + ((((ctid & 3) << 3) + j) ^ ((ctid >> 2) & 31))] / nvf;
    cbar();
    {
        h8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 h, l;
            split(dmean[j], h, l);
            vh[j] = h;
            vl[j] = l;
        }
        const int o = chunk_off<32>(ctid >> 2, ctid & 3);
        *reinterpret_cast<h8*>(dsm.hi + o) = vh;
        *reinterpret_cast<h8*>(dsm.lo + o) = vl;
    }
    float raw_sigma;
    {
        float sg = density_partial(act, dens_w, L);
        sg += __shfl_xor(sg, 1, 64);
        sg += __shfl_xor(sg, 2, 64);
        raw_sigma = sg + lheads[HD_DB];
    }
    // ---- tail GEMMs as one weight stream of 22 k-steps, TD ahead across the stage boundaries:
    //      bottleneck of the view mean (N-tile cw, 8 k-steps, both M-tiles), view layer 0 on [mean bottleneck | mean dir enc]
    //      (N-tile vnt, M-tile vmt, 8 + 2 k-steps), 64 x 64 (4 k-steps) ----
    {
        constexpr int TD = NEO_PC_TD, TS = TD + 1;
        h8 twh[TS], twl[TS];
        auto load_t = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < 22) {
                constexpr int stage = g < 8 ? 0 : g < 18 ? 1 : 2;
                constexpr int ks = stage == 0 ? g : stage == 1 ? g - 8 : g - 18;
                constexpr int KS = stage == 0 ? 8 : stage == 1 ? 10 : 4;
                constexpr uint32_t base = (uint32_t)(stage == 0 ? hoff_b(PE_C) : stage == 1 ? hoff_v0(PE_C) : hoff_v1(PE_C)) * 16u;
                const int nt = stage == 0 ? cw : vnt;
                const uint32_t off = base + (uint32_t)((nt * KS + ks) * 128 + L.lane) * 16u;
                twh[g % TS] = *reinterpret_cast<const h8*>(wb + off);
                twl[g % TS] = *reinterpret_cast<const h8*>(wb + off + 1024u);
            }
        };
        static_for<0, TD>([&](auto gc) { load_t(gc); });
        f32x16 acc2[2], y;
        static_for<0, 22>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            load_t(std::integral_constant<int, g + TD>());
            if constexpr (g < 8) {
                if constexpr (g == 0) {
                    bias_tile(acc2[0], lbias + B_B, cw, L);
                    acc2[1] = acc2[0];
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int o = chunk_off<128>(mt * 32 + L.l31, (g << 1) + L.half);
                    const h8 bh = *reinterpret_cast<const h8*>(act.hi + o);
                    const h8 bl = *reinterpret_cast<const h8*>(act.lo + o);
                    acc2[mt] = NEO_MFMA_H(twl[g % TS], bh, acc2[mt]);
                    acc2[mt] = NEO_MFMA_H(twh[g % TS], bl, acc2[mt]);
                    acc2[mt] = NEO_MFMA_H(twh[g % TS], bh, acc2[mt]);
                }
                if constexpr (g == 7) {
                    cbar();
                    store_tile_h<false>(acc2[0], act, cw, 0, L);
                    store_tile_h<false>(acc2[1], act, cw, 1, L);
                    cbar();
                }
            } else {
                constexpr bool v0 = g < 18;
                constexpr int ks = v0 ? g - 8 : g - 18;
                if constexpr (ks == 0) bias_tile(y, lbias + (v0 ? B_V0 : B_V1), vnt, L);
                h8 bh, bl;
                if constexpr (v0 && ks >= 8) {
                    const int o = chunk_off<32>(vmt * 32 + L.l31, ((ks - 8) << 1) + L.half);
                    bh = *reinterpret_cast<const h8*>(dsm.hi + o);
                    bl = *reinterpret_cast<const h8*>(dsm.lo + o);
                } else {
                    const int o = chunk_off<128>(vmt * 32 + L.l31, (ks << 1) + L.half);
                    bh = *reinterpret_cast<const h8*>(act.hi + o);
                    bl = *reinterpret_cast<const h8*>(act.lo + o);
                }
                y = NEO_MFMA_H(twl[g % TS], bh, y);
                y = NEO_MFMA_H(twh[g % TS], bl, y);
                y = NEO_MFMA_H(twh[g % TS], bh, y);
                if constexpr (g == 17) {
                    cbar();
                    store_tile_h<true>(y, act, vnt, vmt, L);
                    cbar();
                }
                if constexpr (g == 21) {
                    cbar();
                    store_tile_h<true>(y, act, vnt, vmt, L);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    cbar();
    {
        const int pt = cw * 16 + (L.lane >> 2), part = L.lane & 3;
        const float* wr = lheads + HD_RW;
        float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int chunk_i = part * 2 + ((c + part) & 1);
            const int o = chunk_off<128>(pt, chunk_i);
            const h8 vh = *reinterpret_cast<const h8*>(act.hi + o);
            const h8 vl = *reinterpret_cast<const h8*>(act.lo + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float h = (float)vh[e] + (float)vl[e];
                r += h * wr[chunk_i * 8 + e];
                g += h * wr[64 + chunk_i * 8 + e];
                b += h * wr[128 + chunk_i * 8 + e];
            }
        }
        r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64);
        g += __shfl_xor(g, 1, 64); g += __shfl_xor(g, 2, 64);
        b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64);
        const long gi = tile0 + pt;
        if (part == 0 && gi < P) {
            out[gi] = make_float4(colour_act(r + lheads[HD_RB]), colour_act(g + lheads[HD_RB + 1]),
                                  colour_act(b + lheads[HD_RB + 2]), density_act(raw_sigma));
        }
    }
    cbar();                          // the activation tile and the direction planes are free for the next tile
    }
    range_commit(L, m.flags);
}

}  // namespace

void launch_tp_mlp_pc(int input_ch, const TpMlpHDev& m, const float* proj, const TpScene& sc, const TpViews& views,
                      const float* rays_o, const float* rays_d, const float* viewdirs, const float* tvals,
                      const float* far, int R, int N, int chunk, uint32_t* flags, float* out, hipStream_t s) {
    const long P = (long)R * N;
    if (P <= 0) return;
    const size_t lds_pc = pc::LDS_WORDS * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_mlp_pc<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pc);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tp_mlp_pc<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pc);
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu < 8) n_cu = 256;
        n_cu = (n_cu / 8) * 8;
    }
    const long ntiles = (P + TM - 1) / TM;
    const long grid_pc = std::min<long>(n_cu, ((ntiles + 7) / 8) * 8);     // one persistent workgroup per CU
    if (input_ch == 3)
        hipLaunchKernelGGL(k_tp_mlp_pc<3>, dim3((unsigned)grid_pc), dim3(512), lds_pc, s, m, proj, sc, views, rays_o, rays_d,
                           viewdirs, tvals, far, R, N, chunk, flags, reinterpret_cast<float4*>(out));
    else
        hipLaunchKernelGGL(k_tp_mlp_pc<4>, dim3((unsigned)grid_pc), dim3(512), lds_pc, s, m, proj, sc, views, rays_o, rays_d,
                           viewdirs, tvals, far, R, N, chunk, flags, reinterpret_cast<float4*>(out));
}

}  // namespace neo

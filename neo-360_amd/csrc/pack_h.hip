// fp32 -> split-fp16 MFMA fragment re-packing shared by the split point evaluators.
#include "split_tile.h"

namespace neo {

namespace {

// fp16 hi/lo fragments of rows [0, rows) of src into N-tiles [nt0, ...) of a stage with KS 16-deep k-steps;
// packed k -> source column through up to three segments, zero elsewhere.
__global__ void k_pack_block_h(const float* __restrict__ src, int ld, int rows, int KS, int nt0, PackSegs sg,
                               _Float16* __restrict__ dst) {
    const int total = (rows / 32) * KS * 512;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, blk = idx >> 9;
        const int ks = blk % KS, ntl = blk / KS;
        const int n = ntl * 32 + (lane & 31);
        const int k = ks * 16 + 8 * (lane >> 5) + e;
        float w = 0.0f;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (k >= sg.k0[q] && k < sg.k0[q] + sg.len[q]) w = src[(long)n * ld + sg.col[q] + (k - sg.k0[q])];
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        const long base = ((long)((nt0 + ntl) * KS + ks) * 2) * 512 + lane * 8 + e;
        dst[base] = hi;
        dst[base + 512] = lo;
    }
}

__global__ void k_pack_block_h_perm(const float* __restrict__ src, int ld, int rows, int KS, int nt0, PackPerm pm,
                                    _Float16* __restrict__ dst) {
    const int total = (rows / 32) * KS * 512;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, blk = idx >> 9;
        const int ks = blk % KS, ntl = blk / KS;
        const int n = ntl * 32 + (lane & 31);
        const int k = ks * 16 + 8 * (lane >> 5) + e;
        const int col = pm.col[k];
        const float w = col >= 0 ? src[(long)n * ld + col] : 0.0f;
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        const long base = ((long)((nt0 + ntl) * KS + ks) * 2) * 512 + lane * 8 + e;
        dst[base] = hi;
        dst[base + 512] = lo;
    }
}

// any fp16 inf / NaN (exponent all ones) among n halves -> flags |= FLAG_SPLIT_RANGE | FLAG_SPLIT_STATIC (bit 2: the operand that
// left the range is a STATIC one - packed weights or an uploaded feature map -, so every frame on this (weights, scene) pair will
// trip again: render.render_rays_test latches the module to the exact kernels until either changes)
__global__ void k_half_range_check(const uint16_t* __restrict__ h, size_t n, uint32_t* __restrict__ flags) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        bad |= (h[i] & 0x7c00u) == 0x7c00u;
    if (bad) atomicOr(flags, FLAG_SPLIT_RANGE | FLAG_SPLIT_STATIC);
}

__global__ void k_f32_range_check(const float* __restrict__ x, size_t n4, size_t n, float limit, uint32_t* __restrict__ flags) {
    bool bad = false;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = x4[i];
        bad = bad || !(fabsf(v[0]) < limit) || !(fabsf(v[1]) < limit) || !(fabsf(v[2]) < limit) || !(fabsf(v[3]) < limit);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= !(fabsf(x[n4 * 4 + threadIdx.x]) < limit);
    if (bad) atomicOr(flags, FLAG_SPLIT_RANGE | FLAG_SPLIT_STATIC);
}

// Bottleneck folded into the layer that consumes it (tp_hp_layout.h NEO_TP_FOLDB; the same algebra for the vanilla NeRFMLP,
// vanilla_nerf/model.py:113-121: bottleneck_layer has no activation and feeds views_linear.0 only):
//   wfold (n_out, nin + extra) = [W_v[:, :nb] . W_b (nb, nin) | W_v[:, nb:nb + extra]],   bfold (n_out) = b_v + W_v[:, :nb] . b_b
// fp64 accumulation, one thread per output element.
__global__ void k_fold_bottleneck(const float* __restrict__ wv, const float* __restrict__ wb, const float* __restrict__ bb,
                                  const float* __restrict__ bv, int n_out, int nb, int nin, int extra,
                                  float* __restrict__ wfold, float* __restrict__ bfold) {
    const int ldv = nb + extra, ldf = nin + extra;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)n_out * ldf) {
        const int o = (int)(idx / ldf), i = (int)(idx % ldf);
        if (i < nin) {
            double acc = 0.0;
            for (int j = 0; j < nb; ++j) acc += (double)wv[(long)o * ldv + j] * (double)wb[(long)j * nin + i];
            wfold[idx] = (float)acc;
        } else {
            wfold[idx] = wv[(long)o * ldv + nb + (i - nin)];
        }
    } else if (idx < (long)n_out * ldf + n_out) {
        const int o = (int)(idx - (long)n_out * ldf);
        double acc = (double)bv[o];
        for (int j = 0; j < nb; ++j) acc += (double)wv[(long)o * ldv + j] * (double)bb[j];
        bfold[o] = (float)acc;
    }
}

}  // namespace

void launch_fold_bottleneck(const float* wv, const float* wb, const float* bb, const float* bv, int n_out, int nb, int nin,
                            int extra, float* wfold, float* bfold, hipStream_t s) {
    const long total = (long)n_out * (nin + extra) + n_out;
    hipLaunchKernelGGL(k_fold_bottleneck, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wv, wb, bb, bv, n_out, nb, nin,
                       extra, wfold, bfold);
}

void launch_half_range_check(const void* halves, size_t n, uint32_t* flags, hipStream_t s) {
    if (n == 0) return;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_half_range_check, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, s,
                       reinterpret_cast<const uint16_t*>(halves), n, flags);
}

void launch_f32_range_check(const float* x, size_t n, float limit, uint32_t* flags, hipStream_t s) {
    if (n == 0) return;
    const size_t n4 = n / 4, blocks = (n4 + 255) / 256 + 1;
    hipLaunchKernelGGL(k_f32_range_check, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, x, n4, n, limit, flags);
}

void pack_h_perm(const float* src, int ld, int rows, int KS, int nt0, const PackPerm& perm, _Float16* dst, hipStream_t s) {
    const int total = (rows / 32) * KS * 512;
    hipLaunchKernelGGL(k_pack_block_h_perm, dim3((total + 255) / 256), dim3(256), 0, s, src, ld, rows, KS, nt0, perm, dst);
}

void pack_h(const float* src, int ld, int rows, int KS, int nt0, PackSegs sg, _Float16* dst, hipStream_t s) {
    const int total = (rows / 32) * KS * 512;
    hipLaunchKernelGGL(k_pack_block_h, dim3((total + 255) / 256), dim3(256), 0, s, src, ld, rows, KS, nt0, sg, dst);
}


}  // namespace neo

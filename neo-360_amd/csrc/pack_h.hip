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

}  // namespace

void pack_h(const float* src, int ld, int rows, int KS, int nt0, PackSegs sg, _Float16* dst, hipStream_t s) {
    const int total = (rows / 32) * KS * 512;
    hipLaunchKernelGGL(k_pack_block_h, dim3((total + 255) / 256), dim3(256), 0, s, src, ld, rows, KS, nt0, sg, dst);
}


}  // namespace neo

// Micro-benchmark + indexing check of csrc/mip_gemm_h.h: the 8 x 1024 trunk of the Mip-NeRF 360 NeRF MLP as a
// layer-by-layer split-fp16 GEMM over batches of intervals whose activations stay in L2 / Infinity Cache.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
//         -I neo-360_amd/csrc tools/gemm_h_bench.hip -o tools/build/gemm_h_bench
//   tools/build/gemm_h_bench [intervals per batch = 16384] [batches = 40] [scale = 1 (0: all-zero operands)]
// Prints: max relative error of layer 0 / layer 1 / the skip layer against a float64 evaluation of the same (hi + lo)
// operands at sampled (interval, output) pairs, per-layer times of one batch, and the sustained algorithmic TFLOP/s of
// the 8-layer chain (2 x MACs; the matrix pipes execute 3 fp16 products per MAC).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "mip_gemm_h.h"

using namespace neo;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static void split_host(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)(x - (float)h);
}

// tiles x KS x 2 KB fragment buffer from a row-major (rows = 32 * tiles, K = 16 * KS) matrix
static std::vector<_Float16> to_fragments(const std::vector<float>& m, int tiles, int KS) {
    std::vector<_Float16> f((size_t)tiles * KS * 1024);
    const int K = KS * 16;
    for (int t = 0; t < tiles; ++t)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int row = t * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + j;
                    _Float16 h, l;
                    split_host(m[(size_t)row * K + k], h, l);
                    const size_t base = ((size_t)(t * KS + ks) * 2) * 512 + lane * 8 + j;
                    f[base] = h;
                    f[base + 512] = l;
                }
    return f;
}
static double frag_value(const std::vector<_Float16>& f, int KS, int row, int k) {
    const int t = row >> 5, l31 = row & 31, ks = k >> 4, half = (k >> 3) & 1, j = k & 7;
    const size_t base = ((size_t)(t * KS + ks) * 2) * 512 + (half * 32 + l31) * 8 + j;
    return (double)(float)f[base] + (double)(float)f[base + 512];
}

int main(int argc, char** argv) {
    const int Mb = argc > 1 ? atoi(argv[1]) : 16384;
    const int batches = argc > 2 ? atoi(argv[2]) : 40;
    const float scale = argc > 3 ? (float)atof(argv[3]) : 1.0f;
    if (Mb % 2048) { printf("intervals per batch must be a multiple of 2048\n"); return 1; }
    const int n_it = Mb / 32;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(-1.f, 1.f);

    // weights: layer 0 K = 512, 1..4 / 6 / 7 K = 1024, 5 K = 1536 ([h | x0]); He-like scale so activations stay O(1)
    const int KSL[8] = {32, 64, 64, 64, 64, 96, 64, 64};
    std::vector<std::vector<float>> W(8);
    std::vector<std::vector<_Float16>> Wf(8);
    std::vector<float> bias(8 * 1024);
    for (auto& b : bias) b = 0.05f * nd(rng) * scale;
    for (int l = 0; l < 8; ++l) {
        const int K = KSL[l] * 16;
        W[l].resize((size_t)1024 * K);
        const float sd = sqrtf(2.0f / K) * scale;
        for (auto& v : W[l]) v = sd * nd(rng);
        Wf[l] = to_fragments(W[l], 32, KSL[l]);
    }
    std::vector<float> X((size_t)Mb * 512);
    for (auto& v : X) v = ud(rng) * scale;
    std::vector<_Float16> Xf = to_fragments(X, n_it, 32);

    char *dW[8], *dX, *dA, *dB;
    float* dBias;
    uint32_t* dFlags;
    for (int l = 0; l < 8; ++l) {
        CK(hipMalloc(&dW[l], Wf[l].size() * 2));
        CK(hipMemcpy(dW[l], Wf[l].data(), Wf[l].size() * 2, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&dX, Xf.size() * 2));
    CK(hipMemcpy(dX, Xf.data(), Xf.size() * 2, hipMemcpyHostToDevice));
    const size_t act_bytes = (size_t)n_it * 64 * 2048;
    CK(hipMalloc(&dA, act_bytes));
    CK(hipMalloc(&dB, act_bytes));
    CK(hipMalloc(&dBias, bias.size() * 4));
    CK(hipMemcpy(dBias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dFlags, 4));
    CK(hipMemset(dFlags, 0, 4));
#ifndef GEMM_A_LDS
#define GEMM_A_LDS false
#endif
    constexpr int LDS_BYTES = GEMM_A_LDS ? 2 * MG_LDS_BYTES : MG_LDS_BYTES;
    auto kern = k_mip_gemm_h<true, GEMM_A_LDS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));

    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto layer = [&](int l, const char* in, char* out) {
        MipGemmArgs a{};
        a.w = dW[l];
        a.bias = dBias + l * 1024;
        a.x0 = l == 0 ? dX : in;
        a.ks0 = l == 0 ? 32 : 64;
        a.x1 = l == 5 ? dX : nullptr;
        a.ks1 = l == 5 ? 32 : 0;
        a.y = out;
        a.n_it = n_it;
        a.flags = dFlags;
        hipLaunchKernelGGL(kern, dim3(mip_gemm_grid(n_it)), dim3(MG_THREADS), LDS_BYTES, st, a);
    };
    auto chain = [&]() {
        char* bufs[2] = {dA, dB};
        for (int l = 0; l < 8; ++l) layer(l, bufs[(l + 1) & 1], bufs[l & 1]);      // layer l writes bufs[l & 1], reads the other
    };

    // ---- indexing check: layers 0, 1 and 5 on sampled entries, float64 on the same split operands ----
    {
        std::vector<_Float16> Y0(act_bytes / 2), Y1(act_bytes / 2), Y5(act_bytes / 2);
        // Y0 -> host, Y1 -> host, then the skip layer from Y1 (+ x0) into the first buffer
        layer(0, nullptr, dA);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(Y0.data(), dA, act_bytes, hipMemcpyDeviceToHost));
        layer(1, dA, dB);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(Y1.data(), dB, act_bytes, hipMemcpyDeviceToHost));
        layer(5, dB, dA);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(Y5.data(), dA, act_bytes, hipMemcpyDeviceToHost));
        double e0 = 0, e1 = 0, e5 = 0;
        std::mt19937 pick(3);
        for (int t = 0; t < 400; ++t) {
            const int i = pick() % Mb, o = pick() % 1024;
            double s0 = bias[o], s1 = bias[1024 + o], s5 = bias[5 * 1024 + o];
            for (int k = 0; k < 512; ++k) s0 += frag_value(Wf[0], 32, o, k) * frag_value(Xf, 32, i, k);
            for (int k = 0; k < 1024; ++k) s1 += frag_value(Wf[1], 64, o, k) * frag_value(Y0, 64, i, k);
            for (int k = 0; k < 1024; ++k) s5 += frag_value(Wf[5], 96, o, k) * frag_value(Y1, 64, i, k);
            for (int k = 0; k < 512; ++k) s5 += frag_value(Wf[5], 96, o, 1024 + k) * frag_value(Xf, 32, i, k);
            s0 = s0 > 0 ? s0 : 0; s1 = s1 > 0 ? s1 : 0; s5 = s5 > 0 ? s5 : 0;
            e0 = fmax(e0, fabs(frag_value(Y0, 64, i, o) - s0) / (1e-3 + fabs(s0)));
            e1 = fmax(e1, fabs(frag_value(Y1, 64, i, o) - s1) / (1e-3 + fabs(s1)));
            e5 = fmax(e5, fabs(frag_value(Y5, 64, i, o) - s5) / (1e-3 + fabs(s5)));
        }
        printf("check (400 samples each): layer0 %.2e  layer1 %.2e  skip layer %.2e  (relative, vs float64 on the same operands)\n", e0, e1, e5);
    }

    // ---- the same trunk as ONE launch (k_mip_chain_h, slab-local barriers): bitwise the 8-launch result ----
    unsigned* dArrive;
    const size_t arrive_bytes = (size_t)(n_it / 8) * 8 * sizeof(unsigned);
    CK(hipMalloc(&dArrive, arrive_bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mip_chain_h<true>), hipFuncAttributeMaxDynamicSharedMemorySize, MG_LDS_BYTES));
    auto chain1 = [&]() {
        MipChainArgs c{};
        for (int l = 0; l < 8; ++l) { c.w[l] = dW[l]; c.ks0[l] = l == 0 ? 32 : 64; c.ks1[l] = l == 5 ? 32 : 0; }
        c.bias = dBias; c.x0 = dX; c.y[0] = dA; c.y[1] = dB; c.n_it = n_it; c.layers = 8; c.flags = dFlags; c.arrive = dArrive;
        CK(hipMemsetAsync(dArrive, 0, arrive_bytes, st));
        hipLaunchKernelGGL(k_mip_chain_h<true>, dim3(mip_gemm_grid(n_it)), dim3(MG_THREADS), MG_LDS_BYTES, st, c);
    };
    {
        std::vector<char> ref(act_bytes), got(act_bytes);
        chain();
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(ref.data(), dB, act_bytes, hipMemcpyDeviceToHost));       // layer 7 writes bufs[1]
        CK(hipMemset(dB, 0xff, act_bytes));
        chain1();
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(got.data(), dB, act_bytes, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < act_bytes; ++i) bad += ref[i] != got[i];
        printf("one-launch chain vs eight launches: %zu differing bytes of %zu\n", bad, act_bytes);
    }

    hipEvent_t ev[10];
    for (auto& e : ev) CK(hipEventCreate(&e));
    chain();
    CK(hipStreamSynchronize(st));
    {   // per-layer times of one batch
        char* bufs[2] = {dA, dB};
        for (int l = 0; l < 8; ++l) {
            CK(hipEventRecord(ev[l], st));
            layer(l, bufs[(l + 1) & 1], bufs[l & 1]);
        }
        CK(hipEventRecord(ev[8], st));
        CK(hipStreamSynchronize(st));
        printf("per layer (us):");
        for (int l = 0; l < 8; ++l) {
            float ms;
            CK(hipEventElapsedTime(&ms, ev[l], ev[l + 1]));
            printf(" %.1f", ms * 1e3);
        }
        printf("\n");
    }
    double macs = 0;
    for (int l = 0; l < 8; ++l) macs += (double)KSL[l] * 16 * 1024;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(ev[0], st));
        for (int b = 0; b < batches; ++b) chain();
        CK(hipEventRecord(ev[1], st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
        const double tf = 2.0 * macs * Mb * batches / (ms * 1e-3) / 1e12;
        printf("chain: %d intervals x %d batches  %.3f ms per batch  %.1f algorithmic TFLOP/s (%.1f %% of 833; executed %.0f)  scale %g\n",
               Mb, batches, ms / batches, tf, tf / 833.0 * 100.0, tf * 3, scale);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(ev[0], st));
        for (int b = 0; b < batches; ++b) chain1();
        CK(hipEventRecord(ev[1], st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
        const double tf = 2.0 * macs * Mb * batches / (ms * 1e-3) / 1e12;
        printf("ONE LAUNCH (k_mip_chain_h): %d intervals x %d batches  %.3f ms per batch  %.1f algorithmic TFLOP/s (%.1f %% of 833; executed %.0f)  scale %g\n",
               Mb, batches, ms / batches, tf, tf / 833.0 * 100.0, tf * 3, scale);
    }
    uint32_t fl;
    CK(hipMemcpy(&fl, dFlags, 4, hipMemcpyDeviceToHost));
    printf("flags %u\n", fl);
    return 0;
}

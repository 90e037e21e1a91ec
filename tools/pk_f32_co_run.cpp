// Loads hand-edited variants of tools/pk_f32_repro.hip's device code and checks them against the scalar kernel.
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o repro.s tools/pk_f32_repro.hip      (then edit repro.s)
//   /opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c repro.s -o repro.o
//   /opt/rocm/lib/llvm/bin/ld.lld -shared repro.o -o repro.co
//   pk_f32_co_run repro.co [kernel ...]        (kernel = mangled k_mix<...> names; default: the failing BLEND 4)
// Prints, per kernel and occupancy, the number of threads whose result differs from k_mix<0,false> of the same file.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: %s file.co [kernel ...]\n", argv[0]); return 2; }
    const int iters = 400, blocks = 4096, reps = 3;
    const unsigned n16 = 1u << 20;
    std::vector<float> h(n16 * 4);
    unsigned s = 777u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.3f; }
    void *table, *out, *sink;
    CHECK(hipMalloc(&table, n16 * 16));
    CHECK(hipMemcpy(table, h.data(), n16 * 16, hipMemcpyHostToDevice));
    const size_t n_out = (size_t)blocks * 256;
    CHECK(hipMalloc(&out, n_out * 16));
    CHECK(hipMalloc(&sink, 4));
    hipModule_t mod;
    CHECK(hipModuleLoad(&mod, argv[1]));
    auto launch = [&](const char* name, unsigned lds) {
        hipFunction_t f;
        CHECK(hipModuleGetFunction(&f, mod, name));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        struct { void* table; unsigned n16; int iters; void* out; void* sink; } args{table, n16, iters, out, sink};
        size_t size = sizeof(args);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        CHECK(hipModuleLaunchKernel(f, blocks, 1, 1, 256, 1, 1, lds, 0, nullptr, cfg));
        CHECK(hipDeviceSynchronize());
    };
    std::vector<float> ref(n_out * 4), got(n_out * 4);
    launch("_Z5k_mixILi0ELb0EEvPKDv4_fjiPS0_Pf", 100 * 1024);
    CHECK(hipMemcpy(ref.data(), out, n_out * 16, hipMemcpyDeviceToHost));
    const char* dflt[] = {"_Z5k_mixILi4ELb1EEvPKDv4_fjiPS0_Pf"};
    const char** names = argc > 2 ? const_cast<const char**>(argv + 2) : dflt;
    const int n_names = argc > 2 ? argc - 2 : 1;
    for (int k = 0; k < n_names; ++k)
        for (unsigned lds : {60u * 1024u, 100u * 1024u})
            for (int r = 0; r < reps; ++r) {
                launch(names[k], lds);
                CHECK(hipMemcpy(got.data(), out, n_out * 16, hipMemcpyDeviceToHost));
                size_t bad = 0, hist[4] = {0, 0, 0, 0};
                for (size_t i = 0; i < n_out; ++i)
                    if (memcmp(&got[i * 4], &ref[i * 4], 16) != 0) { ++bad; ++hist[(i & 63) >> 4]; }
                printf("%s %s %d WG/CU: differing threads %zu  lanes[0-15 16-31 32-47 48-63] = %zu %zu %zu %zu\n", argv[1], names[k],
                       lds > 80 * 1024 ? 1 : 2, bad, hist[0], hist[1], hist[2], hist[3]);
            }
    return 0;
}

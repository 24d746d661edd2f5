// fir_ablate.hip -- times the north-star analysis kernel (fir_decimate_kernel16<4096, u8, 8,
// late>) with parts switched off (template parameter EXP of the kernel), to see which phase
// the launch waits on.  Tables hold zeros / constants: the arithmetic is the real
// instruction stream, the results are not.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I go-sdr_amd/csrc tools/fir_ablate.hip -o /tmp/fir_ablate
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "hz_chain_dev.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using namespace hz;

template <int EXP> static float run(const void *in, float2 *spec, const float2 *hf, FvTabs tabs, PolyTabs pt, size_t n,
                                     int reps) {
    constexpr int N = 4096, D = 8;
    const unsigned off = 1024, hop = 3072;
    const size_t nblocks = (n + hop - 1) / hop;
    EwProgram P{};
    LateFilters late{};
    late.h[0] = hf;
    SlowBlocks slow{};
    slow.n = 2;
    slow.idx[0] = 0;
    slow.idx[1] = (unsigned)nblocks - 1;
    auto k = fir_decimate_kernel16<N, HZSDR_FMT_U8, D, true, EXP>;
    const size_t lds = fir_lds_bytes(N, D);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    for (int r = 0; r < reps + 3; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3((unsigned)(2 + 8 * ((nblocks - 2 + 7) / 8))), dim3(256), lds, 0, in, spec + (n / 3072 + 2) * 512, (const float2 *)nullptr,
                           (float2 *)nullptr, hf, tabs, spec, nblocks, n, hop, off, (unsigned)D, P, late, pt, slow);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("EXP %2d: min %.1f us  avg %.1f us\n", EXP, best * 1e3f, sum / reps * 1e3f);
    return best;
}

// BASELINE config 3 as a FIR: fir_decimate_kernel16<4096, c64, FOLD 0, late>, no elementwise stages, D = 1
template <int EXP> static float run_c64(const void *in, float2 *out, const float2 *hf, FvTabs tabs, PolyTabs pt, size_t n, int reps) {
    constexpr int N = 4096;
    const unsigned off = 1024, hop = 3072;
    const size_t nblocks = (n + hop - 1) / hop;
    EwProgram P{};
    LateFilters late{};
    late.h[0] = hf;
    SlowBlocks slow{};
    slow.n = 2;
    slow.idx[0] = 0;
    slow.idx[1] = (unsigned)nblocks - 1;
    auto k = fir_decimate_kernel16<N, HZSDR_FMT_C64, 0, true, EXP>;
    const size_t lds = fir_lds_bytes(N, 0);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    for (int r = 0; r < reps + 3; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3((unsigned)(2 + 8 * ((nblocks - 2 + 7) / 8))), dim3(256), lds, 0, in, out, (const float2 *)nullptr,
                           (float2 *)nullptr, hf, tabs, (float2 *)nullptr, nblocks, n, hop, off, 1u, P, late, pt, slow);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("c64 D = 1, EXP %2d: min %.1f us  avg %.1f us\n", EXP, best * 1e3f, sum / reps * 1e3f);
    return best;
}

int main() {
    const size_t n = (size_t)1 << 24;
    void *in;
    float2 *spec, *hf;
    cf4 *t;
    CK(hipMalloc(&in, n * 2 + 65536));
    CK(hipMalloc(&spec, (n / 3072 + 2) * 512 * 8 * 2));  // folded spectra, then outputs
    CK(hipMalloc(&hf, 4096 * 8));
    CK(hipMalloc(&t, 1 << 20));
    std::vector<unsigned char> h(n * 2);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(i * 2654435761u >> 24);
    CK(hipMemcpy(in, h.data(), h.size(), hipMemcpyHostToDevice));
    std::vector<float> ones(1 << 18, 0.5f);
    CK(hipMemcpy(t, ones.data(), 1 << 20, hipMemcpyHostToDevice));
    CK(hipMemcpy(hf, ones.data(), 4096 * 8, hipMemcpyHostToDevice));
    FvTabs tabs{t, t};
    PolyTabs pt{t, t + 4096};
    run<0>(in, spec, hf, tabs, pt, n, 10);
    run<1>(in, spec, hf, tabs, pt, n, 10);
    run<2>(in, spec, hf, tabs, pt, n, 10);
    run<8>(in, spec, hf, tabs, pt, n, 10);
    run<16>(in, spec, hf, tabs, pt, n, 10);
    run<24>(in, spec, hf, tabs, pt, n, 10);
    run<7>(in, spec, hf, tabs, pt, n, 10);
    run<31>(in, spec, hf, tabs, pt, n, 10);
    run<0>(in, spec, hf, tabs, pt, n, 10);
    {   // config 3: 1 = no input loads, 2 = no filter loads, 4 = no output stores
        void *cin;
        float2 *cout;
        CK(hipMalloc(&cin, n * 8 + 65536));
        CK(hipMalloc(&cout, n * 8 + 65536));
        {   // (noise-like samples: zeros let the chip clock higher than real data does)
            std::vector<float> hx(n * 2);
            unsigned s = 99;
            for (auto &v : hx) {
                s = s * 1664525u + 1013904223u;
                v = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
            }
            CK(hipMemcpy(cin, hx.data(), n * 8, hipMemcpyHostToDevice));
        }
        run_c64<0>(cin, cout, hf, tabs, pt, n, 20);
        run_c64<1>(cin, cout, hf, tabs, pt, n, 20);
        run_c64<2>(cin, cout, hf, tabs, pt, n, 20);
        run_c64<4>(cin, cout, hf, tabs, pt, n, 20);
        run_c64<5>(cin, cout, hf, tabs, pt, n, 20);
        run_c64<7>(cin, cout, hf, tabs, pt, n, 20);
        run_c64<0>(cin, cout, hf, tabs, pt, n, 20);
    }
    return 0;
}

// mfma_fir.hip -- times hz::mm::fir_mm_kernel<u8, 8> (hz_firmm.h) in isolation on 12 rotating
// 2^24-sample buffers (384 MiB: every launch reads its input from HBM), with phases switched
// off (template parameter EXP).  Tables hold constants: the instruction stream is the real one,
// the results are not.  (The first version of this file was the prototype that pinned the lane
// maps of v_mfma_i32_*_i8 with exact integer data against a float64 direct form.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I go-sdr_amd/csrc tools/mfma_fir.hip -o tools/bin/mfma_fir
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "hz_firmm.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using namespace hz;

static unsigned long long *g_stamps = nullptr;

template <int EXP> static void run(void *const *in, float2 *out, const float2 *taps, const void *tab, size_t n, int ntaps,
                                   bool shift) {
    constexpr int D = 8;
    mm::Geom g{};
    g.ntaps = ntaps;
    g.w0 = (ntaps - 1 + 7) / 8 * 8;
    g.ks = ((2 * (g.w0 + D * (mm::kT - 1) + 1) + 31) / 32 + D - 1) / D * D;
    g.e0 = 2 * (g.ks + 4);
    g.ne = g.e0 + (D / 8) * (mm::kT - 1) + 1;
    g.shift = 40;
    g.off = (unsigned)((ntaps - 1 + D - 1) / D * D);
    EwProgram P{};
    if (shift) {
        P.n = 1;
        P.op[0].kind = EW_SHIFT;
        P.op[0].tau_shift = -1.5707963e7;
        P.segs.n = 1;
        P.segs.first[0] = 0;
        P.segs.t0[0] = 0.25;
        P.segs.step[0] = 5e-8;
    }
    mm::Runs R{};
    mm::Fix F{};
    const uint32_t n_out = (uint32_t)(n / D), lo = (uint32_t)((ntaps - 1 + D - 1) / D);
    R.n = 1;
    R.tab[0] = tab;
    // steady state: run 0 continues the previous call's run (no fix-up workgroups); FIX=1: a first call
    const bool fix = getenv("FIX") != nullptr;
    R.cont = fix ? 0 : 1;
    R.m_lo[0] = fix ? (lo + mm::kT - 1) / mm::kT * mm::kT : 0;
    R.m_hi[0] = n_out;
    R.n_wg = (int)((n_out + mm::chunk_out(mm::blocks_for(D)) - 1) / mm::chunk_out(mm::blocks_for(D)));
    if (fix) {
        F.n = 1;
        F.m_a[0] = 0;
        F.m_b[0] = R.m_lo[0];
        F.n_wg = (int)((R.m_lo[0] + mm::kFixOut - 1) / mm::kFixOut);
    }
    const size_t lds = mm::chunk_bytes(D, g.ks) + mm::kLookAhead;
    const unsigned grid = (unsigned)R.n_wg;
    auto k = mm::fir_mm_kernel<HZSDR_FMT_U8, D, EXP>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    const int reps = 24;
    for (int r = 0; r < reps + 4; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(mm::kThreads), lds, 0, (const void *)in[r % 12], out, (const float2 *)nullptr,
                           out + n_out, (const uint8_t *)taps, (uint8_t *)(out + n_out + 4096), taps, n, g, P, R, F, g_stamps);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 4) { best = ms < best ? ms : best; sum += ms; }
    }
    CK(hipGetLastError());
    printf("EXP %2d%s: grid %u, LDS %zu, min %.1f us  avg %.1f us\n", EXP, shift ? " +Shift" : "       ", grid, lds, best * 1e3f,
           sum / reps * 1e3f);
    if (EXP & 64) {
        // phase durations per workgroup of the last launch, in s_memtime ticks (the counters of
        // different XCDs are not aligned: only differences inside a workgroup mean anything)
        std::vector<unsigned long long> st((size_t)grid * 8);
        CK(hipMemcpy(st.data(), g_stamps, st.size() * 8, hipMemcpyDeviceToHost));
        const char *names[6] = {"input loads + landing", "barrier", "matrix loop", "exchange + mixer", "stores", "whole workgroup"};
        for (int k = 0; k < 6; k++) {
            std::vector<double> v;
            for (unsigned b = 0; b < grid; b++)
                v.push_back(k < 5 ? (double)(st[b * 8 + k + 1] - st[b * 8 + k]) : (double)(st[b * 8 + 5] - st[b * 8]));
            std::sort(v.begin(), v.end());
            printf("    %-22s min %7.0f  p10 %7.0f  median %7.0f  p90 %7.0f  max %7.0f ticks\n", names[k], v[0], v[v.size() / 10],
                   v[v.size() / 2], v[v.size() * 9 / 10], v.back());
        }
        if (EXP & 256) {  // s_memrealtime is one clock for the chip: when the workgroups start and end
            unsigned long long t0 = ~0ull;
            for (unsigned b = 0; b < grid; b++) t0 = st[b * 8] < t0 ? st[b * 8] : t0;
            for (int k = 0; k < 6; k += 5) {
                std::vector<double> v;
                for (unsigned b = 0; b < grid; b++) v.push_back((double)(st[b * 8 + k] - t0));
                std::sort(v.begin(), v.end());
                printf("    %-22s min %7.0f  p10 %7.0f  median %7.0f  p90 %7.0f  max %7.0f ticks after the first start\n",
                       k ? "workgroup end" : "workgroup start", v[0], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
            }
        }
    }
}

int main(int argc, char **argv) {
    const size_t n = (size_t)1 << 24;
    const int ntaps = argc > 1 ? atoi(argv[1]) : 1024;
    void *in[12];
    float2 *out, *taps;
    void *tab;
    std::vector<unsigned char> h(n * 2);
    for (int b = 0; b < 12; b++) {
        CK(hipMalloc(&in[b], n * 2 + 65536));
        for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)((i + b) * 2654435761u >> 24);
        CK(hipMemcpy(in[b], h.data(), h.size(), hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&out, (n / 8 + 8192) * 8));
    CK(hipMalloc(&taps, ntaps * 8 + 65536));
    CK(hipMemset(taps, 0, ntaps * 8 + 65536));
    std::vector<unsigned char> t(1 << 20);
    for (size_t i = 0; i < t.size(); i++) t[i] = (unsigned char)(i * 40503u >> 8);
    CK(hipMalloc(&tab, t.size()));
    CK(hipMemcpy(tab, t.data(), t.size(), hipMemcpyHostToDevice));
    printf("fir_mm_kernel<u8, 8>, 2^24 samples, %d taps; EXP 1 = no input loads, 2 = no matrix loop, 4 = no elementwise program, 8 = no stores\n", ntaps);
    run<0>(in, out, taps, tab, n, ntaps, true);
    run<0>(in, out, taps, tab, n, ntaps, false);
    run<2>(in, out, taps, tab, n, ntaps, true);
    run<8>(in, out, taps, tab, n, ntaps, true);
    run<4>(in, out, taps, tab, n, ntaps, true);
    CK(hipMalloc(&g_stamps, 8 * 8 * 4096));
    run<64>(in, out, taps, tab, n, ntaps, true);
    run<64 + 128>(in, out, taps, tab, n, ntaps, true);
    run<64 + 256>(in, out, taps, tab, n, ntaps, true);  // the same phases in 10 ns ticks
    return 0;
}

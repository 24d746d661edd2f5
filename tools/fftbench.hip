// fftbench.hip -- standalone check + timing of the two workgroup FFT cores
// (hz_fft.h radix-4, hz_fft16.h radix-16) on 2^24 complex points:
// forward -> pointwise multiply -> backward per block, c64 in / c64 out.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/fftbench.hip -o build/fftbench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <complex>
#include <vector>

#include "../go-sdr_amd/csrc/hz_fft.h"
#include "../go-sdr_amd/csrc/hz_fft16.h"

using namespace hz;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

template <int N>
__global__ __launch_bounds__(fft_block(N), fft_waves(N)) void conv_old(const float2 *in, float2 *out,
                                                                       const float2 *__restrict__ filt,
                                                                       const float2 *__restrict__ tw,
                                                                       size_t nblocks) {
    constexpr int TPT = fft_tpt(N), CNT = N / TPT;
    __shared__ float2 lds[N];
    const int lane = threadIdx.x;
    const size_t b = blockIdx.x;
    if (b >= nblocks) return;
    FftRegs<N> R;
#pragma unroll
    for (int q = 0; q < CNT; q++) R.v[q] = in[b * N + (fft_odd(N) ? edge2_index<N>(q, lane) : edge4_index<N>(q, lane))];
    fft_forward_regs<N>(R, lds, tw, lane);
#pragma unroll
    for (int q = 0; q < CNT; q++) R.v[q] = cmulf(R.v[q], filt[edge4_index<N>(q, lane)]);
    fft_backward_regs<N>(R, lds, tw, lane);
#pragma unroll
    for (int q = 0; q < CNT; q++) out[b * N + (fft_odd(N) ? edge2_index<N>(q, lane) : edge4_index<N>(q, lane))] = R.v[q];
}

template <int N, int MODE>  // MODE 0: fwd+mul+bwd, 1: forward only, 2: forward only, compute floor (all blocks read block 0, nothing written)
__global__ __launch_bounds__(f16::block(N)) void conv_new(const float2 *in, float2 *out,
                                                          const float2 *__restrict__ filt,
                                                          const float2 *__restrict__ tw, size_t nblocks) {
    constexpr int TPT = f16::tpt(N), XPB = f16::xpb(N), R0 = f16::first_radix(N);
    __shared__ float2 lds_all[XPB * f16::lds_elems(N)];
    const int sub = threadIdx.x / TPT, lane = threadIdx.x % TPT;
    float2 *lds = lds_all + sub * f16::lds_elems(N);
    const size_t b = (size_t)blockIdx.x * XPB + sub;
    const bool live = b < nblocks;
    float2 v[16];
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = live ? in[(MODE == 2 ? 0 : b * N) + f16::edge_index<N, R0>(q, lane)] : make_float2(0.f, 0.f);
    f16::forward<N>(v, lds, tw, lane);
    if (MODE == 2) {
        float acc = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) acc += v[q].x + v[q].y;
        if (acc == 123.456f) out[b] = make_float2(acc, acc);  // keeps the transform live, never taken
        return;
    }
    if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = f16::cmul(v[q], filt[f16::edge_index<N, 16>(q, lane)]);
        f16::backward<N>(v, lds, tw, lane);
        if (live) {
#pragma unroll
            for (int q = 0; q < 16; q++) out[b * N + f16::edge_index<N, R0>(q, lane)] = v[q];
        }
    } else if (live) {
#pragma unroll
        for (int q = 0; q < 16; q++) out[b * N + f16::edge_index<N, 16>(q, lane)] = v[q];
    }
}

static void cpu_fft(std::vector<std::complex<double>> &a, bool inv) {
    size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        double ang = (inv ? 2 : -2) * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; k++) {
                std::complex<double> w(cos(ang * k), sin(ang * k));
                auto x = a[i + k + len / 2] * w;
                a[i + k + len / 2] = a[i + k] - x;
                a[i + k] += x;
            }
    }
}

template <int N> static void run(size_t total, const char *which) {
    const size_t nblocks = total / N;
    std::vector<float2> h(total), hf(N), htw(N), ho(total);
    srand(1);
    for (auto &x : h) x = make_float2(rand() / (float)RAND_MAX * 2 - 1, rand() / (float)RAND_MAX * 2 - 1);
    for (auto &x : hf) x = make_float2(rand() / (float)RAND_MAX * 2 - 1, rand() / (float)RAND_MAX * 2 - 1);
    for (int m = 0; m < N; m++) htw[m] = make_float2((float)cos(-2 * M_PI * m / N), (float)sin(-2 * M_PI * m / N));
    float2 *din, *dout, *dfilt, *dtw;
    CK(hipMalloc(&din, total * 8)); CK(hipMalloc(&dout, total * 8)); CK(hipMalloc(&dfilt, N * 8)); CK(hipMalloc(&dtw, N * 8));
    CK(hipMemcpy(din, h.data(), total * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dfilt, hf.data(), N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtw, htw.data(), N * 8, hipMemcpyHostToDevice));
    // reference for block 0 and the last block
    auto ref_block = [&](size_t b, bool conv, std::vector<std::complex<double>> &a) {
        a.resize(N);
        for (int i = 0; i < N; i++) a[i] = {h[b * N + i].x, h[b * N + i].y};
        cpu_fft(a, false);
        if (conv) {
            for (int i = 0; i < N; i++) a[i] *= std::complex<double>((float)a[i].real() * 0 + hf[i].x, hf[i].y);
            cpu_fft(a, true);
        }
    };
    auto check = [&](const char *name, bool conv) {
        CK(hipMemcpy(ho.data(), dout, total * 8, hipMemcpyDeviceToHost));
        double worst = 0;
        for (size_t b : {(size_t)0, nblocks - 1}) {
            std::vector<std::complex<double>> a;
            ref_block(b, conv, a);
            double num = 0, den = 0;
            for (int i = 0; i < N; i++) {
                std::complex<double> g(ho[b * N + i].x, ho[b * N + i].y);
                num += std::norm(g - a[i]);
                den += std::norm(a[i]);
            }
            worst = fmax(worst, sqrt(num / den));
        }
        printf("  %-28s rel-L2 err %.3e %s\n", name, worst, worst < 5e-6 ? "ok" : "WRONG");
    };
    auto time_it = [&](const char *name, auto launch) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int i = 0; i < 3; i++) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9, sum = 0;
        for (int i = 0; i < 10; i++) {
            CK(hipEventRecord(a));
            launch();
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            best = fminf(best, ms);
            sum += ms;
        }
        printf("  %-28s N=%d  avg %.1f us  min %.1f us  (%.2f TB/s at 16 B/pt)\n", name, N, sum * 100, best * 1000,
               total * 16.0 / (best * 1e-3) / 1e12);
    };
    printf("N = %d, %zu blocks (%s)\n", N, nblocks, which);
    if constexpr (N <= 8192) {
        auto l_old = [&] { hipLaunchKernelGGL((conv_old<N>), dim3(nblocks), dim3(fft_block(N)), 0, 0, din, dout, dfilt, dtw, nblocks); };
        l_old();
        CK(hipDeviceSynchronize());
        check("radix-4 core conv", true);
        time_it("radix-4 core conv", l_old);
    }
    if constexpr (f16::ok(N)) {
        constexpr int XPB = f16::xpb(N);
        auto l_new = [&] { hipLaunchKernelGGL((conv_new<N, 0>), dim3((nblocks + XPB - 1) / XPB), dim3(f16::block(N)), 0, 0, din, dout, dfilt, dtw, nblocks); };
        CK(hipMemset(dout, 0, total * 8));
        l_new();
        CK(hipDeviceSynchronize());
        check("radix-16 core conv", true);
        time_it("radix-16 core conv", l_new);
        auto l_fwd = [&] { hipLaunchKernelGGL((conv_new<N, 1>), dim3((nblocks + XPB - 1) / XPB), dim3(f16::block(N)), 0, 0, din, dout, dfilt, dtw, nblocks); };
        l_fwd();
        CK(hipDeviceSynchronize());
        check("radix-16 core forward", false);
        time_it("radix-16 core forward", l_fwd);
        auto l_cf = [&] { hipLaunchKernelGGL((conv_new<N, 2>), dim3((nblocks + XPB - 1) / XPB), dim3(f16::block(N)), 0, 0, din, dout, dfilt, dtw, nblocks); };
        time_it("radix-16 forward, no HBM", l_cf);
    }
    CK(hipFree(din)); CK(hipFree(dout)); CK(hipFree(dfilt)); CK(hipFree(dtw));
}

int main(int argc, char **argv) {
    size_t total = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 24);
    run<256>(total, "");
    run<512>(total, "");
    run<1024>(total, "");
    run<2048>(total, "");
    run<4096>(total, "");
    run<8192>(total, "");
    return 0;
}

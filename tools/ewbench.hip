// ewbench.hip -- where does the Shift kernel's time go?  Times, on 2^24 c64 samples:
//   copy (out of place), scale in place / out of place, and the NCO rotate
//   (exact Sincos vs FMA fast path) with and without memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ewbench.hip -o build/ewbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../go-sdr_amd/csrc/hz_device.h"

using namespace hz;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE, int U>  // MODE 0 copy, 1 scale, 2 nco exact, 3 nco fast
__global__ __launch_bounds__(256) void k(const float4 *in, float4 *out, size_t nvec, double t0, double step, double tau_shift) {
    const size_t tile = (size_t)256 * U;
    for (size_t b = (size_t)blockIdx.x * tile; b < nvec; b += (size_t)gridDim.x * tile) {
        float4 a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = in[b + u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = b + u * 256 + threadIdx.x;
            float4 r = a[u];
            if (MODE == 1) r = make_float4(r.x * 0.5f, r.y * 0.5f, r.z * 0.5f, r.w * 0.5f);
            if (MODE >= 2) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    float2 v = h ? make_float2(r.z, r.w) : make_float2(r.x, r.y);
                    if (MODE == 4) {
                        double ts = __fma_rn(u32_to_f64((uint32_t)(2 * i + h)), step, t0);
                        v = go_rotate_by_phase(v, __dmul_rn(tau_shift, ts));
                    } else {
                    double ts = __fma_rn((double)(uint32_t)(2 * i + h), step, t0);
                    double ph = __dmul_rn(tau_shift, ts);
                    float s, c;
                    if (MODE == 2) { double sd, cd; go_sincos(ph, sd, cd); s = (float)sd; c = (float)cd; }
                    else go_sincos_f32(ph, s, c);
                    v = go_cmul(v, make_float2(c, s));
                    }
                    if (h) { r.z = v.x; r.w = v.y; } else { r.x = v.x; r.y = v.y; }
                }
            }
            out[i] = r;
        }
    }
}

// compute only: no loads, one store per lane at the end
template <int MODE>
__global__ __launch_bounds__(256) void kc(float4 *out, size_t nvec, double t0, double step, double tau_shift) {
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float2 v;
            if (MODE == 4) {
                double ts = __fma_rn(u32_to_f64((uint32_t)(2 * i + h)), step, t0);
                v = go_rotate_by_phase(make_float2(acc.x + 1.0f, acc.y), __dmul_rn(tau_shift, ts));
            } else {
            double ts = __fma_rn((double)(uint32_t)(2 * i + h), step, t0);
            double ph = __dmul_rn(tau_shift, ts);
            float s, c;
            if (MODE == 2) { double sd, cd; go_sincos(ph, sd, cd); s = (float)sd; c = (float)cd; }
            else go_sincos_f32(ph, s, c);
            v = go_cmul(make_float2(acc.x + 1.0f, acc.y), make_float2(c, s));
            }
            acc.x += v.x; acc.y += v.y;
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class F> static void timeit(const char *name, size_t bytes, F f) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int i = 0; i < 10; i++) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-44s %7.1f us  %5.2f TB/s\n", name, best * 1000, bytes / (best * 1e-3) / 1e12);
}

int main() {
    const size_t n = (size_t)1 << 24, nvec = n / 2;
    float4 *a, *b, *big;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&big, (size_t)1 << 30));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
    const double t0 = 4.5, step = 5e-8, tau_shift = 6.283185307179586 * 2.5e6;
    const size_t bytes = n * 16;
    auto flush = [&] { CK(hipMemsetAsync(big, 1, (size_t)1 << 30, 0)); };  // evict the 256 MiB Infinity Cache
    for (int cold = 0; cold < 2; cold++) {
        printf("---- %s ----\n", cold ? "cache flushed before every launch (timed incl. flush? no: see below)" : "back to back (Infinity-Cache warm)");
#define RUN(name, MODE, U, IN, OUT, G)                                                                      \
    timeit(name, bytes, [&] { if (cold) { flush(); } hipLaunchKernelGGL((k<MODE, U>), dim3(G), dim3(256), 0, 0, IN, OUT, nvec, t0, step, tau_shift); })
        if (!cold) {
            RUN("copy out-of-place U=4 grid 2048", 0, 4, a, b, 2048);
            RUN("copy out-of-place U=4 grid 8192", 0, 4, a, b, 8192);
            RUN("scale in-place U=1 grid 2048", 1, 1, a, a, 2048);
            RUN("scale in-place U=4 grid 2048", 1, 4, a, a, 2048);
            RUN("scale in-place U=4 grid 8192", 1, 4, a, a, 8192);
            RUN("scale out-of-place U=4 grid 2048", 1, 4, a, b, 2048);
            RUN("nco exact in-place U=1 grid 2048", 2, 1, a, a, 2048);
            RUN("nco exact in-place U=4 grid 2048", 2, 4, a, a, 2048);
            RUN("nco fast in-place U=1 grid 2048", 3, 1, a, a, 2048);
            RUN("nco fast in-place U=2 grid 2048", 3, 2, a, a, 2048);
            RUN("nco fast in-place U=4 grid 2048", 3, 4, a, a, 2048);
            RUN("nco fast in-place U=4 grid 8192", 3, 4, a, a, 8192);
            RUN("nco fast out-of-place U=4 grid 2048", 3, 4, a, b, 2048);
            RUN("nco q in-place U=1 grid 2048", 4, 1, a, a, 2048);
            RUN("nco q in-place U=2 grid 2048", 4, 2, a, a, 2048);
            RUN("nco q in-place U=4 grid 2048", 4, 4, a, a, 2048);
            RUN("nco q out-of-place U=4 grid 2048", 4, 4, a, b, 2048);
            timeit("nco q compute only", bytes, [&] { hipLaunchKernelGGL((kc<4>), dim3(2048), dim3(256), 0, 0, b, nvec, t0, step, tau_shift); });
            timeit("nco exact compute only", bytes, [&] { hipLaunchKernelGGL((kc<2>), dim3(2048), dim3(256), 0, 0, b, nvec, t0, step, tau_shift); });
            timeit("nco fast compute only", bytes, [&] { hipLaunchKernelGGL((kc<3>), dim3(2048), dim3(256), 0, 0, b, nvec, t0, step, tau_shift); });
        }
    }
    return 0;
}

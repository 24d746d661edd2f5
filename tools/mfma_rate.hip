// mfma_rate.hip -- cycles per int8 MFMA (16x16x64 and 32x32x32), one and two waves per SIMD,
// measured with s_memtime around a register-only loop, plus the wall-clock rate of a full grid.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_rate.hip -o tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void rate(int iters, int *sink, unsigned long long *cyc, int seed) {
    v4i a = {seed + (int)threadIdx.x * 0x01020304, seed * 3, (int)threadIdx.x * 77, seed ^ 0x5a5a5a5a};
    v4i b = {seed * 7 + (int)threadIdx.x * 0x11213141, seed * 5, (int)threadIdx.x * 91, seed ^ 0x3c3c3c3c};
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    int r = 0;
    if constexpr (SHAPE == 16) {
        v4i acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = v4i{0, 0, 0, 0};
#pragma unroll 1
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) r += acc[j][0] + acc[j][3];
    } else {
        v16i acc[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[j][q] = 0;
#pragma unroll 1
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) r += acc[j][0] + acc[j][15];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (r == 0x7fffffff) sink[0] = r;
}

template <int SHAPE> static void run(int wg_per_cu, int iters) {
    int *sink;
    unsigned long long *cyc;
    const int grid = 256 * wg_per_cu;
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&cyc, 8 * grid));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(rate<SHAPE>, dim3(grid), dim3(256), 0, 0, iters, sink, cyc, 12345 + r);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    unsigned long long c0;
    CK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
    const int per_iter = SHAPE == 16 ? 8 : 4;
    const double n_mfma = (double)iters * per_iter;            // per wave
    const double macs = SHAPE == 16 ? 16.0 * 16 * 64 : 32.0 * 32 * 32;
    printf("%dx%d i8, %d waves/SIMD: %.1f memtime ticks per MFMA (one wave), wall %.1f us -> %.2f Pop/s, %.1f ns per MFMA per SIMD\n",
           SHAPE, SHAPE, wg_per_cu, (double)c0 / n_mfma, best * 1e3, 2.0 * macs * n_mfma * grid * 4 / (best * 1e-3) * 1e-15,
           best * 1e6 / (n_mfma * wg_per_cu));
}

int main() {
    run<16>(1, 4096);
    run<16>(2, 4096);
    run<32>(1, 4096);
    run<32>(2, 4096);
    return 0;
}

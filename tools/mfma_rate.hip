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

// the accumulator pattern of fir_mm_kernel: 8 accumulators of 16 registers, 2 A x 4 B operands
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void rate8(int iters, int *sink, int seed) {
    v4i a[2], b[4];
    for (int q = 0; q < 2; q++) a[q] = v4i{seed + q + (int)threadIdx.x * 0x01020304, seed * 3 + q, (int)threadIdx.x * 77, seed ^ 0x5a5a5a5a};
    for (int q = 0; q < 4; q++) b[q] = v4i{seed * 7 + q + (int)threadIdx.x * 0x11213141, seed * 5, (int)threadIdx.x * 91 + q, seed ^ 0x3c3c3c3c};
    v16i acc[2][4];
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[d][j][q] = 0;
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[d][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[d], b[j], acc[d][j], 0, 0, 0);
    }
    int r = 0;
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int j = 0; j < 4; j++) r += acc[d][j][0] + acc[d][j][15];
    if (r == 0x7fffffff) sink[0] = r;
}

// the same with the B operands read from LDS two steps ahead (MODE 1), the A operands from global
// memory four steps ahead (MODE 2), or both (MODE 3)
template <int MODE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void rate8m(int iters, int *sink, const v4i *gtab, int seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = threadIdx.x; q < 34816 / 16; q += 128) reinterpret_cast<v4i *>(lds)[q] = v4i{q * seed, q, seed, 7};
    __syncthreads();
    v4i a[4][2], b[4][4];
    for (int r = 0; r < 4; r++)
        for (int q = 0; q < 2; q++) a[r][q] = v4i{seed + q + l * 0x01020304, seed * 3 + q + r, l * 77, seed ^ 0x5a5a5a5a};
    for (int r = 0; r < 4; r++)
        for (int q = 0; q < 4; q++) b[r][q] = v4i{seed * 7 + q + l * 0x11213141, seed * 5 + r, l * 91 + q, seed ^ 0x3c3c3c3c};
    v16i acc[2][4];
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[d][j][q] = 0;
    const unsigned char *bp = lds + 16 * l;
    const v4i *ap = gtab + wave * 4096 + l;
#pragma unroll 1
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (MODE & 1) {
#pragma unroll
                for (int q = 0; q < 4; q++) b[(j + 2) & 3][q] = *reinterpret_cast<const v4i *>(bp + q * 8192 + ((i + j) & 7) * 1024);
            }
#pragma unroll
            for (int d = 0; d < 2; d++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[d][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[j][d], b[j][q], acc[d][q], 0, 0, 0);
            if (MODE & 2) {
                a[j][0] = ap[((i + j) & 31) * 64];
                a[j][1] = ap[2048 + ((i + j) & 31) * 64];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int r = 0;
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int j = 0; j < 4; j++) r += acc[d][j][0] + acc[d][j][15];
    if (r == 0x7fffffff) sink[0] = r;
}

template <int MODE> static void run8m(int iters) {
    int *sink;
    v4i *gtab;
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&gtab, 1 << 20));
    CK(hipMemset(gtab, 1, 1 << 20));
    CK(hipFuncSetAttribute((const void *)rate8m<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 34816));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(rate8m<MODE>, dim3(1024), dim3(128), 34816, 0, iters, sink, gtab, 12345 + r);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    printf("fir_mm pattern, mode %d (1 = B from LDS, 2 = A from global), %d steps: wall %.1f us, %.1f ns per MFMA per SIMD\n", MODE, iters,
           best * 1e3, best * 1e6 / (2.0 * iters * 8));
}

static void run8(int iters) {
    int *sink;
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(rate8, dim3(1024), dim3(128), 34816, 0, iters, sink, 12345 + r);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    // per SIMD: 2 waves x iters x 8 MFMAs
    printf("fir_mm pattern (8 accumulators x 16 regs, 2 waves/SIMD, %d steps): wall %.1f us, %.1f ns per MFMA per SIMD\n", iters,
           best * 1e3, best * 1e6 / (2.0 * iters * 8));
}

// VALU beside MFMA.  WAVES = 1: one wave per SIMD issues 4 MFMAs and NV independent float fmas per trip (its own
// vector work in the shadow of its own matrix instructions).  WAVES = 2: the SIMD's first wave issues MFMAs only, its
// second wave the fmas only (NV per 4 MFMA times, as long as the first wave runs): a partner's epilogue.
template <int NV, int WAVES>
__global__ __launch_bounds__(64 * 4 * WAVES) void mix_rate(int iters, float *fsink, int *sink, unsigned long long *cyc, int seed) {
    const int wave = threadIdx.x >> 6;
    v4i a = {seed + (int)threadIdx.x * 0x01020304, seed * 3, (int)threadIdx.x * 77, seed ^ 0x5a5a5a5a};
    v4i b = {seed * 7 + (int)threadIdx.x * 0x11213141, seed * 5, (int)threadIdx.x * 91, seed ^ 0x3c3c3c3c};
    float f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) f[q] = 1.0f + (float)(threadIdx.x + q) * 1e-3f;
    v16i acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[j][q] = 0;
    const bool do_mfma = WAVES == 1 || wave < 4, do_valu = WAVES == 1 || wave >= 4;  // (uniform per wave)
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        if (do_mfma) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int q = 0; q < NV; q++) f[q & 7] = __fmaf_rn(f[q & 7], 1.0000001f, 1e-7f);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    int r = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) r += acc[j][0] + acc[j][15];
    float fs = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++) fs += f[q];
    if (r == 0x7fffffff) sink[0] = r;
    if (fs == 123.456f) fsink[0] = fs;
}
template <int NV, int WAVES> static void run_mix(int iters) {
    int *sink;
    float *fsink;
    unsigned long long *cyc;
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&fsink, 4));
    CK(hipMalloc(&cyc, 8 * 8 * 256));
    hipLaunchKernelGGL((mix_rate<NV, WAVES>), dim3(256), dim3(64 * 4 * WAVES), 0, 0, iters, fsink, sink, cyc, 777);
    hipLaunchKernelGGL((mix_rate<NV, WAVES>), dim3(256), dim3(64 * 4 * WAVES), 0, 0, iters, fsink, sink, cyc, 778);
    CK(hipDeviceSynchronize());
    unsigned long long c[8];
    CK(hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost));
    if (WAVES == 1)
        printf("one wave per SIMD, 4 MFMAs + %2d fmas per trip: %.1f memtime ticks per trip (4 MFMAs alone: ~%d)\n", NV, (double)c[0] / iters, 4 * 8);
    else
        printf("two waves per SIMD, A: 4 MFMAs per trip, B: %2d fmas per trip: A %.1f ticks per trip, B %.1f ticks per trip = %.1f per fma\n", NV,
               (double)c[0] / iters, (double)c[4] / iters, (double)c[4] / iters / NV);
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
    run_mix<0, 1>(4000);
    run_mix<8, 1>(4000);
    run_mix<16, 1>(4000);
    run_mix<32, 1>(4000);
    run_mix<8, 2>(4000);
    run_mix<32, 2>(4000);
    run8(72);
    run8(720);
    run8m<0>(720);
    run8m<1>(720);
    run8m<2>(720);
    run8m<3>(720);
    run8m<3>(72);
    return 0;
}

// stream_rate.hip -- the library's three plainest streaming kernels (u8 -> c64, boxcar Downsample / 8 from i16, Scale
// in place) re-stated with their memory access as template switches, over a ROTATION of buffers that the 256 MB
// memory-side cache cannot hold: which form of load / store / tile reaches what tools/copy_rate.hip reaches for a plain
// copy (6.2 TB/s: non-temporal both ways, two vectors per lane).  Arithmetic as in the library (exact, not fused).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/stream_rate.hip -o tools/bin/stream_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <int NT, class T> __device__ __forceinline__ T ld(const T *p) {
    if constexpr (NT & 1) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int NT, class T> __device__ __forceinline__ void st(T *p, T v) {
    if constexpr (NT & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}

__device__ __forceinline__ float u8f(unsigned b) { return __fdiv_rn(__fsub_rn((float)b, 127.5f), 127.5f); }

// (b - 127.5) / 127.5 = (2 b - 255) / 255 without the division sequence: quotient estimate by the rounded reciprocal, one
// residual step -- t * c, fma(-q, 255, t), fma(r, c, q) -- correctly rounded for all 256 bytes (checked on the host in main)
__device__ __host__ __forceinline__ float u8f_fast(unsigned b) {
    const float t = (float)(2 * (int)b - 255), c = 1.0f / 255.0f;
    const float q = t * c;
    const float r = __builtin_fmaf(-q, 255.0f, t);
    return __builtin_fmaf(r, c, q);
}
template <int NT> __global__ __launch_bounds__(256) void k_u8_fast(const unsigned *in, v4f *out, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const unsigned w0 = ld<NT>(in + i);
        st<NT>(out + i, v4f{u8f_fast(w0 & 255), u8f_fast((w0 >> 8) & 255), u8f_fast((w0 >> 16) & 255), u8f_fast(w0 >> 24)});
    }
}

// the same arithmetic, U loads of a lane in flight before the first store (a workgroup's tile = U runs of 256 words)
template <int U, int NT> __global__ __launch_bounds__(256) void k_u8_tile(const unsigned *in, v4f *out, size_t nvec) {
    const size_t tile = (size_t)256 * U;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 + tile <= nvec; t0 += (size_t)gridDim.x * tile) {
        unsigned a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = ld<NT>(in + t0 + u * 256 + threadIdx.x);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned w0 = a[u];
            st<NT>(out + t0 + u * 256 + threadIdx.x, v4f{u8f_fast(w0 & 255), u8f_fast((w0 >> 8) & 255), u8f_fast((w0 >> 16) & 255), u8f_fast(w0 >> 24)});
        }
    }
}
// 16 bytes in per lane (8 samples), four 16-byte stores: the wave's store instruction j writes quarter j of each lane's
// 64 bytes -- quarter lines per instruction, whole lines per wave
template <int NT> __global__ __launch_bounds__(256) void k_u8_wide(const v4i *in, v4f *out, size_t nq) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
        const v4i a = ld<NT>(in + i);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned w0 = (unsigned)a[j];
            st<NT>(out + 4 * i + j, v4f{u8f_fast(w0 & 255), u8f_fast((w0 >> 8) & 255), u8f_fast((w0 >> 16) & 255), u8f_fast(w0 >> 24)});
        }
    }
}

// u8 -> c64: a lane converts U groups of 8 bytes (4 samples: one 8-byte load, two 16-byte stores), the groups of a
// trip 256 lanes apart (coalesced)
template <int U, int NT> __global__ __launch_bounds__(256) void k_u8(const v2u *in, v4f *out, size_t ngrp) {
    const size_t tile = (size_t)256 * U;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < ngrp; t0 += (size_t)gridDim.x * tile) {
        v2u a[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * 256 + threadIdx.x;
            if (i < ngrp) a[u] = ld<NT>(in + i);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * 256 + threadIdx.x;
            if (i < ngrp) {
                const unsigned w0 = a[u].x, w1 = a[u].y;
                st<NT>(out + 2 * i, v4f{u8f(w0 & 255), u8f((w0 >> 8) & 255), u8f((w0 >> 16) & 255), u8f(w0 >> 24)});
                st<NT>(out + 2 * i + 1, v4f{u8f(w1 & 255), u8f((w1 >> 8) & 255), u8f((w1 >> 16) & 255), u8f(w1 >> 24)});
            }
        }
    }
}
// the library's form: 4 bytes per lane (one 16-byte store), grid-stride
template <int NT> __global__ __launch_bounds__(256) void k_u8_lib(const unsigned *in, v4f *out, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const unsigned w0 = ld<NT>(in + i);
        st<NT>(out + i, v4f{u8f(w0 & 255), u8f((w0 >> 8) & 255), u8f((w0 >> 16) & 255), u8f(w0 >> 24)});
    }
}

// Downsample / 8 from i16: an output = 8 samples = 32 bytes in, 8 bytes out.  FORM 0: the library's (a lane reads its
// output's two 16-byte vectors: 32-byte lane stride); FORM 1: a lane PAIR shares an output's window -- every load
// instruction of a wave is one contiguous 1 KiB -- and the halves meet by a DPP add... no: the sum is ORDERED from +0
// (stream/downsample.go:99-124), so the second half's lane adds its four samples onto the first half's partial sum
// passed by a shuffle.  U outputs per lane / lane pair per trip.
__device__ __forceinline__ float i16f(int v) { return __fdiv_rn((float)v, 32767.0f); }
template <int FORM, int U, int NT> __global__ __launch_bounds__(256) void k_ds(const v4i *in, float2 *out, size_t count) {
    if constexpr (FORM == 0) {
        const size_t tile = (size_t)256 * U;
        for (size_t t0 = (size_t)blockIdx.x * tile; t0 < count; t0 += (size_t)gridDim.x * tile) {
            v4i a[U][2];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t i = t0 + (size_t)u * 256 + threadIdx.x;
                if (i < count) a[u][0] = ld<NT>(in + 2 * i), a[u][1] = ld<NT>(in + 2 * i + 1);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t i = t0 + (size_t)u * 256 + threadIdx.x;
                if (i < count) {
                    float sr = 0.f, si = 0.f;
#pragma unroll
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int w = a[u][h][k];
                            sr = __fadd_rn(sr, i16f((int)(short)(w & 0xffff)));
                            si = __fadd_rn(si, i16f(w >> 16));
                        }
                    typedef float v2f __attribute__((ext_vector_type(2)));
                    st<NT>(reinterpret_cast<v2f *>(out + i), v2f{__fdiv_rn(sr, 8.0f), __fdiv_rn(si, 8.0f)});
                }
            }
        }
    } else {
        // lane l of a wave: output (l >> 1) of the wave's 32, half (l & 1): vector 2 * out + half = l -- contiguous
        const size_t tile = (size_t)128 * U;  // outputs per workgroup and trip
        for (size_t t0 = (size_t)blockIdx.x * tile; t0 < count; t0 += (size_t)gridDim.x * tile) {
            v4i a[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t v = 2 * (t0 + (size_t)u * 128) + threadIdx.x;
                if (v < 2 * count) a[u] = ld<NT>(in + v);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t o = t0 + (size_t)u * 128 + (threadIdx.x >> 1);
                // first half: 0 + s0 + s1 + s2 + s3; second half continues from the first half's sum
                float sr = 0.f, si = 0.f;
                const bool second = threadIdx.x & 1;
                float c[4][2];
#pragma unroll
                for (int k = 0; k < 4; k++) c[k][0] = i16f((int)(short)(a[u][k] & 0xffff)), c[k][1] = i16f(a[u][k] >> 16);
#pragma unroll
                for (int k = 0; k < 4; k++) sr = __fadd_rn(sr, c[k][0]), si = __fadd_rn(si, c[k][1]);
                const float pr = __shfl_xor(sr, 1), pi = __shfl_xor(si, 1);  // the partner's first-half sum
                if (second) {
                    sr = pr, si = pi;
#pragma unroll
                    for (int k = 0; k < 4; k++) sr = __fadd_rn(sr, c[k][0]), si = __fadd_rn(si, c[k][1]);
                    typedef float v2f __attribute__((ext_vector_type(2)));
                    if (o < count) st<NT>(reinterpret_cast<v2f *>(out + o), v2f{__fdiv_rn(sr, 8.0f), __fdiv_rn(si, 8.0f)});
                }
            }
        }
    }
}

// Scale in place: U vectors of two samples per lane and trip
template <int U, int NT> __global__ __launch_bounds__(256) void k_scale(v4f *buf, size_t nvec, float r) {
    const size_t tile = (size_t)256 * U;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < nvec; t0 += (size_t)gridDim.x * tile) {
        v4f a[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * 256 + threadIdx.x;
            if (i < nvec) a[u] = ld<NT>(buf + i);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * 256 + threadIdx.x;
            if (i < nvec) st<NT>(buf + i, v4f{__fmul_rn(a[u].x, r), __fmul_rn(a[u].y, r), __fmul_rn(a[u].z, r), __fmul_rn(a[u].w, r)});
        }
    }
}

constexpr int kRot = 16;
template <class F> static void timeit(const char *what, double bytes, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 16; r++) launch(r % kRot);
    CK(hipEventRecord(e0, 0));
    const int reps = 64;
    for (int r = 0; r < reps; r++) launch(r % kRot);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    const double us = ms / reps * 1e3;
    printf("  %-64s %6.1f us  %5.2f TB/s\n", what, us, bytes / us * 1e-6);
}

// yardsticks for the one-sided kernels: a pure FILL (16-byte stores only) and a pure READ (16-byte loads, one word
// stored per workgroup so the loads stay) of a c64 buffer
template <int NT> __global__ __launch_bounds__(256) void k_fill(v4f *out, size_t nvec, float v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) st<NT>(out + i, v4f{v, v, v, v});
}
template <int NT> __global__ __launch_bounds__(256) void k_read(const v4f *in, size_t nvec, float *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    v4f acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) acc += ld<NT>(in + i);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x] = acc.x;
}

static unsigned grid_for(size_t items, size_t per_block, unsigned cap) {
    size_t b = (items + per_block - 1) / per_block;
    if (cap && b > cap) b = cap;
    return (unsigned)b;
}

int main() {
    const size_t n = (size_t)1 << 24;
    void *u8[kRot], *c64[kRot], *i16[kRot], *ds[kRot];
    for (int b = 0; b < kRot; b++) {
        CK(hipMalloc(&u8[b], n * 2));
        CK(hipMalloc(&c64[b], n * 8));
        CK(hipMalloc(&i16[b], n * 4));
        CK(hipMalloc(&ds[b], n / 8 * 8));
        CK(hipMemset(u8[b], 0x55 + b, n * 2));
        CK(hipMemset(c64[b], 0, n * 8));
        CK(hipMemset(i16[b], 0x11 + b, n * 4));
    }
    for (int r = 0; r < 200; r++) hipLaunchKernelGGL((k_scale<2, 3>), dim3(16384), dim3(256), 0, 0, (v4f *)c64[r % kRot], n / 2, 1.0f);  // clocks up
    CK(hipDeviceSynchronize());
    printf("2^24 samples per launch, a rotation of %d buffer sets (HBM both ways); NT 1 = non-temporal loads, 2 = stores, 3 = both\n", kRot);
    printf("u8 -> c64 (10 B/sample):\n");
    const size_t ngrp = n / 4, nvec1 = n / 2;
#define U8LIB(NT, CAP) timeit("library form (4 B in, 16 B out per lane), NT " #NT ", grid cap " #CAP, 10.0 * n, [&](int b) { \
        hipLaunchKernelGGL((k_u8_lib<NT>), dim3(grid_for(nvec1, 256, CAP)), dim3(256), 0, 0, (const unsigned *)u8[b], (v4f *)c64[b], nvec1); })
    U8LIB(0, 32768);
    U8LIB(3, 32768);
    U8LIB(2, 32768);
    U8LIB(3, 0);
    {
        int bad = 0;
        for (unsigned b = 0; b < 256; b++) bad += u8f_fast(b) != ((float)b - 127.5f) / 127.5f;
        printf("  (division-free form: %d of 256 bytes differ from (b - 127.5) / 127.5)\n", bad);
    }
#define U8FAST(NT, CAP) timeit("library form, division-free arithmetic, NT " #NT ", grid cap " #CAP, 10.0 * n, [&](int b) { \
        hipLaunchKernelGGL((k_u8_fast<NT>), dim3(grid_for(nvec1, 256, CAP)), dim3(256), 0, 0, (const unsigned *)u8[b], (v4f *)c64[b], nvec1); })
    U8FAST(0, 32768);
    U8FAST(3, 32768);
    U8FAST(3, 0);
#define U8TILE(U, NT, CAP) timeit("division-free, " #U " loads in flight per lane, NT " #NT ", grid cap " #CAP, 10.0 * n, [&](int b) { \
        hipLaunchKernelGGL((k_u8_tile<U, NT>), dim3(grid_for(nvec1, 256 * U, CAP)), dim3(256), 0, 0, (const unsigned *)u8[b], (v4f *)c64[b], nvec1); })
    U8TILE(2, 3, 0);
    U8TILE(4, 3, 0);
    U8TILE(8, 3, 0);
    U8TILE(4, 3, 4096);
    U8TILE(4, 2, 0);
    timeit("division-free, 16 B in / 64 B out per lane, NT 3", 10.0 * n, [&](int b) {
        hipLaunchKernelGGL((k_u8_wide<3>), dim3(grid_for(n / 8, 256, 0)), dim3(256), 0, 0, (const v4i *)u8[b], (v4f *)c64[b], n / 8); });
    timeit("division-free, 16 B in / 64 B out per lane, NT 0", 10.0 * n, [&](int b) {
        hipLaunchKernelGGL((k_u8_wide<0>), dim3(grid_for(n / 8, 256, 0)), dim3(256), 0, 0, (const v4i *)u8[b], (v4f *)c64[b], n / 8); });
#define U8(U, NT, CAP) timeit("8 B in, 32 B out per lane, U " #U ", NT " #NT ", grid cap " #CAP, 10.0 * n, [&](int b) { \
        hipLaunchKernelGGL((k_u8<U, NT>), dim3(grid_for(ngrp, 256 * U, CAP)), dim3(256), 0, 0, (const v2u *)u8[b], (v4f *)c64[b], ngrp); })
    U8(1, 0, 0);
    U8(1, 3, 0);
    U8(2, 3, 0);
    U8(2, 2, 0);
    U8(4, 3, 0);
    U8(2, 3, 4096);
    printf("Downsample / 8 from i16 (5 B per input sample):\n");
    const size_t cnt = n / 8;
#define DS(FORM, U, NT, CAP) timeit("form " #FORM " (0: lane per output, 1: lane pair, contiguous loads), U " #U ", NT " #NT ", grid cap " #CAP, 5.0 * n, [&](int b) { \
        hipLaunchKernelGGL((k_ds<FORM, U, NT>), dim3(grid_for(cnt, (FORM ? 128 : 256) * U, CAP)), dim3(256), 0, 0, (const v4i *)i16[b], (float2 *)ds[b], cnt); })
    DS(0, 1, 0, 0);
    DS(0, 1, 1, 0);
    DS(0, 1, 3, 0);
    DS(0, 2, 3, 0);
    DS(1, 1, 0, 0);
    DS(1, 1, 3, 0);
    DS(1, 2, 3, 0);
    DS(1, 4, 3, 0);
    DS(1, 2, 1, 0);
    printf("Scale in place (16 B/sample):\n");
    const size_t nv = n / 2;
#define SC(U, NT, CAP) timeit("U " #U ", NT " #NT ", grid cap " #CAP, 16.0 * n, [&](int b) { \
        hipLaunchKernelGGL((k_scale<U, NT>), dim3(grid_for(nv, 256 * U, CAP)), dim3(256), 0, 0, (v4f *)c64[b], nv, 0.999f); })
    SC(1, 0, 32768);
    SC(1, 3, 32768);
    SC(2, 3, 0);
    SC(2, 0, 0);
    SC(4, 3, 0);
    SC(2, 2, 0);
    printf("one-sided yardsticks over the c64 buffers (8 B/sample):\n");
    float *sink;
    CK(hipMalloc(&sink, 1 << 20));
#define FILL(NT, CAP) timeit("fill (stores only), NT " #NT ", grid cap " #CAP, 8.0 * n, [&](int b) { \
        hipLaunchKernelGGL((k_fill<NT>), dim3(grid_for(n / 2, 256, CAP)), dim3(256), 0, 0, (v4f *)c64[b], n / 2, 0.5f); })
    FILL(0, 32768);
    FILL(2, 32768);
    FILL(2, 0);
#define READ(NT, CAP) timeit("read (loads only), NT " #NT ", grid cap " #CAP, 8.0 * n, [&](int b) { \
        hipLaunchKernelGGL((k_read<NT>), dim3(grid_for(n / 2, 256, CAP)), dim3(256), 0, 0, (const v4f *)c64[b], n / 2, sink); })
    READ(0, 32768);
    READ(1, 32768);
    READ(1, 0);
    return 0;
}

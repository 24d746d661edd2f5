// ewbench.hip -- where does the Shift kernel's time go?  Times, on 2^24 c64 samples:
//   copy / scale (memory only), the NCO rotate with the library's exact Sincos
//   restatement in place and out of place, and the same arithmetic with NO memory
//   traffic (the kernel's compute floor).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ewbench.hip -o build/ewbench
//
// Round-1 findings (MI355X): copy 34 us (buffers fit the Infinity Cache when run back
// to back), NCO compute floor 44-46 us, NCO with memory 49-51 us.  Two variants that
// were tried here and NOT kept in the library because they changed nothing: FMA-Horner
// polynomials with an exactness guard (16 fma instead of 37 mul/add: floor 46 us) and
// replacing 7 of the 11 conversion instructions per sample (floor 47 us).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../go-sdr_amd/csrc/hz_device.h"

using namespace hz;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ float2 nco_one(float2 v, size_t j, double t0, double step, double tau_shift) {
    const double ts = __fma_rn((double)(uint32_t)j, step, t0);
    double s, c;
    go_sincos(__dmul_rn(tau_shift, ts), s, c);
    return go_cmul(v, make_float2((float)c, (float)s));
}

template <int MODE, int U>  // MODE 0 copy, 1 scale, 2 nco
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
            if (MODE == 2) {
                float2 l = nco_one(make_float2(r.x, r.y), 2 * i, t0, step, tau_shift);
                float2 h = nco_one(make_float2(r.z, r.w), 2 * i + 1, t0, step, tau_shift);
                r = make_float4(l.x, l.y, h.x, h.y);
            }
            out[i] = r;
        }
    }
}

// compute only: no loads, one store per lane at the end
__global__ __launch_bounds__(256) void kc(float4 *out, size_t nvec, double t0, double step, double tau_shift) {
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float2 v = nco_one(make_float2(acc.x + 1.0f, acc.y), 2 * i + h, t0, step, tau_shift);
            acc.x += v.x;
            acc.y += v.y;
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
    float4 *a, *b;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
    const double t0 = 4.5, step = 5e-8, tau_shift = 6.283185307179586 * 2.5e6;
    const size_t bytes = n * 16;
#define RUN(name, MODE, U, IN, OUT, G) \
    timeit(name, bytes, [&] { hipLaunchKernelGGL((k<MODE, U>), dim3(G), dim3(256), 0, 0, IN, OUT, nvec, t0, step, tau_shift); })
    RUN("copy out-of-place U=4 grid 2048", 0, 4, a, b, 2048);
    RUN("scale in-place U=1 grid 2048", 1, 1, a, a, 2048);
    RUN("scale in-place U=4 grid 2048", 1, 4, a, a, 2048);
    RUN("scale out-of-place U=4 grid 2048", 1, 4, a, b, 2048);
    RUN("nco in-place U=1 grid 2048", 2, 1, a, a, 2048);
    RUN("nco in-place U=2 grid 2048", 2, 2, a, a, 2048);
    RUN("nco in-place U=4 grid 2048", 2, 4, a, a, 2048);
    RUN("nco out-of-place U=4 grid 2048", 2, 4, a, b, 2048);
    timeit("nco compute only (no memory)", bytes, [&] { hipLaunchKernelGGL(kc, dim3(2048), dim3(256), 0, 0, b, nvec, t0, step, tau_shift); });
    return 0;
}

// narrow_check.hip -- the claim of sincos_narrow (csrc/hz_device.h) on the DEVICE functions themselves: wherever the
// check accepts, the float32 pair equals complex64(go_sincos(x)) bit for bit.  2^36 phases by default (~7e10), three
// kinds: uniform over +-10^u (u in [-17, 8.7]); within 3 ulp of multiples of pi/4 up to 2^29; tau shift ts as the
// kernel forms it (ts = k / fs, fs and shift random per block of a million).   tools/bin/narrow_check [log2 phases]
#include <stdio.h>
#include <stdlib.h>

#include "hz_device.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using namespace hz;

__device__ __forceinline__ uint64_t sm(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t &s) { return (double)(sm(s) >> 11) * (1.0 / 9007199254740992.0); }

__global__ __launch_bounds__(256) void k_check(unsigned long long *cnt, uint64_t seed, int iters) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t st = seed + gid * 0xD1B54A32D192ED03ull;
    unsigned long long total = 0, acc = 0, wrong = 0;
    double fs = 20e6, tau = 6.283185307179586 * 2.5e6;
    for (int i = 0; i < iters; i++) {
        double x;
        const int kind = i % 3;
        if (kind == 0) {
            x = (u01(st) * 2 - 1) * exp10(u01(st) * 25.7 - 17.0);
        } else if (kind == 1) {
            x = floor(u01(st) * 683565275.0) * 0.78539816339744830962;
            const int d = (int)(sm(st) % 7) - 3;
            x = __longlong_as_double(__double_as_longlong(x) + d);  // d ulps up or down (x > 0)
            if (sm(st) & 1) x = -x;
        } else {
            if ((i & 0x3FF) == 2) {
                fs = exp10(3 + 6 * u01(st));
                tau = 6.283185307179586 * (u01(st) * 2 - 1) * fs;
            }
            x = __dmul_rn(tau, __dmul_rn((double)(sm(st) % (uint64_t)(6.2 * fs + 1)), 1.0 / fs));
        }
        const double ax = fabs(x);
        if (!(ax < 536870912.0) || (ax < 8.673617379884035e-19 && x != 0)) continue;
        total++;
        float sn, cs;
        if (!sincos_narrow(x, sn, cs)) continue;
        acc++;
        double s, c;
        go_sincos(x, s, c);
        if (__float_as_uint((float)s) != __float_as_uint(sn) || __float_as_uint((float)c) != __float_as_uint(cs)) wrong++;
    }
    atomicAdd(cnt, total);
    atomicAdd(cnt + 1, acc);
    atomicAdd(cnt + 2, wrong);
}

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 36;
    unsigned long long *cnt, h[3];
    CK(hipMalloc(&cnt, 24));
    CK(hipMemset(cnt, 0, 24));
    const int iters = 1 << 12;
    const unsigned long long threads_per_launch = 1ull << 24;  // 2^36 phases = 2^24 threads x 2^12, launch by launch
    const unsigned long long launches = (1ull << lg) / (threads_per_launch * iters);
    for (unsigned long long r = 0; r < (launches ? launches : 1); r++) {
        hipLaunchKernelGGL(k_check, dim3((unsigned)(threads_per_launch / 256)), dim3(256), 0, 0, cnt, 0x9E3779B97F4A7C15ull * (r + 1), iters);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h, cnt, 24, hipMemcpyDeviceToHost));
    printf("%llu phases in the straight path's range: %llu accepted (%.3e refused), %llu of the accepted differ from complex64(go_sincos)\n", h[0], h[1],
           (double)(h[0] - h[1]) / (double)h[0], h[2]);
    return h[2] != 0;
}

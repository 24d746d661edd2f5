// copy_rate.hip -- what a 16 B/sample streaming map can reach on this chip when its bytes really come from HBM
// and go to it: 2^24 c64 samples (128 MiB in, 128 MiB out) per launch over a ROTATION of buffer pairs (2 GiB in all,
// eight times the 256 MB memory-side cache), against the same launch over ONE pair (which that cache largely
// holds from launch to launch).  Variants: bytes per lane per trip, grid, non-temporal loads / stores, in place.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, int NT> __global__ __launch_bounds__(256) void k_copy(const float4 *in, float4 *out, size_t nvec) {
    const size_t tile = (size_t)256 * U;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < nvec; t0 += (size_t)gridDim.x * tile) {
        float4 a[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * 256 + threadIdx.x;
            if (i < nvec) {
                if constexpr (NT & 1) {
                    typedef float v4 __attribute__((ext_vector_type(4)));
                    v4 t = __builtin_nontemporal_load((const v4 *)(in + i));
                    a[u] = make_float4(t.x, t.y, t.z, t.w);
                } else a[u] = in[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * 256 + threadIdx.x;
            if (i < nvec) {
                float4 v = make_float4(a[u].x * 0.5f, a[u].y * 0.5f, a[u].z * 0.5f, a[u].w * 0.5f);
                if constexpr (NT & 2) {
                    typedef float v4 __attribute__((ext_vector_type(4)));
                    v4 t = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(t, (v4 *)(out + i));
                } else out[i] = v;
            }
        }
    }
}

constexpr int kPairs = 8;
template <int U, int NT> static void run(float4 **in, float4 **out, size_t nvec, unsigned cap, bool inplace, const char *what) {
    const size_t tile = (size_t)256 * U;
    size_t blocks = (nvec + tile - 1) / tile;
    if (cap && blocks > cap) blocks = cap;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float t[2];
    for (int rot = 0; rot < 2; rot++) {
        const int reps = 64;
        for (int r = 0; r < reps / 4; r++) {
            const int b = rot ? r % kPairs : 0;
            hipLaunchKernelGGL((k_copy<U, NT>), dim3((unsigned)blocks), dim3(256), 0, 0, in[b], inplace ? in[b] : out[b], nvec);
        }
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) {
            const int b = rot ? r % kPairs : 0;
            hipLaunchKernelGGL((k_copy<U, NT>), dim3((unsigned)blocks), dim3(256), 0, 0, in[b], inplace ? in[b] : out[b], nvec);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t[rot], e0, e1));
        t[rot] = t[rot] / reps * 1e3f;
    }
    const double bytes = (double)nvec * 32;
    printf("%-44s U %d grid %6zu: one pair %.1f us (%.2f TB/s)   rotation %.1f us (%.2f TB/s)\n", what, U, blocks, t[0], bytes / t[0] * 1e-6, t[1],
           bytes / t[1] * 1e-6);
}

int main() {
    const size_t n = 1 << 24, nvec = n / 2;
    float4 *in[kPairs], *out[kPairs];
    for (int b = 0; b < kPairs; b++) {
        CK(hipMalloc(&in[b], n * 8));
        CK(hipMalloc(&out[b], n * 8));
        CK(hipMemset(in[b], 0, n * 8));
        CK(hipMemset(out[b], 0, n * 8));
    }
    for (int r = 0; r < 300; r++) hipLaunchKernelGGL((k_copy<4, 0>), dim3(8192), dim3(256), 0, 0, in[0], out[0], nvec);  // clocks up
    CK(hipDeviceSynchronize());
    run<4, 0>(in, out, nvec, 0, false, "plain");
    run<1, 0>(in, out, nvec, 0, false, "plain");
    run<2, 0>(in, out, nvec, 0, false, "plain");
    run<8, 0>(in, out, nvec, 0, false, "plain");
    run<4, 0>(in, out, nvec, 2048, false, "plain, 2048 workgroups");
    run<4, 0>(in, out, nvec, 1024, false, "plain, 1024 workgroups");
    run<4, 0>(in, out, nvec, 512, false, "plain, 512 workgroups");
    run<4, 1>(in, out, nvec, 0, false, "nt loads");
    run<4, 2>(in, out, nvec, 0, false, "nt stores");
    run<4, 3>(in, out, nvec, 0, false, "nt loads and stores");
    run<4, 3>(in, out, nvec, 2048, false, "nt loads and stores, 2048 workgroups");
    run<2, 3>(in, out, nvec, 0, false, "nt loads and stores");
    run<4, 0>(in, out, nvec, 0, true, "in place, plain");
    run<4, 3>(in, out, nvec, 0, true, "in place, nt loads and stores");
    run<4, 2>(in, out, nvec, 0, true, "in place, nt stores");
    {   // hipMemcpyAsync device to device, the runtime's own copy
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int rot = 0; rot < 2; rot++) {
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < 64; r++) CK(hipMemcpyAsync(out[rot ? r % kPairs : 0], in[rot ? r % kPairs : 0], n * 8, hipMemcpyDeviceToDevice, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("hipMemcpyAsync d2d %s: %.1f us\n", rot ? "rotation" : "one pair", ms / 64 * 1e3f);
        }
    }
    return 0;
}

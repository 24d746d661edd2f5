// exchange_ab.hip -- the A/B north_star names and no round had run: a transform pass's exchange between the lanes of
// ONE wave through LDS (what hz_fftv.h does) against the same permutation through DPP lane moves, beside the real
// radix-16 butterflies of the packed-math core.
//
// The unit of a wave-local FFT pass (N <= 1024: conv_blocks_shared_kernel, fft_plan_kernel16) is "sixteen points per
// lane through a radix-16 butterfly, then the 16 x 16 transposition (lane's low four bits <-> register index) that
// hands the next pass its points".  This program runs that unit in a loop, sixteen waves per CU on every CU (the block
// convolution's occupancy), three ways:
//   MODE 0  butterflies only (no exchange: the floor)
//   MODE 1  exchange through LDS: 16 ds_write_b64 + 16 ds_read_b64 per lane, padded, conflict-free
//   MODE 2  exchange through DPP: four block-swap stages; lane distances 8 and 4 as ONE v_mov_b32_dpp per dword
//           (row_shr / row_shl under a bank mask), distances 2 and 1 as a quad permutation and a select per dword
// and checks MODE 2 against MODE 1 bit for bit before timing (the permutation is the same one).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I go-sdr_amd/csrc tools/exchange_ab.hip -o tools/bin/exchange_ab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "hz_fftv.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using hz::fv::cf;

constexpr int kWaves = 16, kRowPitch = 16 * 17;  // a row of sixteen lanes: 16 x 16 elements, padded to 17 per line

// DPP controls (LLVM AMDGPU: quad_perm 0x00-0xFF, row_shl:n 0x100+n, row_shr:n 0x110+n)
template <int CTRL, int BANKS> __device__ __forceinline__ float dpp_into(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xF, BANKS, false));
}
template <int CTRL> __device__ __forceinline__ float dpp_all(float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, 0xF, 0xF, true));
}

// stage S of the in-row transposition on one dword plane: for r with (r & S) == 0, lanes with bit S set exchange
// a[l] <-> b[l - S]  (a = x[r], b = x[r + S])
template <int S> __device__ __forceinline__ void swap_stage(float *x, int lane) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (r & S) continue;
        const float a = x[r], b = x[r + S];
        if constexpr (S >= 4) {
            constexpr int set = S == 4 ? 0xA : 0xC, clear = S == 4 ? 0x5 : 0x3;  // banks of four lanes with bit S set / clear
            x[r] = dpp_into<0x110 + S, set>(a, b);      // a[l] = b[l - S] where bit S of l is set
            x[r + S] = dpp_into<0x100 + S, clear>(b, a);  // b[l] = a[l + S] where it is clear
        } else {
            // inside a quad: lane l takes lane l ^ S of the other register, where its bit S says so
            constexpr int perm = S == 1 ? 0xB1 : 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]
            const float bs = dpp_all<perm>(b), as = dpp_all<perm>(a);
            const bool hi = (lane & S) != 0;
            x[r] = hi ? bs : a;
            x[r + S] = hi ? b : as;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(64 * kWaves) void exchange_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, int iters) {
    __shared__ cf lds_all[kWaves * 4 * kRowPitch];
    const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63, row = wl >> 4, l = wl & 15;
    cf *lds = lds_all + (wave * 4 + row) * kRowPitch;
    cf v[16];
    const size_t base = ((size_t)blockIdx.x * 64 * kWaves + threadIdx.x) * 16;
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = hz::fv::from2(in[base + q]);
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        hz::fv::butterflies<16, false>(v);
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = v[q] * 0.25f;  // (keeps the values finite over thousands of trips)
        if constexpr (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 16; q++) lds[q * 17 + l] = v[q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = lds[l * 17 + q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        } else if constexpr (MODE == 2) {
            float re[16], im[16];
#pragma unroll
            for (int q = 0; q < 16; q++) re[q] = v[q].x, im[q] = v[q].y;
            swap_stage<8>(re, l), swap_stage<8>(im, l);
            swap_stage<4>(re, l), swap_stage<4>(im, l);
            swap_stage<2>(re, l), swap_stage<2>(im, l);
            swap_stage<1>(re, l), swap_stage<1>(im, l);
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = cf{re[q], im[q]};
        }
    }
#pragma unroll
    for (int q = 0; q < 16; q++) out[base + q] = hz::fv::to2(v[q]);
}

template <int MODE> static float run(const float2 *in, float2 *out, int grid, int iters, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(exchange_kernel<MODE>, dim3(grid), dim3(64 * kWaves), 0, 0, in, out, iters);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(exchange_kernel<MODE>, dim3(grid), dim3(64 * kWaves), 0, 0, in, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int grid = cus, iters = 2000;
    const size_t n = (size_t)grid * 64 * kWaves * 16;
    std::vector<float2> h(n);
    uint64_t s = 88172645463325252ull;
    for (auto &x : h) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        x.x = (float)(int)(s & 0xFFFF) / 65536.0f - 0.5f, x.y = (float)(int)((s >> 20) & 0xFFFF) / 65536.0f - 0.5f;
    }
    float2 *in, *o1, *o2;
    CK(hipMalloc(&in, n * 8));
    CK(hipMalloc(&o1, n * 8));
    CK(hipMalloc(&o2, n * 8));
    CK(hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice));
    // the two exchanges are the same permutation: three trips of each, bit for bit
    hipLaunchKernelGGL(exchange_kernel<1>, dim3(grid), dim3(64 * kWaves), 0, 0, in, o1, 3);
    hipLaunchKernelGGL(exchange_kernel<2>, dim3(grid), dim3(64 * kWaves), 0, 0, in, o2, 3);
    CK(hipDeviceSynchronize());
    std::vector<float2> a(n), b(n);
    CK(hipMemcpy(a.data(), o1, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), o2, n * 8, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < n; i++) diff += memcmp(&a[i], &b[i], 8) != 0;
    printf("DPP exchange against LDS exchange after three trips: %zu of %zu points differ\n", diff, n);
    if (diff) return 1;
    printf("%d CUs x %d waves, %d trips of (radix-16 butterflies on 16 points per lane + exchange); best of 5 launches\n", cus, kWaves, iters);
    const float t0 = run<0>(in, o1, grid, iters, 5), t1 = run<1>(in, o1, grid, iters, 5), t2 = run<2>(in, o1, grid, iters, 5);
    const double per = 1e3 / iters;  // ns per trip of a wave's slot
    printf("  butterflies alone            %8.1f us = %6.1f ns per trip\n", t0, t0 * per);
    printf("  + exchange through LDS       %8.1f us = %6.1f ns per trip  (+%.1f)\n", t1, t1 * per, (t1 - t0) * per);
    printf("  + exchange through DPP       %8.1f us = %6.1f ns per trip  (+%.1f)\n", t2, t2 * per, (t2 - t0) * per);
    return 0;
}

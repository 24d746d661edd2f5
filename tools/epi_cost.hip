// epi_cost.hip -- what ONE instruction of an epilogue costs beside the SIMD partner's matrix loop, by kind.
//
// hz_firmm2.h runs two waves per SIMD; while one is in its matrix loop the other works off its epilogue (landing,
// float64 plane combination, mixer, stores).  Round 5 cut that epilogue from ~520 to ~300 vector instructions and its
// duration beside a partner's loop stayed at ~7 us (tools/mfma_fir2.hip, "pass 1 epilog.") -- so the count of
// instructions is not what sets it.  This program puts the question to the hardware directly: wave A of every SIMD
// issues the kernel's loop pattern (4 x v_mfma_i32_32x32x32_i8 + 6 ds_read_b128 per step, or MFMAs alone, or nothing),
// wave B a straight run of ONE kind of instruction on independent registers; s_memtime around both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/epi_cost.hip -o tools/bin/epi_cost
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum Kind { K_FMA32, K_PKFMA32, K_FMA64, K_CVT_F64_I32, K_CVT_F32_F64, K_INT, K_DSREAD, K_MAD64, K_STORE, K_FMA64_CHAIN, K_PKFMA32_CHAIN, K_DSWRITE, K_GLOAD, K_STORE1, K_NKINDS };
static const char *kKindName[K_NKINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_cvt_f64_i32", "v_cvt_f32_f64", "v_lshl_add_u32",
                                          "ds_read_b128 (waited for in fours)", "v_mad_u64_u32", "global_store_dwordx4 nt", "v_fma_f64, dependent chain",
                                          "v_pk_fma_f32, dependent chain", "ds_write_b128", "global_load_dwordx4 nt (waited for in fours)", "global_store_dword"};

// A: 0 idle, 1 MFMAs back to back, 2 MFMAs + six ds_read_b128 per four (the kernel's loop)
// PRIO: s_setprio of wave B (0: as launched, like A); A's loop as straight-line code of 64 MFMAs per trip (no branch
// for B to slip into) when AMODE == 3
template <int KIND, int AMODE, int PRIO = 0>
__global__ __launch_bounds__(512) void epi(int a_trips, int b_reps, float *fsink, int *sink, unsigned long long *cyc, float *gout, int seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), l = threadIdx.x & 63;
    for (int q = threadIdx.x; q < 65536 / 16; q += 512) reinterpret_cast<v4i *>(lds)[q] = v4i{q * seed, q, seed, 7};
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (AMODE == 0) {
            __builtin_amdgcn_s_sleep(1);
        } else {
            v4i a[2] = {v4i{seed + l * 0x01020304, seed * 3, l * 77, seed ^ 0x5a5a5a5a}, v4i{seed, l, 3, 4}};
            v4i b[2] = {v4i{seed * 7 + l * 0x11213141, seed * 5, l * 91, seed ^ 0x3c3c3c3c}, v4i{l, seed, 9, 1}};
            v16i acc[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[j][q] = 0;
            const unsigned char *bp = lds + 16 * l + 4096 * wave;
            if (AMODE == 3) {
#pragma unroll 1
                for (int i = 0; i < a_trips; i += 16) {
#pragma unroll
                    for (int u = 0; u < 16; u++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[j >> 1], b[j & 1], acc[j], 0, 0, 0);
                }
            } else
#pragma unroll 1
            for (int i = 0; i < a_trips; i++) {
                if (AMODE == 2) {
                    a[0] ^= *reinterpret_cast<const v4i *>(bp + (i & 3) * 1024);
                    a[1] ^= *reinterpret_cast<const v4i *>(bp + 16384 + (i & 3) * 1024);
                    b[0] ^= *reinterpret_cast<const v4i *>(bp + 32768 + (i & 3) * 1024);
                    b[1] ^= *reinterpret_cast<const v4i *>(bp + 49152 + (i & 3) * 1024);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[j >> 1], b[j & 1], acc[j], 0, 0, 0);
            }
            int r = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) r += acc[j][0] + acc[j][15];
            if (r == 0x7fffffff) sink[0] = r;
        }
    } else {
        if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
        // 32 independent instructions of the kind per repetition
        float f[16];
        double d[8];
        int n[16];
        v2f p[8];
#pragma unroll
        for (int q = 0; q < 16; q++) f[q] = 1.0f + (float)(l + q) * 1e-3f, n[q] = l * 7 + q + seed;
#pragma unroll
        for (int q = 0; q < 8; q++) d[q] = 1.0 + (double)(l + q) * 1e-3, p[q] = v2f{f[q], f[q + 8]};
        const unsigned char *bp = lds + 16 * l;
        unsigned long long m64 = (unsigned long long)seed * 0x9E3779B97F4A7C15ull + l;
        v4f *gp = reinterpret_cast<v4f *>(gout) + (size_t)blockIdx.x * 512 + threadIdx.x;
        // (B's own stamps, in line: its set-up and the sums behind the loop are compiler-made instructions beside A too)
        uint64_t tb0, tb1;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tb0));
#pragma unroll 1
        for (int r = 0; r < b_reps; r++) {
#pragma unroll
            for (int q = 0; q < 32; q++) {
                if constexpr (KIND == K_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[q & 15]) : "v"(f[(q + 1) & 15]), "v"(f[(q + 2) & 15]));
                if constexpr (KIND == K_PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[q & 7]) : "v"(p[(q + 1) & 7]), "v"(p[(q + 2) & 7]));
                if constexpr (KIND == K_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[q & 7]) : "v"(d[(q + 1) & 7]), "v"(d[(q + 2) & 7]));
                if constexpr (KIND == K_CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[q & 7]) : "v"(n[q & 15]));
                if constexpr (KIND == K_CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[q & 15]) : "v"(d[q & 7]));
                if constexpr (KIND == K_INT) asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(n[q & 15]) : "v"(n[(q + 1) & 15]), "v"(n[(q + 2) & 15]));
                if constexpr (KIND == K_MAD64) asm volatile("v_mad_u64_u32 %0, s[6:7], %1, %2, %0" : "+v"(m64) : "v"(n[q & 15]), "v"(n[(q + 3) & 15]) : "s6", "s7");
                if constexpr (KIND == K_FMA64_CHAIN) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[0]) : "v"(d[1]), "v"(d[2]));
                if constexpr (KIND == K_PKFMA32_CHAIN) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(p[1]), "v"(p[2]));
                if constexpr (KIND == K_DSREAD) {
                    v4i t = *reinterpret_cast<const v4i *>(bp + (q & 31) * 1024);
                    asm volatile("" : "+v"(t));
                    n[q & 15] ^= t[0];
                    if ((q & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)");
                }
                if constexpr (KIND == K_DSWRITE) {
                    v4i t{n[q & 15], n[(q + 1) & 15], n[(q + 2) & 15], n[(q + 3) & 15]};
                    *reinterpret_cast<v4i *>(const_cast<unsigned char *>(bp) + (q & 31) * 1024) = t;
                    asm volatile("" ::: "memory");
                }
                if constexpr (KIND == K_GLOAD) {
                    v4i t = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(gp) + (size_t)((r * 32 + q) & 63) * 131072);
                    asm volatile("" : "+v"(t));
                    n[q & 15] ^= t[0];
                    if ((q & 3) == 3) asm volatile("s_waitcnt vmcnt(0)");
                }
                if constexpr (KIND == K_STORE1) {
                    if ((q & 7) == 0) __builtin_nontemporal_store(f[q & 15], reinterpret_cast<float *>(gp + (size_t)((r * 4 + (q >> 3)) & 63) * 131072));
                }
                if constexpr (KIND == K_STORE) {
                    if ((q & 7) == 0) __builtin_nontemporal_store(v4f{f[q & 15], f[(q + 1) & 15], f[2], f[3]}, gp + (size_t)((r * 4 + (q >> 3)) & 63) * 131072);
                }
            }
        }
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tb1));
        if (l == 0) cyc[2048 + blockIdx.x * 8 + wave] = tb1 - tb0, cyc[4096 + blockIdx.x * 8 + wave] = tb0 - t0;
        float fs = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q++) fs += f[q] + (float)n[q];
#pragma unroll
        for (int q = 0; q < 8; q++) fs += (float)d[q] + p[q].x + p[q].y;
        fs += (float)(m64 & 0xffff);
        if (fs == 123.456f) fsink[0] = fs;
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (l == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND, int AMODE, int PRIO = 0> static void run(int a_trips, int b_reps, float *fsink, int *sink, unsigned long long *cyc, float *gout) {
    CK(hipFuncSetAttribute((const void *)epi<KIND, AMODE, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL((epi<KIND, AMODE, PRIO>), dim3(256), dim3(512), 65536, 0, a_trips, b_reps, fsink, sink, cyc, gout, 777 + r);
    CK(hipDeviceSynchronize());
    unsigned long long c[8], cb[8], cs[8];
    CK(hipMemcpy(c, cyc + 8 * 100, sizeof c, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cb, cyc + 2048 + 8 * 100, sizeof cb, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cs, cyc + 4096 + 8 * 100, sizeof cs, hipMemcpyDeviceToHost));
    c[4] = cb[4];  // (the loop alone)
    const double per = (double)c[4] / ((double)b_reps * (KIND == K_STORE || KIND == K_STORE1 ? 4 : 32));
    printf("  %-36s beside %-34s%s: B %7.2f ticks per instruction (%9.0f in all), A %7.1f ticks per 4 MFMAs\n", kKindName[KIND],
           AMODE == 0 ? "an idle partner" : AMODE == 1 ? "MFMAs back to back" : AMODE == 2 ? "MFMAs + 6 ds_read_b128 per four" : "64 MFMAs per trip, straight-line",
           PRIO ? ", B at s_setprio 3" : "", per, (double)c[4], AMODE ? (double)c[0] / a_trips : 0.0);
    printf("      (B's set-up in front of its loop: %.0f ticks)\n", (double)cs[4]);
}

template <int KIND> static void kind(float *fsink, int *sink, unsigned long long *cyc, float *gout) {
    // B: 10 repetitions of 32 instructions = 320, about an epilogue; A: long enough to outlast it
    run<KIND, 0>(0, 10, fsink, sink, cyc, gout);
    run<KIND, 1>(400, 10, fsink, sink, cyc, gout);
    run<KIND, 2>(400, 10, fsink, sink, cyc, gout);
    run<KIND, 3>(400, 10, fsink, sink, cyc, gout);
    run<KIND, 1, 3>(400, 10, fsink, sink, cyc, gout);
    run<KIND, 3, 3>(400, 10, fsink, sink, cyc, gout);
    run<KIND, 3, 3>(400, 100, fsink, sink, cyc, gout);  // (B as long as A: what A loses to a whole loop's worth of partner instructions)
}

int main() {
    float *fsink, *gout;
    int *sink;
    unsigned long long *cyc;
    CK(hipMalloc(&fsink, 4));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&cyc, 8 * 8192));
    CK(hipMalloc(&gout, (size_t)64 * 131072 * 16 + (size_t)256 * 512 * 16));
    printf("ticks: s_memtime (shader clock; an MFMA 32x32x32 i8 is 32) -- an instruction of wave B (second wave of the SIMD) by kind, beside wave A\n");
    kind<K_FMA32>(fsink, sink, cyc, gout);
    kind<K_PKFMA32>(fsink, sink, cyc, gout);
    kind<K_PKFMA32_CHAIN>(fsink, sink, cyc, gout);
    kind<K_INT>(fsink, sink, cyc, gout);
    kind<K_FMA64>(fsink, sink, cyc, gout);
    kind<K_FMA64_CHAIN>(fsink, sink, cyc, gout);
    kind<K_CVT_F64_I32>(fsink, sink, cyc, gout);
    kind<K_CVT_F32_F64>(fsink, sink, cyc, gout);
    kind<K_MAD64>(fsink, sink, cyc, gout);
    kind<K_DSREAD>(fsink, sink, cyc, gout);
    kind<K_STORE>(fsink, sink, cyc, gout);
    kind<K_STORE1>(fsink, sink, cyc, gout);
    kind<K_DSWRITE>(fsink, sink, cyc, gout);
    kind<K_GLOAD>(fsink, sink, cyc, gout);
    return 0;
}

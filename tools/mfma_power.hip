// mfma_power.hip -- what the chip sustains of v_mfma_i32_32x32x32_i8 as a function of the OPERANDS' CONTENT and of the
// matrix pipe's DUTY CYCLE: every SIMD of every CU runs two waves that each issue bursts of MFMAs (four independent
// accumulators, operands re-read from a register ring so that consecutive MFMAs see different bits) separated by
// s_sleep.  Reports, per (data, duty): the shader clock the chip held (s_memtime cycles per 10 ns of s_memrealtime),
// ns per MFMA per SIMD, and the chip's int8 rate.  Round 6: the north-star kernel runs 33 % slower on random bytes
// than on zeros with the SAME instruction stream (tools/mfma_fir2.hip ZERO=1) -- the chip lowers its clock under
// toggling matrix operands; this tool measures that roofline without the kernel around it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// BURST MFMAs, then s_sleep(SLEEP) (64 SLEEP cycles), TRIPS times.  mode: 0 = all operands zero, 1 = A random / B zero,
// 2 = A zero / B random, 3 = both random
template <int BURST, int SLEEP>
__global__ __launch_bounds__(512) void k(unsigned long long *out, int trips, int mode, int *sink) {
    const unsigned tid = threadIdx.x + blockIdx.x * 512u;
    v4i a[4], b[4];
    for (int i = 0; i < 4; i++) {
        unsigned h = (tid * 2654435761u) ^ (0x9E3779B9u * (i + 1));
        auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (int)h; };
        a[i] = (mode & 1) ? v4i{nx(), nx(), nx(), nx()} : v4i{0, 0, 0, 0};
        b[i] = (mode & 2) ? v4i{nx(), nx(), nx(), nx()} : v4i{0, 0, 0, 0};
    }
    v16i c[4];
    for (int i = 0; i < 4; i++)
        for (int q = 0; q < 16; q++) c[i][q] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int m = 0; m < BURST; m++) c[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(m + (m >> 2)) & 3], b[(m >> 1) & 3], c[m & 3], 0, 0, 0);
        if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int i = 0; i < 4; i++)
        for (int q = 0; q < 16; q++) s += c[i][q];
    if (s == 0x12345678) *sink = s;
    if ((threadIdx.x & 63) == 0) {
        out[2 * (tid >> 6)] = t1 - t0;
        out[2 * (tid >> 6) + 1] = r1 - r0;
    }
}

// The same with the operands read from LDS as the FIR kernel reads them: READS ds_read_b128 per four MFMAs (4 = one per
// MFMA, the kernel's rate: two A and two B fragments per four MFMAs; 2 = half of it), random bytes in 64 KB of LDS,
// two steps ahead of their use.
template <int BURST, int SLEEP, int READS, int GL = 0, int F64 = 0>
__global__ __launch_bounds__(512) void kl(unsigned long long *out, int trips, int mode, int *sink, const v4i *__restrict__ big = nullptr, size_t big_n = 0) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const unsigned tid = threadIdx.x + blockIdx.x * 512u;
    {
        unsigned h = tid * 2654435761u + 12345u;
        for (int i = threadIdx.x; i < 65536 / 4; i += 512) {
            h ^= h << 13, h ^= h >> 17, h ^= h << 5;
            reinterpret_cast<unsigned *>(lds)[i] = mode == 3 ? h : 0u;
        }
    }
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *base = lds + wave * 8192 + lane * 16;  // (conflict-free: 16 consecutive bytes per lane)
    v4i a[4], b[4];
    for (int i = 0; i < 4; i++) a[i] = *reinterpret_cast<const v4i *>(base + 1024 * i), b[i] = *reinterpret_cast<const v4i *>(base + 1024 * (i + 4));
    v16i c[4];
    for (int i = 0; i < 4; i++)
        for (int q = 0; q < 16; q++) c[i][q] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned off = 0;
    [[maybe_unused]] v4i gx[GL > 0 ? GL : 1];
    [[maybe_unused]] size_t gpos = ((size_t)blockIdx.x * 8 + wave) * 65536 + lane;
    [[maybe_unused]] double fd[8] = {1.0, 1.1, 1.2, 1.3, 1.4, 1.5, 1.6, 1.7};
    [[maybe_unused]] float ff[8] = {1.0f, 1.1f, 1.2f, 1.3f, 1.4f, 1.5f, 1.6f, 1.7f};
    [[maybe_unused]] int fi[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    [[maybe_unused]] long long fl[8] = {1, 2, 3, 4, 5, 6, 7, 8};
#pragma unroll 1
    for (int t = 0; t < trips; t++) {
        if constexpr (GL > 0) {  // the next "pass image": GL x 1 KB per wave from a buffer far larger than the caches, non-temporal
#pragma unroll
            for (int u = 0; u < GL; u++) gx[u] = __builtin_nontemporal_load(big + (gpos + 64 * u) % big_n);
            gpos += 64 * GL * 2048 / 8;
        }
#pragma unroll
        for (int m = 0; m < BURST; m += 4) {
            // four MFMAs (2 A x 2 B fragments), READS fragments re-read for the group after next
            c[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b[0], c[0], 0, 0, 0);
            if constexpr (READS >= 1) a[2] = *reinterpret_cast<const v4i *>(base + ((off + 0) & 7168));
            c[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b[1], c[1], 0, 0, 0);
            if constexpr (READS >= 2) b[2] = *reinterpret_cast<const v4i *>(base + ((off + 1024) & 7168));
            c[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1], b[0], c[2], 0, 0, 0);
            if constexpr (READS >= 3) a[3] = *reinterpret_cast<const v4i *>(base + ((off + 2048) & 7168));
            c[3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1], b[1], c[3], 0, 0, 0);
            if constexpr (READS >= 4) b[3] = *reinterpret_cast<const v4i *>(base + ((off + 3072) & 7168));
            off += 4096;
            // rotate: the fragments read two groups ago become the operands
            v4i t0v = a[0], t1v = a[1], t2v = b[0], t3v = b[1];
            a[0] = a[2], a[1] = a[3], b[0] = b[2], b[1] = b[3];
            a[2] = t0v, a[3] = t1v, b[2] = t2v, b[3] = t3v;
        }
        if constexpr (GL > 0) {  // land it: GL ds_write_b128 into the wave's region (what the next trips read)
#pragma unroll
            for (int u = 0; u < GL; u++) *reinterpret_cast<v4i *>(const_cast<unsigned char *>(base) + 1024 * (u & 7)) = mode == 3 ? gx[u] : v4i{0, 0, 0, 0};
        }
        if constexpr (F64 > 0) {  // an epilogue's vector work: F64 % 1000 instructions per lane of kind F64 / 1000 on the accumulators' (random) bits
            constexpr int CNT = F64 % 1000, KIND = F64 / 1000;
#pragma unroll
            for (int u = 0; u < CNT; u++) {
                const int ci = c[u & 3][u & 15];
                if constexpr (KIND == 0) {  // v_fma_f64 (+ the v_cvt_f64_i32 that feeds it: 2 instructions per count)
                    fd[u & 7] = __builtin_fma(fd[u & 7], 1.0000001, (double)ci);
                } else if constexpr (KIND == 1) {  // v_fma_f32 + v_cvt_f32_i32
                    ff[u & 7] = __builtin_fmaf(ff[u & 7], 1.0001f, (float)ci);
                } else if constexpr (KIND == 2) {  // v_cvt_f64_i32 alone (summed by xor of the low words: one int op)
                    const double d = (double)ci;
                    fi[u & 7] ^= (int)__double2loint(d) ^ __double2hiint(d);
                } else if constexpr (KIND == 3) {  // v_mad_i64_i32
                    fl[u & 7] = (long long)ci * (long long)(0x10001 + u) + fl[u & 7];
                } else if constexpr (KIND == 4) {  // v_lshl_add_u32 / v_add3
                    fi[u & 7] = (ci << 8) + fi[u & 7];
                } else if constexpr (KIND == 5) {  // v_fma_f64 on values that stay in registers (no conversion)
                    fd[u & 7] = __builtin_fma(fd[u & 7], 1.0000001, fd[(u + 1) & 7]);
                }
            }
        }
        if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int i = 0; i < 4; i++)
        for (int q = 0; q < 16; q++) s += c[i][q];
    if constexpr (F64 > 0) {
        for (int u = 0; u < 8; u++) s += (int)fd[u] + (int)ff[u] + fi[u] + (int)fl[u];
    }
    if (s == 0x12345678) *sink = s;
    if ((threadIdx.x & 63) == 0) {
        out[2 * (tid >> 6)] = t1 - t0;
        out[2 * (tid >> 6) + 1] = r1 - r0;
    }
}

// a "pass": 272 MFMAs with their LDS reads, then the pass's other work; both waves of a SIMD run it
template <int GL, int F64> static void run_pass(unsigned long long *dout, int *sink, const v4i *big, size_t big_n, double target_us) {
    const int grid = 256, waves = grid * 8, BURST = 272;
    const int trips = (int)(target_us * 1500.0 / (2.0 * BURST * 32.0)) + 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((kl<BURST, 0, 4, GL, F64>), dim3(grid), dim3(512), 0, 0, dout, trips, 3, sink, big, big_n);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(2 * waves);
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int w = 0; w < waves; w++) ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] / 10.0);
    std::sort(ghz.begin(), ghz.end());
    const double mfma_per_simd = 2.0 * (double)trips * BURST;
    const double ns_per = ms * 1e6 / mfma_per_simd;
    static const char *kinds[6] = {"v_cvt_f64_i32 + v_fma_f64", "v_cvt_f32_i32 + v_fma_f32", "v_cvt_f64_i32 + 2 v_xor", "v_mad_i64_i32", "v_lshl_add_u32", "v_fma_f64 (registers)"};
    printf("  passes of 272 MFMAs (random bytes, 1 LDS read per MFMA) + %2d KB per wave and pass from HBM (%.2f TB/s) + %3d x %-26s per lane: %7.1f us  clock %.3f GHz  %.2f ns per MFMA per SIMD  pipe duty %.2f  chip %.2f Pop/s\n",
           GL, (double)GL * 1024.0 * 2048.0 * trips / (ms * 1e-3) * 1e-12, F64 % 1000, kinds[F64 / 1000], ms * 1e3, ghz[waves / 2], ns_per, mfma_per_simd * 32.0 / (ghz[waves / 2] * 1e9 * ms * 1e-3), 1024.0 * 65536.0 / ns_per * 1e-6);
}

template <int BURST, int SLEEP, int READS> static void run_lds(unsigned long long *dout, int *sink, int mode, double target_us) {
    const int grid = 256, waves = grid * 8;
    const double cyc_per_trip = 2.0 * BURST * 32.0 > BURST * 32.0 + 64.0 * SLEEP ? 2.0 * BURST * 32.0 : BURST * 32.0 + 64.0 * SLEEP;
    const int trips = (int)(target_us * 1500.0 / cyc_per_trip) + 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((kl<BURST, SLEEP, READS>), dim3(grid), dim3(512), 0, 0, dout, trips, mode, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(2 * waves);
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int w = 0; w < waves; w++) ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] / 10.0);
    std::sort(ghz.begin(), ghz.end());
    const double mfma_per_simd = 2.0 * (double)trips * BURST;
    const double ns_per = ms * 1e6 / mfma_per_simd;
    printf("  operands from LDS, %d ds_read_b128 per 4 MFMAs, %s, burst %3d sleep %3d: %7.1f us  clock %.3f GHz  %.2f ns per MFMA per SIMD  pipe duty %.2f  chip %.2f Pop/s\n", READS,
           mode == 3 ? "random bytes" : "zeros       ", BURST, SLEEP, ms * 1e3, ghz[waves / 2], ns_per, mfma_per_simd * 32.0 / (ghz[waves / 2] * 1e9 * ms * 1e-3), 1024.0 * 65536.0 / ns_per * 1e-6);
}

// v_mfma_i32_16x16x64_i8 (half the multiply-adds per instruction, a quarter of the accumulators): what does THIS shape
// sustain on random operands?
template <int BURST, int SLEEP>
__global__ __launch_bounds__(512) void k16(unsigned long long *out, int trips, int mode, int *sink) {
    const unsigned tid = threadIdx.x + blockIdx.x * 512u;
    v4i a[4], b[4];
    for (int i = 0; i < 4; i++) {
        unsigned h = (tid * 2654435761u) ^ (0x9E3779B9u * (i + 1));
        auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (int)h; };
        a[i] = (mode & 1) ? v4i{nx(), nx(), nx(), nx()} : v4i{0, 0, 0, 0};
        b[i] = (mode & 2) ? v4i{nx(), nx(), nx(), nx()} : v4i{0, 0, 0, 0};
    }
    v4i c[8];
    for (int i = 0; i < 8; i++) c[i] = v4i{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int m = 0; m < BURST; m++) c[m & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(m + (m >> 2)) & 3], b[(m >> 1) & 3], c[m & 7], 0, 0, 0);
        if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int i = 0; i < 8; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 0x12345678) *sink = s;
    if ((threadIdx.x & 63) == 0) {
        out[2 * (tid >> 6)] = t1 - t0;
        out[2 * (tid >> 6) + 1] = r1 - r0;
    }
}
// ... and with its operands read from LDS in the blocking a FIR pass would use: four A fragments (the four digit planes of
// eight outputs) x four B fragments (four column blocks of sixteen tiles) = sixteen accumulators of four registers,
// eight ds_read_b128 per sixteen MFMAs -- the same LDS bytes per multiply-add as the 32x32x32 loop's two by two.
template <int STEPS, int SLEEP>
__global__ __launch_bounds__(512) void kl16(unsigned long long *out, int trips, int mode, int *sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const unsigned tid = threadIdx.x + blockIdx.x * 512u;
    {
        unsigned h = tid * 2654435761u + 12345u;
        for (int i = threadIdx.x; i < 65536 / 4; i += 512) {
            h ^= h << 13, h ^= h >> 17, h ^= h << 5;
            reinterpret_cast<unsigned *>(lds)[i] = mode == 3 ? h : 0u;
        }
    }
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *base = lds + wave * 8192 + lane * 16;
    v4i a[2][4], b[2][4];
    for (int i = 0; i < 4; i++) a[0][i] = *reinterpret_cast<const v4i *>(base + 1024 * i), b[0][i] = *reinterpret_cast<const v4i *>(base + 1024 * (i + 4));
    v4i c[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) c[i][j] = v4i{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned off = 0;
#pragma unroll 1
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const int cur = s & 1, nxt = cur ^ 1;
#pragma unroll
            for (int i = 0; i < 4; i++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    c[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[cur][i], b[cur][j], c[i][j], 0, 0, 0);
                    // (the next step's fragments, one read per two MFMAs)
                    if (j == 0) a[nxt][i] = *reinterpret_cast<const v4i *>(base + ((off + 1024 * i) & 7168));
                    if (j == 2) b[nxt][i] = *reinterpret_cast<const v4i *>(base + ((off + 1024 * (i + 4)) & 7168));
                }
            }
            off += 3072;
        }
        if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int sacc = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) sacc += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
    if (sacc == 0x12345678) *sink = sacc;
    if ((threadIdx.x & 63) == 0) {
        out[2 * (tid >> 6)] = t1 - t0;
        out[2 * (tid >> 6) + 1] = r1 - r0;
    }
}
template <int STEPS, int SLEEP> static void run_lds16(unsigned long long *dout, int *sink, int mode, double target_us) {
    const int grid = 256, waves = grid * 8;
    const int trips = (int)(target_us * 1500.0 / (2.0 * STEPS * 16 * 16.0)) + 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((kl16<STEPS, SLEEP>), dim3(grid), dim3(512), 0, 0, dout, trips, mode, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(2 * waves);
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int w = 0; w < waves; w++) ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] / 10.0);
    std::sort(ghz.begin(), ghz.end());
    const double mfma_per_simd = 2.0 * (double)trips * STEPS * 16;
    const double ns_per = ms * 1e6 / mfma_per_simd;
    printf("  v_mfma_i32_16x16x64_i8, 4 x 4 fragments from LDS (8 reads per 16 MFMAs), %s, sleep %2d: %7.1f us  clock %.3f GHz  %.2f ns per MFMA per SIMD (%.1f cycles)  chip %.2f Pop/s\n",
           mode == 3 ? "random bytes" : "zeros       ", SLEEP, ms * 1e3, ghz[waves / 2], ns_per, ns_per * ghz[waves / 2], 1024.0 * 32768.0 / ns_per * 1e-6);
}

template <int BURST, int SLEEP> static void run16(unsigned long long *dout, int *sink, int mode, double target_us) {
    const int grid = 256, waves = grid * 8;
    const int trips = (int)(target_us * 1500.0 / (2.0 * BURST * 16.0)) + 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k16<BURST, SLEEP>), dim3(grid), dim3(512), 0, 0, dout, trips, mode, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(2 * waves);
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int w = 0; w < waves; w++) ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] / 10.0);
    std::sort(ghz.begin(), ghz.end());
    const double mfma_per_simd = 2.0 * (double)trips * BURST;
    const double ns_per = ms * 1e6 / mfma_per_simd;
    printf("  v_mfma_i32_16x16x64_i8, %s, burst %3d sleep %3d: %7.1f us  clock %.3f GHz  %.2f ns per MFMA per SIMD (%.1f cycles)  chip %.2f Pop/s\n",
           mode == 3 ? "A random, B random" : "zeros             ", BURST, SLEEP, ms * 1e3, ghz[waves / 2], ns_per, ns_per * ghz[waves / 2], 1024.0 * 32768.0 / ns_per * 1e-6);
}

template <int BURST, int SLEEP> static void run(unsigned long long *dout, int *sink, int mode, double target_us) {
    const int grid = 256, waves = grid * 8;
    // trips for ~target_us: a burst is BURST x 32 cycles per wave, two waves share a SIMD
    const double cyc_per_trip = 2.0 * BURST * 32.0 > BURST * 32.0 + 64.0 * SLEEP ? 2.0 * BURST * 32.0 : BURST * 32.0 + 64.0 * SLEEP;
    const int trips = (int)(target_us * 1500.0 / cyc_per_trip) + 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<BURST, SLEEP>), dim3(grid), dim3(512), 0, 0, dout, trips, mode, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(2 * waves);
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int w = 0; w < waves; w++) ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] / 10.0);
    std::sort(ghz.begin(), ghz.end());
    const double mfma_per_simd = 2.0 * (double)trips * BURST;
    const double ns_per = ms * 1e6 / mfma_per_simd;
    const double duty = mfma_per_simd * 32.0 / (ghz[waves / 2] * 1e9 * ms * 1e-3);
    static const char *names[4] = {"A zero,   B zero  ", "A random, B zero  ", "A zero,   B random", "A random, B random"};
    printf("  %s  burst %3d sleep %3d: %7.1f us  clock %.3f GHz (min %.3f max %.3f)  %.2f ns per MFMA per SIMD  pipe duty %.2f  chip %.2f Pop/s\n", names[mode], BURST, SLEEP, ms * 1e3,
           ghz[waves / 2], ghz[0], ghz.back(), ns_per, duty, 1024.0 * 65536.0 / ns_per * 1e-6);
}

int main(int argc, char **argv) {
    const double us = argc > 1 ? atof(argv[1]) : 400.0;
    unsigned long long *dout;
    int *sink;
    CK(hipMalloc(&dout, 2 * 2048 * 8));
    CK(hipMalloc(&sink, 4));
    printf("v_mfma_i32_32x32x32_i8 on every SIMD of 256 CUs, two waves per SIMD, launches of ~%.0f us; Pop/s = 2 x 32 x 32 x 32 per MFMA (dense peak 5.0 at 2.4 GHz)\n", us);
    for (int mode = 0; mode < 4; mode++) {
        run<16, 0>(dout, sink, mode, us);   // back to back: two waves keep the pipe full
        run<16, 4>(dout, sink, mode, us);   // 512 MFMA cycles + 256 asleep per wave: still full with two waves
        run<16, 8>(dout, sink, mode, us);   // 512 + 512: the pipe just full
        run<16, 12>(dout, sink, mode, us);  // duty 2 x 512 / (512 + 768) = 0.80
        run<16, 16>(dout, sink, mode, us);  // 0.67
        run<16, 24>(dout, sink, mode, us);  // 0.50
    }
    printf("operands read from LDS between the MFMAs (the FIR kernel: 4 reads of 1 KB per 4 MFMAs):\n");
    for (int mode = 0; mode <= 3; mode += 3) {
        run_lds<16, 0, 0>(dout, sink, mode, us);
        run_lds<16, 0, 2>(dout, sink, mode, us);
        run_lds<16, 0, 4>(dout, sink, mode, us);
        run_lds<16, 12, 0>(dout, sink, mode, us);
        run_lds<16, 12, 2>(dout, sink, mode, us);
        run_lds<16, 12, 4>(dout, sink, mode, us);
    }
    {
        printf("the FIR kernel's pass without its filter (what else in a pass draws power):\n");
        v4i *big;
        const size_t big_n = ((size_t)1 << 30) / 16;  // 1 GiB
        CK(hipMalloc(&big, big_n * 16));
        {   // random bytes (a constant fill makes the landed operands quiet: 2.39 GHz instead of 1.73)
            std::vector<unsigned> hb(big_n * 4);
            unsigned hh = 0x12345u;
            for (auto &w : hb) { hh ^= hh << 13; hh ^= hh >> 17; hh ^= hh << 5; w = hh; }
            CK(hipMemcpy(big, hb.data(), big_n * 16, hipMemcpyHostToDevice));
        }
        run_pass<0, 0>(dout, sink, big, big_n, us);
        run_pass<10, 0>(dout, sink, big, big_n, us);
        run_pass<0, 48>(dout, sink, big, big_n, us);        // the FIR kernel's 48 conversions + 48 float64 fmas per pass and lane
        run_pass<0, 128>(dout, sink, big, big_n, us);
        run_pass<0, 1128>(dout, sink, big, big_n, us);      // the same count in float32
        run_pass<0, 2128>(dout, sink, big, big_n, us);      // conversions to float64 alone
        run_pass<0, 3128>(dout, sink, big, big_n, us);      // 64-bit integer multiply-adds
        run_pass<0, 4128>(dout, sink, big, big_n, us);      // 32-bit integer shift-adds
        run_pass<0, 5128>(dout, sink, big, big_n, us);      // float64 fmas without conversions
        run_pass<0, 0>(dout, sink, big, big_n, us);
        CK(hipFree(big));
    }
    printf("the other int8 shape:\n");
    run16<32, 0>(dout, sink, 0, us);
    run16<32, 0>(dout, sink, 3, us);
    run16<32, 8>(dout, sink, 3, us);
    run<16, 0>(dout, sink, 3, us);
    for (int rep = 0; rep < 2; rep++) {
        run_lds16<4, 0>(dout, sink, 3, us);
        run_lds<16, 0, 4>(dout, sink, 3, us);
        run_lds16<4, 4>(dout, sink, 3, us);
        run_lds<16, 4, 4>(dout, sink, 3, us);
    }
    run_lds16<4, 0>(dout, sink, 0, us);
    printf("long launches (4 ms), both operands random:\n");
    run<16, 0>(dout, sink, 3, 4000.0);
    run<16, 12>(dout, sink, 3, 4000.0);
    run<16, 24>(dout, sink, 3, 4000.0);
    return 0;
}

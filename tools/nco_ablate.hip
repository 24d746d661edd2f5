// nco_ablate.hip -- times the Shift map (csrc/hz_nco.hip: nco_shift_vec_kernel, BASELINE config 2) with its parts
// exchanged or switched off: which Sincos, with or without the loads / the stores, how many 16-byte vectors a lane
// keeps in flight.  One clock run (ts = k / fs), 2^24 samples in place; the values are not checked here.
//   SC 0 = sincos_narrow with go_sincos inline behind it, 1 = go_sincos (math.Sincos operation for operation),
//      2 = sincos_late unchecked, 3 = no Sincos (factor from the clock's bits), 4 = check only, no fallback code
//   MEM 0 = loads and stores, 1 = no loads, 2 = no stores, 3 = neither
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "hz_device.h"
#include "hz_nco.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using namespace hz;

// the narrowing check with go_sincos inline behind it (the form that lost: see shift_exact_kernel's comment)
__device__ __forceinline__ void go_sincos_f32(double x, float &sn, float &cs) {
    if (!sincos_narrow(x, sn, cs) || !(fabs(x) < 536870912.0)) {
        double s, c;
        go_sincos(x, s, c);
        sn = (float)s, cs = (float)c;
    }
}

template <int SC> __device__ __forceinline__ float2 rot(float2 v, double ts, double tau_shift) {
    const double ph = __dmul_rn(tau_shift, ts);
    float s, c;
    if constexpr (SC == 0) {
        go_sincos_f32(ph, s, c);
    } else if constexpr (SC == 1) {
        double sd, cd;
        go_sincos(ph, sd, cd);
        s = (float)sd, c = (float)cd;
    } else if constexpr (SC == 2) {
        double sd, cd;
        sincos_late(ph, sd, cd);
        s = (float)sd, c = (float)cd;
    } else if constexpr (SC == 4) {
        if (!sincos_narrow(ph, s, c)) s = 0.f;
    } else {
        s = (float)ph, c = __int_as_float(__float_as_int(s) ^ 0x1234);
    }
    return go_cmul(v, make_float2(c, s));
}

template <int SC, int MEM, int U, int TPB>
__global__ __launch_bounds__(TPB) void k_shift(float4 *buf, size_t nvec, double tau_shift, NcoSegs sg) {
    const size_t tile = (size_t)TPB * U;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < nvec; t0 += (size_t)gridDim.x * tile) {
        const uint64_t j_lo = 2 * t0;
        const NcoWin w = nco_window(sg, j_lo, j_lo + 2 * tile - 1);
        float4 a[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * TPB + threadIdx.x;
            if constexpr ((MEM & 1) == 0) { if (i < nvec) a[u] = buf[i]; }
            else a[u] = make_float4((float)i, 1.f, 2.f, (float)u);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * TPB + threadIdx.x;
            if (i < nvec) {
                const uint64_t j = 2 * i;
                float2 l = rot<SC>(make_float2(a[u].x, a[u].y), nco_ts(sg, w, j), tau_shift);
                float2 h = rot<SC>(make_float2(a[u].z, a[u].w), nco_ts(sg, w, j + 1), tau_shift);
                if constexpr ((MEM & 2) == 0) buf[i] = make_float4(l.x, l.y, h.x, h.y);
                else if (l.x == 1.2345f && h.y == 5.4321f) buf[i] = make_float4(l.x, l.y, h.x, h.y);
            }
        }
    }
}


// ---- the second cut: one clock run per tile as the straight path, the factor's quadrant applied in float32,
// the lanes the narrowing check cannot decide redone at the tile's end by ONE rolled copy of go_sincos
__device__ __forceinline__ bool narrow_q(double x, float &sn, float &cs) {
    const double PI4A = 7.85398125648498535156e-1, PI4B = 3.77489470793079817668e-8, PI4C = 2.69515142907905952645e-15;
    const double M4PI = 1.27323954473516268615107010698;
    const double ax = fabs(x);
    int32_t ji = __double2int_rz(__dmul_rn(ax, M4PI));
    ji += ji & 1;
    const double y = (double)ji;
    double z = __fma_rn(-y, PI4A, ax);  // (y PI4A, y PI4B: exact products, so the fused forms round as Go's do)
    z = __fma_rn(-y, PI4B, z);
    z = __dsub_rn(z, __dmul_rn(y, PI4C));
    const double zz = __dmul_rn(z, z);
    double ps = __fma_rn(1.58962301576546568060e-10, zz, -2.50507477628578072866e-8);
    ps = __fma_rn(ps, zz, 2.75573136213857245213e-6);
    ps = __fma_rn(ps, zz, -1.98412698295895385996e-4);
    ps = __fma_rn(ps, zz, 8.33333333332211858878e-3);
    ps = __fma_rn(ps, zz, -1.66666666666666307295e-1);
    const double s = __fma_rn(__dmul_rn(z, zz), ps, z);
    double pc = __fma_rn(-1.13585365213876817300e-11, zz, 2.08757008419747316778e-9);
    pc = __fma_rn(pc, zz, -2.75573141792967388112e-7);
    pc = __fma_rn(pc, zz, 2.48015872888517045348e-5);
    pc = __fma_rn(pc, zz, -1.38888888888730564116e-3);
    pc = __fma_rn(pc, zz, 4.16666666666665929218e-2);
    const double c = __fma_rn(__dmul_rn(zz, zz), pc, __fma_rn(-0.5, zz, 1.0));
    const double E = 1.4210854715202004e-14;  // 2^-46
    const float s_up = (float)__fma_rn(s, E, s), s_dn = (float)__fma_rn(s, -E, s);
    const float c_up = (float)__fma_rn(c, E, c), c_dn = (float)__fma_rn(c, -E, c);
    const bool swap = ji & 2;
    const float a = swap ? c_up : s_up, b = swap ? s_up : c_up;
    const unsigned t = (unsigned)ji << 29;  // bit 31 = quadrant bit 1, bit 30 = quadrant bit 0
    sn = __uint_as_float(__float_as_uint(a) ^ (((unsigned)__double2hiint(x) ^ t) & 0x80000000u));
    cs = __uint_as_float(__float_as_uint(b) ^ ((t ^ (t << 1)) & 0x80000000u));
    return s_up == s_dn && c_up == c_dn && ax < 536870912.0;
}

template <int MEM, int U, int TPB, int SLOW>
__global__ __launch_bounds__(TPB) void k_shift2(float4 *buf, size_t nvec, double tau_shift, NcoSegs sg) {
    const size_t tile = (size_t)TPB * U;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < nvec; t0 += (size_t)gridDim.x * tile) {
        const uint64_t j_lo = 2 * t0;
        const NcoWin w = nco_window(sg, j_lo, j_lo + 2 * tile - 1);
        float4 a[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * TPB + threadIdx.x;
            if constexpr ((MEM & 1) == 0) { if (i < nvec) a[u] = buf[i]; }
            else a[u] = make_float4((float)i, 1.f, 2.f, (float)u);
        }
        unsigned bad = 0;
        const uint64_t d0 = j_lo - nco_first(sg, w.lo);
        if (w.lo == w.hi && sg.big_n == 0 && d0 + 2 * tile < (1ull << 32)) {
            const double step = sg.step[w.lo], tb = sg.t0[w.lo];
            const double k0 = (double)((uint32_t)d0 + 2u * threadIdx.x);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t i = t0 + (size_t)u * TPB + threadIdx.x;
                if (i < nvec) {
                    float sl, cl, sh, ch;
                    const double kl = k0 + (double)(2 * u * TPB), kh = k0 + (double)(2 * u * TPB + 1);
                    bool ok = narrow_q(__dmul_rn(tau_shift, __fma_rn(kl, step, tb)), sl, cl);
                    ok &= narrow_q(__dmul_rn(tau_shift, __fma_rn(kh, step, tb)), sh, ch);
                    float2 l = go_cmul(make_float2(a[u].x, a[u].y), make_float2(cl, sl));
                    float2 h = go_cmul(make_float2(a[u].z, a[u].w), make_float2(ch, sh));
                    if (!ok) bad |= 1u << u;
                    if constexpr ((MEM & 2) == 0) { if (ok) buf[i] = make_float4(l.x, l.y, h.x, h.y); }
                    else if (l.x == 1.2345f && h.y == 5.4321f) buf[i] = make_float4(l.x, l.y, h.x, h.y);
                }
            }
        } else {
            bad = (1u << U) - 1;
        }
        if constexpr (SLOW) {
            if (bad) {  // (lanes: rare)
#pragma unroll 1
                for (int u = 0; u < U; u++) {
                    const size_t i = t0 + (size_t)u * TPB + threadIdx.x;
                    if (!((bad >> u) & 1) || i >= nvec) continue;
                    const float4 v = buf[i];
                    float r[4];
#pragma unroll 1
                    for (int h = 0; h < 2; h++) {
                        double sd, cd;
                        const double ph = __dmul_rn(tau_shift, nco_ts(sg, w, 2 * i + h));
                        if constexpr (SLOW == 1) go_sincos(ph, sd, cd);
                        else if constexpr (SLOW == 2) sincos_late(ph, sd, cd);
                        else if (fabs(ph) < 536870912.0) go_sincos_tail((uint32_t)(ph * 1.27), ph, ph < 0, sd, cd);
                        else sd = cd = 1.0;
                        const float2 o = go_cmul(h ? make_float2(v.z, v.w) : make_float2(v.x, v.y), make_float2((float)cd, (float)sd));
                        r[2 * h] = o.x, r[2 * h + 1] = o.y;
                    }
                    buf[i] = make_float4(r[0], r[1], r[2], r[3]);
                }
            }
        }
    }
}

template <int MEM, int U, int TPB, int SLOW> static void run2(float4 *buf, size_t nvec, unsigned grid_mult, const char *what) {
    NcoSegs sg{};
    sg.n = 1;
    sg.first[0] = 0, sg.t0[0] = 0.0, sg.step[0] = 1.0 / 20e6;
    const double tau_shift = (M_PI * 2) * 2.5e6;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t tile = (size_t)TPB * U;
    size_t blocks = (nvec + tile - 1) / tile;
    if (grid_mult) blocks = blocks < 256u * grid_mult ? blocks : 256u * grid_mult;
    float best = 1e9f, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 3; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_shift2<MEM, U, TPB, SLOW>), dim3((unsigned)blocks), dim3(TPB), 0, 0, buf, nvec, tau_shift, sg);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("v2 SLOW %d MEM %d U %d TPB %3d grid %6zu  %-24s min %.1f us  avg %.1f us\n", SLOW, MEM, U, TPB, blocks, what, best * 1e3f, sum / reps * 1e3f);
}


// ---- the third cut: the narrowing check on the float64 BITS (the 29 mantissa bits float32 drops, against their half
// point), a wave-private queue in LDS for the vectors a lane cannot decide, worked off behind the tile by one rolled
// copy of go_sincos; tiles that are not one full clock run inside the straight range go to the queue whole
__device__ __forceinline__ bool narrow_b(double x, float &sn, float &cs) { return sincos_narrow(x, sn, cs); }

template <int U, int TPB, int NT = 0>
__global__ __launch_bounds__(TPB) void k_shift3(float4 *buf, size_t nvec, uint64_t base, double tau_shift, NcoSegs sg) {
    constexpr int kW = TPB / 64;
    __shared__ unsigned q_n[kW];
    __shared__ unsigned short q[kW][64 * U];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) q_n[wave] = 0;
    const size_t tile = (size_t)TPB * U;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < nvec; t0 += (size_t)gridDim.x * tile) {
        const uint64_t j_lo = base + 2 * t0;
        const NcoWin w = nco_window(sg, j_lo, j_lo + 2 * tile - 1);
        const uint64_t d0 = j_lo - nco_first(sg, w.lo);
        bool fast = w.lo == w.hi && sg.big_n == 0 && d0 + 2 * tile < (1ull << 32) && t0 + tile <= nvec;
        double step = 0, tb = 0;
        if (fast) {
            step = sg.step[w.lo], tb = sg.t0[w.lo];
            const double ts_lo = __fma_rn((double)(uint32_t)d0, step, tb), ts_hi = __fma_rn((double)((uint32_t)d0 + 2 * (uint32_t)tile - 1), step, tb);
            const double x_lo = fabs(__dmul_rn(tau_shift, ts_lo)), x_hi = fabs(__dmul_rn(tau_shift, ts_hi));
            fast = ts_lo >= 0.0 && step > 0.0 && x_lo >= 8.673617379884035e-19 && x_hi < 536870912.0;  // 2^-60, 2^29
        }
        if (__builtin_amdgcn_readfirstlane((int)fast)) {
            float4 a[U];
#pragma unroll
            for (int u = 0; u < U; u++) a[u] = NT ? nt_load(buf + (t0 + (size_t)u * TPB + threadIdx.x)) : buf[t0 + (size_t)u * TPB + threadIdx.x];
            const double k0 = (double)((uint32_t)d0 + 2u * threadIdx.x);
#pragma unroll
            for (int u = 0; u < U; u++) {
                float sl, cl, sh, ch;
                const double kl = k0 + (double)(2 * u * TPB), kh = k0 + (double)(2 * u * TPB + 1);
                bool ok = narrow_b(__dmul_rn(tau_shift, __fma_rn(kl, step, tb)), sl, cl);
                ok &= narrow_b(__dmul_rn(tau_shift, __fma_rn(kh, step, tb)), sh, ch);
                const float2 l = go_cmul(make_float2(a[u].x, a[u].y), make_float2(cl, sl));
                const float2 h = go_cmul(make_float2(a[u].z, a[u].w), make_float2(ch, sh));
                if (ok) {
                    if (NT) nt_store(buf + (t0 + (size_t)u * TPB + threadIdx.x), make_float4(l.x, l.y, h.x, h.y));
                    else buf[t0 + (size_t)u * TPB + threadIdx.x] = make_float4(l.x, l.y, h.x, h.y);
                } else q[wave][atomicAdd(&q_n[wave], 1u)] = (unsigned short)(u * TPB + threadIdx.x);
            }
        } else {
#pragma unroll 1
            for (int u = 0; u < U; u++)
                if (t0 + (size_t)u * TPB + threadIdx.x < nvec) q[wave][atomicAdd(&q_n[wave], 1u)] = (unsigned short)(u * TPB + threadIdx.x);
        }
        const unsigned nq = q_n[wave];  // (LDS operations of one wave complete in order)
        if (nq) {
#pragma unroll 1
            for (unsigned e = lane; e < nq; e += 64) {
                const size_t i = t0 + q[wave][e];
                float2 *const p2 = reinterpret_cast<float2 *>(buf + i);
#pragma unroll 1
                for (int h = 0; h < 2; h++) {
                    double sd, cd;
                    go_sincos(__dmul_rn(tau_shift, nco_ts(sg, w, base + 2 * i + h)), sd, cd);
                    p2[h] = go_cmul(p2[h], make_float2((float)cd, (float)sd));
                }
            }
            if (lane == 0) q_n[wave] = 0;
        }
    }
}

template <int U, int TPB> static void run3(float4 *buf, size_t nvec, unsigned grid_mult, const char *what) {
    NcoSegs sg{};
    sg.n = 1;
    sg.first[0] = 0, sg.t0[0] = 0.0, sg.step[0] = 1.0 / 20e6;
    const double tau_shift = (M_PI * 2) * 2.5e6;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t tile = (size_t)TPB * U;
    size_t blocks = (nvec + tile - 1) / tile;
    if (grid_mult) blocks = blocks < 256u * grid_mult ? blocks : 256u * grid_mult;
    float best = 1e9f, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 3; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_shift3<U, TPB>), dim3((unsigned)blocks), dim3(TPB), 0, 0, buf, nvec, (uint64_t)0, tau_shift, sg);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("v3 U %d TPB %3d grid %6zu  %-24s min %.1f us  avg %.1f us\n", U, TPB, blocks, what, best * 1e3f, sum / reps * 1e3f);
}

// the third cut in place over a ROTATION of eight buffers (1 GiB: every call's bytes from HBM and to it)
template <int U, int TPB, int NT> static void run3_rot(float4 **bufs, size_t nvec, unsigned grid_mult, const char *what) {
    NcoSegs sg{};
    sg.n = 1;
    sg.first[0] = 0, sg.t0[0] = 0.0, sg.step[0] = 1.0 / 20e6;
    const double tau_shift = (M_PI * 2) * 2.5e6;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t tile = (size_t)TPB * U;
    size_t blocks = (nvec + tile - 1) / tile;
    if (grid_mult) blocks = blocks < 256u * grid_mult ? blocks : 256u * grid_mult;
    const int reps = 64;
    for (int r = 0; r < 16; r++) hipLaunchKernelGGL((k_shift3<U, TPB, NT>), dim3((unsigned)blocks), dim3(TPB), 0, 0, bufs[r % 8], nvec, (uint64_t)0, tau_shift, sg);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_shift3<U, TPB, NT>), dim3((unsigned)blocks), dim3(TPB), 0, 0, bufs[r % 8], nvec, (uint64_t)0, tau_shift, sg);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("v3 rotation U %d TPB %3d NT %d grid %6zu  %-20s %.1f us per call\n", U, TPB, NT, blocks, what, ms / reps * 1e3f);
}

template <int SC, int MEM, int U, int TPB> static void run(float4 *buf, size_t nvec, unsigned grid_mult, const char *what) {
    NcoSegs sg{};
    sg.n = 1;
    sg.first[0] = 0, sg.t0[0] = 0.0, sg.step[0] = 1.0 / 20e6;
    const double tau_shift = (M_PI * 2) * 2.5e6;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t tile = (size_t)TPB * U;
    size_t blocks = (nvec + tile - 1) / tile;
    if (grid_mult) blocks = blocks < 256u * grid_mult ? blocks : 256u * grid_mult;
    float best = 1e9f, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 3; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_shift<SC, MEM, U, TPB>), dim3((unsigned)blocks), dim3(TPB), 0, 0, buf, nvec, tau_shift, sg);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("SC %d MEM %d U %d TPB %3d grid %6zu  %-28s min %.1f us  avg %.1f us\n", SC, MEM, U, TPB, blocks, what, best * 1e3f, sum / reps * 1e3f);
}

int main() {
    const size_t n = 1 << 24, nvec = n / 2;
    float4 *buf;
    CK(hipMalloc(&buf, n * 8));
    CK(hipMemset(buf, 0, n * 8));
    run<0, 0, 4, 256>(buf, nvec, 0, "narrow, go_sincos inline");
    run<1, 0, 4, 256>(buf, nvec, 0, "math.Sincos op for op");
    run<2, 0, 4, 256>(buf, nvec, 0, "late, unchecked");
    run<4, 0, 4, 256>(buf, nvec, 0, "check, no fallback code");
    run<3, 0, 4, 256>(buf, nvec, 0, "no Sincos");
    run<0, 3, 4, 256>(buf, nvec, 0, "narrow inline, no memory");
    run<1, 3, 4, 256>(buf, nvec, 0, "op for op, no memory");
    run<2, 3, 4, 256>(buf, nvec, 0, "late, no memory");
    run<3, 3, 4, 256>(buf, nvec, 0, "no Sincos, no memory");
    run<0, 1, 4, 256>(buf, nvec, 0, "narrow inline, no loads");
    run<0, 2, 4, 256>(buf, nvec, 0, "narrow inline, no stores");
    run<0, 0, 2, 256>(buf, nvec, 0, "narrow inline U 2");
    run<0, 0, 8, 256>(buf, nvec, 0, "narrow inline U 8");
    run<0, 0, 1, 256>(buf, nvec, 0, "narrow inline U 1");
    run<0, 0, 4, 256>(buf, nvec, 8, "narrow inline, 2048 workgroups");
    run<0, 0, 4, 256>(buf, nvec, 16, "narrow inline, 4096 workgroups");
    run<0, 0, 2, 256>(buf, nvec, 16, "narrow inline U 2, 4096 wgs");
    run<0, 0, 4, 128>(buf, nvec, 0, "narrow inline, 128 threads");
    run<0, 0, 4, 512>(buf, nvec, 0, "narrow inline, 512 threads");
    run2<0, 4, 256, 1>(buf, nvec, 0, "second cut");
    run2<0, 4, 256, 0>(buf, nvec, 0, "no slow code");
    run2<0, 4, 256, 2>(buf, nvec, 0, "slow code = late");
    run2<0, 4, 256, 3>(buf, nvec, 0, "slow code = tail only");
    run2<3, 4, 256, 2>(buf, nvec, 0, "no memory, slow code = late");
    run2<3, 4, 256, 3>(buf, nvec, 0, "no memory, slow code = tail only");
    run2<3, 4, 256, 1>(buf, nvec, 0, "no memory");
    run2<3, 4, 256, 0>(buf, nvec, 0, "no memory, no slow code");
    run2<0, 2, 256, 1>(buf, nvec, 0, "U 2");
    run2<0, 8, 256, 1>(buf, nvec, 0, "U 8");
    run2<0, 4, 256, 1>(buf, nvec, 16, "4096 workgroups");
    run2<0, 4, 256, 1>(buf, nvec, 8, "2048 workgroups");
    run2<0, 4, 128, 1>(buf, nvec, 0, "128 threads");
    run2<0, 2, 128, 1>(buf, nvec, 0, "128 threads U 2");
    {  // the third cut against math.Sincos operation for operation, bit for bit, on random samples
        float4 *b1, *b2;
        CK(hipMalloc(&b1, n * 8));
        CK(hipMalloc(&b2, n * 8));
        std::vector<float> h(2 * n);
        uint64_t st = 12345;
        for (auto &v : h) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = (float)((int64_t)(st >> 20) % 2000001 - 1000000) * 1e-6f; }
        CK(hipMemcpy(b1, h.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(b2, h.data(), n * 8, hipMemcpyHostToDevice));
        NcoSegs sg{};
        sg.n = 1;
        sg.first[0] = 0, sg.t0[0] = 0.0, sg.step[0] = 1.0 / 20e6;
        const double tau_shift = (M_PI * 2) * 2.5e6;
        hipLaunchKernelGGL((k_shift<1, 0, 4, 256>), dim3((unsigned)(nvec / 1024)), dim3(256), 0, 0, b1, nvec, tau_shift, sg);
        hipLaunchKernelGGL((k_shift3<4, 256>), dim3((unsigned)(nvec / 1024)), dim3(256), 0, 0, b2, nvec, (uint64_t)0, tau_shift, sg);
        std::vector<uint32_t> r1(2 * n), r2(2 * n);
        CK(hipMemcpy(r1.data(), b1, n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r2.data(), b2, n * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < 2 * n; i++) bad += r1[i] != r2[i];
        printf("third cut vs op for op: %zu of %zu floats differ\n", bad, 2 * n);
        CK(hipFree(b1));
        CK(hipFree(b2));
    }
    run3<4, 256>(buf, nvec, 0, "third cut");
    run3<2, 256>(buf, nvec, 0, "third cut");
    run3<8, 256>(buf, nvec, 0, "third cut");
    run3<4, 256>(buf, nvec, 8, "third cut");
    run3<4, 256>(buf, nvec, 16, "third cut");
    run3<4, 128>(buf, nvec, 0, "third cut");
    run3<4, 512>(buf, nvec, 0, "third cut");
    {
        float4 *bufs[8];
        for (int b = 0; b < 8; b++) { CK(hipMalloc(&bufs[b], n * 8)); CK(hipMemset(bufs[b], 0, n * 8)); }
        run3_rot<4, 256, 0>(bufs, nvec, 0, "plain");
        run3_rot<4, 256, 1>(bufs, nvec, 0, "non-temporal");
        run3_rot<2, 256, 0>(bufs, nvec, 0, "plain");
        run3_rot<2, 256, 1>(bufs, nvec, 0, "non-temporal");
        run3_rot<4, 128, 0>(bufs, nvec, 0, "plain");
        run3_rot<4, 128, 1>(bufs, nvec, 0, "non-temporal");
        run3_rot<2, 128, 1>(bufs, nvec, 0, "non-temporal");
        run3_rot<4, 256, 1>(bufs, nvec, 16, "non-temporal");
        run3_rot<8, 256, 1>(bufs, nvec, 0, "non-temporal");
        run3_rot<4, 256, 0>(bufs, nvec, 0, "plain (again)");
    }
    return 0;
}

// mfma_fir2.hip -- times hz::mm2::fir_mm2_kernel<u8, 8> (hz_firmm2.h) in isolation on 12 rotating
// 2^24-sample buffers (384 MiB: every launch reads its input from HBM), with phases switched off
// (template parameter EXP) and with per-pass time stamps.  Tables hold constants: the instruction
// stream is the real one, the results are not.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "hz_firmm2.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using namespace hz;

static unsigned long long *g_stamps = nullptr;
static int g_grid = 256;

static void pct(const char *name, std::vector<double> v) {
    if (v.empty()) return;
    std::sort(v.begin(), v.end());
    printf("    %-34s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f us\n", name, v[0] / 100, v[v.size() / 10] / 100,
           v[v.size() / 2] / 100, v[v.size() * 9 / 10] / 100, v.back() / 100);
}

static int g_batch = 1;  // buffers per launch (BATCH=k: the kernel over k of the 12 buffers as one call, hzsdr_chain_run_batch's form)

template <int EXP, int MIXT = 1, int NGT = 17> static void run(void *const *in, float2 *out, const float2 *taps, const void *tab, size_t n_each, int ntaps,
                                   bool shift) {
    constexpr int D = 8;
    const size_t n = n_each * (size_t)g_batch;
    const unsigned off = (unsigned)((ntaps - 1 + D - 1) / D * D);
    mm2::Geom g = mm2::make_geom(ntaps, D, off, 40);
    EwProgram P{};
    if (shift) {
        P.n = 1;
        P.op[0].kind = EW_SHIFT;
        P.op[0].tau_shift = -1.5707963e7;
        P.segs.n = 1;
        P.segs.first[0] = 0;
        P.segs.t0[0] = 0.25;
        P.segs.step[0] = 5e-8;
    }
    mm2::Plan R{};
    mm2::Fix F{};
    const uint32_t n_out = (uint32_t)(n / D);
    R.n = 1;
    R.run[0].tab = tab;
    R.cont = 1;
    R.run[0].m_lo = 0;
    R.run[0].m_hi = n_out;
    R.run[0].pass_first = 0;
    R.n_pass = (int)((n_out + mm2::pass_out(D) - 1) / mm2::pass_out(D));
    R.shift_op = shift ? 0 : -1;
    R.run[0].pass_end = R.n_pass;
    R.pass_first[0] = 0;
    R.pass_end[0] = R.n_pass;
    R.n_ops = P.n;
    R.grid = g_grid;
    if (shift) mm2::phase_fix(P.op[0].tau_shift, P.segs.t0[0], P.segs.step[0], 0, &R.run[0].phi, &R.run[0].dphi);
    if (getenv("FIXTASKS")) {  // N fix-up tasks over outputs that the passes compute as well (timing only)
        F.n = 1;
        F.m_a[0] = 4096;
        F.n_task = atoi(getenv("FIXTASKS"));
        F.m_b[0] = 4096 + mm2::kFixOut * F.n_task;
        R.n_task = F.n_task;
    }
    const size_t lds = mm2::lds_bytes(D, g.ks, g.ne, g.ntaps) + ((EXP & 64) ? 2048 : 0);  // (+ the stamps' staging area)
    const unsigned grid = (unsigned)g_grid;
    auto k = mm2::fir_mm2_kernel<HZSDR_FMT_U8, D, NGT, EXP>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    std::vector<float> all;
    const int reps = getenv("AB") ? 96 : 24;
    for (int r = 0; r < reps + 4; r++) {
        const void *ins[mm2::kMaxBatch];
        void *outs[mm2::kMaxBatch];
        for (int j = 0; j < g_batch; j++) ins[j] = in[getenv("ONEBUF") ? 0 : (r * g_batch + j) % 12], outs[j] = out + (size_t)j * (n_each / D);
        const mm2::Batch B = g_batch == 1 ? mm2::one_buffer(ins[0], out, n, D) : mm2::make_batch(ins, outs, (size_t)g_batch, n_each, D);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(mm2::kThreads), lds, 0, ins[0], out, (const float2 *)nullptr,
                           out + n_out, (const uint8_t *)taps, (uint8_t *)(out + n_out + 4096), taps, n, g, R, P, F, B, g_stamps);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 4) { best = ms < best ? ms : best; sum += ms; all.push_back(ms); }
    }
    CK(hipGetLastError());
    std::sort(all.begin(), all.end());
    printf("MIX %d EXP %6d%s: grid %u, LDS %zu, ks %d, passes %d, %d buffer(s) per launch: min %.1f us  median %.1f us  avg %.1f us PER BUFFER\n", MIXT, EXP, shift ? " +Shift" : "       ", grid, lds,
           g.ks, R.n_pass, g_batch, best * 1e3f / g_batch, all[all.size() / 2] * 1e3f / g_batch, sum / reps * 1e3f / g_batch);
    if (EXP & 64) {
        const size_t nw = (size_t)grid * mm2::kWaves;
        std::vector<unsigned long long> st(nw * 32);
        CK(hipMemcpy(st.data(), g_stamps, st.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (size_t w = 0; w < nw; w++) t0 = std::min(t0, st[w * 32]);
        std::vector<double> start, end, l1s, l1d, ld1, c1d, m1d, s1d, l2s, l2d, ld2, c2d, m2d, s2d;
        for (size_t w = 0; w < nw; w++) {
            const unsigned long long *s = &st[w * 32];
            start.push_back((double)(s[0] - t0));
            end.push_back((double)(s[7] - t0));
            const unsigned long long *p1 = s + 8, *p2 = s + 16;
            if (p1[1]) {
                l1s.push_back((double)(p1[1] - t0));
                l1d.push_back((double)(p1[2] - p1[1]));
                ld1.push_back((double)(p1[3] - p1[2]));
                c1d.push_back((double)(p1[4] - p1[3]));
                m1d.push_back((double)(p1[5] - p1[4]));
                s1d.push_back((double)(p1[6] - p1[5]));
            }
            if (p2[1]) {
                l2s.push_back((double)(p2[1] - t0));
                l2d.push_back((double)(p2[2] - p2[1]));
                c2d.push_back((double)(p2[4] - p2[3]));
                m2d.push_back((double)(p2[5] - p2[4]));
                s2d.push_back((double)(p2[6] - p2[5]));
            }
        }
        pct("wave start", start);
        pct("pass 1: loop starts at", l1s);
        pct("pass 1: loop", l1d);
        pct("pass 1: landing the next pass", ld1);
        pct("pass 1: planes -> float", c1d);
        pct("pass 1: mixer", m1d);
        pct("pass 1: stores", s1d);
        pct("pass 2: loop starts at", l2s);
        pct("pass 2: loop", l2d);
        pct("pass 2: planes -> float", c2d);
        pct("pass 2: mixer", m2d);
        pct("pass 2: stores", s2d);
        pct("wave end", end);
        for (int wgx = (getenv("FIXTASKS") ? 0 : 100); wgx < (getenv("FIXTASKS") ? 2 : 102) && (unsigned)wgx < grid; wgx++)
            for (int w = 0; w < 8; w++) {
                const unsigned long long *s = &st[((size_t)wgx * 8 + w) * 32];
                printf("    wg %d wave %d: start %.2f tasks %.2f x-issued %.2f tab-issued %.2f tab-landed %.2f barrier %.2f landed %.2f |", wgx, w, (double)(s[0] - t0) / 100, (double)(s[1] - t0) / 100, (double)(s[5] - t0) / 100, (double)(s[6] - t0) / 100, (double)(s[2] - t0) / 100, (double)(s[3] - t0) / 100, (double)(s[4] - t0) / 100);
                for (int pp = 1; pp <= 3; pp++) {
                    const unsigned long long *q = s + 8 * pp;
                    if (q[1]) printf(" top %.2f grabbed %.2f", (double)(q[0] - t0) / 100, (double)(q[7] - t0) / 100);
                    if (q[1]) printf(" loop %.2f-%.2f land %.2f planes %.2f mixer %.2f stores %.2f |", (double)(q[1] - t0) / 100, (double)(q[2] - t0) / 100,
                                     (double)(q[3] - t0) / 100, (double)(q[4] - t0) / 100, (double)(q[5] - t0) / 100, (double)(q[6] - t0) / 100);
                }
                printf(" end %.2f\n", (double)(s[7] - t0) / 100);
            }
        {   // the workgroups that end last: where their time went
            std::vector<std::pair<double, int>> ends;
            for (unsigned wg = 0; wg < grid; wg++) {
                double e = 0;
                for (int w = 0; w < 8; w++) e = std::max(e, (double)(st[((size_t)wg * 8 + w) * 32 + 7] - t0));
                ends.push_back({e, (int)wg});
            }
            std::sort(ends.begin(), ends.end());
            printf("    workgroup ends: min %.2f median %.2f max %.2f us; the last twelve:\n", ends[0].first / 100, ends[grid / 2].first / 100, ends.back().first / 100);
            for (unsigned q = grid >= 12 ? grid - 12 : 0; q < grid; q++) {
                const int wg = ends[q].second;
                double s0 = 1e18, l0 = 1e18, l4 = 1e18, bar = 0;
                for (int w = 0; w < 8; w++) {
                    const unsigned long long *sp = &st[((size_t)wg * 8 + w) * 32];
                    s0 = std::min(s0, (double)(sp[0] - t0));
                    bar = std::max(bar, (double)(sp[3] - t0));
                    if (sp[8 + 1]) (w < 4 ? l0 : l4) = std::min(w < 4 ? l0 : l4, (double)(sp[8 + 1] - t0));
                }
                printf("      wg %3d: start %.2f barrier %.2f first loops %.2f / %.2f end %.2f\n", wg, s0 / 100, bar / 100, l0 / 100, l4 / 100, ends[q].first / 100);
            }
        }
        {   // by XCD (workgroup index mod 8): the second pass's loop (nobody waits for memory there), the epilogue beside it, the end
            printf("    by XCD:        ");
            for (int x = 0; x < 8; x++) printf("%8d", x);
            const char *rows[4] = {"first loop at ", "pass 2 loop   ", "pass 1 epilog.", "end           "};
            for (int row = 0; row < 4; row++) {
                printf("\n      %s", rows[row]);
                for (int x = 0; x < 8; x++) {
                    std::vector<double> v;
                    for (size_t w = 0; w < nw; w++) {
                        if ((int)((w / 8) % 8) != x) continue;
                        const unsigned long long *sp = &st[w * 32];
                        if (row == 0 && sp[8 + 1]) v.push_back((double)(sp[8 + 1] - t0));
                        if (row == 1 && sp[16 + 1]) v.push_back((double)(sp[16 + 2] - sp[16 + 1]));
                        if (row == 2 && sp[8 + 1]) v.push_back((double)(sp[8 + 6] - sp[8 + 2]));
                        if (row == 3) v.push_back((double)(sp[7] - t0));
                    }
                    std::sort(v.begin(), v.end());
                    printf("%8.2f", v.empty() ? 0.0 : v[v.size() / 2] / 100);
                }
            }
            printf("\n");
        }
        // waves 0-3 vs 4-7
        for (int half = 0; half < 2; half++) {
            std::vector<double> a, b;
            for (size_t w = 0; w < nw; w++)
                if ((int)((w % 8) / 4) == half) {
                    a.push_back((double)(st[w * 32 + 8 + 1] - t0));
                    b.push_back((double)(st[w * 32 + 7] - t0));
                }
            pct(half ? "waves 4-7: first loop starts" : "waves 0-3: first loop starts", a);
            pct(half ? "waves 4-7: end" : "waves 0-3: end", b);
        }
    }
}

int main(int argc, char **argv) {
    const size_t n = (size_t)1 << 24;
    const int ntaps = argc > 1 ? atoi(argv[1]) : 1024;
    if (argc > 2) g_grid = atoi(argv[2]);
    void *in[12];
    float2 *out, *taps;
    void *tab;
    std::vector<unsigned char> h(n * 2);
    for (int b = 0; b < 12; b++) {
        CK(hipMalloc(&in[b], n * 2 + 65536));
        for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)((i + b) * 2654435761u >> 24);
        CK(hipMemcpy(in[b], h.data(), h.size(), hipMemcpyHostToDevice));
    }
    if (getenv("BATCH")) g_batch = atoi(getenv("BATCH"));
    if (g_batch < 1 || g_batch > mm2::kMaxBatch) g_batch = 1;
    CK(hipMalloc(&out, ((size_t)g_batch * n / 8 + 8192) * 8));
    CK(hipMalloc(&taps, ntaps * 8 + 65536));
    CK(hipMemset(taps, 0, ntaps * 8 + 65536));
    std::vector<unsigned char> t(1 << 20);
    for (size_t i = 0; i < t.size(); i++) t[i] = (unsigned char)(i * 40503u >> 8);
    CK(hipMalloc(&tab, t.size()));
    CK(hipMemcpy(tab, t.data(), t.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&g_stamps, 8 * 32 * 8 * 1024));
    CK(hipMemset(g_stamps, 0, 8 * 32 * 8 * 1024));
    printf("fir_mm2_kernel<u8, 8>, 2^24 samples, %d taps; EXP 1 = no input loads, 2 = no matrix loop, 4 = no mixer, 8 = no stores, 32 = wave priorities, 128 = no explicit vmcnt(0)\n", ntaps);
    if (getenv("AB")) {  // A/B of the build's switches on one box, interleaved
        // round 5: the instruction cuts one by one and together (hz_firmm2.h, EXP): 8192 packed mixer, 16384 int32 plane
        // pair, 32768 sign flip by the LDS, 65536 constant-C first step, 131072 cold path fenced; 4 = no mixer at all
        constexpr int ALL = 8192 | 16384 | 65536 | 131072 | 262144 | 524288;  // (kLibExp)
        for (int r = 0; r < 3; r++) {
            run<0, 0>(in, out, taps, tab, n, ntaps, true);
            run<ALL & ~524288, 0>(in, out, taps, tab, n, ntaps, true);
            run<ALL, 0>(in, out, taps, tab, n, ntaps, true);
            run<524288 | 131072, 0>(in, out, taps, tab, n, ntaps, true);
            run<ALL | 4, 0>(in, out, taps, tab, n, ntaps, true);
            run<ALL | 8, 0>(in, out, taps, tab, n, ntaps, true);
        }
        return 0;
    }
    if (getenv("BATCH")) {  // the shipped form, k buffers per launch
        constexpr int ALL = 8192 | 16384 | 65536 | 131072 | 262144 | 524288;
        for (int r = 0; r < 3; r++) {
            run<ALL, 0>(in, out, taps, tab, n, ntaps, true);
            run<ALL | (1 << 22), 0>(in, out, taps, tab, n, ntaps, true);                // the mixer's products in scalar float32
            run<(ALL | (1 << 22)) & ~524288, 0>(in, out, taps, tab, n, ntaps, true);   // ... and the stores in the epilogue
            run<((ALL | (1 << 22)) & ~524288) | (1 << 23), 0>(in, out, taps, tab, n, ntaps, true);  // ... the stores at the next loop's top
            run<ALL | 4, 0>(in, out, taps, tab, n, ntaps, true);                        // no mixer at all
            run<0, 0>(in, out, taps, tab, n, ntaps, true);                              // round 4's kernel
        }
        return 0;
    }
    run<8192 | 16384 | 65536 | 131072 | 262144 | (1 << 22), 0>(in, out, taps, tab, n, ntaps, true);  // (what the library ships: kLibExp)
    run<0, 0>(in, out, taps, tab, n, ntaps, true);
    run<128, 0>(in, out, taps, tab, n, ntaps, true);
    run<0, 0>(in, out, taps, tab, n, ntaps, true);
    run<0, 0>(in, out, taps, tab, n, ntaps, false);
    run<1, 0>(in, out, taps, tab, n, ntaps, true);
    run<2, 0>(in, out, taps, tab, n, ntaps, true);
    run<8, 0>(in, out, taps, tab, n, ntaps, true);
    run<32, 0>(in, out, taps, tab, n, ntaps, true);
    run<64 | 8192 | 16384 | 65536 | 131072 | 262144 | (1 << 22), 0>(in, out, taps, tab, n, ntaps, true);  // (kLibExp with stamps)
    if (getenv("BISECT")) {  // which part of the epilogue waits for the partner's loop: without the mixer, without the stores
        run<64 | 4 | 8192 | 16384 | 65536 | 131072 | 262144 | 524288, 0>(in, out, taps, tab, n, ntaps, true);
        run<64 | 8 | 8192 | 16384 | 65536 | 131072 | 262144 | 524288, 0>(in, out, taps, tab, n, ntaps, true);
        run<64 | 4 | 8 | 8192 | 16384 | 65536 | 131072 | 262144 | 524288, 0>(in, out, taps, tab, n, ntaps, true);
        constexpr int ALLB = 64 | 8192 | 16384 | 65536 | 131072 | 262144 | 524288;
        run<ALLB | (1 << 20), 0>(in, out, taps, tab, n, ntaps, true);
        run<ALLB | (1 << 21), 0>(in, out, taps, tab, n, ntaps, true);
        run<ALLB | (1 << 22), 0>(in, out, taps, tab, n, ntaps, true);
        run<ALLB | (1 << 20) | (1 << 21), 0>(in, out, taps, tab, n, ntaps, true);
    }
    return 0;
}

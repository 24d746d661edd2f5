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

// PASSES=1: a row of stamps for every pass of every wave (EXP 64 | 1 << 24; s_memrealtime, 10 ns ticks, staged in LDS and
// written out at the wave's end).  Row p of a wave: [0] the pass's top, [7] queue read, [1] the next pass's loads issued =
// the matrix loop's start, [2] its end, [3] the next pass landed, [4] planes combined, [5] mixer done, [6] stores issued.
// Reduced to what VERDICT r05 asked for: a per-phase table over all passes of the launch, the same split by whether the
// SIMD partner (wave w ^ 4 of the workgroup) was inside ITS matrix loop for most of the phase, and per SIMD the share
// of its busy window in which both / one / neither wave was inside a matrix loop.
static void analyze_passes(unsigned grid) {
    constexpr int R = 16;
    const size_t nw = (size_t)grid * mm2::kWaves;
    std::vector<unsigned long long> st(nw * 8 * R);
    CK(hipMemcpy(st.data(), g_stamps, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (size_t w = 0; w < nw; w++) t0 = std::min(t0, st[w * 8 * R]);
    struct Iv { double a, b; };
    auto loops_of = [&](size_t w) {
        std::vector<Iv> v;
        for (int p = 1; p < R; p++) {
            const unsigned long long *q = &st[(w * R + p) * 8];
            if (q[1] && q[2]) v.push_back({(double)(q[1] - t0), (double)(q[2] - t0)});
        }
        return v;
    };
    auto overlap = [](const std::vector<Iv> &v, double a, double b) {
        double s = 0;
        for (const Iv &i : v) s += std::max(0.0, std::min(b, i.b) - std::max(a, i.a));
        return s;
    };
    const char *names[7] = {"queue + next pass's loads issued", "matrix loop", "landing the next pass", "planes -> float32", "mixer", "stores issued", "whole pass (top to stores)"};
    std::vector<double> all[7], with[7], without[7];
    std::vector<double> first_loop, last_end, passes_per_wave;
    double sum_window = 0, sum_both = 0, sum_one = 0, sum_none = 0, sum_loop = 0;
    size_t simds = 0;
    for (size_t wg = 0; wg < grid; wg++)
        for (int w = 0; w < mm2::kWaves; w++) {
            const size_t me = wg * mm2::kWaves + w, pa = wg * mm2::kWaves + (w ^ 4);
            const std::vector<Iv> mine = loops_of(me), theirs = loops_of(pa);
            passes_per_wave.push_back((double)mine.size());
            for (int p = 1; p < R; p++) {
                const unsigned long long *q = &st[(me * R + p) * 8];
                if (!q[1] || !q[2] || !q[6]) continue;
                const double tq[7] = {(double)(q[0] - t0), (double)(q[1] - t0), (double)(q[2] - t0), (double)(q[3] - t0), (double)(q[4] - t0), (double)(q[5] - t0), (double)(q[6] - t0)};
                for (int k = 0; k < 7; k++) {
                    const double a = k == 6 ? tq[0] : tq[k], b = k == 6 ? tq[6] : tq[k + 1];
                    if (!(b >= a) || (k >= 3 && k <= 5 && q[4] == 0)) continue;
                    all[k].push_back(b - a);
                    const double ov = b > a ? overlap(theirs, a, b) / (b - a) : 0.0;
                    (ov >= 0.5 ? with[k] : without[k]).push_back(b - a);
                }
            }
            if (w < 4 && !mine.empty() && !theirs.empty()) {  // one SIMD: the pair's window from its first loop's start to its last loop's end
                const double a = std::min(mine[0].a, theirs[0].a), b = std::max(mine.back().b, theirs.back().b);
                // sweep
                std::vector<std::pair<double, int>> ev;
                for (const Iv &i : mine) ev.push_back({i.a, 1}), ev.push_back({i.b, -1});
                for (const Iv &i : theirs) ev.push_back({i.a, 1}), ev.push_back({i.b, -1});
                std::sort(ev.begin(), ev.end());
                double t = a, both = 0, one = 0, none = 0;
                int in = 0;
                for (auto &e : ev) {
                    const double dt = e.first - t;
                    (in >= 2 ? both : in == 1 ? one : none) += dt;
                    t = e.first, in += e.second;
                }
                sum_window += b - a, sum_both += both, sum_one += one, sum_none += none;
                for (const Iv &i : mine) sum_loop += i.b - i.a;
                for (const Iv &i : theirs) sum_loop += i.b - i.a;
                first_loop.push_back(a), last_end.push_back(b);
                simds++;
            }
        }
    auto row = [&](const char *nm, std::vector<double> v) {
        if (v.empty()) { printf("    %-36s (none)\n", nm); return; }
        std::sort(v.begin(), v.end());
        double m = 0;
        for (double x : v) m += x;
        printf("    %-36s n %7zu  mean %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f us\n", nm, v.size(), m / v.size() / 100, v[v.size() / 10] / 100, v[v.size() / 2] / 100, v[v.size() * 9 / 10] / 100);
    };
    printf("  per-pass phases over ALL passes of the launch (every wave of every workgroup):\n");
    for (int k = 0; k < 7; k++) row(names[k], all[k]);
    printf("  ... the same, phases during which the SIMD partner was inside its matrix loop (>= half of the phase):\n");
    for (int k = 0; k < 7; k++) row(names[k], with[k]);
    printf("  ... and phases during which it was not:\n");
    for (int k = 0; k < 7; k++) row(names[k], without[k]);
    row("passes per wave (x 0.01)", passes_per_wave);
    printf("  per SIMD (%zu pairs of waves), from the pair's first loop start to its last loop end: window %.2f us mean;\n"
           "    both waves inside a matrix loop %.1f %%, exactly one %.1f %%, NEITHER %.1f %% of the window; the two waves' loop time adds up to %.2f us per SIMD\n",
           simds, sum_window / simds / 100, 100 * sum_both / sum_window, 100 * sum_one / sum_window, 100 * sum_none / sum_window, sum_loop / simds / 100);
    row("SIMD: first loop starts at", first_loop);
    row("SIMD: last loop ends at", last_end);
    std::vector<double> ends;
    for (size_t w = 0; w < nw; w++) ends.push_back((double)(st[w * 8 * R + 7] - t0));
    row("wave end", ends);
    {   // the shader clock under this kernel: s_memtime cycles per 10 ns tick of s_memrealtime, per workgroup
        std::vector<unsigned long long> ck(2 * (size_t)grid);
        CK(hipMemcpy(ck.data(), g_stamps + nw * 8 * R, ck.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> ghz;
        for (unsigned wg = 0; wg < grid; wg++)
            if (ck[2 * wg + 1]) ghz.push_back((double)ck[2 * wg] / (double)ck[2 * wg + 1] / 10.0);
        std::sort(ghz.begin(), ghz.end());
        if (!ghz.empty())
            printf("  shader clock over the workgroups' lives (s_memtime / s_memrealtime): min %.3f  median %.3f  max %.3f GHz;\n"
                   "    at the median clock a pass's 272 MFMAs x 32 cycles are %.2f us, a SIMD's %d passes %.1f us of matrix pipe\n",
                   ghz[0], ghz[ghz.size() / 2], ghz.back(), 8704.0 / ghz[ghz.size() / 2] / 1000.0, (int)(all[1].size() / (4 * (size_t)grid)),
                   (double)(all[1].size() / (4 * (size_t)grid)) * 8704.0 / ghz[ghz.size() / 2] / 1000.0);
    }
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
    const size_t lds = mm2::lds_bytes(D, g.ks, g.ne, g.ntaps) + ((EXP & 64) ? ((EXP & (1 << 24)) ? 8192 : 2048) : 0);  // (+ the stamps' staging area)
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
    if constexpr ((EXP & 64) != 0 && (EXP & (1 << 24)) != 0) {
        analyze_passes(grid);
        return;
    }
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
    if (getenv("ZERO") || getenv("ZERO_IN")) {  // quiet data: every sample 0x80 (0 behind the u8 sign flip), the table below all zeros -- the same instruction stream, no toggling operands
        std::fill(h.begin(), h.end(), (unsigned char)0x80);
        for (int b = 0; b < 12; b++) CK(hipMemcpy(in[b], h.data(), h.size(), hipMemcpyHostToDevice));
    }
    if (getenv("BATCH")) g_batch = atoi(getenv("BATCH"));
    if (g_batch < 1 || g_batch > mm2::kMaxBatch) g_batch = 1;
    CK(hipMalloc(&out, ((size_t)g_batch * n / 8 + 8192) * 8));
    CK(hipMalloc(&taps, ntaps * 8 + 65536));
    CK(hipMemset(taps, 0, ntaps * 8 + 65536));
    std::vector<unsigned char> t(1 << 20);
    for (size_t i = 0; i < t.size(); i++) t[i] = (getenv("ZERO") || getenv("ZERO_TAB")) ? 0 : (unsigned char)(i * 40503u >> 8);
    CK(hipMalloc(&tab, t.size()));
    CK(hipMemcpy(tab, t.data(), t.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&g_stamps, 8 * 128 * 8 * 1024));
    CK(hipMemset(g_stamps, 0, 8 * 128 * 8 * 1024));
    printf("fir_mm2_kernel<u8, 8>, 2^24 samples, %d taps; EXP 1 = no input loads, 2 = no matrix loop, 4 = no mixer, 8 = no stores, 32 = wave priorities, 128 = no explicit vmcnt(0)\n", ntaps);
    constexpr int LIB5 = 8192 | 16384 | 65536 | 131072 | 262144 | (1 << 22);  // round 5's library
    constexpr int LIB6 = LIB5 | (1 << 25) | (1 << 26);                          // round 6: the landing's flip as a branch, the one-Shift program peeled
    if (getenv("PASSES")) {  // the per-pass stamp table (BATCH=4 for the benchmarked form)
        run<LIB6, 0>(in, out, taps, tab, n, ntaps, true);
        run<LIB6 | 64 | (1 << 24), 0>(in, out, taps, tab, n, ntaps, true);
        if (getenv("SPLIT")) {  // where the power goes: the same launch without its mixer, without its stores, without both, without the input
            run<LIB6 | 4, 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB6 | 8, 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB6 | 4 | 8, 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB6 | 1, 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB6 | 2, 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB6, 0>(in, out, taps, tab, n, ntaps, true);
        }
        return 0;
    }
    if (getenv("CLOCKS")) {  // the shader clock the chip holds under the kernel and under the kernel without one of its parts
        constexpr int ST = 64 | (1 << 24);
        for (int r = 0; r < 2; r++) {
            run<LIB6 | ST, 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB6 | ST | 4, 0>(in, out, taps, tab, n, ntaps, true);       // no mixer
            run<LIB6 | ST | 8, 0>(in, out, taps, tab, n, ntaps, true);       // no stores
            run<LIB6 | ST | (1 << 27), 0>(in, out, taps, tab, n, ntaps, true);  // the planes' combination in float32 (timing only: not exact)
            run<LIB6 | ST, 0>(in, out, taps, tab, n, ntaps, false);          // no Shift stage at all
        }
        return 0;
    }
    if (getenv("AB8")) {  // the MFMAs of a step in snake order (one operand changes between neighbours)
        for (int r = 0; r < 5; r++) {
            run<LIB6, 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB6 | (1 << 28), 0>(in, out, taps, tab, n, ntaps, true);
        }
        return 0;
    }
    if (getenv("AB7")) {  // what the float64 plane combination costs in the benchmarked form: the same kernel with a float32 one (NOT exact: timing only)
        for (int r = 0; r < 5; r++) {
            run<LIB6, 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB6 | (1 << 27), 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB6 | 4, 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB6 | 4 | (1 << 27), 0>(in, out, taps, tab, n, ntaps, true);
        }
        return 0;
    }
    if (getenv("AB6")) {  // round 6's instruction cuts against round 5's library, interleaved (BATCH=4 for the benchmarked form)
        for (int r = 0; r < 4; r++) {
            run<LIB5, 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB5 | (1 << 25), 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB5 | (1 << 26), 0>(in, out, taps, tab, n, ntaps, true);
            run<LIB6, 0>(in, out, taps, tab, n, ntaps, true);
        }
        return 0;
    }
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

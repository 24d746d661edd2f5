// plan_fuzz.cpp -- the host-only planners under AddressSanitizer + UndefinedBehaviorSanitizer (the GPU box offers
// no device sanitizers; the reference runs `go test -race` on everything, Makefile:35,42).  Built by
// tests/test_host_sanitizers.py:
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I go-sdr_amd/csrc -I include
//         tests/host/plan_fuzz.cpp go-sdr_amd/csrc/hz_host.cpp -o plan_fuzz
// Random sample rates (incl. power-of-two rates, whose clock steps never change), clock starts (0, binade
// edges, just below the 2 pi wrap), buffer lengths, tap counts and tables present / absent; asserts
//   * hzsdr_nco_segments: the counts add up to n, run k starts where run k - 1 ended, and the runs reproduce the
//     serial recurrence of stream/shifter.go:76-79 (ts += 1/fs; if ts > 2 pi { ts -= 2 pi }) sample for sample;
//   * mm2::plan_call (hz_firmm2_plan.h): every output lies in EXACTLY one run's valid range or one fix-up
//     interval; ranges and intervals ascend; a run's passes hold its outputs; tile alignment; task counts;
//     windows of valid outputs lie inside their run (or the raw history when the call continues a run);
//   * mm::plan_chunks (hz_firmm_plan.h, the chunk form's planner, factors 8 ... 64): every output lies in EXACTLY
//     one place -- its chunk's owner run's valid range or one fix-up interval --, the owners ascend with the chunks,
//     a run's first chunk is where its ownership starts, task counts add up;
//   * mm::digit_table (hz_firmm_plan.h): the four balanced base-256 digits of every table entry recombine to the
//     quantised tap they were cut from, |q| <= 2^30, in both kernels' layouts; the constant term; step factors of
//     unit modulus to 2^-23.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "hz_firmm_plan.h"
#include "hzsdr.h"

using namespace hz;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double urand() { return (double)(rnd() >> 11) / 9007199254740992.0; }

#define REQUIRE(cond)                                                                  \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            printf("FAIL %s:%d: %s (case %d)\n", __FILE__, __LINE__, #cond, g_case);   \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)
static int g_case = 0;

int main(int argc, char **argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 3000;
    const double tau = 6.283185307179586476925286766559;
    const uint64_t rates[] = {250000, 1000000, 1048576, 1800000, 2048000, 2097152, 2400000, 8000000, 20000000, 200000000};
    long planned = 0, fell_back = 0, chunks_planned = 0, tables = 0, combine_refused = 0;
    for (g_case = 0; g_case < cases; g_case++) {
        const uint64_t fs = rates[rnd() % (sizeof rates / sizeof rates[0])];
        double ts0;
        switch (rnd() % 6) {
        case 0: ts0 = 0.0; break;
        case 1: ts0 = ldexp(1.0, -(int)(rnd() % 30)); break;                         // a binade edge
        case 2: ts0 = tau - (double)(rnd() % 100000) / (double)fs; break;            // just below the wrap
        case 3: ts0 = nextafter(ldexp(1.0, (int)(rnd() % 3)), 0.0); break;           // just below a binade edge
        default: ts0 = urand() * tau; break;
        }
        const uint64_t n = (rnd() % 8 == 0) ? (rnd() % 70000) : ((uint64_t)1 << (15 + rnd() % 10)) + 8 * (rnd() % 4096);
        std::vector<hzsdr_nco_segment> segs(4096);
        size_t need = 0;
        double ts_end = 0;
        const int rc = hzsdr_nco_segments(fs, ts0, n, segs.data(), segs.size(), &need, &ts_end);
        REQUIRE(rc == HZSDR_OK);
        REQUIRE(need <= segs.size());
        // ---- the clock runs against the serial recurrence
        uint64_t total = 0;
        for (size_t k = 0; k < need; k++) {
            REQUIRE(segs[k].first == total);
            REQUIRE(segs[k].count > 0);
            total += segs[k].count;
        }
        REQUIRE(total == n);
        if (n <= (1u << 21)) {  // sample for sample (the long buffers: at the runs' ends only)
            double ts = ts0;
            const double inc = 1.0 / (double)fs;
            size_t k = 0;
            for (uint64_t j = 0; j < n; j++) {
                ts += inc;
                if (ts > tau) ts -= tau;
                while (j >= segs[k].first + segs[k].count) k++;
                const double planned_ts = fma((double)(j - segs[k].first), segs[k].step, segs[k].t0);
                REQUIRE(planned_ts == ts);
            }
            REQUIRE(ts == ts_end);
        }
        // ---- the chunk planner (hz_firmm.h's form) over these runs, any of its factors
        if (need <= (size_t)mm::kMaxRuns) {
            static const unsigned factors[] = {8, 16, 24, 32, 40, 48, 64};
            const unsigned Dc = factors[rnd() % 7];
            const int nt = 16 + (int)(rnd() % (Dc == 8 ? 1500 : 4000));
            std::vector<uint64_t> first(need);
            std::vector<const void *> tab(need);
            static const char dummy_c = 0;
            for (size_t k = 0; k < need; k++) {
                first[k] = segs[k].first;
                tab[k] = (segs[k].count >= 8ull * (uint64_t)nt && rnd() % 16 != 0) ? &dummy_c : nullptr;
            }
            const mm::ChunkPlanIn cin{n / Dc * Dc, Dc, nt, rnd() % 2 == 0, (int)need, first.data(), tab.data()};
            mm::Runs R;
            mm::Fix F;
            uint64_t fix_outputs = 0;
            if (mm::plan_chunks(cin, &R, &F, &fix_outputs)) {
                chunks_planned++;
                const uint64_t n_out = cin.n_in / Dc, chunk = (uint64_t)mm::chunk_out(mm::blocks_for((int)Dc));
                REQUIRE(R.n == (int)need && (uint64_t)R.n_wg == (n_out + chunk - 1) / chunk && F.n <= mm::kMaxFix);
                REQUIRE(fix_outputs <= 16384 && fix_outputs * 8 <= n_out);
                std::vector<uint8_t> cover(n_out, 0);
                uint64_t prev_hi = 0;
                for (int r = 0; r < R.n; r++) {
                    if (R.m_hi[r] == 0) continue;
                    REQUIRE(R.tab[r] != nullptr && R.m_lo[r] < R.m_hi[r] && R.m_hi[r] <= n_out && R.m_lo[r] >= prev_hi);
                    REQUIRE(R.m_lo[r] % mm::kT == 0 && (R.m_hi[r] % mm::kT == 0 || R.m_hi[r] == n_out));
                    const uint64_t run_end = (size_t)r + 1 < need ? first[r + 1] : cin.n_in;
                    const int64_t w_lo = (int64_t)Dc * R.m_lo[r] - (nt - 1);
                    REQUIRE(w_lo >= (int64_t)first[r] || (r == 0 && cin.cont));
                    REQUIRE((uint64_t)Dc * (R.m_hi[r] - 1) < run_end);
                    prev_hi = R.m_hi[r];
                }
                // a chunk's owner: the last run whose first chunk is at or before it (the kernel's lookup)
                int prev_owner = 0;
                for (uint64_t ch = 0; ch < (uint64_t)R.n_wg; ch++) {
                    int owner = -1;
                    for (int r = 0; r < R.n; r++)
                        if ((uint64_t)R.wg_first[r] <= ch) owner = r;
                    REQUIRE(owner >= prev_owner);
                    prev_owner = owner;
                    const uint64_t cs = ch * chunk, ce = cs + chunk < n_out ? cs + chunk : n_out;
                    const uint64_t vlo = R.m_lo[owner] > cs ? R.m_lo[owner] : cs, vhi = R.m_hi[owner] < ce ? R.m_hi[owner] : ce;
                    for (uint64_t m = vlo; m < vhi; m++) cover[m]++;
                }
                int tasks = 0;
                uint64_t fixed = 0;
                uint32_t prev_b = 0;
                for (int k = 0; k < F.n; k++) {
                    REQUIRE(F.m_a[k] < F.m_b[k] && F.m_b[k] <= n_out && F.m_a[k] >= prev_b && F.wg_first[k] == tasks);
                    tasks += (int)((F.m_b[k] - F.m_a[k] + mm::kFixOut - 1) / mm::kFixOut);
                    for (uint64_t m = F.m_a[k]; m < F.m_b[k]; m++) cover[m]++;
                    fixed += F.m_b[k] - F.m_a[k];
                    prev_b = F.m_b[k];
                }
                REQUIRE(tasks == F.n_wg && fixed == fix_outputs);
                for (uint64_t m = 0; m < n_out; m++) REQUIRE(cover[m] == 1);
            }
        }
        // ---- a digit table (every eighth case: the tables are 20 ... 70 KB)
        if (g_case % 8 == 0) {
            static const unsigned factors[] = {8, 16, 32, 64};
            const bool v2 = rnd() % 2 == 0;
            const int Dt = v2 ? (rnd() % 2 ? 8 : 16) : (int)factors[rnd() % 4];
            const int nt = 16 + (int)(rnd() % (v2 ? (Dt == 16 ? 1024 : 1100) : 2000));
            std::vector<double> taps(2 * (size_t)nt);
            const double amp = ldexp(1.0, (int)(rnd() % 40) - 30);
            for (auto &t : taps) t = (double)(float)((urand() - 0.5) * amp);
            const double scale = rnd() % 2 ? 1.0 / 127.5 : 1.0 / 128.0;
            const int S = mm::digit_shift(taps.data(), (size_t)nt, scale);
            const unsigned off = (unsigned)((nt - 1 + Dt - 1) / Dt * Dt);
            mm::Geom g;
            if (v2) {
                const mm2::Geom g2 = mm2::make_geom(nt, Dt, off, S);
                g.ntaps = g2.ntaps, g.w0 = g2.w0, g.ks = g2.ks, g.ne = g2.ne, g.e0 = g2.e0, g.shift = g2.shift, g.off = g2.off;
            } else {
                g = mm::make_geom(nt, Dt, off, S);
            }
            const double stepv = 1.0 / (double)fs, omega = tau * 1e6 * (urand() - 0.5);
            std::vector<int64_t> q;
            const std::vector<uint8_t> tab = mm::digit_table(g, Dt, taps.data(), scale, stepv, omega, scale != 1.0 / 128.0, v2, &q);
            REQUIRE(tab.size() == mm::table_bytes(g.ne));
            int64_t sr = 0, si = 0;
            for (int k = 0; k < nt; k++) {
                REQUIRE(llabs(q[2 * k]) <= (1ll << 30) && llabs(q[2 * k + 1]) <= (1ll << 30));
                sr += q[2 * k], si += q[2 * k + 1];
            }
            size_t seen = 0;
            int64_t top2[2] = {0, 0};  // sum over an output part's entries of |256 d0 + d1|
            for (int E = 0; E < g.ne; E++)
                for (int pout = 0; pout < 2; pout++)
                    for (int e = 0; e < 16; e++) {
                        const int kap = 8 * (E - g.e0) + g.w0 - (e >> 1), pin = e & 1;
                        int64_t v = 0;
                        for (int d = 0; d < 4; d++) {
                            const size_t at = v2 ? ((((size_t)(d >> 1) * g.ne + E) * 2 + pout) * 2 + (d & 1)) : (((size_t)d * g.ne + E) * 2 + pout);
                            v = v * 256 + (int8_t)tab[at * 16 + e];
                            if (d == 1) top2[pout] += llabs(v);
                        }
                        if (kap < 0 || kap >= nt) {
                            REQUIRE(v == 0);
                            continue;
                        }
                        const int64_t want = pout == 0 ? (pin == 0 ? q[2 * kap] : -q[2 * kap + 1]) : (pin == 0 ? q[2 * kap + 1] : q[2 * kap]);
                        REQUIRE(v == want);
                        seen++;
                    }
            REQUIRE(seen == 4 * (size_t)nt);  // every tap, (re, im) x (I, Q), exactly once
            // mm::int32_combine_ok: where it says yes, the int32 sum of the two top planes holds for every input byte
            if (mm::int32_combine_ok(taps.data(), (size_t)nt, scale, S)) REQUIRE(128 * top2[0] < (1ll << 31) && 128 * top2[1] < (1ll << 31));
            else combine_refused++;
            double dc[2];
            memcpy(dc, tab.data() + (size_t)4 * g.ne * 32, 16);
            REQUIRE(scale == 1.0 / 128.0 ? (dc[0] == 0.0 && dc[1] == 0.0) : (dc[0] == 0.5 * (double)(sr - si) && dc[1] == 0.5 * (double)(sr + si)));
            if (v2) {
                float wf[8][4];
                memcpy(wf, tab.data() + (size_t)4 * g.ne * 32 + 16, sizeof wf);
                REQUIRE(wf[0][0] == 1.0f && wf[0][1] == 0.0f);  // (k = 0)
                for (int i = 0; i < 8; i++) {
                    const double c = (double)wf[i][0] + (double)wf[i][2], s = (double)wf[i][1] + (double)wf[i][3];
                    REQUIRE(fabs(c * c + s * s - 1.0) < 1e-12);
                }
            }
            tables++;
        }
        // ---- the pass planner over these runs
        if (need > (size_t)mm2::kMaxClockRuns) continue;
        const unsigned D = rnd() % 3 == 0 ? 16 : 8;  // (the persistent-pass kernel's factors: hz_firmm2_plan.h factor_ok)
        const int ntaps = 16 + (int)(rnd() % (D == 16 ? 1024 : 1140));
        std::vector<uint64_t> first(need);
        std::vector<double> t0(need), step(need);
        std::vector<const void *> tab(need);
        static const char dummy = 0;
        for (size_t k = 0; k < need; k++) {
            first[k] = segs[k].first, t0[k] = segs[k].t0, step[k] = segs[k].step;
            // tables exist for the long runs (prepare_mm_tables), sometimes not at all
            tab[k] = (segs[k].count >= 8ull * (uint64_t)ntaps && rnd() % 16 != 0) ? &dummy : nullptr;
        }
        mm2::ClockRuns cr{(int)need, first.data(), t0.data(), step.data(), tab.data()};
        mm2::PlanIn in{};
        in.n_in = n / D * D, in.D = D, in.ntaps = ntaps, in.has_shift = true;
        in.back = rnd() % 3 == 0 ? 0 : (rnd() % 2 ? (uint64_t)ntaps + rnd() % 64 : 1 + rnd() % (uint64_t)ntaps);  // none, all, part of a window
        in.shift_op = rnd() % 4 ? 0 : -1, in.tau = tau * 2.5e6 * (urand() - 0.5), in.n_ops = 1, in.max_grid = 1 + (int)(rnd() % 256);
        mm2::Plan L;
        mm2::Fix F;
        uint64_t fix_outputs = 0;
        const bool ok = mm2::plan_call(in, cr, &L, &F, &fix_outputs);
        if (!ok) {
            fell_back++;
            continue;
        }
        planned++;
        const uint64_t n_out = in.n_in / D, pass = (uint64_t)mm2::pass_out((int)D);
        REQUIRE(L.n >= 1 && L.n <= mm2::kMaxRuns && F.n <= mm2::kMaxFix);
        REQUIRE((uint64_t)L.n_pass == (n_out + pass - 1) / pass);
        REQUIRE(L.grid >= 1 && L.grid <= in.max_grid && L.grid <= L.n_pass);
        REQUIRE(F.n_task == L.n_task && F.n_task <= 4 * L.grid);
        std::vector<uint8_t> cover(n_out, 0);
        uint64_t prev_hi = 0, fixed = 0;
        int tasks = 0;
        for (int r = 0; r < L.n; r++) {
            const mm2::Run &u = L.run[r];
            REQUIRE(u.tab != nullptr);
            REQUIRE(u.m_lo < u.m_hi && u.m_hi <= n_out && u.m_lo >= prev_hi);
            REQUIRE(u.m_lo % mm2::kT == 0 && (u.m_hi % mm2::kT == 0 || u.m_hi == n_out));
            REQUIRE((uint64_t)u.pass_first == u.m_lo / pass && (uint64_t)u.pass_end == (u.m_hi + pass - 1) / pass);
            REQUIRE(L.pass_first[r] == u.pass_first && L.pass_end[r] == u.pass_end);
            REQUIRE(u.seg >= 0 && u.seg < (int)need && u.first == first[u.seg]);
            // every valid output's window lies in the run (run 0 of a continuing call: or in the raw history)
            const uint64_t run_end = (size_t)u.seg + 1 < need ? first[u.seg + 1] : in.n_in;
            const int64_t w_lo = (int64_t)D * u.m_lo - (ntaps - 1);
            REQUIRE(w_lo >= (int64_t)u.first || (u.seg == 0 && w_lo >= -(int64_t)in.back));
            REQUIRE((uint64_t)D * (u.m_hi - 1) < run_end);
            REQUIRE(mm2::run_line(t0[u.seg], step[u.seg]).full / D >= mm2::kShortRun);  // (short lines go to the tasks)
            for (uint64_t m = u.m_lo; m < u.m_hi; m++) cover[m]++;
            prev_hi = u.m_hi;
        }
        uint32_t prev_b = 0;
        for (int k = 0; k < F.n; k++) {
            REQUIRE(F.m_a[k] < F.m_b[k] && F.m_b[k] <= n_out && F.m_a[k] >= prev_b);
            REQUIRE(F.task_first[k] == tasks);
            tasks += (int)((F.m_b[k] - F.m_a[k] + mm2::kFixOut - 1) / mm2::kFixOut);
            for (uint64_t m = F.m_a[k]; m < F.m_b[k]; m++) cover[m]++;
            fixed += F.m_b[k] - F.m_a[k];
            prev_b = F.m_b[k];
        }
        REQUIRE(tasks == F.n_task && fixed == fix_outputs);
        for (uint64_t m = 0; m < n_out; m++) REQUIRE(cover[m] == 1);
        // the workgroups' pass ranges (hz_firmm2.h) partition the passes
        const uint32_t NP = (uint32_t)L.n_pass, per = NP / (uint32_t)L.grid, rem = NP - per * (uint32_t)L.grid;
        uint32_t at = 0;
        for (uint32_t wb = 0; wb < (uint32_t)L.grid; wb++) {
            const uint32_t pb0 = wb * per + (wb < rem ? wb : rem), pb1 = pb0 + per + (wb < rem ? 1u : 0u);
            REQUIRE(pb0 == at && pb1 > pb0);
            at = pb1;
        }
        REQUIRE(at == NP);
    }
    {
        // directed: the benchmarked call in which the clock wraps (20 Msps, 2^24 samples, the wrap 0.4 s in).  The
        // binades behind the wrap double from a few samples up: the short ones belong to the fix-up tasks (one per
        // workgroup) and not, with a table each, to the call's first workgroup.
        const uint64_t fs = 20000000, n = (uint64_t)1 << 24;
        std::vector<hzsdr_nco_segment> segs(4096);
        size_t need = 0;
        double ts_end = 0;
        REQUIRE(hzsdr_nco_segments(fs, tau - 0.4, n, segs.data(), segs.size(), &need, &ts_end) == HZSDR_OK);
        REQUIRE(need <= (size_t)mm2::kMaxClockRuns);
        std::vector<uint64_t> first(need);
        std::vector<double> t0(need), step(need);
        std::vector<const void *> tab(need);
        static const char dummy2 = 0;
        for (size_t k = 0; k < need; k++) first[k] = segs[k].first, t0[k] = segs[k].t0, step[k] = segs[k].step, tab[k] = segs[k].count >= 8 * 1024 ? &dummy2 : nullptr;
        mm2::ClockRuns cr{(int)need, first.data(), t0.data(), step.data(), tab.data()};
        mm2::PlanIn in{};
        in.n_in = n, in.D = 8, in.ntaps = 1024, in.has_shift = true, in.back = 1024, in.shift_op = 0, in.tau = -tau * 2.5e6, in.n_ops = 1, in.max_grid = 256;
        mm2::Plan L;
        mm2::Fix F;
        REQUIRE(mm2::plan_call(in, cr, &L, &F, nullptr));
        REQUIRE(F.n_task <= L.grid);
        int with_table = 0;
        for (size_t k = 0; k < need; k++) with_table += tab[k] != nullptr;
        REQUIRE(L.n < with_table);  // some runs with a table went to the tasks ...
        int short_planned = 0;  // (no run whose line is short stays on the matrix path)
        for (int r = 0; r < L.n; r++) short_planned += mm2::run_line(t0[L.run[r].seg], step[L.run[r].seg]).full / 8 < mm2::kShortRun;
        REQUIRE(short_planned == 0);
        printf("plan_fuzz wrap call: %zu clock runs, %d with a table, %d on the matrix path (%d of them short), %d fix-up tasks\n", need, with_table, L.n, short_planned, F.n_task);
    }
    // ---- a stream cut two ways: ONE call over k buffers against k single calls (hzsdr_chain_run_batch against
    // hzsdr_chain_run).  What the kernel derives from the plan must be the same for every output: matrix path or task,
    // the run's table (its step), and the mixer's 64-bit phase at the output -- then the two forms agree bit for bit.
    // The chain's state between calls (hz_chain_fir.hip: fir_run / mm2_plan) is restated here: the clock, and how many
    // samples of the current run the raw history holds.
    long cut_cases = 0, cut_outputs = 0, sure_cases = 0;
    for (g_case = 0; g_case < cases / 4; g_case++) {
        const uint64_t fs = rates[rnd() % (sizeof rates / sizeof rates[0])];
        const unsigned D = rnd() % 3 == 0 ? 16 : 8;
        const int ntaps = 16 + (int)(rnd() % (D == 16 ? 1024 : 1140));
        const unsigned off = (unsigned)((ntaps - 1 + D - 1) / D * D);
        const size_t nbuf = 2 + rnd() % 7;
        const uint64_t ppb = 8 + rnd() % 120, n_each = ppb * (uint64_t)mm2::pass_out((int)D) * D;
        double ts0;
        switch (rnd() % 4) {
        case 0: ts0 = tau - (double)(rnd() % (n_each * nbuf)) / (double)fs; break;  // the wrap somewhere in the stream
        case 1: ts0 = ldexp(1.0, -(int)(rnd() % 6)) - (double)(rnd() % (n_each * nbuf)) / (double)fs; if (ts0 < 0) ts0 = 0.3; break;  // a binade edge
        default: ts0 = urand() * tau; break;
        }
        const double tsh = tau * 2.5e6 * (urand() - 0.5);
        struct OutInfo { uint8_t kind; double step; uint64_t phase; };
        uint64_t pp_fix_total = 0;
        int pp_fn = 0, pp_ntask = 0;  // (of the last piece planned)
        auto plan_piece = [&](double ts_in, uint64_t n, uint64_t back, std::vector<OutInfo> &info, double *ts_out, uint64_t *last_len, bool *last_is_run0, bool *ok) {
            std::vector<hzsdr_nco_segment> segs(4096);
            size_t need = 0;
            REQUIRE(hzsdr_nco_segments(fs, ts_in, n, segs.data(), segs.size(), &need, ts_out) == HZSDR_OK);
            *ok = false;
            if (need > (size_t)mm2::kMaxClockRuns) return;
            std::vector<uint64_t> first(need);
            std::vector<double> t0(need), step(need);
            std::vector<const void *> tab(need);
            static const char dummy3 = 0;
            for (size_t k = 0; k < need; k++) {
                first[k] = segs[k].first, t0[k] = segs[k].t0, step[k] = segs[k].step;
                // (prepare_mm_tables: a table per STEP whose line is long -- a property of the stream, not of the call)
                tab[k] = mm2::run_line(t0[k], step[k]).full >= 8ull * (uint64_t)ntaps ? &dummy3 : nullptr;
            }
            mm2::ClockRuns cr{(int)need, first.data(), t0.data(), step.data(), tab.data()};
            mm2::PlanIn in{};
            in.n_in = n, in.D = D, in.ntaps = ntaps, in.has_shift = true, in.shift_op = 0, in.tau = tsh, in.n_ops = 1, in.max_grid = 256;
            in.back = (back && mm2::continues(ts_in, t0[0], step[0])) ? (back < off ? back : off) : 0;
            mm2::Plan L;
            mm2::Fix F;
            if (!mm2::plan_call(in, cr, &L, &F, &pp_fix_total)) return;
            pp_fn = F.n, pp_ntask = F.n_task;
            *ok = true;
            info.assign(n / D, OutInfo{0, 0.0, 0});
            for (int r = 0; r < L.n; r++)
                for (uint64_t m = L.run[r].m_lo; m < L.run[r].m_hi; m++)
                    info[m] = OutInfo{1, step[L.run[r].seg], L.run[r].phi + (uint64_t)D * m * L.run[r].dphi};
            const size_t last = need - 1;
            *last_len = n - first[last];
            *last_is_run0 = last == 0 && in.back > 0;
        };
        // the batch: one call (the chain's state in front of it: nothing -- a fresh stream or a set_time)
        std::vector<OutInfo> big;
        double ts_big;
        uint64_t ll;
        bool l0, ok_big;
        if (!mm2::batch_ok(n_each, D, nbuf)) continue;
        plan_piece(ts0, n_each * nbuf, 0, big, &ts_big, &ll, &l0, &ok_big);
        if (!ok_big) continue;
        // (hz_chain_fir.hip: mm2_plan's shortcut -- when even ALL the call's fix-up outputs in one buffer would pass that
        // buffer's checks, every single call plans: checked below)
        const uint64_t out_each = n_each / D;
        const int grid_each = (int)(ppb < 256 ? ppb : 256);
        const bool surely = pp_fix_total * 8 <= out_each && pp_fn + 1 <= mm2::kMaxFix && pp_ntask + 2 <= 4 * grid_each;
        // the single calls
        double ts = ts0;
        uint64_t rh_len = 0;
        bool all_ok = true;
        std::vector<OutInfo> cat;
        for (size_t j = 0; j < nbuf && all_ok; j++) {
            std::vector<OutInfo> piece;
            double ts_next;
            bool okp;
            plan_piece(ts, n_each, j == 0 ? 0 : rh_len, piece, &ts_next, &ll, &l0, &okp);
            if (!okp) {
                all_ok = false;  // (a single call that keeps the transform kernels: other arithmetic, nothing to compare)
                break;
            }
            rh_len = l0 ? rh_len + ll : ll;
            ts = ts_next;
            cat.insert(cat.end(), piece.begin(), piece.end());
        }
        REQUIRE(all_ok || !surely);
        if (surely) sure_cases++;
        if (!all_ok) continue;
        REQUIRE(ts == ts_big && cat.size() == big.size());
        for (size_t m = 0; m < big.size(); m++) {
            REQUIRE(big[m].kind == cat[m].kind);
            if (big[m].kind) REQUIRE(big[m].step == cat[m].step && big[m].phase == cat[m].phase);
        }
        cut_cases++, cut_outputs += (long)big.size();
    }
    REQUIRE(cut_cases > cases / 16);
    // ---- a pass's buffer by the host's reciprocal (hz_firmm2.h: buf_of): exact wherever batch_ok says yes
    for (g_case = 0; g_case < 2000; g_case++) {
        const size_t nbuf = 1 + rnd() % 8;
        const uint64_t ppb = g_case < 64 ? 1 + (uint64_t)g_case : 1 + rnd() % 40000, n_each = ppb * 512 * 8;
        if (!mm2::batch_ok(n_each, 8, nbuf)) {
            REQUIRE(ppb * 512 < 4096 || nbuf * ppb * ppb >= (1ull << 32) || n_each * nbuf / 8 >= (1ull << 31));
            continue;
        }
        std::vector<const void *> ins(nbuf, (const void *)(uintptr_t)0x10000000);
        std::vector<void *> outs(nbuf, (void *)(uintptr_t)0x20000000);
        const mm2::Batch B = mm2::make_batch(ins.data(), outs.data(), nbuf, n_each, 8);
        REQUIRE(B.ppb == ppb);
        const uint64_t np = nbuf * ppb, stride = np > 200000 ? 7 : 1;
        for (uint64_t p = 0; p < np; p += stride) REQUIRE((((uint64_t)(uint32_t)p * B.rcp) >> 32) == p / ppb);
        for (uint64_t j = 1; j <= nbuf; j++) REQUIRE((((j * ppb - 1) * B.rcp) >> 32) == j - 1);  // (every buffer's last pass)
    }
    printf("plan_fuzz cuts: %ld streams planned as one call and as single calls agree output for output (%ld outputs); %ld decided by the call's own counts\n", cut_cases, cut_outputs, sure_cases);
    printf("plan_fuzz ok: %d cases, %ld planned, %ld kept the transforms; %ld chunk plans, %ld digit tables (%ld refused the int32 plane sum)\n", cases, planned, fell_back, chunks_planned, tables, combine_refused);
    return 0;
}

// mm2_glitch.hip -- hunts the non-repeatable pass of hz::mm2::fir_mm2_kernel ("A hazard,
// and a known issue") at the level of the int32 accumulators, thousands of launches per second.
//
//   leg 1 (EXP 256): the library's kernel with two checksums of a lane's 64 accumulator registers per pass --
//          sum and index-weighted sum, taken right behind the matrix loop and again behind the landing of the
//          next pass.  The first launch on a buffer stores them, every later launch compares in the kernel
//          and appends a record (pass, lane, wave, workgroup, HW_ID, XCC_ID, both pairs, the stored pairs) on
//          a mismatch.  The host then says for every record: which register (weighted / plain difference),
//          early and late alike (the sum IS wrong) or early only (a stale read), and -- with the exact
//          per-step contributions recomputed from the table and the input bytes -- which window step and
//          which half of its bytes the difference equals.
//   leg 2 (EXP 0): the unmodified kernel; the float outputs of every launch compared bit for bit with the
//          first launch on the same buffer by a second kernel (no instrumentation in the pass loop at all).
// The table holds random digits in all four planes at every step (a dropped or corrupted term is visible
// wherever it happens, not only where the real filter's taps are tiny).
//   tools/bin/mm2_glitch [launches per leg = 20000] [taps = 1024]
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#include "hz_firmm2.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using namespace hz;

constexpr int D = 8, NBUF = 12;
static std::vector<unsigned char> g_in[NBUF], g_tab;
static mm2::Geom g_geom;

// exact accumulator (f, blk, reg) of lane `lane` of pass `pass` on buffer b; per-step/per-half contributions in `part` (68 x 2)
static long long model_acc(int b, uint32_t pass, int f, int blk, int reg, int lane, size_t n_in, std::vector<long long> *part = nullptr) {
    const mm2::Geom &g = g_geom;
    const int row = 8 * (reg >> 2) + 4 * (lane >> 5) + (reg & 3), col = lane & 31;
    const int i_row = 4 * ((row >> 2) & 1) + (row >> 3);
    const long long p0 = 2 * ((long long)D * ((long long)pass * mm2::pass_out(D)) - g.w0), n_bytes = 2 * (long long)n_in;
    long long sum = 0;
    if (part) part->assign((size_t)g.ks * 2, 0);
    for (int s = 0; s < g.ks; s++)
        for (int k = 0; k < 32; k++) {
            const int hh = k >> 4;
            const long long a_addr = 64ll * ((D / 8) * i_row - hh + g.e0) + 16 * (row & 3) + (long long)f * g.ne * 64 - 128ll * s + (k & 15);
            const int A = (signed char)g_tab[(size_t)a_addr];
            const long long p = p0 + 128ll * (col + 32 * blk) + 32ll * s + k;
            const int B = (p >= 0 && p < n_bytes) ? (signed char)(g_in[b][(size_t)p] ^ 0x80) : (signed char)0x80;
            sum += (long long)A * B;
            if (part) (*part)[(size_t)s * 2 + hh] += (long long)A * B;
        }
    return sum;
}

__global__ void compare_kernel(const uint32_t *__restrict__ a, const uint32_t *__restrict__ ref, size_t nwords, unsigned long long *rec, unsigned launch) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        if (a[i] != ref[i]) {
            const unsigned long long s = atomicAdd(&rec[0], 1ull);
            if (s < 100000) {
                rec[8 + 2 * s] = i | ((unsigned long long)launch << 32);
                rec[9 + 2 * s] = (unsigned long long)a[i] | ((unsigned long long)ref[i] << 32);
            }
        }
    }
}


// ---- leg 2: the kernel without instrumentation, outputs compared ---------------------------------------------
template <int EXP>
static void leg2(const char *name, int launches, void *const *in, float2 *out, const float2 *taps, size_t n, const mm2::Geom &g, const mm2::Plan &R,
                 const EwProgram &P, const mm2::Fix &F, size_t lds) {
    const uint32_t n_out = (uint32_t)(n / D);
    auto k = mm2::fir_mm2_kernel<HZSDR_FMT_U8, D, 17, EXP>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint32_t *refo[NBUF];
    unsigned long long *rec;
    const size_t rec_words = 8 + 2 * 100000;
    CK(hipMalloc(&rec, rec_words * 8));
    CK(hipMemset(rec, 0, rec_words * 8));
    for (int b = 0; b < NBUF; b++) CK(hipMalloc(&refo[b], (size_t)n_out * 8));
    unsigned long long *no_stamps = nullptr;
    for (int r = 0; r < launches; r++) {
        const int b = r % NBUF;
        hipLaunchKernelGGL(k, dim3(R.grid), dim3(mm2::kThreads), lds, 0, (const void *)in[b], out, (const float2 *)nullptr, out + n_out,
                           (const uint8_t *)taps, (uint8_t *)(out + n_out + 4096), taps, n, g, R, P, F, mm2::one_buffer(in[b], out, n, D), no_stamps);
        if (r < NBUF) CK(hipMemcpyAsync(refo[b], out, (size_t)n_out * 8, hipMemcpyDeviceToDevice, 0));
        else hipLaunchKernelGGL(compare_kernel, dim3(1024), dim3(256), 0, 0, (const uint32_t *)out, (const uint32_t *)refo[b], (size_t)n_out * 2, rec, (unsigned)r);
    }
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    std::vector<unsigned long long> c(rec_words);
    CK(hipMemcpy(c.data(), rec, rec_words * 8, hipMemcpyDeviceToHost));
    // events = (launch, pass); per event: rows, tiles, words
    struct Ev { int words = 0; unsigned rows = 0; unsigned long long tiles = 0; unsigned maxulp = 0; };
    std::map<unsigned long long, Ev> ev;
    const unsigned long long nrec = std::min<unsigned long long>(c[0], 100000);
    for (unsigned long long r = 0; r < nrec; r++) {
        const unsigned long long i = c[8 + 2 * r] & 0xffffffffull, launch = c[8 + 2 * r] >> 32, m = i / 2;
        Ev &e = ev[(launch << 32) | (m / 512)];
        e.words++;
        e.rows |= 1u << (m % 8);
        e.tiles |= 1ull << ((m % 512) / 8);
        const unsigned a = (unsigned)c[9 + 2 * r], bq = (unsigned)(c[9 + 2 * r] >> 32);
        e.maxulp = std::max(e.maxulp, a > bq ? a - bq : bq - a);
    }
    printf("leg 2, %s: %llu differing output words in %d compared launches: %zu events (launch, pass)\n", name, c[0], launches - NBUF, ev.size());
    std::map<unsigned, int> by_wg, by_slot, by_rows;
    std::map<unsigned long long, int> by_tiles;
    for (auto &kv : ev) {
        const unsigned pass = (unsigned)kv.first;
        by_wg[pass / 16]++;
        by_slot[pass % 16]++;
        by_rows[kv.second.rows]++;
        by_tiles[kv.second.tiles]++;
    }
    printf("    by pass %% 16 (a workgroup's k-th pass):");
    for (auto &kv : by_slot) printf(" %u:%d", kv.first, kv.second);
    printf("\n    by workgroup:");
    for (auto &kv : by_wg) printf(" %u:%d", kv.first, kv.second);
    printf("\n    by set of output rows (bit i = row i):");
    for (auto &kv : by_rows) printf(" %02x:%d", kv.first, kv.second);
    printf("\n    by set of tiles (bit t = tile t of 64):");
    for (auto &kv : by_tiles) printf(" %016llx:%d", kv.first, kv.second);
    printf("\n");
    int shown = 0;
    for (auto &kv : ev) {
        if (shown++ >= 6) break;
        printf("    launch %llu pass %u: %d words, rows %02x, tiles %016llx, largest difference %u ulp\n", kv.first >> 32, (unsigned)kv.first, kv.second.words,
               kv.second.rows, kv.second.tiles, kv.second.maxulp);
    }
    for (int b = 0; b < NBUF; b++) CK(hipFree(refo[b]));
    CK(hipFree(rec));
}

int main(int argc, char **argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20000;
    const int ntaps = argc > 2 ? atoi(argv[2]) : 1024;
    const size_t n = (size_t)1 << 24;
    const unsigned off = (unsigned)((ntaps - 1 + D - 1) / D * D);
    g_geom = mm2::make_geom(ntaps, D, off, 40);
    const mm2::Geom g = g_geom;
    if (g.ks != 68) {
        printf("the straight-line kernel is the 1024-tap window (ks 68), this is ks %d\n", g.ks);
        return 1;
    }
    void *in[NBUF];
    for (int b = 0; b < NBUF; b++) {
        g_in[b].resize(n * 2);
        CK(hipMalloc(&in[b], n * 2 + 65536));
        unsigned s = 77u + (unsigned)b;
        for (size_t i = 0; i < g_in[b].size(); i++) {
            s = s * 1664525u + 1013904223u;
            g_in[b][i] = (unsigned char)(s >> 24);
        }
        CK(hipMemcpy(in[b], g_in[b].data(), n * 2, hipMemcpyHostToDevice));
    }
    g_tab.resize(1 << 20);
    {
        unsigned s = 4242;
        for (auto &v : g_tab) {
            s = s * 1664525u + 1013904223u;
            v = (unsigned char)(s >> 24);
        }
        // (the two doubles behind the entries: finite)
        const double dc[2] = {0.25, -0.5};
        memcpy(&g_tab[(size_t)g.ne * 128], dc, 16);
    }
    void *tab;
    CK(hipMalloc(&tab, g_tab.size()));
    CK(hipMemcpy(tab, g_tab.data(), g_tab.size(), hipMemcpyHostToDevice));
    const uint32_t n_out = (uint32_t)(n / D);
    float2 *out, *taps;
    CK(hipMalloc(&out, ((size_t)n_out + 8192) * 8));
    CK(hipMalloc(&taps, ntaps * 8 + 65536));
    CK(hipMemset(taps, 0, ntaps * 8 + 65536));

    EwProgram P{};
    P.n = 1;
    P.op[0].kind = EW_SHIFT;
    P.op[0].tau_shift = -1.5707963e7;
    P.segs.n = 1;
    P.segs.first[0] = 0;
    P.segs.t0[0] = 0.25;
    P.segs.step[0] = 5e-8;
    mm2::Plan R{};
    mm2::Fix F{};
    R.n = 1;
    R.run[0].tab = tab;
    R.cont = 1;
    R.run[0].m_lo = 0;
    R.run[0].m_hi = n_out;
    R.run[0].pass_first = 0;
    R.n_pass = (int)((n_out + mm2::pass_out(D) - 1) / mm2::pass_out(D));
    R.shift_op = 0;
    R.run[0].pass_end = R.n_pass;
    R.pass_first[0] = 0;
    R.pass_end[0] = R.n_pass;
    R.n_ops = P.n;
    R.grid = 256;
    mm2::phase_fix(P.op[0].tau_shift, P.segs.t0[0], P.segs.step[0], 0, &R.run[0].phi, &R.run[0].dphi);
    const size_t lds = mm2::lds_bytes(D, g.ks, g.ne, g.ntaps);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d kHz; %d launches per leg, %d taps, %d passes per launch\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate, launches, ntaps, R.n_pass);

    // ---- leg 1: checksums -------------------------------------------------------------------------------------
    {
        auto k = mm2::fir_mm2_kernel<HZSDR_FMT_U8, D, 17, 256>;
        CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const size_t ref_bytes = (size_t)R.n_pass * 64 * 16, ctl_words = 8 + 8 * 4000;
        unsigned long long *ctl[NBUF];
        void *ref[NBUF];
        for (int b = 0; b < NBUF; b++) {
            CK(hipMalloc(&ref[b], ref_bytes));
            CK(hipMemset(ref[b], 0, ref_bytes));
            CK(hipMalloc(&ctl[b], ctl_words * 8));
            CK(hipMemset(ctl[b], 0, ctl_words * 8));
            const unsigned long long head[2] = {0ull, (unsigned long long)(uintptr_t)ref[b]};
            CK(hipMemcpy(ctl[b], head, 16, hipMemcpyHostToDevice));
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int r = 0; r < launches; r++) {
            const int b = r % NBUF;
            if (r == NBUF) {
                CK(hipDeviceSynchronize());
                const unsigned long long one = 1;
                for (int q = 0; q < NBUF; q++) CK(hipMemcpy(ctl[q], &one, 8, hipMemcpyHostToDevice));
                CK(hipEventRecord(e0, 0));
            }
            hipLaunchKernelGGL(k, dim3(R.grid), dim3(mm2::kThreads), lds, 0, (const void *)in[b], out, (const float2 *)nullptr, out + n_out,
                               (const uint8_t *)taps, (uint8_t *)(out + n_out + 4096), taps, n, g, R, P, F, mm2::one_buffer(in[b], out, n, D), ctl[b]);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        float ms = 0;
        if (launches > NBUF) CK(hipEventElapsedTime(&ms, e0, e1));
        // the model against the stored checksums (two lanes of a middle pass)
        {
            std::vector<int> rf((size_t)R.n_pass * 64 * 4);
            CK(hipMemcpy(rf.data(), ref[0], ref_bytes, hipMemcpyDeviceToHost));
            int ok = 0, tot = 0;
            for (uint32_t pass : {5u, 2049u})
                for (int lane : {3, 50}) {
                    long long s1 = 0, s2 = 0;
                    for (int f = 0; f < 2; f++)
                        for (int blk = 0; blk < 2; blk++)
                            for (int q = 0; q < 16; q++) {
                                const long long a = model_acc(0, pass, f, blk, q, lane, n);
                                s1 += a, s2 += (1 + q + 16 * (blk + 2 * f)) * a;
                            }
                    const int *rr = &rf[((size_t)pass * 64 + lane) * 4];
                    tot++;
                    ok += (int)s1 == rr[0] && (int)s2 == rr[1] && rr[2] == rr[0] && rr[3] == rr[1];
                }
            printf("leg 1: host model of the accumulators agrees with the stored checksums on %d of %d probes\n", ok, tot);
        }
        unsigned long long total = 0;
        std::map<unsigned long long, int> by_cu;
        for (int b = 0; b < NBUF; b++) {
            std::vector<unsigned long long> c(ctl_words);
            CK(hipMemcpy(c.data(), ctl[b], ctl_words * 8, hipMemcpyDeviceToHost));
            total += c[2];
            for (unsigned long long r = 0; r < std::min<unsigned long long>(c[2], 4000); r++) {
                const unsigned long long *q = &c[8 + 8 * r];
                const uint32_t pass = (uint32_t)q[0];
                const int lane = (int)((q[0] >> 32) & 0xff), wave = (int)(q[0] >> 40);
                const unsigned hwid = (unsigned)(q[1] >> 32), xcc = (unsigned)q[7];
                const int e1g = (int)(uint32_t)q[2], e2g = (int)(uint32_t)(q[2] >> 32), l1g = (int)(uint32_t)q[3], l2g = (int)(uint32_t)(q[3] >> 32);
                const int e1r = (int)(uint32_t)q[4], e2r = (int)(uint32_t)(q[4] >> 32), l1r = (int)(uint32_t)q[5], l2r = (int)(uint32_t)(q[5] >> 32);
                by_cu[((unsigned long long)xcc << 32) | (hwid & 0xfffff0f0u)]++;  // (wave slot and queue bits masked roughly)
                if (r >= 40) continue;
                printf("  buf %d pass %u lane %d (quarter %d) wave %d wg %u hw_id %08x xcc %u: early d1 %d d2 %d | late d1 %d d2 %d", b, pass, lane,
                       lane >> 4, wave, (unsigned)q[1], hwid, xcc, e1g - e1r, e2g - e2r, l1g - l1r, l2g - l2r);
                const int d1 = l1g - l1r ? l1g - l1r : e1g - e1r, d2 = l1g - l1r ? l2g - l2r : e2g - e2r;
                if (d1 != 0 && d2 % d1 == 0 && d2 / d1 >= 1 && d2 / d1 <= 64) {
                    const int idx = d2 / d1 - 1, f = idx / 32, blk = (idx / 16) & 1, reg = idx & 15;
                    std::vector<long long> part;
                    model_acc(b, pass, f, blk, reg, lane, n, &part);
                    printf(" -> ONE register: f %d block %d reg %d (row %d, tile %d)", f, blk, reg, 8 * (reg >> 2) + 4 * (lane >> 5) + (reg & 3), (lane & 31) + 32 * blk);
                    bool hit = false;
                    for (int s = 0; s < g.ks && !hit; s++) {
                        const long long c0 = part[(size_t)s * 2], c1 = part[(size_t)s * 2 + 1];
                        if (-d1 == c0 + c1) printf(", = MINUS step %d's whole term", s), hit = true;
                        else if (-d1 == c0) printf(", = MINUS step %d's bytes 0-15", s), hit = true;
                        else if (-d1 == c1) printf(", = MINUS step %d's bytes 16-31", s), hit = true;
                        else if (d1 == c0 + c1) printf(", = PLUS step %d's whole term (counted twice)", s), hit = true;
                    }
                    if (!hit) {
                        long long tail = 0;
                        for (int s = g.ks - 1; s >= 0 && !hit; s--) {
                            tail += part[(size_t)s * 2] + part[(size_t)s * 2 + 1];
                            if (-d1 == tail) printf(", = MINUS the terms of steps %d .. %d", s, g.ks - 1), hit = true;
                        }
                    }
                    if (!hit) printf(", no single-step explanation (d %d)", d1);
                } else {
                    printf(" -> several registers");
                }
                printf("\n");
            }
        }
        printf("leg 1: %llu mismatching (pass, lane) records in %d compared launches x %d passes x 64 lanes  (%.1f us per launch)\n", total,
               launches - NBUF, R.n_pass, launches > NBUF ? ms * 1e3 / (launches - NBUF) : 0.0);
        for (auto &kv : by_cu) printf("    xcc %llu hw_id&mask %08llx: %d records\n", kv.first >> 32, kv.first & 0xffffffffull, kv.second);
        for (int b = 0; b < NBUF; b++) {
            CK(hipFree(ref[b]));
            CK(hipFree(ctl[b]));
        }
    }
    // ---- leg 3: the first step factor, formed twice (EXP 1024) ---------------------------------------------------
    {
        auto k = mm2::fir_mm2_kernel<HZSDR_FMT_U8, D, 17, 1024>;
        CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const size_t ctl_words = 8 + 8 * 4000;
        unsigned long long *ctl;
        CK(hipMalloc(&ctl, ctl_words * 8));
        CK(hipMemset(ctl, 0, ctl_words * 8));
        for (int r = 0; r < launches; r++)
            hipLaunchKernelGGL(k, dim3(R.grid), dim3(mm2::kThreads), lds, 0, (const void *)in[r % NBUF], out, (const float2 *)nullptr, out + n_out,
                               (const uint8_t *)taps, (uint8_t *)(out + n_out + 4096), taps, n, g, R, P, F, mm2::one_buffer(in[r % NBUF], out, n, D), ctl);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        std::vector<unsigned long long> c(ctl_words);
        CK(hipMemcpy(c.data(), ctl, ctl_words * 8, hipMemcpyDeviceToHost));
        printf("leg 3 (EXP 1024): %llu lanes whose first step factor differs from its second evaluation, %d launches\n", c[2], launches);
        auto f = [](unsigned long long v, int hi) { unsigned u = hi ? (unsigned)(v >> 32) : (unsigned)v; float x; memcpy(&x, &u, 4); return x; };
        for (unsigned long long r = 0; r < std::min<unsigned long long>(c[2], 24); r++) {
            const unsigned long long *q = &c[8 + 8 * r];
            const float c0 = f(q[2], 0), s0 = f(q[2], 1), cs1 = f(q[3], 0), sn1 = f(q[3], 1), cs2 = f(q[4], 0), sn2 = f(q[4], 1);
            const float wx = f(q[5], 0), wy = f(q[5], 1), wz = f(q[6], 0), ww = f(q[6], 1);
            const float cs_hi = fmaf(c0, wx, -(s0 * wy)), sn_hi = fmaf(c0, wy, s0 * wx), cs_lo = fmaf(c0, wz, -(s0 * ww)), sn_lo = fmaf(c0, ww, s0 * wz);
            printf("    pass %u lane %d wave %d wg %u: c0 %.9g s0 %.9g | used cs %.9g sn %.9g | again cs %.9g sn %.9g | w %.9g %.9g %.9g %.9g (again lo %.9g %.9g)\n"
                   "        hi parts %.9g %.9g, lo parts %.9g %.9g; used - hi = %.9g %.9g\n",
                   (unsigned)q[0], (int)((q[0] >> 32) & 0xff), (int)(q[0] >> 40), (unsigned)q[1], c0, s0, cs1, sn1, cs2, sn2, wx, wy, wz, ww, f(q[7], 0),
                   f(q[7], 1), cs_hi, sn_hi, cs_lo, sn_lo, cs1 - cs_hi, sn1 - sn_hi);
        }
        CK(hipFree(ctl));
    }
    leg2<0>("the unmodified kernel", launches, in, out, taps, n, g, R, P, F, lds);
    leg2<4>("no mixer (EXP 4)", launches, in, out, taps, n, g, R, P, F, lds);
    leg2<512>("the mixer's step factors waited for and pinned (EXP 512)", launches, in, out, taps, n, g, R, P, F, lds);
    leg2<128>("no explicit vmcnt(0) (EXP 128)", launches, in, out, taps, n, g, R, P, F, lds);
    return 0;
}

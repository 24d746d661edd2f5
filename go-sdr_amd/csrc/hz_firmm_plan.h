// hz_firmm_plan.h -- the HOST side of the chunk form of the int8 matrix FIR (hz_firmm.h): geometry, the kernel's
// argument structures, the planner that cuts one call into chunks and fix-up tasks, and the builder of a clock
// run's digit table (both kernels' layouts).  No HIP in here: hz_firmm.h includes it, and tests/host/plan_fuzz.cpp
// builds it with gcc -fsanitize=address,undefined and fuzzes the invariants (the GPU box offers no device sanitizers).
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "hz_firmm2_plan.h"

namespace hz {
namespace mm {

constexpr int kT = 16;                        // outputs per tile
// NB 32-tile column blocks per workgroup: 4 at D = 8 (2048 outputs, 34 KB of input: four workgroups per
// CU), 2 at D = 16 (1024 outputs, the same 34 KB: with 4 a chunk was 67 KB, two workgroups per CU --
// one wave per SIMD -- and the matrix form no faster than the transforms at 1024 taps), 1 at D = 32 (512
// outputs, 34 KB: the table loads are no longer shared between blocks, but the matrix work is a quarter)
constexpr int blocks_for(int D) { return D >= 32 ? 1 : D >= 16 ? 2 : 4; }  // (D = 24: 48 KB, 40 / 48 / 64: 40 / 48 / 64 KB)
constexpr int chunk_tiles(int nb) { return 32 * nb; }
constexpr int chunk_out(int nb) { return chunk_tiles(nb) * kT; }
constexpr int kThreads = 128;                 // two waves
constexpr int kFixOut = 16;                   // outputs per fix-up task
constexpr int kMaxRuns = 32;                  // = kNcoMaxSegs (hz_nco.h; hz_firmm.h asserts it)
constexpr int kMaxFix = kMaxRuns + 2;

// geometry of one chain (host: make_geom)
struct Geom {
    int ntaps;
    int w0;      // window start before the tile's first output sample: ntaps-1 rounded up to 8
    int ks;      // 32-byte steps over a tile's window (a multiple of D: whole groups of the matrix loop)
    int ne;      // entries E of one digit of the table
    int e0;      // E of (i = 0, h = 0, s = 0): 2 (ks + 4), the prefetch runs four steps past the end
    int shift;   // S: taps are q = round(h' 2^S)
    unsigned off;  // history length (the chain's `off`)
};

// per clock run: the table, the outputs that take the matrix path (tile-aligned inside the call) and
// the first chunk (2048 outputs of the call's grid) the host gave to the run
struct Runs {
    int n;
    int n_wg;                 // chunk workgroups in total
    int cont;                 // run 0 continues the previous call's last run: windows may reach into the raw history
    const void *tab[kMaxRuns];
    uint32_t m_lo[kMaxRuns], m_hi[kMaxRuns];
    int wg_first[kMaxRuns];
};
// the output ranges computed in reference order, and the first fix-up workgroup of each
struct Fix {
    int n;
    int n_wg;
    uint32_t m_a[kMaxFix], m_b[kMaxFix];
    int wg_first[kMaxFix];
};

constexpr int tile_bytes(int D) { return 2 * D * kT; }
constexpr int pieces_per_tile(int D) { return tile_bytes(D) / 16; }
constexpr size_t chunk_bytes(int D, int ks) {
    return (size_t)(chunk_tiles(blocks_for(D)) - 1) * tile_bytes(D) + 32 * (size_t)ks;
}
// the matrix loop's look-ahead reads two steps past the last window: LDS allocated behind the image
// (a conditional in the unrolled loop instead cost 1.8 us of the kernel)
constexpr size_t kLookAhead = 1024;
// table bytes: 4 digits x ne x 2 parts x 16, then (dc_re, dc_im) as two doubles, then hz_firmm2.h's step factors
constexpr size_t table_bytes(int ne) { return (size_t)4 * ne * 32 + 16 + 128; }
static_assert(table_bytes(100) == mm2::table_bytes(100), "one size for both layouts");

// geometry of a chain on the chunk form (the persistent-pass form: mm2::make_geom); `shift` is S of digit_shift
inline Geom make_geom(int ntaps, int D, unsigned off, int shift) {
    Geom g{};
    g.ntaps = ntaps;
    g.w0 = (ntaps - 1 + 7) / 8 * 8;
    const int window = g.w0 + D * (kT - 1) + 1;  // samples a tile's outputs reach back over
    g.ks = (2 * window + 31) / 32;
    g.ks = (g.ks + D - 1) / D * D;  // whole groups of PPT / 2 = D steps (hz_firmm.h)
    g.e0 = 2 * (g.ks + 4);
    g.ne = g.e0 + (D / 8) * (kT - 1) + 1;
    g.shift = shift;
    g.off = off;
    return g;
}

// S with q = round(h' 2^S), |q| <= 2^30: |h'[k]| <= |h[k]| * scale for every modulation (taps: interleaved re, im)
inline int digit_shift(const double *taps, size_t ntaps, double scale) {
    double hmax = 0.0;
    for (size_t k = 0; k < ntaps; k++) hmax = std::max(hmax, hypot(taps[2 * k], taps[2 * k + 1]));
    hmax *= scale;
    int S = 0;
    if (hmax > 0.0) {
        int e;
        frexp(hmax, &e);  // hmax < 2^e
        S = 30 - e;
    }
    return S < -900 ? -900 : S > 900 ? 900 : S;
}

// The persistent-pass kernel adds the two top digit planes in int32 before the float64 step: hi = 256 sum_0 + sum_1 =
// sum over the 2 ntaps (coefficient, byte) products of an output part of A x, A = 256 d0 + d1.  With q = A 2^16 + (d2 2^8 + d3),
// |d2 2^8 + d3| <= 2^15 + 2^7 and |q| <= |h' component| 2^S + 1/2:  |A| <= |h' component| 2^(S-16) + 0.52, and over a tap's two
// components |re| + |im| <= sqrt(2) |h'| whatever the run's modulation.  |x| <= 128.  True when |hi| < 2^31 for EVERY
// input and modulation (a long flat filter -- 1024 equal taps -- does not pass: such a chain takes the chunk form).
inline bool int32_combine_ok(const double *taps, size_t ntaps, double scale, int S) {
    double sum = 0.0;
    for (size_t k = 0; k < ntaps; k++) sum += hypot(taps[2 * k], taps[2 * k + 1]);
    const double bound = 128.0 * (1.4142135623730951 * sum * scale * ldexp(1.0, S - 16) + 2.0 * (double)ntaps);
    return bound < 2147483648.0;
}

// The digit table of taps[k] * exp(-i omega k step) * scale: 32-bit fixed point q = round(h' 2^S) in four balanced
// base-256 digits, d = 0 most significant.  Chunk form (v2 = false): F[digit][E][part][16]; persistent-pass form:
// T[f][E][part][pl][16] with digit = 2 f + pl (a fragment row holds two digit planes).  Entry E, byte e of part
// `pout` (0: the output's real part, 1: imaginary) is the coefficient of input byte `pin` = e & 1 (I or Q) of tap
// kap = 8 (E - e0) + w0 - (e >> 1):  y_re = h_re x_re - h_im x_im,  y_im = h_im x_re + h_re x_im.
// Then the constant term of a u8 stream (x = (b - 128) + 0.5 (1 + i)) in units of 2^-S, then (v2) the mixer's step
// factors of the run.  `q_out` (optional): the quantised taps (re, im) the digits were cut from.
inline std::vector<uint8_t> digit_table(const Geom &g, int D, const double *taps, double scale, double step, double omega, bool u8_source,
                                        bool v2, std::vector<int64_t> *q_out = nullptr) {
    const size_t ntaps = (size_t)g.ntaps;
    std::vector<int64_t> qr(ntaps), qi(ntaps);
    int64_t sr = 0, si = 0;
    for (size_t k = 0; k < ntaps; k++) {
        const double ph = -omega * ((double)k * step);
        const double cr = cos(ph), ci = sin(ph);
        const double hr = taps[2 * k], hi = taps[2 * k + 1];
        qr[k] = llround(ldexp((hr * cr - hi * ci) * scale, g.shift));
        qi[k] = llround(ldexp((hr * ci + hi * cr) * scale, g.shift));
        sr += qr[k];
        si += qi[k];
    }
    std::vector<uint8_t> tab(table_bytes(g.ne), 0);
    for (int E = 0; E < g.ne; E++)
        for (int pout = 0; pout < 2; pout++)
            for (int e = 0; e < 16; e++) {
                const int kap = 8 * (E - g.e0) + g.w0 - (e >> 1), pin = e & 1;
                if (kap < 0 || kap >= g.ntaps) continue;
                int64_t q = pout == 0 ? (pin == 0 ? qr[kap] : -qi[kap]) : (pin == 0 ? qi[kap] : qr[kap]);
                for (int d = 3; d >= 0; d--) {  // balanced base-256 digits, d = 0 most significant
                    const int64_t r = ((q + 128) & 255) - 128;
                    const size_t at = v2 ? ((((size_t)(d >> 1) * g.ne + E) * 2 + pout) * 2 + (d & 1)) : (((size_t)d * g.ne + E) * 2 + pout);
                    tab[at * 16 + e] = (uint8_t)(int8_t)r;
                    q = (q - r) >> 8;
                }
            }
    double dc[2] = {0.0, 0.0};
    if (u8_source) {
        dc[0] = 0.5 * (double)(sr - si);
        dc[1] = 0.5 * (double)(sr + si);
    }
    memcpy(tab.data() + (size_t)4 * g.ne * 32, dc, 16);
    if (v2) {  // (one Shift stage: omega is its tau; otherwise unused)
        float wf[8][4];
        mm2::step_factors(omega, step, D, wf);
        memcpy(tab.data() + (size_t)4 * g.ne * 32 + 16, wf, sizeof wf);
    }
    if (q_out) {
        q_out->resize(2 * ntaps);
        for (size_t k = 0; k < ntaps; k++) (*q_out)[2 * k] = qr[k], (*q_out)[2 * k + 1] = qi[k];
    }
    return tab;
}

// ---- the planner ---------------------------------------------------------------------------------------------
// The exactly-linear runs of the NCO clock over the call's samples: run r starts at sample first[r]; tab[r] its
// digit table (null: none was prepared).  A chain without a Shift stage passes nr = 1, first[0] = 0.
struct ChunkPlanIn {
    uint64_t n_in;
    unsigned D;
    int ntaps;
    bool cont;   // run 0 continues the run the previous call ended in: its windows may reach into the raw history
    int nr;
    const uint64_t *first;
    const void *const *tab;
};
// Splits the outputs of one call between the matrix path (per clock run) and the fix-up workgroups: the call's
// outputs in chunks of 2048 (D = 16: 1024) on ONE grid, a chunk goes to the run that holds most of it (none: the run
// of the chunk before), what that run does not hold -- windows across a run boundary, the stream's start, runs
// without a table -- goes to fix-up tasks of kFixOut outputs.  false: the call stays on the transform kernels.
inline bool plan_chunks(const ChunkPlanIn &in, Runs *R, Fix *F, uint64_t *fix_outputs = nullptr) {
    memset(R, 0, sizeof *R);
    memset(F, 0, sizeof *F);
    if (fix_outputs) *fix_outputs = 0;
    const uint64_t D = in.D, n_out = in.n_in / D, nt = (uint64_t)in.ntaps;
    const uint64_t chunk = (uint64_t)chunk_out(blocks_for((int)D));  // 2048 outputs at D = 8, 1024 at D = 16
    const uint64_t tile = kT;
    if (n_out < 4096 || n_out >= (1ull << 31)) return false;  // (a call this short is launch-bound either way)
    const int nr = in.nr;
    if (nr < 1 || nr > kMaxRuns) return false;
    R->n = nr;
    R->cont = in.cont ? 1 : 0;
    // outputs whose whole window lies in run r (and that has a table): [lo, hi), on the tile grid
    bool any = false;
    for (int r = 0; r < nr; r++) {
        const uint64_t a = in.first[r];
        const uint64_t b = r + 1 < nr ? in.first[r + 1] : in.n_in;
        uint64_t lo = (r == 0 && in.cont) ? 0 : (a + nt - 1 + D - 1) / D, hi = std::min((b + D - 1) / D, n_out);
        lo = (lo + tile - 1) / tile * tile;
        if (hi < n_out) hi = hi / tile * tile;
        R->m_lo[r] = R->m_hi[r] = 0;
        R->tab[r] = in.tab[r];
        if (!in.tab[r] || hi < lo + 64) continue;  // a run without a table, or too short to bother
        R->m_lo[r] = (uint32_t)lo;
        R->m_hi[r] = (uint32_t)hi;
        any = true;
    }
    if (!any) return false;
    const uint64_t n_chunks = (n_out + chunk - 1) / chunk;
    uint64_t fix_total = 0, fix_a = 0, fix_b = 0;  // the open fix interval [fix_a, fix_b)
    bool too_many = false;
    auto flush_fix = [&]() {
        if (fix_b > fix_a && F->n < kMaxFix) {
            F->m_a[F->n] = (uint32_t)fix_a;
            F->m_b[F->n] = (uint32_t)fix_b;
            F->wg_first[F->n] = F->n_wg;
            F->n_wg += (int)((fix_b - fix_a + kFixOut - 1) / kFixOut);
            F->n++;
        } else if (fix_b > fix_a) {
            too_many = true;  // too many intervals: the call stays on the transform kernels
        }
        fix_a = fix_b = 0;
    };
    auto add_fix = [&](uint64_t a, uint64_t b) {
        if (b <= a) return;
        fix_total += b - a;
        if (fix_b == a && fix_b > fix_a) {
            fix_b = b;
        } else {
            flush_fix();
            fix_a = a;
            fix_b = b;
        }
    };
    int owner = 0, prev_owner = -1;
    for (int r = 0; r < nr; r++) R->wg_first[r] = (int)n_chunks;
    for (uint64_t ch = 0; ch < n_chunks; ch++) {
        const uint64_t cs = ch * chunk, ce = std::min(cs + chunk, n_out);
        // the valid ranges ascend with r, so the owner never goes back and the scan stops at the first
        // run that starts behind the chunk (runs without a range are [0, 0)): linear in chunks + runs
        uint64_t best = 0;
        for (int r = owner; r < nr; r++) {
            if (R->m_hi[r] == 0) continue;
            if (R->m_lo[r] >= ce) break;
            const uint64_t lo = std::max<uint64_t>(R->m_lo[r], cs), hi = std::min<uint64_t>(R->m_hi[r], ce);
            if (hi > lo && hi - lo > best) {
                best = hi - lo;
                owner = r;
            }
        }
        if (owner != prev_owner) {
            for (int r = prev_owner + 1; r <= owner; r++) R->wg_first[r] = (int)ch;  // (runs skipped over own nothing)
            prev_owner = owner;
        }
        const uint64_t vlo = std::max<uint64_t>(R->m_lo[owner], cs), vhi = std::min<uint64_t>(R->m_hi[owner], ce);
        if (vlo < vhi) {
            add_fix(cs, vlo);
            add_fix(vhi, ce);
        } else {
            add_fix(cs, ce);
        }
    }
    flush_fix();
    R->n_wg = (int)n_chunks;
    if (fix_outputs) *fix_outputs = fix_total;
    // the fix-up tasks are the slow way: a call that is mostly boundaries keeps the transforms
    return !too_many && fix_total <= 16384 && fix_total * 8 <= n_out;
}

}  // namespace mm
}  // namespace hz

// hz_firmm2_plan.h -- the HOST side of the persistent-pass matrix FIR (hz_firmm2.h): geometry, the kernel's
// argument structures and the planner that cuts one call into clock runs, passes and fix-up tasks.  No HIP in
// here: the library includes it through hz_firmm2.h, and tests/host/ builds it with gcc -fsanitize=address,
// undefined and fuzzes the planner's invariants (the GPU box offers no device sanitizers).
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace hz {
namespace mm2 {

constexpr int kT = 8;        // outputs per tile
constexpr int kWaves = 8;    // two per SIMD
constexpr int kThreads = 64 * kWaves;
constexpr int kFixOut = 16;  // outputs per fix-up task
constexpr int kHistPer = 64; // samples per history task
constexpr int kMaxRuns = 16;  // clock runs of a call that take the matrix path (more: the call keeps the transforms)
constexpr int kMaxClockRuns = 32;  // hz_nco.h: kNcoMaxSegs, the runs of the clock that travel in the kernel arguments
constexpr int kMaxFix = kMaxClockRuns + 2;
// outputs: a clock run whose WHOLE line (run_line below: the binade's, not the call's piece of it) is shorter than this is
// left to the fix-up tasks.  (Rounds 3-5: 2560, "while there are workgroups left to take a task each" -- a budget
// counted over the call, so the same run could go one way in a call over four buffers and the other way in the four
// single calls.  By the run's own length the choice belongs to the stream: behind a 2 pi wrap the binades double from
// one sample up, the runs below 1280 outputs add up to ~160 tasks, one per workgroup as before.)
constexpr uint64_t kShortRun = 1280;
constexpr int kU = 12;       // 16-byte pieces per lane of a pass image, any window (768 pieces = 12 KB at most)

// (round 6: factor 16 as well -- one column block per pass, 256 outputs; 24 ... 64 keep the chunk kernel: their tiles
// are 24 ... 64 pieces wide and the matrix loop's operand rings are written for groups of 4 or 8 steps)
constexpr bool factor_ok(unsigned D) { return D == 8 || D == 16; }
constexpr int blocks_for(int D) { return D >= 16 ? 1 : 2; }
constexpr int tile_bytes(int D) { return 2 * D * kT; }
constexpr int pass_tiles(int D) { return 32 * blocks_for(D); }
constexpr int pass_out(int D) { return pass_tiles(D) * kT; }
// bytes of a pass: the windows of its tiles, 32 ks = (ks / GS) tile_bytes
constexpr size_t image_bytes(int D, int ks) { return (size_t)(pass_tiles(D) - 1) * tile_bytes(D) + 32 * (size_t)ks; }
// In LDS a tile is followed by 16 bytes of padding: a fragment read (one piece of 16 consecutive tiles per
// lane group) then falls on 16 different 16-byte bank groups (9 t + p mod 16 for 128-byte tiles), and --
// unlike an XOR swizzle -- every address of the matrix loop is ONE per-lane base plus a constant: tile
// n + g, piece 2 j + h, block b sit at base + tile_stride g + 32 j + 32 tile_stride b.
constexpr int tile_stride(int D) { return tile_bytes(D) + 16; }
// (the loop's look-ahead reads two steps past the last window: one tile behind the image)
constexpr size_t slot_bytes(int D, int ks) {
    return ((image_bytes(D, ks) / tile_bytes(D) + 1) * tile_stride(D) + 255) / 256 * 256;
}
// table: T[f][E][part][pl] of 16 bytes (digit plane 2 f + pl, most significant first), then (dc_re, dc_im), then the
// mixer's eight step factors of the run (step_factors below)
constexpr size_t table_bytes(int ne) { return (size_t)ne * 128 + 16 + 128; }
constexpr size_t table_lds(int ne) { return (table_bytes(ne) + 255) / 256 * 256; }
constexpr size_t kCtlBytes = 512;  // the queue's counter
// two tables, the queue's counter, a slot per wave, the fix-up task's window (ntaps + D (kFixOut - 1) samples) and taps
constexpr size_t lds_bytes(int D, int ks, int ne, int ntaps) {
    return 2 * table_lds(ne) + kCtlBytes + kWaves * slot_bytes(D, ks) + ((size_t)(2 * ntaps + D * (kFixOut - 1)) * 8 + 255) / 256 * 256;
}

struct Geom {
    int ntaps;
    int w0;     // window start before a tile's first output sample: ntaps - 1 rounded up to 8
    int ks;     // 32-byte steps over a tile's window (a multiple of D / 2: whole groups)
    int ne;     // entries E of one fragment row of the table
    int e0;     // E of (i = 0, h = 0, s = 0)
    int shift;  // S: taps are q = round(h' 2^S)
    unsigned off;
};

// What a call hands the kernel.  The kernel arguments are read through the scalar cache, where a miss costs
// ~0.9 us while every wave of the chip is asking (measured: five dependent groups of reads in front of the
// first matrix loop took 4.5 us): the plan keeps what the main path reads in its first line and ONE 64-byte
// line per clock run; the program P and the fix-up intervals F are read by the small tasks only.
struct Run {
    const void *tab;       // the run's digit table
    uint32_t m_lo, m_hi;   // the outputs that take the matrix path (tile-aligned inside the call)
    int pass_first;        // the passes (512 outputs of the call's grid) that hold outputs of the run: a pass that
    int pass_end;          // straddles a boundary belongs to BOTH runs and is multiplied twice, once per table
    // programs with exactly ONE Shift stage: the stage's phase tau ts / 2 pi at the BUFFER's first sample by
    // the run's line, and its increment per sample, in 2^-64 turns (mod 1)
    uint64_t phi, dphi;
    uint64_t first;        // the run's first sample
    int seg;               // the run's index in the program's clock table (programs with several Shift stages)
    int pad[3];
};
static_assert(sizeof(Run) == 64, "one scalar-cache line per run");
struct Plan {
    int n;          // clock runs with outputs on the matrix path, in stream order (at most kMaxRuns)
    int n_pass;     // passes of the call
    int cont;       // run 0 continues the previous call's last run: windows may reach into the raw history
    int shift_op;   // index of the program's only Shift stage, -1: none or several
    int n_task;     // fix-up tasks (F)
    int n_ops;      // stages of the program
    float gain;     // (unused)
    int grid;       // workgroups of the launch (gridDim.x is a read of the dispatch packet: another miss)
    int pass_first[kMaxRuns], pass_end[kMaxRuns];  // (copies: a workgroup finds its runs without reading their lines)
    Run run[kMaxRuns];
};
struct Fix {
    int n;
    int n_task;
    uint32_t m_a[kMaxFix], m_b[kMaxFix];
    int task_first[kMaxFix];
};
// A call over SEVERAL buffers of equal length (hzsdr_chain_run_batch): the kernel works on their concatenation -- the
// planner sees one call of nbuf * n samples -- and finds a sample's (an output's) buffer by its index.  Buffer j's
// VIRTUAL base is its address minus j buffers' worth of bytes, so that base + (index in the concatenation) is the
// sample's address; a pass (512 outputs) never straddles two buffers (the host: n / D is a multiple of pass_out), a
// pass IMAGE does at the first pass of every buffer but the first (its window reaches back into the previous
// buffer: the edge landing, piece by piece).  An ordinary call is nbuf = 1 with the buffers' own addresses.
constexpr int kMaxBatch = 8;
struct Batch {
    int nbuf;
    uint32_t ppb;        // passes per buffer
    uint32_t out_each;   // outputs per buffer
    uint32_t rcp;        // floor(2^32 / ppb) + 1: a pass's buffer is mulhi(pass, rcp) -- ONE scalar instruction (batch_ok bounds ppb)
    uint64_t nb;         // bytes per input buffer
    uint64_t vin[kMaxBatch], vout[kMaxBatch];
};
inline Batch one_buffer(const void *in, void *out, uint64_t n_in, unsigned D) {
    Batch B{};
    B.nbuf = 1;
    B.ppb = 0xffffffffu;
    B.out_each = (uint32_t)(n_in / D);
    B.nb = 2 * n_in;
    B.vin[0] = (uint64_t)(uintptr_t)in;
    B.vout[0] = (uint64_t)(uintptr_t)out;
    return B;
}
// nbuf buffers of n_each samples (n_each / D a multiple of the pass: the caller checks batch_ok)
inline bool batch_ok(uint64_t n_each, unsigned D, size_t nbuf) {
    if (!(nbuf >= 1 && nbuf <= (size_t)kMaxBatch && n_each % D == 0 && (n_each / D) % (uint64_t)pass_out((int)D) == 0 && n_each * nbuf / D < (1ull << 31)))
        return false;
    // pass / ppb as mulhi(pass, floor(2^32 / ppb) + 1): exact for pass < nbuf ppb as long as nbuf ppb^2 < 2^32
    const uint64_t ppb = n_each / D / (uint64_t)pass_out((int)D);
    // (ppb = 1: floor(2^32 / 1) + 1 does not fit 32 bits -- the reciprocal came out as 1 and every pass landed in
    // buffer 0, past its end; such buffers, 512 outputs each, go one by one)
    // (and a buffer of fewer than 4096 outputs does not take the matrix path by itself -- plan_call -- so a call over
    // several of them must not either: hzsdr_chain_run_batch promises the bits of single calls)
    return n_each / D >= 4096 && nbuf * ppb * ppb < (1ull << 32);
}
inline Batch make_batch(const void *const *ins, void *const *outs, size_t nbuf, uint64_t n_each, unsigned D) {
    Batch B{};
    B.nbuf = (int)nbuf;
    B.out_each = (uint32_t)(n_each / D);
    B.ppb = B.out_each / (uint32_t)pass_out((int)D);
    B.rcp = (uint32_t)((1ull << 32) / B.ppb) + 1u;
    B.nb = 2 * n_each;
    for (size_t j = 0; j < nbuf; j++) {
        B.vin[j] = (uint64_t)(uintptr_t)ins[j] - (uint64_t)j * B.nb;
        B.vout[j] = (uint64_t)(uintptr_t)outs[j] - (uint64_t)j * B.out_each * 8;
    }
    return B;
}

// The LINE a clock run lies on.  Inside one binade [2^e, 2^(e+1)) of the clock every value is a multiple of
// u = 2^(e-52) and a run advances by step = S u (hz_host.cpp: hzsdr_nco_segments), so the clock values of a run are
// T u with T = T0 + k S.  Where a CALL first meets the run is the caller's business -- a stream cut into other buffers
// meets it elsewhere -- but the line is the stream's: its anchor is the line's first point in the binade,
// T_a = T0 - floor((T0 - 2^52) / S) S, `before` = (T0 - T_a) / S samples in front of t0, and `full` = the samples of
// the line inside the binade (up to 2 pi in the binade that holds it).  Everything a kernel derives from a run --
// the mixer's phase, whether the run is worth a table -- is derived from the line, so the results do not depend on
// how the stream was cut into calls (hzsdr_chain_run_batch == the same buffers through hzsdr_chain_run, bit for
// bit; the reference's ts recurrence, stream/shifter.go:68-79, does not know about buffers either).
struct Line {
    double t_anchor;   // the clock at the anchor (exact)
    uint64_t before;   // samples between the anchor and t0
    uint64_t full;     // samples of the line in its binade (1: not a line -- a run of one sample, a clock that stands)
};
inline Line run_line(double t0, double step) {
    Line ln{t0, 0, 1};
    if (!(step > 0.0) || !isnormal(t0) || !(t0 > 0.0)) return ln;
    int e2;
    (void)frexp(t0, &e2);
    const int e = e2 - 1;  // t0 in [2^e, 2^(e+1))
    const double u = ldexp(1.0, e - 52);
    const double Td = t0 / u, Sd = step / u;  // (exact: powers of two)
    if (!(Sd >= 1.0) || Sd != floor(Sd) || Td != floor(Td) || !(Td < 9007199254740992.0) || !(Sd < 9007199254740992.0)) return ln;
    const uint64_t T0 = (uint64_t)Td, S = (uint64_t)Sd, Tmin = 1ull << 52;
    if (T0 < Tmin) return ln;
    const uint64_t k = (T0 - Tmin) / S, Ta = T0 - k * S;
    uint64_t Tmax = (1ull << 53) - 1;
    if (e == 2) {  // (the binade of 2 pi: the clock wraps there)
        const uint64_t tu = (uint64_t)(6.283185307179586476925286766559 / u);
        if (tu < Tmax) Tmax = tu;
    }
    ln.t_anchor = (double)Ta * u;  // (exact)
    ln.before = k;
    ln.full = Tmax >= Ta ? (Tmax - Ta) / S + 1 : 1;
    return ln;
}

// (t_prev, t0): the clocks of two consecutive samples of a stream; true when they lie on one line of the clock -- one
// binade, exactly `step` apart (the difference of two values of one binade is exact).  What a chain asks at a call
// boundary: does this call's run 0 continue the run the previous call ended in?
inline bool continues(double t_prev, double t0, double step) {
    if (!(step > 0.0) || !(t_prev > 0.0) || !(t0 > t_prev)) return false;
    int ea, eb;
    (void)frexp(t_prev, &ea);
    (void)frexp(t0, &eb);
    return ea == eb && t0 - t_prev == step;
}

// phase accumulator constants of a run (host): frac(tau step / 2 pi) and frac(tau t0 / 2 pi) - first * the increment,
// in 2^-64 turns.
// Extended precision: the increment is multiplied by up to 2^27 samples.
inline void phase_fix(double tau, double t0, double step, uint64_t first, uint64_t *phi, uint64_t *dphi) {
    const long double inv2pi = 0.159154943091895335768883763372514362L;
    auto fix = [](long double turns) {
        turns -= floorl(turns);  // [0, 1)
        const long double v = turns * 18446744073709551616.0L;
        return v >= 18446744073709551615.0L ? ~0ull : (uint64_t)v;
    };
    *dphi = fix((long double)tau * (long double)step * inv2pi);
    *phi = fix((long double)tau * (long double)t0 * inv2pi) - first * *dphi;  // (mod 2^64: by the run's line at sample 0)
}
// The same constants from the run's LINE (run_line above): the phase at the line's anchor, rounded once, plus whole
// increments -- the phase of a stream sample is then the same 64-bit number whichever call holds the sample and
// wherever that call starts.  (phase_fix rounds tau t0 at the CALL's first sample of the run: 2^-37 turns of rounding
// that differed from cut to cut and moved one pass in thirty across a boundary of sincos_turns32's 2^-32 grid --
// round 5's "a batch equals single calls within an ulp or two of the factor".)
inline void phase_line(double tau, double t0, double step, uint64_t first, uint64_t *phi, uint64_t *dphi) {
    const Line ln = run_line(t0, step);
    uint64_t pa;
    phase_fix(tau, ln.t_anchor, step, 0, &pa, dphi);
    *phi = pa + (ln.before - first) * *dphi;  // (mod 2^64: call sample j of the run is sample before + j - first of the line)
}

// The mixer's step factors of a clock run (chains with exactly one Shift stage): a lane forms ONE Sincos per pass, for
// its first output, and turns it on to its other seven outputs -- (32 kT b + a) D samples further, b = 0, 1 the
// column block, a = 0 .. 3 -- by exp(2 pi i k dphi), kept as float pairs (cos hi, sin hi, cos lo, sin lo).  They depend
// on the run's phase increment alone, so they are a property of the run's table and travel at its end (the kernel
// computed them in its first lines until round 4: two hundred float64 instructions in front of the first barrier).
inline void step_factors(double tau, double step, int D, float out[8][4]) {
    uint64_t phi, dphi;
    phase_fix(tau, 0.0, step, 0, &phi, &dphi);
    for (int i = 0; i < 8; i++) {
        const uint64_t k = (uint64_t)(D * (32 * kT * (i >> 2) + (i & 3)));
        const long double turns = (long double)(k * dphi) * 5.42101086242752217003726400434970855712890625e-20L;  // 2^-64 (k dphi mod 2^64)
        const long double ang = turns * 6.283185307179586476925286766559005768L;
        const long double cs = cosl(ang), sn = sinl(ang);
        const float ch = (float)cs, sh = (float)sn;
        out[i][0] = ch, out[i][1] = sh, out[i][2] = (float)(cs - (long double)ch), out[i][3] = (float)(sn - (long double)sh);
    }
}

// geometry of a chain with `ntaps` taps at decimation D (host)
inline Geom make_geom(int ntaps, int D, unsigned off, int shift) {
    Geom g{};
    g.ntaps = ntaps;
    g.w0 = (ntaps - 1 + 7) / 8 * 8;
    const int window = g.w0 + D * (kT - 1) + 1;  // samples a tile's outputs reach back over
    const int gs = D / 2;                        // steps per group
    g.ks = ((2 * window + 31) / 32 + gs - 1) / gs * gs;
    g.e0 = 2 * (g.ks + 4);
    g.ne = g.e0 + (D / 8) * (kT - 1) + 1;
    g.shift = shift;
    g.off = off;
    return g;
}


// ---- the planner ---------------------------------------------------------------------------------------------
// The exactly-linear runs of the NCO clock over the call's samples (hzsdr_nco_segments): run r starts at sample
// first[r] with clock t0[r] and advances by step[r]; tab[r]: the run's digit table (null: none was prepared).
// A chain without a Shift stage passes n = 1, first[0] = 0.
struct ClockRuns {
    int n;
    const uint64_t *first;
    const double *t0, *step;
    const void *const *tab;
};
struct PlanIn {
    uint64_t n_in;     // samples of the call (a multiple of D is consumed)
    unsigned D;
    int ntaps;
    bool has_shift;
    // run 0 continues the run the previous call ended in, and the last `back` samples in front of the call lie in
    // that run AND in the raw history (at most `off` of them): windows of run 0 may reach that far back.  0: they
    // may not reach in front of the call at all.  (Rounds 2-5 knew all or nothing -- `cont`, set when the run already
    // held ntaps samples; a run that starts less than ntaps samples in front of a buffer boundary then gave its first
    // outputs to the tasks in single calls and to the matrix path in a call over several buffers.)
    uint64_t back;
    int shift_op;      // the program's only Shift stage (-1: none or several) and its fl(2 pi f)
    double tau;
    int n_ops;
    int max_grid;      // CUs
};
// Every clock run with a table gets the outputs whose whole window lies in it (tile-aligned) and the passes of
// the call's 512-output grid that hold them -- a pass that straddles a boundary is multiplied once per run, each
// time with that run's table and valid range.  What no run holds -- windows that cross a boundary, the stream's
// start, runs without a table -- are fix-up tasks of kFixOut outputs.  false: the call keeps the transform
// kernels (too short, too many runs or intervals, mostly boundaries).
inline bool plan_call(const PlanIn &in, const ClockRuns &cr, Plan *L, Fix *F, uint64_t *fix_outputs = nullptr) {
    memset(L, 0, sizeof *L);
    memset(F, 0, sizeof *F);
    const uint64_t D = in.D, n_out = in.n_in / D, nt = (uint64_t)in.ntaps, tile = kT, pass = (uint64_t)pass_out((int)D);
    if (fix_outputs) *fix_outputs = 0;
    if (n_out < 4096 || n_out >= (1ull << 31)) return false;  // (a call this short is launch-bound either way)
    if (cr.n < 1 || cr.n > kMaxClockRuns) return false;
    L->cont = in.back > 0 ? 1 : 0;
    L->n_ops = in.n_ops;
    L->gain = 1.0f;
    L->shift_op = in.shift_op;
    uint64_t fix_total = 0, covered = 0;  // outputs below `covered` are planned
    bool too_many = false;
    auto add_fix = [&](uint64_t a, uint64_t b) {
        if (b <= a) return;
        fix_total += b - a;
        if (F->n >= kMaxFix) {
            too_many = true;
            return;
        }
        F->m_a[F->n] = (uint32_t)a;
        F->m_b[F->n] = (uint32_t)b;
        F->task_first[F->n] = F->n_task;
        F->n_task += (int)((b - a + kFixOut - 1) / kFixOut);
        F->n++;
    };
    for (int r = 0; r < cr.n; r++) {
        const uint64_t a = cr.first[r];
        const uint64_t b = r + 1 < cr.n ? cr.first[r + 1] : in.n_in;
        uint64_t lo = (a + nt - 1 + D - 1) / D, hi = (b + D - 1) / D < n_out ? (b + D - 1) / D : n_out;
        if (r == 0 && in.back > 0) lo = in.back >= nt - 1 ? 0 : (nt - 1 - in.back + D - 1) / D;  // (a = 0: D m - (nt - 1) >= -back)
        lo = (lo + tile - 1) / tile * tile;
        if (hi < n_out) hi = hi / tile * tile;
        // a run without a table, or nothing of it on the matrix path.  (Rounds 3-5 also left pieces of fewer than 64
        // outputs to the tasks -- "too short to bother" -- which is a property of the call's cut, not of the run)
        if (!cr.tab[r] || hi <= lo) continue;
        // A SHORT run -- the binades behind a 2*pi wrap of the clock double from a few samples up -- would put its
        // table and one more multiplication of every pass it shares on ONE workgroup (the call's first: 24 items
        // for its eight waves instead of 16, a whole pass time longer than the rest).  As fix-up tasks its outputs
        // spread over the workgroups, one task each.  Short by the run's LINE, not by the call's piece of it.
        if (in.has_shift && run_line(cr.t0[r], cr.step[r]).full / D < kShortRun) continue;
        if (L->n >= kMaxRuns) return false;
        Run &u = L->run[L->n];
        u.tab = cr.tab[r], u.m_lo = (uint32_t)lo, u.m_hi = (uint32_t)hi;
        u.pass_first = (int)(lo / pass), u.pass_end = (int)((hi + pass - 1) / pass);
        u.seg = r;
        u.first = a;
        if (in.shift_op >= 0) phase_line(in.tau, cr.t0[r], cr.step[r], u.first, &u.phi, &u.dphi);
        L->pass_first[L->n] = u.pass_first, L->pass_end[L->n] = u.pass_end;
        add_fix(covered, lo);
        L->n++;
        covered = hi;
    }
    if (L->n == 0) return false;
    add_fix(covered, n_out);
    L->n_pass = (int)((n_out + pass - 1) / pass);
    L->n_task = F->n_task;
    L->grid = in.max_grid < L->n_pass ? in.max_grid : L->n_pass;
    if (L->grid < 1) L->grid = 1;
    if (fix_outputs) *fix_outputs = fix_total;
    // the fix-up tasks are the slow way: a call that is mostly boundaries keeps the transforms; a workgroup takes
    // them one at a time, behind its passes
    return !too_many && F->n_task <= 4 * L->grid && fix_total * 8 <= n_out;
}

}  // namespace mm2
}  // namespace hz

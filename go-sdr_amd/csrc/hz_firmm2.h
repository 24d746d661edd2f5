// hz_firmm2.h -- the int8 matrix FIR (hz_firmm.h) re-cut so that its phases overlap: one PERSISTENT
// workgroup per CU whose eight waves pull small passes from a queue.
//
// hz_firmm.h runs 1024 workgroups, each of which loads 34 KB, multiplies for 24 us and finishes 2048
// outputs -- all of them in the same phase at the same time (accumulators for ALL 2^21 outputs of a
// 2^24-sample call are resident at once, so there is exactly one round): the matrix pipe idles through
// the 9 us input burst and the 5 us epilogue, 44 % busy over the kernel.  A round can only be cut into
// shorter ones with FEWER accumulators per wave, and fewer accumulators mean less operand re-use; what
// makes that affordable is the LDS (256 B/clk/CU for ds_read_b128, a quarter of it used by hz_firmm.h)
// once the tap table lives there too -- which needs ONE copy per CU, i.e. one workgroup per CU:
//
//   * a tile = EIGHT consecutive outputs x (re, im) x TWO digit planes = 32 rows of the Toeplitz
//     matrix (hz_firmm.h: 16 outputs x (re, im), the planes split over two waves).  One wave holds all
//     four planes of its outputs (two A fragments), nothing is exchanged between waves, and a tile's
//     window is 56 samples wider than the taps instead of 120 (68 instead of 72 MFMA steps at 1024 taps);
//   * a PASS = 2 column blocks of 32 tiles = 512 outputs (D = 8): 4 accumulators (64 registers), 10 KB of
//     input in a slot of LDS that belongs to the wave.  Per 32-byte step of the window: two A fragments
//     from the table in LDS, two B fragments from the slot, four MFMAs -- 1 KB of LDS reads per MFMA,
//     half of what the LDS delivers, no global load in the loop;
//   * a wave's life: take a pass from the workgroup's queue (an LDS counter), and while it multiplies,
//     the NEXT pass's bytes are in flight into registers (landed in the slot when the loop is done);
//     the finished pass's float64 combination, mixer and stores then run beside the partner wave's
//     loop on the same SIMD.  The eight waves de-phase by themselves (the queue) and by construction
//     (waves 4-7, the second wave of each SIMD, start half a pass late);
//   * the table (19 KB per clock run at 1024 taps) is loaded once per workgroup; a workgroup whose
//     range of passes spans a run boundary reloads it between the runs (a barrier; one or two of the
//     256 workgroups per call);
//   * fix-up tasks (outputs in reference order) and history tasks are wave-sized here and ride on the
//     waves of all workgroups before their first pass.
//
// Geometry, table contents and the arithmetic (four balanced base-256 digit planes of 32-bit
// fixed-point taps, exact int32 sums, one float64 combination, one rounding) are those of hz_firmm.h.
#pragma once
#include <type_traits>

#include "hz_firmm.h"

namespace hz {
namespace mm2 {

using mm::v16i;
using mm::v4i;

constexpr int kT = 8;        // outputs per tile
constexpr int kWaves = 8;    // two per SIMD
constexpr int kThreads = 64 * kWaves;
constexpr int kFixOut = 8;   // outputs per fix-up task
constexpr int kHistPer = 64; // samples per history task
constexpr int kMaxRuns = 8;   // clock runs of a call that take the matrix path (more: the call keeps the transforms)
constexpr int kMaxFix = kNcoMaxSegs + 2;
constexpr int kU = 12;       // 16-byte pieces per lane of a pass image, any window (768 pieces = 12 KB at most)

constexpr bool factor_ok(unsigned D) { return D == 8; }
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
// table: T[f][E][part][pl] of 16 bytes (digit plane 2 f + pl, most significant first), then (dc_re, dc_im)
constexpr size_t table_bytes(int ne) { return (size_t)ne * 128 + 16; }
constexpr size_t table_lds(int ne) { return (table_bytes(ne) + 255) / 256 * 256; }
constexpr size_t kCtlBytes = 256;  // the queue's counter
constexpr size_t lds_bytes(int D, int ks, int ne) { return 2 * table_lds(ne) + kCtlBytes + kWaves * slot_bytes(D, ks); }  // two tables

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
    const void *tab;       // the run's digit table (null: no table -- its outputs are fix-up tasks)
    uint32_t m_lo, m_hi;   // the outputs that take the matrix path (tile-aligned inside the call)
    int pass_first;        // the passes (512 outputs of the call's grid) that hold outputs of the run: a pass that
    int pass_end;          // straddles a boundary belongs to BOTH runs and is multiplied twice, once per table
    // programs with exactly ONE Shift stage: the stage's phase tau ts / 2 pi at the BUFFER's first sample by
    // the run's line, and its increment per sample, in 2^-64 turns (mod 1)
    uint64_t phi, dphi;
    uint64_t first;        // the run's first sample, its clock there and the clock's step (programs with
    double t0, step;       // several Shift stages)
};
static_assert(sizeof(Run) == 64, "one scalar-cache line per run");
struct Plan {
    int n;          // clock runs with outputs on the matrix path, in stream order (at most kMaxRuns)
    int n_pass;     // passes of the call
    int cont;       // run 0 continues the previous call's last run: windows may reach into the raw history
    int shift_op;   // index of the program's only Shift stage, -1: none or several
    int n_task;     // fix-up tasks (F)
    int n_ops;      // stages of the program
    float gain;     // MIX kernels: the Gain behind the Shift (n_ops == 2)
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

// sin and cos of 2 pi t / 2^32, float32 arithmetic only: within 0.8 ulp (0.30 ulp rms: a correctly rounded
// result has 0.29).  The nearest quarter turn comes off the integer phase exactly; the rest r, |r| <= 1/8
// turn, keeps 24 bits in r and its last six in e; odd / even polynomials in r with the leading coefficients
// split, one rounding at the end of each.  (float64 Sincos here would queue behind the SIMD partner's MFMAs.)
__device__ __forceinline__ void sincos_turns(uint32_t t, float &sn, float &cs) {
    const uint32_t q = (t + 0x20000000u) >> 30;
    const int32_t rf = (int32_t)(t - (q << 30));
    const float r = (float)(rf & ~63) * 2.3283064365386963e-10f;  // 2^-32: exact
    const float e = (float)(rf & 63) * 1.4629180792671596e-09f;   // 2 pi 2^-32
    const float zh = r * r, zl = __fmaf_rn(r, r, -zh);
    float ps = __fmaf_rn(zh, 42.058692932128906f, -76.70585632324219f);
    ps = __fmaf_rn(ps, zh, 81.6052474975586f);
    ps = __fmaf_rn(ps, zh, -41.34170150756836f);
    float pc = __fmaf_rn(zh, -26.42625617980957f, 60.2446403503418f);
    pc = __fmaf_rn(pc, zh, -85.45681762695312f);
    pc = __fmaf_rn(pc, zh, 64.93939208984375f);
    const float rin = r * __fmaf_rn(zh, ps, -1.7484555314695172e-07f);
    const float small = __fmaf_rn(zh, zh * pc, __fmaf_rn(zl, -19.739208221435547f, zh * -5.807431762150372e-07f));
    const float s0 = __fmaf_rn(r, 6.2831854820251465f, rin), c0 = __fmaf_rn(zh, -19.739208221435547f, small) + 1.0f;
    const float s = __fmaf_rn(r, 6.2831854820251465f, __fmaf_rn(e, c0, rin));
    const float c = __fmaf_rn(zh, -19.739208221435547f, __fmaf_rn(-e, s0, small)) + 1.0f;
    const float a = (q & 1) ? c : s, b = (q & 1) ? s : c;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
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

// The same computation cut into seven short stages (with the product and an optional Gain as the last): the
// mixer of a pass rides inside the NEXT pass's matrix loop, one stage per step, so that each gap between two
// MFMAs takes two or three vector instructions and hides them (five is what a gap hides; a whole output per
// step, ~50 instructions, made the step vector-bound: +1.4 us per pass, measured).
struct MixState {
    uint32_t q;
    float r, e, zh, zl, ps, pc, rin, small, s, c;
};
template <int K>
__device__ __forceinline__ void mix_stage(MixState &m, uint32_t t, float2 &y, float gain, bool has_gain) {
    if constexpr (K == 0) {
        m.q = (t + 0x20000000u) >> 30;
        const int32_t rf = (int32_t)(t - (m.q << 30));
        m.r = (float)(rf & ~63) * 2.3283064365386963e-10f;
        m.e = (float)(rf & 63) * 1.4629180792671596e-09f;
    } else if constexpr (K == 1) {
        m.zh = m.r * m.r;
        m.zl = __fmaf_rn(m.r, m.r, -m.zh);
        m.ps = __fmaf_rn(m.zh, 42.058692932128906f, -76.70585632324219f);
        m.pc = __fmaf_rn(m.zh, -26.42625617980957f, 60.2446403503418f);
    } else if constexpr (K == 2) {
        m.ps = __fmaf_rn(m.ps, m.zh, 81.6052474975586f);
        m.ps = __fmaf_rn(m.ps, m.zh, -41.34170150756836f);
        m.pc = __fmaf_rn(m.pc, m.zh, -85.45681762695312f);
        m.pc = __fmaf_rn(m.pc, m.zh, 64.93939208984375f);
    } else if constexpr (K == 3) {
        m.rin = m.r * __fmaf_rn(m.zh, m.ps, -1.7484555314695172e-07f);
        m.small = __fmaf_rn(m.zh, m.zh * m.pc, __fmaf_rn(m.zl, -19.739208221435547f, m.zh * -5.807431762150372e-07f));
    } else if constexpr (K == 4) {
        const float s0 = __fmaf_rn(m.r, 6.2831854820251465f, m.rin), c0 = __fmaf_rn(m.zh, -19.739208221435547f, m.small) + 1.0f;
        m.s = __fmaf_rn(m.r, 6.2831854820251465f, __fmaf_rn(m.e, c0, m.rin));
        m.c = __fmaf_rn(m.zh, -19.739208221435547f, __fmaf_rn(-m.e, s0, m.small)) + 1.0f;
    } else if constexpr (K == 5) {
        const float a = (m.q & 1) ? m.c : m.s, b = (m.q & 1) ? m.s : m.c;
        m.s = (m.q & 2) ? -a : a;
        m.c = ((m.q + 1) & 2) ? -b : b;
    } else if constexpr (K == 6) {
        float2 v = make_float2(__fmaf_rn(y.x, m.c, -(y.y * m.s)), __fmaf_rn(y.x, m.s, y.y * m.c));
        if (has_gain) v = make_float2(__fmul_rn(v.x, gain), __fmul_rn(v.y, gain));
        y = v;
    }
}

using mm::find_le;
using mm::task_window;

// EXP (tools/mfma_fir2.hip; 0 in the library): 1 = no input loads, 2 = no matrix loop, 4 = no mixer,
// 8 = no stores, 16 = no stagger, 64 = stamps.
// NG: the window's groups (ks / GS) when the instantiation is for ONE tap count -- the matrix loop is then
// straight-line code (a loop header drains the operand pipeline: the compiler cannot count outstanding
// loads across a back edge); 0: any window, a loop over the groups.
// MIX = 1: the elementwise program is ONE Shift stage, optionally followed by one Gain -- the program of the
// headline chain.  Vector instructions of a wave that runs BESIDE its SIMD partner's matrix loop cost both
// waves dearly (measured: ~600 of them took the one wave 7 us instead of 1.8, and the partner's loop twice its
// time), while the same instructions between a wave's OWN MFMAs are nearly free (24 of an MFMA's 32 issue
// cycles are open).  So with MIX = 1 a pass leaves its matrix loop with 16 floats per lane (the planes
// combined), and its mixer and stores ride inside the NEXT pass's matrix loop, one output per eight steps.
// MIX = 0: any program, applied behind the loop.
template <int FMT, int D, int NG = 0, int MIX = 0, int EXP = 0>
__global__ __launch_bounds__(kThreads) void fir_mm2_kernel(
    const void *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ hist,
    float2 *__restrict__ new_hist, const uint8_t *__restrict__ rhist, uint8_t *__restrict__ new_rhist,
    const float2 *__restrict__ taps, size_t n_in, Geom G, Plan L, EwProgram P, Fix F,
    unsigned long long *stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t mm_lds[];
    constexpr int TB = tile_bytes(D), TS = tile_stride(D), PPT = TB / 16, GS = PPT / 2, NB = blocks_for(D);
    constexpr int kPassOut = pass_out(D);
    // pieces per lane of a pass image: known when the window is (NG groups: pass_tiles - 1 + NG tiles of PPT pieces)
    constexpr int KU = NG > 0 ? ((pass_tiles(D) - 1 + NG) * PPT + 63) / 64 : kU;
    constexpr bool kWhole = NG > 0 && ((pass_tiles(D) - 1 + NG) * PPT) % 64 == 0;  // every lane has KU pieces
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int wb = blockIdx.x;
    const int n = l & 31, h = l >> 5;
    const size_t tab_lds = table_lds(G.ne), slot_sz = slot_bytes(D, G.ks);
    uint8_t *const tabp = mm_lds;
    unsigned *const ctr = reinterpret_cast<unsigned *>(mm_lds + 2 * tab_lds);
    uint8_t *const slot = mm_lds + 2 * tab_lds + kCtlBytes + (size_t)wave * slot_sz;
    const int64_t n_bytes = 2 * (int64_t)n_in;
    const uint8_t *src = (const uint8_t *)in;
    const int pieces = (int)(image_bytes(D, G.ks) / 16);
    // (EXP & 64: s_memrealtime stamps, 10 ns ticks: slot 0 of a wave = its start and end, slots 1 .. 3 its passes)
    int stamp_pass = 0;
    auto stamp = [&](int k) {
        if constexpr ((EXP & 64) != 0) {
            if (l == 0 && stamp_pass < 4)
                stamps[(((size_t)wb * kWaves + wave) * 4 + (k == 0 || k == 7 ? 0 : k >= 8 ? 0 : stamp_pass)) * 8 + (k & 7)] = __builtin_amdgcn_s_memrealtime();
        }
    };
    stamp(0);
    if constexpr ((EXP & 32) == 0) __builtin_amdgcn_s_setprio(1);
    const Run run0 = L.run[0];  // (read with the header: most workgroups of most calls are in run 0)

    // ---- the small tasks, one wave each --------------------------------------------------------------
    const int n_hist_tasks = new_hist ? (int)((G.off + kHistPer - 1) / kHistPer) : 0;
    const int n_tasks = L.n_task + n_hist_tasks;
    auto small_task = [&](int task) {
        using RW = typename Raw<FMT>::t;
        if (task < L.n_task) {
            // Up to 16 outputs in reference order: their window's samples (load, convert, elementwise
            // program) go to the wave's slot, a chunk of the taps at a time; four lanes per output run the
            // direct form in float64.
            const int k = find_le(F.task_first, F.n, task);
            const uint32_t m0 = F.m_a[k] + (uint32_t)(task - F.task_first[k]) * kFixOut;
            const int cnt = (int)min((uint32_t)kFixOut, F.m_b[k] - m0);
            float2 *xs = reinterpret_cast<float2 *>(slot);
            const int kc = ((int)(slot_sz / 8) - D * (kFixOut - 1)) & ~7;  // taps per chunk
            // eight lanes per output, four independent accumulation chains per lane (a single chain per component
            // runs at the latency of a float64 fma, a quarter of the issue rate)
            constexpr int kLanesPer = 64 / kFixOut;
            const int o = l / kLanesPer, sl = l % kLanesPer;
            double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
#pragma unroll 1
            for (int k0 = 0; k0 < G.ntaps; k0 += kc) {
                const int k1 = min(G.ntaps, k0 + kc);
                const int64_t p_lo = (int64_t)D * m0 - (k1 - 1);
                const int n_s = (k1 - k0) + D * (cnt - 1);
                const NcoWin tw = task_window(P, p_lo, p_lo + n_s - 1);
                constexpr int W = 3;
#pragma unroll 1
                for (int i0 = l; i0 < n_s; i0 += W * 64) {
                    float2 v[W];
#pragma unroll
                    for (int u = 0; u < W; u++) {
                        const int64_t pu = p_lo + i0 + u * 64;
                        const bool ok = i0 + u * 64 < n_s && pu >= 0;
                        v[u] = Raw<FMT>::cvt(ok ? ((const RW *)in)[pu] : RW{});
                    }
                    ew_apply_n<W, true>(P, v, (uint64_t)(p_lo + i0), tw, (uint64_t)64);
#pragma unroll
                    for (int u = 0; u < W; u++) {
                        const int idx = i0 + u * 64;
                        const int64_t pu = p_lo + idx;
                        if (idx < n_s)
                            xs[idx] = pu >= 0 ? v[u] : ((hist && pu + (int64_t)G.off >= 0) ? hist[pu + (int64_t)G.off] : make_float2(0.f, 0.f));
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (o < cnt) {
                    const float2 *xo = xs + D * o + (k1 - 1);
#pragma unroll 4
                    for (int kk = k0 + sl; kk < k1; kk += kLanesPer) {
                        const float2 hk = taps[kk], x = xo[-kk];
                        const double xr = x.x, xi = x.y, hr = hk.x, hi = hk.y;
                        ar = __fma_rn(xr, hr, ar);
                        br = __fma_rn(-xi, hi, br);
                        ai = __fma_rn(xr, hi, ai);
                        bi = __fma_rn(xi, hr, bi);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            ar += br;
            ai += bi;
#pragma unroll
            for (int d = 1; d < kLanesPer; d <<= 1) {
                ar += __shfl_xor(ar, d);
                ai += __shfl_xor(ai, d);
            }
            if (o < cnt && sl == 0) out[m0 + o] = make_float2((float)ar, (float)ai);
        } else {
            // 64 samples of the history: the last `off` samples after the elementwise program, and the
            // same samples as raw bytes
            const unsigned idx = (unsigned)(task - L.n_task) * kHistPer + l;
            const int64_t h_lo = (int64_t)n_in - (int64_t)G.off + (int64_t)(task - L.n_task) * kHistPer;
            const NcoWin tw = task_window(P, h_lo, h_lo + kHistPer - 1);
            if (idx < G.off) {
                const int64_t p = (int64_t)n_in - (int64_t)G.off + idx;
                new_hist[idx] = mm::ordered_sample<FMT>(in, P, p, hist, G.off, tw);
                reinterpret_cast<RW *>(new_rhist)[idx] = p >= 0 ? ((const RW *)in)[p] : reinterpret_cast<const RW *>(rhist)[p + (int64_t)G.off];
            }
        }
    };
    // Task t goes to workgroup t % grid, wave 7 - (t / grid) % 8: the YOUNG waves first (they only fill the gaps
    // of their SIMD partners' matrix loops), one task per workgroup before two, and behind the workgroup's
    // first barrier -- a wave that is late for it holds up all eight.
    bool tasks_done = false;
    auto run_tasks = [&]() {
        if (tasks_done) return;
        tasks_done = true;
        for (int task = (kWaves - 1 - wave) * L.grid + wb; task < n_tasks; task += kWaves * L.grid) small_task(task);  // (uniform per wave)
    };
    stamp(9);
    // ---- this workgroup's passes [pb0, pb1) of the call -------------------------------------------------
    const uint32_t NP = (uint32_t)L.n_pass;
    // (NP / grid passes each, the first NP % grid workgroups one more; 32-bit arithmetic)
    const uint32_t per = NP / (uint32_t)L.grid, rem = NP - per * (uint32_t)L.grid;
    const uint32_t pb0 = (uint32_t)wb * per + min((uint32_t)wb, rem), pb1 = pb0 + per + ((uint32_t)wb < rem ? 1u : 0u);

    // The bytes of a pass: lane l takes pieces l, l + 64, ...  A pass that lies inside the buffer (all but the
    // call's first and last) is prefetched into registers while the previous pass multiplies (issue, then
    // land); the others are brought in piece by piece when their turn comes (land_edge: bytes before the
    // buffer are the previous call's last samples when the clock run continues, zeros otherwise).
    auto pass_p0 = [&](uint32_t pass) { return 2 * ((int64_t)D * ((int64_t)pass * kPassOut) - G.w0); };
    auto inside = [&](uint32_t pass) {
        const int64_t p0 = pass_p0(pass);
        return p0 >= 0 && p0 + 16 * (int64_t)pieces <= n_bytes;
    };
    auto issue = [&](v4i(&x)[KU], uint32_t pass) {
        const uint8_t *p = src + pass_p0(pass);
#pragma unroll
        for (int u = 0; u < KU; u++) {
            x[u] = v4i{0, 0, 0, 0};  // (pieces past the image: the last piece again, not landed)
            if constexpr ((EXP & 1) != 0) continue;
            if constexpr (kWhole) x[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p + 16 * l + 1024 * u));
            else x[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p + 16 * min(l + u * 64, pieces - 1)));
        }
    };
    auto put = [&](int q, v4i v) {
        if constexpr (FMT == HZSDR_FMT_U8) v ^= (int)0x80808080;  // b - 128 as int8
        *reinterpret_cast<v4i *>(slot + TS * (q / PPT) + 16 * (q % PPT)) = v;
    };
    auto land = [&](v4i(&x)[KU]) {
        // piece l + 64 u is piece l % PPT of tile l / PPT + (64 / PPT) u: one address and constants
        uint8_t *lp = slot + TS * (l / PPT) + 16 * (l % PPT);
#pragma unroll
        for (int u = 0; u < KU; u++)
            if (kWhole || l + u * 64 < pieces) {
                if constexpr (FMT == HZSDR_FMT_U8) x[u] ^= (int)0x80808080;  // b - 128 as int8
                *reinterpret_cast<v4i *>(lp + u * (64 / PPT) * TS) = x[u];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto land_edge = [&](uint32_t pass) {
        const int64_t p0 = pass_p0(pass);
#pragma unroll 1
        for (int q = l; q < pieces; q += 64) {
            const int64_t p = p0 + (int64_t)q * 16;
            v4i v{0, 0, 0, 0};
            if (p >= 0 && p + 16 <= n_bytes) {
                v = *reinterpret_cast<const v4i *>(src + p);
            } else if (p < 0 && L.cont && p + 2 * (int64_t)G.off >= 0) {
                v = *reinterpret_cast<const v4i *>(rhist + (p + 2 * (int64_t)G.off));  // (2 off is a multiple of 16)
            } else if (p + 16 > 0 && p < n_bytes) {  // straddles the buffer's end: byte by byte
                uint64_t w0 = 0, w1 = 0;  // (no byte array: it would live in scratch memory)
#pragma unroll 1
                for (int e = 0; e < 16; e++)
                    if (p + e >= 0 && p + e < n_bytes) {
                        const uint64_t bb = src[p + e];
                        if (e < 8) w0 |= bb << (8 * e);
                        else w1 |= bb << (8 * (e - 8));
                    }
                v = v4i{(int)(uint32_t)w0, (int)(uint32_t)(w0 >> 32), (int)(uint32_t)w1, (int)(uint32_t)(w1 >> 32)};
            }
            put(q, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // A fragments: lane n is row (a = n >> 3, c = (n >> 2) & 1, part = (n >> 1) & 1, pl = n & 1) = output
    // i = 4 c + a of the tile -- the accumulator then holds FOUR CONSECUTIVE outputs per lane (row =
    // 8 (q >> 2) + 4 (l >> 5) + (q & 3)): entry (D/8) i - h - 2 s + e0, 64 bytes per entry and fragment row
    const int i_row = 4 * ((n >> 2) & 1) + (n >> 3);
    const int a_base = 64 * ((D / 8) * i_row - h + G.e0) + 16 * (n & 3);
    const int f_stride = G.ne * 64;
    const int groups = G.ks / GS;  // (NG > 0: the host launches this instantiation for ks = NG GS only)

    const Run run1 = L.run[1];
    bool first_seg = true, first_stamp = true;
    int r = 0;
#pragma unroll 1
    while (r < L.n && (uint32_t)L.pass_first[r] < pb1) {
        // The passes of this workgroup that hold outputs of run r, and of the run behind it: TWO tables fit the
        // LDS, so a workgroup whose range holds a clock boundary (one or two of the 256 in most calls) works
        // through both runs' passes as ONE queue instead of draining the first behind a barrier.
        const uint32_t a0 = max(pb0, (uint32_t)L.pass_first[r]), a1 = min(pb1, (uint32_t)L.pass_end[r]);
        if (a0 >= a1) {
            r++;
            continue;
        }
        uint32_t b0 = 0, b1 = 0;
        if (r + 1 < L.n) {
            b0 = max(pb0, (uint32_t)L.pass_first[r + 1]);
            b1 = min(pb1, (uint32_t)L.pass_end[r + 1]);
            if (b0 >= b1) b0 = b1 = 0;
        }
        const bool has_b = b1 > b0;  // uniform
        const uint32_t n_a = a1 - a0, seg_b = n_a + (b1 - b0);  // items [0, seg_b): pass a0 + q, then b0 + (q - n_a)
        auto pass_of = [&](uint32_t q) { return q < n_a ? a0 + q : b0 + (q - n_a); };
        if (!first_seg) __syncthreads();  // every wave is done with the previous tables and queue
        const int first_seg_prio = first_seg ? 1 : 0;
        first_seg = false;
        // Waves 0-3 (one per SIMD) take the queue's first passes and put their bytes in flight at once;
        // waves 4-7 start behind the barrier: the first burst is half as large, and the two waves of a SIMD
        // start out of phase.
        constexpr int kEarly = kWaves / 2;
        auto grab = [&]() -> uint32_t {
            unsigned p = 0;
            if (l == 0) p = atomicAdd(ctr, 1u);
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
        };
        v4i x[KU];
        uint32_t cur = seg_b;
        bool in0 = false;
        constexpr int kTU = 4;  // table pieces per thread (32 KB at most: the host)
        if (wave < kEarly) {
            cur = min((uint32_t)wave, seg_b);
            in0 = cur < seg_b && inside(pass_of(cur));  // uniform
            if (in0) issue(x, pass_of(cur));
        }
        if (first_stamp) stamp(13);
        // (everything the queue reads from the kernel arguments by run, at once)
        Run ru = run0, rv = run1;
        if (r == 1) ru = run1;
        else if (r > 1) ru = L.run[r];  // (one line)
        if (has_b && r > 0) rv = L.run[r + 1];
        const float gain = L.gain;
        const bool has_gain = MIX && L.n_ops > 1;  // uniform
        {
            const int tp = (int)(table_bytes(G.ne) + 15) / 16;
            const v4i *tg = (const v4i *)ru.tab, *th = (const v4i *)rv.tab;
            v4i tq[kTU], tr[kTU];
#pragma unroll
            for (int u = 0; u < kTU; u++) tq[u] = tid + u * kThreads < tp ? tg[tid + u * kThreads] : v4i{0, 0, 0, 0};
            if (has_b) {
#pragma unroll
                for (int u = 0; u < kTU; u++) tr[u] = tid + u * kThreads < tp ? th[tid + u * kThreads] : v4i{0, 0, 0, 0};
            }
            if (first_stamp) stamp(14);
#pragma unroll
            for (int u = 0; u < kTU; u++)
                if (tid + u * kThreads < tp) *reinterpret_cast<v4i *>(tabp + 16 * (size_t)(tid + u * kThreads)) = tq[u];
            if (has_b) {
#pragma unroll
                for (int u = 0; u < kTU; u++)
                    if (tid + u * kThreads < tp) *reinterpret_cast<v4i *>(tabp + tab_lds + 16 * (size_t)(tid + u * kThreads)) = tr[u];
            }
        }
        if (first_stamp) stamp(10);
        if (tid == 0) *ctr = kEarly;
        __syncthreads();
        if (first_stamp) stamp(11);
        // (a task is ~1500 float64 instructions per lane: beside a SIMD partner's matrix loop at equal or lower
        // priority it took ~40 us, measured; first at the issue port it is done in a fraction of that)
        if constexpr ((EXP & 32) == 0) __builtin_amdgcn_s_setprio(3);
        run_tasks();
        if constexpr ((EXP & 32) == 0) __builtin_amdgcn_s_setprio(0);
        if constexpr ((EXP & 32) == 0) {
            if (first_seg_prio) __builtin_amdgcn_s_setprio(1);
        }
        if (wave >= kEarly) {
            cur = grab();
            in0 = cur < seg_b && inside(pass_of(cur));
            if (in0) issue(x, pass_of(cur));
            if constexpr ((EXP & 16) != 0) __builtin_amdgcn_s_sleep(60);  // (experiment: a longer offset)
        }
        if (cur < seg_b) {
            if (in0) land(x);
            else land_edge(pass_of(cur));
        }
        if (first_stamp) stamp(12);
        first_stamp = false;
        // (MIX) the pass whose mixer and stores are still owed: its combined planes, first output, valid range
        float2 yp[NB][4];
        uint32_t p_mb = 0, p_lo = 0, p_hi = 0;  // p_lo >= p_hi: nothing owed
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int q = 0; q < 4; q++) yp[b][q] = make_float2(0.f, 0.f);
        // one output of the owed pass: phase accumulator in turns, sincos_turns, product (+ Gain)
        uint64_t p_ph = 0, p_dphi = 0;
        MixState ms{};
        auto mix_phase = [&](int b, int q) { return (uint32_t)((p_ph + (uint64_t)(D * (32 * kT * b + q)) * p_dphi) >> 32); };
        auto mix_one = [&](int b, int q) {  // all stages of one output at once (drains)
            const uint32_t t = mix_phase(b, q);
            mix_stage<0>(ms, t, yp[b][q], gain, has_gain);
            mix_stage<1>(ms, t, yp[b][q], gain, has_gain);
            mix_stage<2>(ms, t, yp[b][q], gain, has_gain);
            mix_stage<3>(ms, t, yp[b][q], gain, has_gain);
            mix_stage<4>(ms, t, yp[b][q], gain, has_gain);
            mix_stage<5>(ms, t, yp[b][q], gain, has_gain);
            mix_stage<6>(ms, t, yp[b][q], gain, has_gain);
        };
        auto store_block = [&](float2(&y)[4], uint32_t mo, uint32_t lo, uint32_t hi) {
            if constexpr ((EXP & 8) != 0) return;
            if (mo >= lo && mo + 4 <= hi) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f *o4 = reinterpret_cast<v4f *>(out + mo);
                __builtin_nontemporal_store(v4f{y[0].x, y[0].y, y[1].x, y[1].y}, o4);
                __builtin_nontemporal_store(v4f{y[2].x, y[2].y, y[3].x, y[3].y}, o4 + 1);
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (mo + q >= lo && mo + q < hi) out[mo + q] = y[q];
            }
        };
#pragma unroll 1
        while (cur < seg_b) {
            const uint32_t nxt = grab();
            const bool has_next = nxt < seg_b, in_next = has_next && inside(pass_of(nxt));  // uniform
            if (in_next) issue(x, pass_of(nxt));
            stamp_pass++;
            stamp(1);
            const bool sel = cur >= n_a;  // uniform: the pass belongs to the second run of the queue
            const uint32_t m_lo = sel ? rv.m_lo : ru.m_lo, m_hi = sel ? rv.m_hi : ru.m_hi;
            const uint64_t dphi = sel ? rv.dphi : ru.dphi, phi_r = sel ? rv.phi : ru.phi;
            const uint32_t m_start = pass_of(cur) * (uint32_t)kPassOut;
            const uint32_t v_lo = max(m_lo, m_start), v_hi = min(m_hi, m_start + (uint32_t)kPassOut);
            const bool active = v_lo < v_hi;  // uniform
            v16i acc[2][NB];
#pragma unroll
            for (int f = 0; f < 2; f++)
#pragma unroll
                for (int b = 0; b < NB; b++)
#pragma unroll
                    for (int q = 0; q < 16; q++) acc[f][b][q] = 0;
            if (active && (EXP & 2) == 0) {
                // step s = GS g + j of the window: the A entries 2 s below the lane's first, B piece 2 j + h of
                // tile n + g -- constants off two per-lane addresses (NG > 0) or off two running ones
                constexpr int KS = NG * GS;
                // (NG > 0: the lane's LAST entry, and the compiler kept from folding the base back to the first
                // one: DS offsets are unsigned, negative ones cost an address add per load)
                int a_off = a_base - (NG > 0 ? 128 * KS : 0) + (sel ? (int)tab_lds : 0);
                if constexpr (NG > 0) asm volatile("" : "+v"(a_off));
                const uint8_t *ap = tabp + a_off;
                const uint8_t *bp = slot + TS * n + 16 * h;
                int s_done = 0, g_done = 0;  // (compile-time values when the groups are unrolled)
                auto load_a = [&](v4i(&a)[2], int s) {
#pragma unroll
                    for (int f = 0; f < 2; f++) {
                        if constexpr (NG > 0) a[f] = *reinterpret_cast<const v4i *>(ap + f * (2 * (KS + 4) + (D / 8) * (kT - 1) + 1) * 64 + 128 * (KS - s));
                        else a[f] = *reinterpret_cast<const v4i *>(ap + f * f_stride - 128 * (s - s_done));
                    }
                };
                auto load_b = [&](v4i(&b)[NB], int g, int j) {
#pragma unroll
                    for (int q = 0; q < NB; q++) b[q] = *reinterpret_cast<const v4i *>(bp + TS * (g - g_done) + 32 * j + q * 32 * TS);
                };
                // operands two steps ahead of their MFMAs, in rings of three (straight-line code) or four
                constexpr int RG = NG > 0 ? 3 : 4;
                v4i a[RG][2], b[RG][NB];
                load_a(a[0], 0);
                load_b(b[0], 0, 0);
                load_a(a[1], 1);
                load_b(b[1], 0, 1);
                static_assert(!MIX || NG == 0 || 8 * 4 * NB <= KS, "the owed pass's outputs take eight steps each");
                // one step; SC: the step as a compile-time value (straight-line form), or -1
                auto step = [&](auto sc, int g, int j) {
                    constexpr int SC = decltype(sc)::value;
                    const int s = SC >= 0 ? SC : GS * g + j;
                    const int sx = SC >= 0 ? SC : j;  // (ring positions: the step mod 3, or j mod 4 of every group)
                    load_a(a[(sx + 2) % RG], s + 2);
                    if (j + 2 < GS) load_b(b[(sx + 2) % RG], g, j + 2);
                    else load_b(b[(sx + 2) % RG], g + 1, j + 2 - GS);
#pragma unroll
                    for (int f = 0; f < 2; f++)
#pragma unroll
                        for (int q = 0; q < NB; q++)
                            acc[f][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[sx % RG][f], b[sx % RG][q], acc[f][q], 0, 0, 0);
                    // (MIX) output SC / 8 of the owed pass, stage SC % 8 (the eighth is empty); the stores at the end
                    constexpr bool kMixHere = MIX && SC >= 0 && SC / 8 < 4 * NB && SC % 8 < 7;
                    if constexpr (kMixHere) {
                        constexpr int O = SC / 8;
                        mix_stage<SC % 8>(ms, mix_phase(O / 4, O % 4), yp[O / 4][O % 4], gain, has_gain);
                    }
                    if constexpr (MIX && SC >= 0 && SC == (8 * 4 * NB < KS ? 8 * 4 * NB : KS - 1)) {
#pragma unroll
                        for (int bb = 0; bb < NB; bb++) store_block(yp[bb], p_mb + (uint32_t)(32 * kT) * bb, p_lo, p_hi);
                    }
                    // the step's loads (and the mixer's instructions) between its MFMAs, nothing moved across steps
#pragma unroll
                    for (int q = 0; q < 2 + NB; q++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // LDS read
                        if constexpr (MIX && SC >= 0) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NB - 2 - NB > 0 ? 2 * NB - 2 - NB : 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                static_assert(GS == 4 || GS == 8, "ring indices repeat per group");
                if constexpr (NG > 0) {
                    auto all = [&](auto self, auto sc) {
                        constexpr int S = decltype(sc)::value;
                        if constexpr (S < KS) {
                            step(sc, S / GS, S % GS);
                            self(self, std::integral_constant<int, S + 1>{});
                        }
                    };
                    all(all, std::integral_constant<int, 0>{});
                } else {
#pragma unroll 1
                    for (int g = 0; g < groups; g++) {
#pragma unroll
                        for (int j = 0; j < GS; j++) step(std::integral_constant<int, -1>{}, g, j);
                        ap -= 128 * GS;
                        bp += TS;
                        s_done += GS;
                        g_done++;
                    }
                }
            } else if constexpr (MIX && NG > 0) {
                // (no matrix work in this pass: what is owed, plainly)
                if (p_lo < p_hi) {
#pragma unroll
                    for (int bb = 0; bb < NB; bb++) {
#pragma unroll
                        for (int q = 0; q < 4; q++) mix_one(bb, q);
                        store_block(yp[bb], p_mb + (uint32_t)(32 * kT) * bb, p_lo, p_hi);
                    }
                }
            }
            if constexpr (MIX && NG > 0) p_lo = p_hi = 0;  // paid
            stamp(2);
            if (in_next) land(x);
            else if (has_next) land_edge(pass_of(nxt));
            stamp(3);
            if (active) {
                // a lane holds outputs 4 h + a (a = 0 .. 3) of tile n of each block: planes 2 f + pl at q = 4 a + 2 part + pl.
                // The planes meet in float64 with power-of-two weights (exact; the constant part of a u8 stream and
                // the scale 2^-S folded in), one rounding to float32.
                const double k3 = __hiloint2double((1023 - G.shift) << 20, 0), k2 = k3 * 256.0, k1 = k3 * 65536.0, k0 = k3 * 16777216.0;
                const double *dc = reinterpret_cast<const double *>(tabp + (sel ? tab_lds : 0) + (size_t)G.ne * 128);
                const double dcr = dc[0] * k3, dci = dc[1] * k3;
                float2 y[NB][4];
#pragma unroll
                for (int b = 0; b < NB; b++)
#pragma unroll
                    for (int aa = 0; aa < 4; aa++) {
                        float c2[2];
#pragma unroll
                        for (int pt = 0; pt < 2; pt++) {
                            const int q = 4 * aa + 2 * pt;
                            double v = __fma_rn((double)acc[1][b][q + 1], k3, pt ? dci : dcr);
                            v = __fma_rn((double)acc[1][b][q], k2, v);
                            v = __fma_rn((double)acc[0][b][q + 1], k1, v);
                            v = __fma_rn((double)acc[0][b][q], k0, v);
                            c2[pt] = (float)v;
                        }
                        y[b][aa] = make_float2(c2[0], c2[1]);
                    }
                stamp(4);
                const uint32_t mb = m_start + (uint32_t)n * kT + 4 * h;
                if constexpr (MIX && NG > 0) {
                    // owed: the next pass's matrix loop (or the drain behind the segment) mixes and stores them
#pragma unroll
                    for (int b = 0; b < NB; b++)
#pragma unroll
                        for (int q = 0; q < 4; q++) yp[b][q] = y[b][q];
                    p_mb = mb;
                    p_lo = v_lo;
                    p_hi = v_hi;
                    p_dphi = dphi;
                    p_ph = phi_r + (uint64_t)D * mb * dphi;
                } else {
                    // the elementwise program: a lane's outputs m = mb + 256 b + a are equally spaced in two
                    // directions inside one exactly-linear clock run (see hz_firmm.h)
#pragma unroll 1
                    for (int oi = 0; oi < ((EXP & 4) ? 0 : L.n_ops); oi++) {  // uniform
                        const EwOp &o = P.op[oi];
                        if (o.kind == EW_SCALE) {
#pragma unroll
                            for (int b = 0; b < NB; b++)
#pragma unroll
                                for (int q = 0; q < 4; q++) y[b][q] = make_float2(__fmul_rn(y[b][q].x, o.a), __fmul_rn(y[b][q].y, o.a));
                        } else if (o.kind == EW_ROTATE) {
#pragma unroll
                            for (int b = 0; b < NB; b++)
#pragma unroll
                                for (int q = 0; q < 4; q++) y[b][q] = go_cmul(y[b][q], make_float2(o.a, o.b));
                        } else if (L.shift_op == oi) {
                            // the stage's phase as a 64-bit accumulator in turns: exact increments, no float64
                            const uint64_t ph0 = phi_r + (uint64_t)D * mb * dphi;
#pragma unroll
                            for (int b = 0; b < NB; b++)
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    const uint64_t ph = ph0 + (uint64_t)(D * (32 * kT * b + q)) * dphi;
                                    float sn, cs;
                                    sincos_turns((uint32_t)(ph >> 32), sn, cs);
                                    y[b][q] = make_float2(__fmaf_rn(y[b][q].x, cs, -(y[b][q].y * sn)), __fmaf_rn(y[b][q].x, sn, y[b][q].y * cs));
                                }
                        } else {
                            // (programs with several Shift stages: float64 phases, as hz_firmm.h)
                            const double step = sel ? rv.step : ru.step;
                            const int64_t dj = (int64_t)((uint64_t)D * mb) - (int64_t)(sel ? rv.first : ru.first);
                            const double ts0 = __fma_rn((double)dj, step, sel ? rv.t0 : ru.t0);
                            double cs0, cc0, fs, fc;
                            sincos_late(__dmul_rn(o.tau_shift, ts0), cs0, cc0);
                            const double mult = l == 0 ? (double)D : (double)(32 * kT * D);
                            sincos_late(__dmul_rn(o.tau_shift, __dmul_rn(mult, step)), fs, fc);
                            const double e_s = __shfl(fs, 0), e_c = __shfl(fc, 0), b_s = __shfl(fs, 1), b_c = __shfl(fc, 1);
                            auto mul32 = [](float2 v, double c, double sn) {
                                const float cr = (float)c, ci = (float)sn;
                                return make_float2(__fmaf_rn(v.x, cr, -(v.y * ci)), __fmaf_rn(v.x, ci, v.y * cr));
                            };
#pragma unroll
                            for (int b = 0; b < NB; b++) {
                                double zc = cc0, zs = cs0;
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    y[b][q] = mul32(y[b][q], zc, zs);
                                    if (q < 3) {
                                        const double nc = __fma_rn(zc, e_c, -(zs * e_s)), ns = __fma_rn(zc, e_s, zs * e_c);
                                        zc = nc;
                                        zs = ns;
                                    }
                                }
                                if (b + 1 < NB) {
                                    const double nc = __fma_rn(cc0, b_c, -(cs0 * b_s)), ns = __fma_rn(cc0, b_s, cs0 * b_c);
                                    cc0 = nc;
                                    cs0 = ns;
                                }
                            }
                        }
                    }
                    stamp(5);
#pragma unroll
                    for (int b = 0; b < NB; b++) store_block(y[b], mb + (uint32_t)(32 * kT) * b, v_lo, v_hi);
                }
            }
            stamp(6);
            // The matrix pipe goes to the wave of higher priority, then to the OLDER one -- it is not shared: the
            // second wave of a SIMD only fills the first one's gaps, and with equal priorities it would run its
            // last pass alone at the end.  A wave on its first pass outranks one that has finished a pass, so the
            // two alternate pass by pass and finish together (measured: see DESIGN.md).
            if constexpr ((EXP & 32) == 0) __builtin_amdgcn_s_setprio(0);
            cur = nxt;
        }
        if constexpr (MIX && NG > 0) {
            // the segment's last pass of this wave
            if (p_lo < p_hi) {
#pragma unroll
                for (int bb = 0; bb < NB; bb++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) mix_one(bb, q);
                    store_block(yp[bb], p_mb + (uint32_t)(32 * kT) * bb, p_lo, p_hi);
                }
            }
        }
        r += has_b ? 2 : 1;
    }
    run_tasks();  // (a workgroup without a pass on the matrix path)
    stamp(7);
}

// hz_firmm2.hip
int launch_fir(hipStream_t stream, int num_cus, int fmt, unsigned D, const void *in, float2 *out, const float2 *hist,
               float2 *new_hist, const uint8_t *rhist, uint8_t *new_rhist, const float2 *taps, size_t n, const Geom &g,
               const Plan &L, const EwProgram &P, const Fix &F);

}  // namespace mm2
}  // namespace hz

// hz_nco.h -- NCO clock tables shared by hz_nco.hip and hz_chain.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/hzsdr.h"

struct hzsdr_ctx;

namespace hz {

constexpr int kNcoMaxSegs = 32;

// The exactly-linear runs of the NCO clock for one buffer (see
// hzsdr_nco_segments).  Up to kNcoMaxSegs runs travel by value in the kernel
// arguments (scalar loads, uniform scan); longer tables (tiny sample rates,
// tie binades) are read from device memory with a binary search.
struct NcoSegs {
    int n;      // runs held inline (0 when `big` is used)
    int big_n;  // runs in `big`
    const hzsdr_nco_segment *big;
    uint64_t first[kNcoMaxSegs];
    double t0[kNcoMaxSegs];
    double step[kNcoMaxSegs];
};

// The runs that can contain samples [j_lo, j_hi] (wave-uniform bounds): found
// ONCE per workgroup-sized span with scalar code; lanes then only look inside
// the window, which is a single run except at a run boundary.  Keeping the scan
// rolled and scalar matters: unrolled over 32 runs it pulled the whole table
// into SGPRs and spilled them through v_readlane/v_writelane.
struct NcoWin {
    int lo, hi;
};

__device__ __forceinline__ uint64_t nco_first(const NcoSegs &sg, int s) {
    return sg.big_n ? sg.big[s].first : sg.first[s];
}

__device__ __forceinline__ NcoWin nco_window(const NcoSegs &sg, uint64_t j_lo, uint64_t j_hi) {
    NcoWin w{0, 0};
    if (sg.big_n == 0) {
#pragma unroll 1
        for (int s = 1; s < sg.n; s++) {
            const uint64_t f = sg.first[s];
            if (f <= j_lo) w.lo = s;
            if (f <= j_hi) w.hi = s;
        }
    } else {
        int lo = 0, hi = sg.big_n - 1;  // last run with first <= j_lo
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (sg.big[mid].first <= j_lo) lo = mid; else hi = mid - 1;
        }
        w.lo = lo;
        hi = sg.big_n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (sg.big[mid].first <= j_hi) lo = mid; else hi = mid - 1;
        }
        w.hi = lo;
    }
    return w;
}

// The run that holds sample j and the run that holds sample j2 >= j, found by the lanes
// together: lane i compares entry i (ONE vector load of the table, 32 entries), a ballot counts
// the entries at or below.  `group` lanes (32 or 64, a power of two) starting at a multiple of
// `group` must call with the same j, j2 (a block's lanes); the inline table only.  The scalar
// scan of nco_window waits for a dependent scalar load per entry: 24 x ~200 cycles in front
// of EVERY block of a launch whose clock wraps (25 runs) -- measured, 44 us instead of 34.
__device__ __forceinline__ NcoWin nco_window_ballot(const NcoSegs &sg, uint64_t j, uint64_t j2, int group) {
    const int l = (int)(threadIdx.x & (unsigned)(group - 1));
    const bool valid = l < sg.n && l < kNcoMaxSegs;
    const uint64_t f = sg.first[valid ? l : 0];
    const uint64_t le1 = __ballot(valid && f <= j), le2 = __ballot(valid && f <= j2);
    const int sh = (int)((threadIdx.x & 63u) & ~(unsigned)(group - 1));  // this group's bits of the wave mask
    const uint64_t m = group >= 64 ? ~0ull : ((1ull << group) - 1);
    NcoWin w;
    w.lo = __popcll((le1 >> sh) & m) - 1;  // first[0] = 0 <= j: at least one bit
    w.hi = __popcll((le2 >> sh) & m) - 1;
    return w;
}

// every run: for callers whose sample index is not confined to a span
__device__ __forceinline__ NcoWin nco_window_all(const NcoSegs &sg) {
    return NcoWin{0, (sg.big_n ? sg.big_n : sg.n) - 1};
}

// ts for sample j (which must lie in the window's span)
__device__ __forceinline__ double nco_ts(const NcoSegs &sg, NcoWin w, uint64_t j) {
    uint64_t first;
    double t0, step;
    if (sg.big_n == 0 && w.hi - w.lo <= 3) {
        first = sg.first[w.lo];
        t0 = sg.t0[w.lo];
        step = sg.step[w.lo];
#pragma unroll 1
        for (int s = w.lo + 1; s <= w.hi; s++) {
            if (j >= sg.first[s]) {
                first = sg.first[s];
                t0 = sg.t0[s];
                step = sg.step[s];
            }
        }
    } else if (sg.big_n == 0) {
        // a wide window (a span over the short runs right behind the clock's 2*pi wrap): the scan above
        // waits for a dependent scalar load per run and sample -- 15 runs cost a fix-up task of the
        // matrix FIR 17 us; a per-lane binary search is five vector loads
        int lo = w.lo, hi = w.hi;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (sg.first[mid] <= j) lo = mid; else hi = mid - 1;
        }
        first = sg.first[lo];
        t0 = sg.t0[lo];
        step = sg.step[lo];
    } else {
        int lo = w.lo, hi = w.hi;  // last run with first <= j, per lane
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (sg.big[mid].first <= j) lo = mid; else hi = mid - 1;
        }
        first = sg.big[lo].first;
        t0 = sg.big[lo].t0;
        step = sg.big[lo].step;
    }
    // exact: the true value is representable.  The run offset is < 2^32 for any
    // buffer that fits in memory, so one 32-bit conversion replaces the 64-bit one.
    const uint64_t d = j - first;
    double k = (double)(uint32_t)d;
    if (d >> 32) k = __fma_rn((double)(uint32_t)(d >> 32), 4294967296.0, k);
    return __fma_rn(k, step, t0);
}

// Plans the next n clock values from *ts (advancing it) into `sg`; a table too
// long for the inline form is uploaded to a context scratch slot.
int nco_plan(hzsdr_ctx *ctx, uint64_t sample_rate, double *ts, size_t n, NcoSegs *sg);
// In-place shift of n samples at device pointer buf, advancing *ts.
// (ulp1: the float32 factor within one ulp of the reference's instead of bit-identical, hzsdr_nco_set_ulp1)
int nco_shift_device(hzsdr_ctx *ctx, uint64_t sample_rate, double *ts, double shift_hz, void *buf,
                     size_t n, bool ulp1 = false);

// hz_chain.hip: the <= 1-ulp Shift over four samples per lane (chain_map_kernel<c64, 4, SHAPE_SHIFT_ULP1>: four
// interleaved factor chains, 32-byte loads) on `nvec4` groups of four samples at a 32-byte aligned `buf`, the
// clock table `sg` counted from sample `base` of the call
// hz_chain.hip: the bit-exact Shift over `nvec2` vectors of two samples at a 16-byte aligned `buf`, in place
// (shift_exact_kernel: the factor by sincos_narrow, go_sincos where that cannot tell), the clock table counted
// from sample `base` of the call
void nco_shift_exact_map2(hzsdr_ctx *ctx, void *buf, size_t nvec2, uint64_t base, double tau_shift, const NcoSegs &sg);
void nco_shift_ulp1_map4(hzsdr_ctx *ctx, void *buf, size_t nvec4, uint64_t base, double tau_shift, const NcoSegs &sg);

}  // namespace hz

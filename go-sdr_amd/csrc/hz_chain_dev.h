// hz_chain_dev.h -- device side of the fused operator chains (see hz_chain.hip for the
// host side and the C-ABI): the elementwise program, the streaming terminals, the
// block-circular convolution kernel and the overlap-save FIR-decimate kernels.  A header
// so that tools/ can instantiate ONE kernel and look at its ISA / time it in isolation.
#pragma once
#include <math.h>

#include "hz_common.h"
#include "hz_device.h"
#include "hz_fft.h"
#include "hz_fftv.h"
#include "hz_nco.h"

namespace hz {

// ---- elementwise program -------------------------------------------------------------

constexpr int kMaxEw = 6;
enum EwKind { EW_SCALE = 1, EW_ROTATE = 2, EW_SHIFT = 3 };
struct EwOp {
    int kind;
    float a, b;        // scale: a = r; rotate: a + ib
    double tau_shift;  // shift: fl(2*pi * shift_hz)
};
struct EwProgram {
    int n;
    EwOp op[kMaxEw];
    NcoSegs segs;  // one clock serves every Shift stage: same rate, same start, same length
};

// `w`: the NCO runs that can contain sample j (nco_window of the caller's span)
__device__ __forceinline__ float2 ew_apply(const EwProgram &P, float2 v, uint64_t j, NcoWin w) {
    double ts = 0.0;
    bool have_ts = false;
#pragma unroll 1
    for (int i = 0; i < P.n; i++) {  // uniform
        const EwOp &o = P.op[i];
        if (o.kind == EW_SCALE) {
            v = make_float2(__fmul_rn(v.x, o.a), __fmul_rn(v.y, o.a));  // stream/gain.go:39-48
        } else if (o.kind == EW_ROTATE) {
            v = go_cmul(v, make_float2(o.a, o.b));  // stream/multiply.go:46-70
        } else {
            if (!have_ts) {
                ts = nco_ts(P.segs, w, j);
                have_ts = true;
            }
            double ph = __dmul_rn(o.tau_shift, ts);  // stream/shifter.go:81
            float s, c;
            go_sincos32(ph, s, c);  // (complex64(math.Sincos): hz_device.h)
            v = go_cmul(v, make_float2(c, s));  // :82
        }
    }
    return v;
}

// The same program over W consecutive samples j0 .. j0+W-1 at once: the op loop is
// outside, the sample loop inside and unrolled, so the W independent Sincos /
// multiply chains interleave (instruction-level parallelism within a lane; the
// one-sample form serialises them behind the rolled op loop).
// LATE: the call mixes FILTER OUTPUTS of a late block (error-bounded, not bit-defined):
// Shift uses sincos_late instead of the operation-for-operation math.Sincos.
// LATE = 2: the Sincos of a late call through the phase in turns (sincos_turns32 below): the float64 product
// tau ts as the reference forms it, reduced to a 32-bit fraction of a turn, float32 polynomials -- half the issue
// cycles of sincos_late (the fix-up tasks of hz_firmm2.h, whose outputs are held to the FIR bound).
template <int W, int LATE = 0>
__device__ __forceinline__ void ew_apply_n(const EwProgram &P, float2 (&v)[W], uint64_t j0, NcoWin w,
                                           uint64_t stride = 1) {
    double ts[W];
    bool have_ts = false;
#pragma unroll 1
    for (int i = 0; i < P.n; i++) {  // uniform
        const EwOp &o = P.op[i];
        if (o.kind == EW_SCALE) {
#pragma unroll
            for (int l = 0; l < W; l++) v[l] = make_float2(__fmul_rn(v[l].x, o.a), __fmul_rn(v[l].y, o.a));
        } else if (o.kind == EW_ROTATE) {
#pragma unroll
            for (int l = 0; l < W; l++) v[l] = go_cmul(v[l], make_float2(o.a, o.b));
        } else {
            if (!have_ts) {
#pragma unroll
                for (int l = 0; l < W; l++) ts[l] = nco_ts(P.segs, w, j0 + l * stride);
                have_ts = true;
            }
            float s[W], c[W];
#pragma unroll
            for (int l = 0; l < W; l++) {
                if constexpr (LATE == 2) {
                    sincos_turns32(turns32(__dmul_rn(o.tau_shift, ts[l])), s[l], c[l]);
                } else if constexpr (LATE == 1) {
                    double sd, cd;
                    sincos_late(__dmul_rn(o.tau_shift, ts[l]), sd, cd);
                    s[l] = (float)sd, c[l] = (float)cd;
                } else {
                    go_sincos32(__dmul_rn(o.tau_shift, ts[l]), s[l], c[l]);  // (reference order: bit for bit, hz_device.h)
                }
            }
#pragma unroll
            for (int l = 0; l < W; l++) v[l] = go_cmul(v[l], make_float2(c[l], s[l]));
        }
    }
}

// The late mixer over W filter outputs at time indices idx[l] + lane of a block whose first
// input sample is p0 (output i sits at stream position p0 + i*D): ew_apply_n<W, true> with a
// position per slot instead of a common stride.
template <int W>
__device__ __forceinline__ void ew_apply_at(const EwProgram &P, float2 (&v)[W], int64_t p0, int lane,
                                            const unsigned (&idx)[W], unsigned D, NcoWin w) {
    double ts[W];
    bool have_ts = false;
#pragma unroll 1
    for (int i = 0; i < P.n; i++) {  // uniform
        const EwOp &o = P.op[i];
        if (o.kind == EW_SCALE) {
#pragma unroll
            for (int l = 0; l < W; l++) v[l] = make_float2(__fmul_rn(v[l].x, o.a), __fmul_rn(v[l].y, o.a));
        } else if (o.kind == EW_ROTATE) {
#pragma unroll
            for (int l = 0; l < W; l++) v[l] = go_cmul(v[l], make_float2(o.a, o.b));
        } else {
            if (!have_ts) {
#pragma unroll
                for (int l = 0; l < W; l++) ts[l] = nco_ts(P.segs, w, (uint64_t)(p0 + (int64_t)(idx[l] + lane) * D));
                have_ts = true;
            }
            double s[W], c[W];
#pragma unroll
            for (int l = 0; l < W; l++) sincos_late(__dmul_rn(o.tau_shift, ts[l]), s[l], c[l]);
#pragma unroll
            for (int l = 0; l < W; l++) v[l] = go_cmul(v[l], make_float2((float)c[l], (float)s[l]));
        }
    }
}

// The late mixer over the outputs of ONE lane that are equally spaced in time: slot m
// (m0 <= m < W, v[m] valid from m0 on) sits at stream position pos0 + m*stride, all inside the
// clock run `w` (a late block lies in one run), where the clock is exactly linear.  A Shift
// stage is then a geometric sequence, exp(i tau ts_m) = exp(i tau ts_m0) * exp(i tau stride step)^(m-m0):
// two Sincos per lane and stage and one float64 complex product per output, instead of a
// Sincos per output.  The phases are those of exact arithmetic on the run's (t0, step); the
// reference rounds tau*ts_m per sample (|tau ts| * 1e-16 <= 4e-8 rad), a difference of a few
// 1e-8 relative in an output held to an FFT error bound of 6e-7.
template <int W>
__device__ __forceinline__ void ew_apply_seq(const EwProgram &P, float2 (&v)[W], int m0, uint64_t pos0,
                                             uint64_t stride, NcoWin w) {
#pragma unroll 1
    for (int i = 0; i < P.n; i++) {  // uniform
        const EwOp &o = P.op[i];
        if (o.kind == EW_SCALE) {
#pragma unroll
            for (int m = 0; m < W; m++) v[m] = make_float2(__fmul_rn(v[m].x, o.a), __fmul_rn(v[m].y, o.a));
        } else if (o.kind == EW_ROTATE) {
#pragma unroll
            for (int m = 0; m < W; m++) v[m] = go_cmul(v[m], make_float2(o.a, o.b));
        } else {
            // z = exp(i tau ts(pos0 + m0 stride)), q = exp(i tau stride step)
            double zs, zc, qs, qc;
            const double ts0 = nco_ts(P.segs, w, pos0 + (uint64_t)m0 * stride);
            const double step = P.segs.big_n ? P.segs.big[w.lo].step : P.segs.step[w.lo];
            sincos_late(__dmul_rn(o.tau_shift, ts0), zs, zc);
            sincos_late(__dmul_rn(o.tau_shift, __dmul_rn((double)stride, step)), qs, qc);
#pragma unroll
            for (int m = 0; m < W; m++) {
                if (m >= m0) {  // uniform
                    v[m] = go_cmul(v[m], make_float2((float)zc, (float)zs));
                    const double nc = __fma_rn(zc, qc, -(zs * qs)), ns = __fma_rn(zc, qs, zs * qc);
                    zc = nc;
                    zs = ns;
                }
            }
        }
    }
}

// The two most common programs -- Shift, and Shift then Gain (BASELINE config 2) -- spelt
// out, without the op loop: same operations in the same order, but straight-line code the
// scheduler can overlap with the loads and stores around it.  SHAPE 0 = interpret.
// SHAPE_*_ULP1 (hzsdr_chain_shift_ulp1): the rotation factor through the phase in turns and float32 polynomials
// (sincos_turns32) -- within one float32 ulp of the factor the reference forms from math.Sincos of the same
// float64 product, a third of the vector instructions.
enum EwShape { SHAPE_ANY = 0, SHAPE_SHIFT = 1, SHAPE_SHIFT_GAIN = 2, SHAPE_SHIFT_ULP1 = 3, SHAPE_SHIFT_GAIN_ULP1 = 4 };
template <int W, int SHAPE>
__device__ __forceinline__ void ew_apply_shape(const EwProgram &P, float2 (&v)[W], uint64_t j0, NcoWin w) {
    if constexpr (SHAPE == SHAPE_ANY) {
        ew_apply_n<W>(P, v, j0, w);
    } else if constexpr (SHAPE == SHAPE_SHIFT_ULP1 || SHAPE == SHAPE_SHIFT_GAIN_ULP1) {
        float s[W], c[W];
#pragma unroll
        for (int l = 0; l < W; l++) sincos_turns32(turns32(__dmul_rn(P.op[0].tau_shift, nco_ts(P.segs, w, j0 + l))), s[l], c[l]);
#pragma unroll
        for (int l = 0; l < W; l++) {
            v[l] = go_cmul(v[l], make_float2(c[l], s[l]));
            if constexpr (SHAPE == SHAPE_SHIFT_GAIN_ULP1)
                v[l] = make_float2(__fmul_rn(v[l].x, P.op[1].a), __fmul_rn(v[l].y, P.op[1].a));
        }
    } else {
        float s[W], c[W];
#pragma unroll
        for (int l = 0; l < W; l++) go_sincos32(__dmul_rn(P.op[0].tau_shift, nco_ts(P.segs, w, j0 + l)), s[l], c[l]);
#pragma unroll
        for (int l = 0; l < W; l++) {
            v[l] = go_cmul(v[l], make_float2(c[l], s[l]));
            if constexpr (SHAPE == SHAPE_SHIFT_GAIN)
                v[l] = make_float2(__fmul_rn(v[l].x, P.op[1].a), __fmul_rn(v[l].y, P.op[1].a));
        }
    }
}

template <int FMT> struct Raw;
template <> struct Raw<HZSDR_FMT_C64> {
    using t = float2;
    static __device__ __forceinline__ float2 cvt(float2 r) { return r; }
    static __device__ __forceinline__ float2 cvt_late(float2 r) { return r; }
};
template <> struct Raw<HZSDR_FMT_U8> {
    using t = uint16_t;
    static __device__ __forceinline__ float2 cvt(uint16_t r) { return make_float2(u8_to_f32(r & 0xFF), u8_to_f32(r >> 8)); }
    // b - 127.5, the converter without its division (late blocks: the 1/127.5 is in the filter)
    static __device__ __forceinline__ float2 cvt_late(uint16_t r) {
        return make_float2(__fsub_rn((float)(r & 0xFF), 127.5f), __fsub_rn((float)(r >> 8), 127.5f));
    }
};
template <> struct Raw<HZSDR_FMT_I8> {
    using t = uint16_t;
    static __device__ __forceinline__ float2 cvt(uint16_t r) { return make_float2(i8_to_f32((int8_t)(r & 0xFF)), i8_to_f32((int8_t)(r >> 8))); }
    static __device__ __forceinline__ float2 cvt_late(uint16_t r) { return cvt(r); }
};
template <> struct Raw<HZSDR_FMT_I16> {
    using t = uint32_t;
    static __device__ __forceinline__ float2 cvt(uint32_t r) { return make_float2(i16_to_f32((int16_t)(r & 0xFFFF)), i16_to_f32((int16_t)(r >> 16))); }
    static __device__ __forceinline__ float2 cvt_late(uint32_t r) { return cvt(r); }
};

// sample j of the buffer after conversion and the elementwise stages
template <int FMT>
__device__ __forceinline__ float2 chain_sample(const void *in, const EwProgram &P, uint64_t j) {
    using R = typename Raw<FMT>::t;
    return ew_apply(P, Raw<FMT>::cvt(((const R *)in)[j]), j, nco_window_all(P.segs));
}

// ---- streaming terminals ---------------------------------------------------------------

// TERM 0: out[j] = f(j).  W samples per lane per step (vector load / store).
template <int FMT, int W, int SHAPE = SHAPE_ANY>
__global__ __launch_bounds__(kThreads) void chain_map_kernel(const void *__restrict__ in,
                                                             float2 *__restrict__ out, size_t nvec,
                                                             uint64_t base, EwProgram P) {
    using R = typename Raw<FMT>::t;
    struct alignas(sizeof(R) * W) RV { R v[W]; };
    struct alignas(8 * W) OV { float2 v[W]; };
    // a workgroup owns a contiguous tile per trip and issues its U loads per lane
    // back to back before the arithmetic (memory-level parallelism, see hz_nco.hip)
#ifdef HZ_MAP_U4
    constexpr int U = W >= 4 ? HZ_MAP_U4 : 2;
#else
    constexpr int U = W >= 4 ? 1 : 2;
#endif
    const size_t tile = (size_t)kThreads * U;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < nvec; t0 += (size_t)gridDim.x * tile) {
        const uint64_t j_lo = base + t0 * W;
        const NcoWin w = nco_window(P.segs, j_lo, j_lo + tile * W - 1);
        RV x[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * kThreads + threadIdx.x;
            if (i < nvec) x[u] = ((const RV *)in)[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * kThreads + threadIdx.x;
            if (i < nvec) {
                OV o;
#pragma unroll
                for (int l = 0; l < W; l++) o.v[l] = Raw<FMT>::cvt(x[u].v[l]);
                ew_apply_shape<W, SHAPE>(P, o.v, base + i * W, w);
                ((OV *)out)[i] = o;
            }
        }
    }
}

// out[j] = Shift(convert(in[j])) (and Gain behind it, GAIN) -- BASELINE config 2, the ShiftReader / ShiftBuffer
// map and hzsdr_nco_shift -- bit for bit as chain_map_kernel forms it from math.Sincos operation for operation,
// at two thirds of its vector instructions (which bound it: 45 float64 instructions per sample there, 38 here,
// and the quadrant logic in float32).  A lane holds U vectors of two samples; a tile that is ONE clock run whose
// phases stay inside sincos_narrow's range takes the straight path: the clock by one exact fma from the lane's
// run offset, the factor from sincos_narrow.  The vectors a lane cannot decide there (2^-19 of them), and every
// vector of any other tile (a run boundary inside, the ragged last tile, tiny or huge phases, the long clock
// table), go to the wave's queue in LDS -- their input is left where it is: the map may be in place -- and the
// wave works the queue off behind the tile through ONE rolled copy of go_sincos.  (With go_sincos inline behind
// each check the straight path lost more to the compiler's register and code layout than the check saves;
// tools/nco_ablate.hip.)
#ifndef HZ_SHIFT_U
#define HZ_SHIFT_U 4
#endif
#ifndef HZ_SHIFT_TPB
#define HZ_SHIFT_TPB 128
#endif
constexpr int kShiftU = HZ_SHIFT_U, kShiftThreads = HZ_SHIFT_TPB;  // (two waves per workgroup: 46.5 us where four take 47.8, tools/nco_ablate.hip)
// Round 6, measured and left off (-DHZ_SHIFT_PF builds it: csrc/Makefile EXTRA): a workgroup WALKS several tiles and
// asks for the NEXT tile's vectors before it works on the ones it holds.  From HBM (a rotation of six buffer pairs,
// tools/shift_time.py, profiles/r06_shift_time.txt) 55-57 us per 2^24 samples with 12 walking workgroups per CU, 59
// with 6, 51-53 with 24 -- against 51.7-53 for one tile per workgroup: what the map lacks from HBM is not overlap
// inside a wave but workgroups in flight, and a walk has fewer.
#ifdef HZ_SHIFT_PF
constexpr bool kShiftPrefetch = true;
#else
constexpr bool kShiftPrefetch = false;
#endif
template <int FMT, bool GAIN, bool NT = false>
__global__ __launch_bounds__(kShiftThreads) void shift_exact_kernel(const void *in, float4 *out, size_t nvec, uint64_t base, EwProgram P) {
    using R = typename Raw<FMT>::t;
    struct alignas(sizeof(R) * 2) RV { R v[2]; };
    constexpr int U = kShiftU, kW = kShiftThreads / 64, TPB = kShiftThreads;
    __shared__ unsigned q_n[kW];
    __shared__ unsigned short q[kW][64 * U];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) q_n[wave] = 0;
    const double tau_shift = P.op[0].tau_shift;
    const size_t tile = (size_t)TPB * U, hop = (size_t)gridDim.x * tile;
    auto fetch = [&](RV(&a)[U], size_t t0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const RV *const p = (const RV *)in + (t0 + (size_t)u * TPB + threadIdx.x);
            a[u] = NT ? nt_load(p) : *p;
        }
    };
    RV a[U], nx[U];
    size_t t0 = (size_t)blockIdx.x * tile;
    if (kShiftPrefetch && t0 + tile <= nvec) fetch(a, t0);  // (whole tiles only: the ragged last one goes to the queue)
    for (; t0 < nvec; t0 += hop) {
#ifdef HZ_SHIFT_EARLY
        // (-DHZ_SHIFT_EARLY, measured and left off: the tile's vectors asked for BEFORE the tile's clock run is looked up
        // -- a scan of the run table and a dozen float64 instructions on the tile's bounds stand between a workgroup's
        // start and its loads.  From HBM 49.8-50.1 us out of place against 51.0-51.2, 52.4-52.7 in place against
        // 50.8-51.3: nothing)
        const bool whole_early = !kShiftPrefetch && t0 + tile <= nvec;
        if (whole_early) fetch(a, t0);
#endif
        const uint64_t j_lo = base + 2 * t0;
        const NcoWin w = nco_window(P.segs, j_lo, j_lo + 2 * tile - 1);
        const uint64_t d0 = j_lo - nco_first(P.segs, w.lo);
        bool straight = w.lo == w.hi && P.segs.big_n == 0 && d0 + 2 * tile < (1ull << 32) && t0 + tile <= nvec;
        double step = 0, tb = 0;
        if (straight) {
            step = P.segs.step[w.lo], tb = P.segs.t0[w.lo];
            // the tile's phases: tau ts is monotonic in ts, ts in the sample; the smallest NONZERO one counts below
            const double ts_lo = __fma_rn((double)(uint32_t)d0, step, tb), ts_1 = __fma_rn((double)((uint32_t)d0 + 1u), step, tb);
            const double ts_hi = __fma_rn((double)((uint32_t)d0 + 2u * (uint32_t)tile - 1u), step, tb);
            const double x_lo = fabs(__dmul_rn(tau_shift, ts_lo > 0.0 ? ts_lo : ts_1)), x_hi = fabs(__dmul_rn(tau_shift, ts_hi));
            straight = ts_lo >= 0.0 && step > 0.0 && (x_lo >= 8.673617379884035e-19 || tau_shift == 0.0) && x_hi < 536870912.0;  // 2^-60, 2^29
        }
        // (the next tile of this workgroup's walk, in flight under this tile's arithmetic: a whole tile, so that the
        // loads need no mask -- the map may be in place, but tiles do not overlap)
        const bool more = kShiftPrefetch && t0 + hop + tile <= nvec;  // uniform
        if (more) fetch(nx, t0 + hop);
        if (__builtin_amdgcn_readfirstlane((int)straight)) {
#ifdef HZ_SHIFT_EARLY
            if constexpr (false) fetch(a, t0);
#else
            if constexpr (!kShiftPrefetch) fetch(a, t0);
#endif
            const double k0 = (double)((uint32_t)d0 + 2u * threadIdx.x);
#pragma unroll
            for (int u = 0; u < U; u++) {
                float sl, cl, sh, ch;
                const double kl = k0 + (double)(2 * u * TPB), kh = k0 + (double)(2 * u * TPB + 1);  // (exact)
                bool ok = sincos_narrow(__dmul_rn(tau_shift, __fma_rn(kl, step, tb)), sl, cl);  // stream/shifter.go:81
                ok &= sincos_narrow(__dmul_rn(tau_shift, __fma_rn(kh, step, tb)), sh, ch);
                float2 l = go_cmul(Raw<FMT>::cvt(a[u].v[0]), make_float2(cl, sl));  // :82
                float2 h = go_cmul(Raw<FMT>::cvt(a[u].v[1]), make_float2(ch, sh));
                if constexpr (GAIN) {  // stream/gain.go:39-48
                    l = make_float2(__fmul_rn(l.x, P.op[1].a), __fmul_rn(l.y, P.op[1].a));
                    h = make_float2(__fmul_rn(h.x, P.op[1].a), __fmul_rn(h.y, P.op[1].a));
                }
                if (ok) {
                    float4 *const p = out + (t0 + (size_t)u * TPB + threadIdx.x);
                    if (NT) nt_store(p, make_float4(l.x, l.y, h.x, h.y)); else *p = make_float4(l.x, l.y, h.x, h.y);
                } else q[wave][atomicAdd(&q_n[wave], 1u)] = (unsigned short)(u * TPB + threadIdx.x);
            }
        } else {
#pragma unroll 1
            for (int u = 0; u < U; u++)
                if (t0 + (size_t)u * TPB + threadIdx.x < nvec) q[wave][atomicAdd(&q_n[wave], 1u)] = (unsigned short)(u * TPB + threadIdx.x);
        }
        const unsigned nq = q_n[wave];  // (the LDS operations of one wave complete in order)
        if (nq) {
#pragma unroll 1
            for (unsigned e = lane; e < nq; e += 64) {
                const size_t i = t0 + q[wave][e];
                const RV x = ((const RV *)in)[i];
                float2 *const o2 = reinterpret_cast<float2 *>(out + i);
#pragma unroll 1
                for (int h = 0; h < 2; h++) {
                    double sd, cd;
                    go_sincos(__dmul_rn(tau_shift, nco_ts(P.segs, w, base + 2 * i + h)), sd, cd);
                    float2 v = go_cmul(Raw<FMT>::cvt(h ? x.v[1] : x.v[0]), make_float2((float)cd, (float)sd));
                    if constexpr (GAIN) v = make_float2(__fmul_rn(v.x, P.op[1].a), __fmul_rn(v.y, P.op[1].a));
                    o2[h] = v;
                }
            }
            if (lane == 0) q_n[wave] = 0;
        }
        if constexpr (kShiftPrefetch) {
            if (more) {
#pragma unroll
                for (int u = 0; u < U; u++) a[u] = nx[u];
            }
        }
    }
}

// DecimateReader: 32 Ki-sample blocks, `per` = 32768 / factor outputs per block,
// out[blk*per + i] = f(blk*32768 + i*factor)  (stream/decimate.go:34-101)
template <int FMT>
__global__ __launch_bounds__(kThreads) void chain_decimate_kernel(const void *__restrict__ in,
                                                                  float2 *__restrict__ out,
                                                                  size_t n_out, size_t per,
                                                                  size_t factor, EwProgram P) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += stride) {
        const size_t blk = o / per, i = o - blk * per;
        out[o] = chain_sample<FMT>(in, P, blk * kReaderBlock + i * factor);
    }
}

// DownsampleReader: boxcar over `factor` samples inside 32 Ki-sample blocks
// (stream/downsample.go:47-127), accumulated in order from +0.
template <int FMT>
__global__ __launch_bounds__(kThreads) void chain_downsample_kernel(const void *__restrict__ in,
                                                                    float2 *__restrict__ out,
                                                                    size_t n_out, size_t per,
                                                                    unsigned factor, EwProgram P) {
    const float div = (float)factor;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += stride) {
        const size_t blk = o / per, i = o - blk * per;
        const size_t j0 = blk * kReaderBlock + i * factor;
        float sr = 0.0f, si = 0.0f;
        for (unsigned k = 0; k < factor; k++) {
            float2 c = chain_sample<FMT>(in, P, j0 + k);
            sr = __fadd_rn(sr, c.x);
            si = __fadd_rn(si, c.y);
        }
        out[o] = make_float2(__fdiv_rn(sr, div), __fdiv_rn(si, div));
    }
}

// ---- reference convolution: block-circular, one workgroup pass per block ----------------

// out stream position g -> optional DecimateReader pick over the conv output.
__device__ __forceinline__ void conv_store(float2 *out, size_t g, float2 v, unsigned dec, size_t per) {
    if (dec <= 1) {
        out[g] = v;
    } else {
        const size_t blk = g / kReaderBlock, i = g - blk * kReaderBlock;
        const size_t q = i / dec;
        if (q * dec == i && q < per) out[blk * per + q] = v;
    }
}

// Stage N post-elementwise samples of one block into LDS in natural order with a
// ROLLED loop (two samples per lane per trip: one 16-B LDS store, the float64
// sincos of the Shift stage instantiated twice, not N/TPT times -- the unrolled
// register-direct form needs 256 VGPRs and drops to one wave per SIMD).
// FIR history for the next run = the last `off` samples of (old history ++ this
// run): every staged position p in [n_in - off, n_in) is written to
// new_hist[p - (n_in - off)] by whichever block stages it (blocks overlap by
// N - hop positions; both write the same value), so no second kernel is needed.
__device__ __forceinline__ void keep_history(float2 *new_hist, int64_t p, size_t n_in, unsigned off,
                                             bool live, float2 a, float2 b) {
    if (!new_hist) return;  // uniform
    const int64_t h0 = (int64_t)n_in - (int64_t)off;
    if (live && p >= h0 && p < (int64_t)n_in) new_hist[p - h0] = a;
    if (live && p + 1 >= h0 && p + 1 < (int64_t)n_in) new_hist[p + 1 - h0] = b;
}

// lds[i] = sample at stream position p0 + i; positions < 0 come from `hist`
// (`off` entries, may be null = zeros), positions >= n_in are zero.
template <int N, int FMT>
__device__ __forceinline__ void stage_block(float2 *lds, const void *in, const EwProgram &P, int64_t p0,
                                            size_t n_in, const float2 *hist, unsigned off, int lane,
                                            bool live, float2 *new_hist = nullptr) {
    using R = typename Raw<FMT>::t;
    constexpr int TPT = fft_tpt(N);
    struct alignas(sizeof(R) * 2) RV { R v[2]; };
    // pair loads need 2*sizeof(R) alignment of in + p0 (i0 is even); mod-2 arithmetic wraps safely
    const bool vec_ok = ((((uintptr_t)in / sizeof(R)) + (uint64_t)p0) & 1) == 0;
#pragma unroll 1
    for (int ib = 0; ib < N; ib += TPT * 2) {
        const int i0 = ib + lane * 2;
        const int64_t p = p0 + i0;
        const int64_t span_lo = p0 + ib;  // uniform: the TPT*2 samples this trip covers
        const NcoWin w = nco_window(P.segs, span_lo < 0 ? 0 : (uint64_t)span_lo,
                                    span_lo + TPT * 2 <= 0 ? 0 : (uint64_t)(span_lo + TPT * 2 - 1));
        float2 a = make_float2(0.f, 0.f), b = a;
        if (live) {
            // ONE instantiation of the elementwise program for the pair (p, p+1):
            // lanes whose position lies outside [0, n_in) run it on a dummy value and
            // are overridden afterwards (history or zero), instead of a second and
            // third inlined copy of the float64 Sincos on a divergent path.
            const bool in0 = p >= 0 && (size_t)p < n_in, in1 = p + 1 >= 0 && (size_t)(p + 1) < n_in;
            RV x{};
            if (vec_ok && in0 && in1) {
                x = *reinterpret_cast<const RV *>((const R *)in + p);
            } else {
                if (in0) x.v[0] = ((const R *)in)[p];
                if (in1) x.v[1] = ((const R *)in)[p + 1];
            }
            float2 ab[2] = {Raw<FMT>::cvt(x.v[0]), Raw<FMT>::cvt(x.v[1])};
            ew_apply_n<2>(P, ab, (uint64_t)p, w);  // (uint64)(-1) + 1 wraps to sample 0
            if (in0) a = ab[0]; else if (p < 0 && hist) a = hist[p + off];
            if (in1) b = ab[1]; else if (p + 1 < 0 && hist) b = hist[p + 1 + off];
        }
        keep_history(new_hist, p, n_in, off, live, a, b);
        *reinterpret_cast<float4 *>(lds + i0) = make_float4(a.x, a.y, b.x, b.y);
    }
}

// first-pass register image from staged LDS
template <int N> __device__ __forceinline__ void load_edge_from_lds(FftRegs<N> &R, const float2 *lds, int lane) {
    if constexpr (fft_odd(N)) r2_load_lds<N>(R, lds, lane); else r4_load_lds<N>(R, lds, lane);
}

// STAGED = the source needs the elementwise program (or a non-c64 format):
// samples go through stage_block; otherwise c64 samples are loaded straight
// into the first pass's registers.
template <int N, int FMT, bool STAGED>
__global__ __launch_bounds__(fft_block(N), fft_waves(N)) void conv_blocks_kernel(const void *in, float2 *out,
                                                                   const float2 *__restrict__ filt,
                                                                   const float2 *__restrict__ tw,
                                                                   size_t nblocks, unsigned dec,
                                                                   size_t per, EwProgram P) {
    constexpr int TPT = fft_tpt(N), XPB = fft_xpb(N), CNT = N / TPT;
    __shared__ __attribute__((aligned(16))) float2 lds_all[XPB * N];
    const int sub = threadIdx.x / TPT, lane = threadIdx.x % TPT;
    float2 *lds = lds_all + sub * N;
    {   // one workgroup per XPB blocks, no grid-stride loop (see hz_fft.hip: LICM of twiddles)
        const size_t b0 = (size_t)blockIdx.x * XPB;
        const size_t b = b0 + sub;
        const bool live = b < nblocks;
        FftRegs<N> R;
        __syncthreads();
        if constexpr (STAGED) {
            stage_block<N, FMT>(lds, in, P, (int64_t)(b * N), ~(size_t)0, nullptr, 0, lane, live);
            __syncthreads();
            load_edge_from_lds<N>(R, lds, lane);
            fft_forward_regs<N, true>(R, lds, tw, lane);
        } else {
#pragma unroll
            for (int q = 0; q < CNT; q++) {
                const int idx = fft_odd(N) ? edge2_index<N>(q, lane) : edge4_index<N>(q, lane);
                R.v[q] = live ? Raw<FMT>::cvt(((const typename Raw<FMT>::t *)in)[b * N + idx]) : make_float2(0.f, 0.f);
            }
            fft_forward_regs<N>(R, lds, tw, lane);
        }
#pragma unroll
        for (int q = 0; q < CNT; q++)  // freq1[i] = freq1[i] * freq[i], fft/convolution.go:187-189
            R.v[q] = go_cmul(R.v[q], filt[edge4_index<N>(q, lane)]);
        fft_backward_regs<N>(R, lds, tw, lane);
        if (live) {
#pragma unroll
            for (int q = 0; q < CNT; q++) {
                const int idx = fft_odd(N) ? edge2_index<N>(q, lane) : edge4_index<N>(q, lane);
                conv_store(out, b * N + idx, R.v[q], dec, per);
            }
        }
    }
}

// ---- north-star FIR-decimate: overlap-save ------------------------------------------------

// Block b covers input positions [b*hop - off, b*hop - off + N); positions < 0
// come from `hist` (the last `off` post-elementwise samples of the previous
// run), positions >= n_in are zero.  Circular outputs at idx in [off, off+hop)
// on the decimation grid are y[(b*hop + idx - off) / D].
// FOLD = D when the decimation folds into the spectrum (D a power of two that
// divides the per-lane bin count): y[D i] = IFFT_{N/D}( sum_q Y[k + q N/D] )[i],
// so the backward transform is N/D points instead of N, and the fold itself is
// lane-local in the edge4 register image.  FOLD = 0: full backward transform,
// outputs picked on the decimation grid.
template <int N, int FMT, int FOLD>
__global__ __launch_bounds__(fft_block(N), fft_waves(N)) void fir_decimate_kernel(
    const void *in, float2 *out, const float2 *__restrict__ hist, float2 *__restrict__ new_hist,
    const float2 *__restrict__ hfreq,
    const float2 *__restrict__ tw, const float2 *__restrict__ tw_small, size_t nblocks, size_t n_in,
    unsigned hop, unsigned off, unsigned D, EwProgram P) {
    constexpr int TPT = fft_tpt(N), CNT = N / TPT;
    static_assert(fft_xpb(N) == 1, "fir_decimate_kernel: one block per workgroup");
    __shared__ __attribute__((aligned(16))) float2 lds[N];
    const int lane = threadIdx.x;
    {   // one workgroup per overlap-save block (no grid-stride loop: see hz_fft.hip)
        const size_t b = blockIdx.x;
        FftRegs<N> R;
        __syncthreads();
        stage_block<N, FMT>(lds, in, P, (int64_t)(b * hop) - (int64_t)off, n_in, hist, off, lane, true, new_hist);
        __syncthreads();
        load_edge_from_lds<N>(R, lds, lane);
        fft_forward_regs<N, true>(R, lds, tw, lane);
#pragma unroll
        for (int q = 0; q < CNT; q++) R.v[q] = cmulf(R.v[q], hfreq[edge4_index<N>(q, lane)]);
        if constexpr (FOLD == 0) {
            fft_backward_regs<N>(R, lds, tw, lane);
#pragma unroll
            for (int q = 0; q < CNT; q++) {
                const unsigned idx = fft_odd(N) ? edge2_index<N>(q, lane) : edge4_index<N>(q, lane);
                if (idx >= off && idx < off + hop && ((idx - off) % D) == 0) {
                    const size_t p = b * hop + (idx - off);
                    if (p < n_in) out[p / D] = R.v[q];
                }
            }
        } else {
            constexpr int M = N / FOLD, S = CNT / FOLD, TPTM = fft_tpt(M), CNTM = M / TPTM;
            static_assert(S >= 1 && M >= 4, "fold geometry");
            // register slot q holds bin lane + TPT*m(q), m(q) = (q>>2) + (q&3)*(CNT/4);
            // folded bin lane + TPT*s collects every m with m % S == s
            float2 z[S];
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) z[s2] = make_float2(0.f, 0.f);
#pragma unroll
            for (int q = 0; q < CNT; q++) {
                constexpr int B4 = CNT / 4;
                const int m = (q >> 2) + (q & 3) * B4;
                z[m % S] = cadd(z[m % S], R.v[q]);
            }
            __syncthreads();  // forward's last-pass LDS reads are done
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) lds[lane + TPT * s2] = z[s2];
            __syncthreads();
            const int g = lane / TPTM, l2 = lane % TPTM;
            FftRegs<M> Q;
#pragma unroll
            for (int q = 0; q < CNTM; q++) Q.v[q] = lds[edge4_index<M>(q, l2)];
            fft_backward_regs<M>(Q, lds + g * M, tw_small, l2);  // groups > 0 redo it in their own region
            if (g == 0) {
                const unsigned i_lo = off / FOLD, i_hi = (off + hop) / FOLD;
#pragma unroll
                for (int q = 0; q < CNTM; q++) {
                    const unsigned i = fft_odd(M) ? edge2_index<M>(q, l2) : edge4_index<M>(q, l2);
                    if (i >= i_lo && i < i_hi) {
                        const size_t m_out = b * (hop / FOLD) + (i - i_lo);
                        if (m_out * FOLD < n_in) out[m_out] = Q.v[q];
                    }
                }
            }
        }
    }
}

// ==== the same two kernels on the packed-math core (hz_fftv.h), N = 256 .. 8192 ===============
// LDS is dynamic (`extern __shared__`): N = 8192 needs 68 KiB per workgroup, above the 64 KiB
// static limit and well inside the 160 KiB a CU has (launch_fv sets the attribute).

using fv::cf;
using fv::cf4;

using fv::FvTabs;
using fv::PolyTabs;

__device__ __forceinline__ cf *fv_lds() {
    extern __shared__ __attribute__((aligned(16))) unsigned char hz_dyn_lds[];
    return reinterpret_cast<cf *>(hz_dyn_lds);
}

// stage_block for the padded LDS image of the workgroup core
// FILTERED: the staged samples feed a FIR whose outputs are held to an error bound (the
// north-star FIR-decimate), so Shift may use sincos_late; the reference's own
// ConvolutionReader chain keeps the operation-for-operation math.Sincos.
template <int N, int FMT, bool FILTERED = false>
__device__ __forceinline__ void stage_block16(float2 *lds, const void *in, const EwProgram &P, int64_t p0,
                                              size_t n_in, const float2 *hist, unsigned off, int lane,
                                              bool live, float2 *new_hist = nullptr) {
    using R = typename Raw<FMT>::t;
    constexpr int TPT = fv::tpt(N), STEP = TPT * 2;
    struct alignas(sizeof(R) * 2) RV { R v[2]; };
    const bool vec_ok = ((((uintptr_t)in / sizeof(R)) + (uint64_t)p0) & 1) == 0;
    // raw pair at block offset ib (zeros outside [0, n_in)): the only memory access of a trip
    auto load_raw = [&](int ib) {
        RV x{};
        const int64_t p = p0 + ib + lane * 2;
        if (live && ib < N) {
            const bool in0 = p >= 0 && (size_t)p < n_in, in1 = p + 1 >= 0 && (size_t)(p + 1) < n_in;
            if (vec_ok && in0 && in1) {
                x = *reinterpret_cast<const RV *>((const R *)in + p);
            } else {
                if (in0) x.v[0] = ((const R *)in)[p];
                if (in1) x.v[1] = ((const R *)in)[p + 1];
            }
        }
        return x;
    };
    // Software prefetch two trips ahead (named registers, no indexed array): the
    // ~1-2 us HBM/L2 latency of a trip's load hides behind the ~600-cycle Sincos work
    // of the two trips before it instead of being paid eight times per block.
    RV x0 = load_raw(0), x1 = load_raw(STEP);
#pragma unroll 1
    for (int ib = 0; ib < N; ib += STEP) {
        const RV x = x0;
        x0 = x1;
        x1 = load_raw(ib + 2 * STEP);
        const int i0 = ib + lane * 2;
        const int64_t p = p0 + i0;
        const int64_t span_lo = p0 + ib;  // uniform: the STEP samples this trip covers
        const NcoWin w = nco_window(P.segs, span_lo < 0 ? 0 : (uint64_t)span_lo,
                                    span_lo + STEP <= 0 ? 0 : (uint64_t)(span_lo + STEP - 1));
        float2 a = make_float2(0.f, 0.f), b = a;
        if (live) {
            // ONE instantiation of the elementwise program for the pair (p, p+1):
            // lanes whose position lies outside [0, n_in) run it on a dummy value and
            // are overridden afterwards (history or zero), instead of a second and
            // third inlined copy of the float64 Sincos on a divergent path.
            const bool in0 = p >= 0 && (size_t)p < n_in, in1 = p + 1 >= 0 && (size_t)(p + 1) < n_in;
            float2 ab[2] = {Raw<FMT>::cvt(x.v[0]), Raw<FMT>::cvt(x.v[1])};
            ew_apply_n<2, FILTERED>(P, ab, (uint64_t)p, w);  // (uint64)(-1) + 1 wraps to sample 0
            if (in0) a = ab[0]; else if (p < 0 && hist) a = hist[p + off];
            if (in1) b = ab[1]; else if (p + 1 < 0 && hist) b = hist[p + 1 + off];
        }
        keep_history(new_hist, p, n_in, off, live, a, b);
        const int q = fv::pad(i0);  // i0 even: i0 and i0+1 share a 16-element row
        lds[q] = a;
        lds[q + 1] = b;
    }
}

// waves per SIMD the register allocator is asked to leave room for
constexpr int conv_occupancy(int n, bool staged = true) { return fv::block(n) >= 512 ? 2 : 4; }

template <int N, int FMT, bool STAGED>
__global__ __launch_bounds__(fv::block(N), conv_occupancy(N, STAGED)) void conv_blocks_kernel16(
    const void *in, float2 *out, const float2 *__restrict__ filt, FvTabs tabs, size_t nblocks, unsigned dec,
    size_t per, EwProgram P) {
    constexpr int TPT = fv::tpt(N), XPB = fv::xpb(N), R0 = fv::first_radix(N);
    // (one block per workgroup: the block index is uniform, and the addresses are a scalar base plus a lane offset)
    const int sub = XPB == 1 ? 0 : threadIdx.x / TPT, lane = XPB == 1 ? (int)threadIdx.x : threadIdx.x % TPT;
    cf *lds = fv_lds() + sub * fv::lds_elems(N);
    cf v[16];
    auto tail = [&](size_t b, bool live, int lane) {
        fv::backward<N>(v, lds, tabs.bwd, lane);
        if (live) {
            if (dec <= 1) {  // (uniform)
                float2 *ob = out + b * N;
#pragma unroll
                for (int q = 0; q < 16; q++) ob[fv::edge_index<N, R0>(q, lane)] = fv::to2(v[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 16; q++) conv_store(out, b * N + fv::edge_index<N, R0>(q, lane), fv::to2(v[q]), dec, per);
            }
        }
    };
    auto finish = [&](size_t b, bool live, int lane) {
        {
            // freq1[i] = freq1[i] * freq[i] (fft/convolution.go:187-189).  Go forms this product in
            // float64 and narrows once; between two float32 transforms that are themselves only
            // error-bounded the extra half ulp buys nothing, and the float64 form cost 13 % of the
            // kernel's vector instructions: a float32 product here (the closures of fft.Convolve,
            // whose results ARE the reference's own arithmetic apart from the transforms, keep it)
            // (the direct form in two halves: with the next block waiting in registers, sixteen bins in flight at
            // once cost the fourth wave per SIMD its registers)
            const cf *fl = (const cf *)filt + lane;
            constexpr int HALF = STAGED ? 16 : 8;
#pragma unroll
            for (int q0 = 0; q0 < 16; q0 += HALF) {
#pragma unroll
                for (int q = q0; q < q0 + HALF; q++) v[q] = fv::cmul(v[q], fl[fv::edge_off<N, 16>(q)]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        tail(b, live, lane);
    };
    if constexpr (STAGED) {
        const size_t b = (size_t)blockIdx.x * XPB + sub;
        const bool live = b < nblocks;
        stage_block16<N, FMT>((float2 *)lds, in, P, (int64_t)(b * N), ~(size_t)0, nullptr, 0, lane, live);
        __syncthreads();
        fv::load_lds<N, R0>(v, lds, lane);
        fv::forward<N, true>(v, lds, tabs.fwd, lane);
        finish(b, live, lane);
    } else {
        // c64 blocks straight from memory: the workgroup walks the blocks blockIdx.x, + gridDim.x, ... and has the
        // NEXT block's sixteen loads per lane in flight while it transforms the current one -- load, transform and
        // store phases of one block per workgroup left the memory pipes idle whenever the resident waves happened to
        // compute (4.3 TB/s -> 4.7).  (Loads return in order: the filter's bins, read behind the forward transform, wait
        // for the prefetch too -- it has the forward transform to arrive in.  Keeping the bins in registers across the
        // blocks instead costs a wave per SIMD and was slower: 73 us against 57.  Non-temporal loads and stores, for a
        // call whose bytes exceed the memory-side cache: 67.8 us instead of 69.6 over a rotation of buffers, 67.6 instead
        // of 56.5 over one pair -- not worth having.  The bins asked for in front of the prefetch, behind the forward
        // transform, so that the product waits for L2 and not for HBM: 152 registers, three waves per SIMD, 57.3 / 70.1 us
        // -- the same; with four waves it spills, 108 us.)
        using RT = typename Raw<FMT>::t;
        const size_t stride = (size_t)gridDim.x * XPB;
        size_t b = (size_t)blockIdx.x * XPB + sub;
        RT nx[16];
        {
            const RT *pb = (const RT *)in + b * N;
#pragma unroll
            for (int q = 0; q < 16; q++) nx[q] = b < nblocks ? pb[fv::edge_index<N, R0>(q, lane)] : RT{};
        }
#pragma unroll 1
        for (size_t b0 = (size_t)blockIdx.x * XPB; b0 < nblocks; b0 += stride, b += stride) {  // (uniform trip count)
            const bool live = b < nblocks;
            // (the lane index as a value the compiler cannot prove loop-invariant: hoisted out of the loop, the
            // trip's ~50 registers of addresses -- twiddle tables, filter, stores -- were spilled and reloaded)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            // The previous trip's backward transform ends in reads of the LDS (no barrier behind them) and this
            // trip's forward transform starts with writes: where several waves share one transform's LDS a fast
            // wave would overwrite what a slow one still reads.  (One wave per transform -- N <= 1024 -- orders
            // its own LDS operations.)
            if constexpr (fv::block(N) > 64) __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = fv::from2(Raw<FMT>::cvt(nx[q]));
            const size_t bn = b + stride;
            if (bn < nblocks) {
                const RT *pb = (const RT *)in + bn * N;
#pragma unroll
                for (int q = 0; q < 16; q++) nx[q] = pb[fv::edge_index<N, R0>(q, ln)];
            }
            fv::forward<N>(v, lds, tabs.fwd, ln);
            finish(b, live, ln);
        }
    }
}

// Blocks of N <= 1024 points (a transform inside ONE wave) with nothing elementwise in front but the conversion: W waves of a workgroup
// walk the blocks side by side and share ONE copy of the twiddle tables and of the filter's bins IN LDS.
// Why: a wave's vector-memory loads return in order.  conv_blocks_kernel16 keeps the next block's sixteen loads in flight
// under the current block's transforms, but its table and filter reads (L2 hits) queue up BEHIND those loads, so the
// first twiddle a block needs waits for the prefetch to land from HBM -- a hundred instructions into the block the
// prefetch hid nothing (ISA: `s_waitcnt vmcnt(5)` behind six table loads behind the sixteen prefetches).  LDS reads
// count on lgkmcnt: here the vector-memory queue of a wave holds its prefetch and its stores, nothing else.
template <int N> struct ConvShared {
    static constexpr int TPT = fv::tpt(N), XPW = 64 / TPT;  // transforms per wave
    static constexpr int FL = fv::fwd_tab_len(N), BL = fv::bwd_tab_len(N);
    static constexpr size_t table_bytes = (size_t)(FL + BL) * sizeof(cf4) + (size_t)N * sizeof(cf);
    static constexpr size_t wave_bytes = (size_t)XPW * fv::lds_elems(N) * sizeof(cf);
    static constexpr size_t lds_bytes(int waves) { return table_bytes + wave_bytes * waves; }
    static_assert(TPT <= 64 && TPT >= 16, "one wave per transform");
};

template <int N, int W, int FMT = HZSDR_FMT_C64>
__global__ __launch_bounds__(64 * W) void conv_blocks_shared_kernel(const typename Raw<FMT>::t *__restrict__ in, float2 *__restrict__ out,
                                                                    const float2 *__restrict__ filt, FvTabs tabs, size_t nblocks) {
    using G = ConvShared<N>;
    constexpr int TPT = G::TPT, XPW = G::XPW, R0 = fv::first_radix(N);
    cf4 *lf = reinterpret_cast<cf4 *>(fv_lds());
    cf4 *lb = lf + G::FL;
    cf *lh = reinterpret_cast<cf *>(lb + G::BL);
    const int wave = threadIdx.x / 64, wl = threadIdx.x % 64;
    const int sub = XPW == 1 ? 0 : wl / TPT, lane = XPW == 1 ? wl : wl % TPT;
    cf *lds = lh + N + (size_t)(wave * XPW + sub) * fv::lds_elems(N);
    for (int i = threadIdx.x; i < G::FL; i += 64 * W) lf[i] = tabs.fwd[i];
    for (int i = threadIdx.x; i < G::BL; i += 64 * W) lb[i] = tabs.bwd[i];
    for (int i = threadIdx.x; i < N; i += 64 * W) lh[i] = fv::from2(filt[i]);
    __syncthreads();  // the only barrier: from here on the waves never meet
    // consecutive blocks side by side: the workgroup reads W * XPW blocks = W * 8 KiB in one run per trip
    const size_t stride = (size_t)gridDim.x * W * XPW;
    size_t b = ((size_t)blockIdx.x * W + wave) * XPW + sub;
    using RT = typename Raw<FMT>::t;  // (a byte or i16 source converts on its way into the first pass's registers)
    cf v[16];
    RT nx[16];
    {
        const RT *pb = in + b * N + lane;
#pragma unroll
        for (int q = 0; q < 16; q++) nx[q] = b < nblocks ? pb[fv::edge_off<N, R0>(q)] : RT{};
    }
#pragma unroll 1
    for (size_t b0 = ((size_t)blockIdx.x * W + wave) * XPW; b0 < nblocks; b0 += stride, b += stride) {  // (uniform in a wave)
        const bool live = XPW == 1 || b < nblocks;
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = fv::from2(Raw<FMT>::cvt(nx[q]));
        const size_t bn = b + stride;
        if (XPW == 1 ? b0 + stride < nblocks : bn < nblocks) {
            const RT *pb = in + bn * N + lane;
#pragma unroll
            for (int q = 0; q < 16; q++) nx[q] = pb[fv::edge_off<N, R0>(q)];
        }
        fv::forward<N, false, true>(v, lds, lf, lane);
        // freq1[i] *= freq[i] (fft/convolution.go:187-189) as a float32 product, like conv_blocks_kernel16
        const cf *fl = lh + lane;
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = fv::cmul(v[q], fl[fv::edge_off<N, 16>(q)]);
        fv::backward<N, true>(v, lds, lb, lane);
        if (live) {
            float2 *ob = out + b * N + lane;
#pragma unroll
            for (int q = 0; q < 16; q++) ob[fv::edge_off<N, R0>(q)] = fv::to2(v[q]);
        }
    }
}

// FOLD = D (power of two <= 16 with N/D >= 256): lane-local spectral fold to M = N/D
// bins; the M-point inverse runs in fir_synth_kernel16.  (One kernel did both at first:
// the inverse kept one group of M/16 lanes busy and parked the rest of the workgroup,
// and cost as much as the whole forward transform -- 33 of 98 us.)  FOLD = 0: full
// backward transform + pick, in this kernel.
//
// LATE: the mixer commutes with the filter.  Inside one exactly-linear run of the NCO
// clock ts[n-k] = ts[n] - k*step, so with every elementwise stage a multiplication by a
// complex scalar (Gain, Multiply) or by exp(i*tau_s*ts[n]) (Shift),
//     sum_k h[k] * ew(x, n-k)  =  ew( sum_k (h[k] * exp(-i*Omega*k*step)) * x[n-k], n ),
// Omega = sum of the Shift stages' tau_s: filter the CONVERTED samples with the run's
// modulated taps (late.h[run], prepared by the host per distinct step) and run the
// unchanged elementwise program on the hop/D outputs instead of the N inputs -- the
// float64 Sincos at the decimated rate.  A block takes this path only when its whole
// input span lies in one run, inside [0, n_in - off) (no history read or written);
// every other block -- the first, the last, any block across a run boundary or a
// 2*pi wrap of the clock -- mixes in reference order before the filter, as before.
struct LateFilters {
    const float2 *h[kNcoMaxSegs];
};
// the blocks of a run that do NOT take the late path, ascending (host: slow_blocks)
constexpr int kMaxSlowBlocks = 96;
struct SlowBlocks {
    int n;
    unsigned idx[kMaxSlowBlocks];
};

// POLYPHASE form of the folded analysis (N = 4096, D in {2, 4, 8, 16}).  The fold sums the
// D aliases of every output bin, sum_q H[k + M q] X[k + M q] (M = N/D).  Writing the
// N-point transform by its first decimation-in-time stage, X[k] = sum_r W_N^(r k) U_r[k mod M]
// with U_r the M-point transform of branch u_r[m] = x[D m + r], the aliases collapse:
//     Z[k] = sum_r G_r[k] U_r[k],   G_r = FFT_M(g_r) / M,   g_r[j] = h[D j - r]
// -- D transforms of M points and one multiply-accumulate, no N-point last pass.  Lane p of
// the workgroup is lane p / D of branch p % D, which makes the first pass (radix 16 over
// x[p + 256 q]) the very loads and butterflies of the N-point form; the second radix-16 pass
// stays inside a branch; the last pass of the M-point transforms is radix 16/D at Ns = 256,
// so lane p can run it for ALL D branches at bins p + 256 i (D * 16/D = 16 values), multiply
// by the D filter spectra and add up.  `hfreq` and `late.h[]` hold G[r][k] (D x M) for these
// chains.  (N = 2048 and 1024 -- 128 and 64 lanes -- work the same way with a second pass of
// radix N/256 = 8 or 4 and the fused last pass at Ns = N/16.)
// Tables (get_fv_poly_tables): pass 2 = rows k < 16 of R2 entries w^(r k), w = exp(-2 pi i /
// (16 R2)); last pass = rows `lane` < N/16 of RL entries exp(-2 pi i * i * lane / (RL N/16)).
constexpr bool fold_poly(int n, int fold) {
    return (n == 4096 || n == 2048 || n == 1024) && (fold == 2 || fold == 4 || fold == 8 || fold == 16) &&
           n / fold >= 256;
}

// the FFT sizes whose blocks can mix late (one overlap-save block per workgroup)
constexpr bool late_capable(unsigned nfft) { return nfft >= 1024 && nfft <= 8192; }

// workgroup-uniform: does block b (input span [p0, p0 + N)) take the late-mixer path?
// `group`: lanes per block when they are a whole 32- or 64-lane group of the wave (the run
// lookup is then one vector load + ballot, nco_window_ballot), 0 = scalar scan.
__device__ __forceinline__ bool late_block(const EwProgram &P, const LateFilters &late, int64_t p0, int N,
                                           unsigned off, size_t n_in, NcoWin *run, int group = 0) {
    *run = NcoWin{0, 0};
    const bool inside = !(p0 < 0 || (uint64_t)p0 + (uint64_t)N + off > n_in);
    // (a table of a few runs is scanned faster than the vector load's round trip)
    if (group != 0 && P.segs.big_n == 0 && P.segs.n > 4) {
        // every lane takes part in the ballot, also those of blocks outside the buffer
        const NcoWin w = nco_window_ballot(P.segs, inside ? (uint64_t)p0 : 0, inside ? (uint64_t)p0 + N - 1 : 0, group);
        if (!inside) return false;
        *run = w;
    } else {
        if (!inside) return false;
        *run = nco_window(P.segs, (uint64_t)p0, (uint64_t)p0 + N - 1);
    }
    return run->lo == run->hi && late.h[run->lo] != nullptr;
}

// dynamic LDS of fir_decimate_kernel16: the padded block (+ 32 elements of branch skew) and,
// for the polyphase form, pass 2's twiddle rows
constexpr size_t fir_lds_bytes(int n, int fold) {
    return ((size_t)fv::xpb(n) * fv::lds_elems(n) + 32) * 8 + (fold_poly(n, fold) ? 16 * (n / 256 + 1) * 16 : 0);
}

constexpr int fir_occupancy(int n, int fold, bool late) {
    return !late ? 1 : fv::block(n) >= 512 ? 2 : fold != 0 ? 4 : 3;
}

// EXP: ablation switches for tools/fir_ablate.hip (0 in the library): 1 = no input loads,
// 2 = no filter loads, 4 = no output store, 8 = no pass-3 LDS reads, 16 = no pass-1/2 LDS
// traffic.  Results are wrong by construction with any of them set.  (Round 4, config 3's form -- c64, full backward
// transform, D = 1 -- in that tool: 66 us of which input loads 7, filter loads 5, stores 9, the transforms alone 50;
// tried and dropped, no gain either: the filter's bins asked for in front of the forward transform, and persistent
// workgroups that keep the bins in registers from block to block -- 150 registers, one wave per SIMD less; and
// workgroups that WALK consecutive blocks with the next block's samples in flight under the transforms, the N - hop
// samples two blocks share passed on in registers (12 loads per lane and block instead of 16, read amplification 1.0):
// 73.3 us over one buffer pair and 88.7 over a rotation against 71.6-75.5 and 89-97 -- the input is not what it waits for.)
template <int N, int FMT, int FOLD, bool LATE, int EXP = 0>
__global__ __launch_bounds__(fv::block(N), fir_occupancy(N, FOLD, LATE)) void fir_decimate_kernel16(
    const void *in, float2 *out, const float2 *__restrict__ hist, float2 *__restrict__ new_hist,
    const float2 *__restrict__ hfreq, FvTabs tabs, float2 *__restrict__ spec, size_t nblocks, size_t n_in,
    unsigned hop, unsigned off, unsigned D, EwProgram P, LateFilters late, PolyTabs ptabs, SlowBlocks slow) {
    constexpr int R0 = fv::first_radix(N), TPT = fv::tpt(N);
    static_assert(fv::xpb(N) == 1 || FOLD == 0, "fold path assumes one block per workgroup");
    static_assert(!LATE || fv::xpb(N) == 1, "the late mixer assumes one block per workgroup");
    // (the polyphase form shifts each branch's region by 32/D elements: + 32 elements, see below)
    const int sub = threadIdx.x / TPT;
    int lane = threadIdx.x % TPT;
    if constexpr (fold_poly(N, FOLD) && FOLD == 8) {
        // Polyphase form, D = 8: position p = 8 bl + br (branch br, lane bl inside the branch).
        // With p = hardware lane, a 16-lane LDS write group holds 8 branches x 2 bl and the
        // branch regions (8 dwords apart mod 64 banks for the 32-lane READ groups) collide
        // pairwise on the 32 write banks: measured, 43 % of the LDS cycles were conflicts.
        // Permute inside the wave so that a write group holds 4 branches x 4 consecutive bl
        // (conflict-free) while a read half still holds 8 branches x 4 bl and the wave still
        // covers 64 consecutive samples: p bits [5 | 4:3 | 2 | 1:0] = lane bits [5 | 3:2 | 4 | 1:0].
        const int l = lane & 63;
        lane = (lane & ~63) | (l & 0x23) | ((l & 0xc) << 1) | ((l & 0x10) >> 2);
    }
    cf *lds = fv_lds() + sub * fv::lds_elems(N);
    size_t b = (size_t)blockIdx.x * fv::xpb(N) + sub;
    if constexpr (LATE) {
        // blocks that mix in reference order (the stream's edges, any block across a boundary
        // of the clock's runs or inside a short run) cost several times a late block: in
        // stream order they would be the kernel's tail whenever a boundary falls late in the
        // buffer, so the host lists them (sorted) and they are dispatched first
        static_assert(fv::xpb(N) == 1, "block reordering assumes one block per workgroup");
        // The other blocks go to the XCDs in contiguous runs: workgroups are dealt round-robin
        // over the eight XCDs (a placement used for speed only), so workgroup i takes the
        // (i / 8)-th block of run i % 8 and stream neighbours share an L2 -- the N - hop
        // samples two consecutive blocks have in common are then fetched from HBM once, not
        // twice (measured at the fabric: 45.5 MB read for 33.5 MB of input before).  The grid
        // is padded to a whole number of rounds; the surplus workgroups leave at once.
        if (b >= (size_t)slow.n) {
            const size_t rest = nblocks - slow.n, chunk = (rest + 7) / 8, i = b - slow.n;
            const size_t c = (i % 8) * chunk + i / 8;
            if (i / 8 >= chunk || c >= rest) return;  // whole workgroup
            b = c + slow.n;
        }
        if (b < (size_t)slow.n) {
            b = slow.idx[b];
        } else if (slow.n > 0 && slow.n <= 4) {
            size_t c = b - slow.n;  // the c-th block that is not in the list: a short scalar scan
#pragma unroll 1
            for (int i = 0; i < slow.n; i++)
                if ((size_t)slow.idx[i] <= c) c++;
            b = c;
        } else if (slow.n > 0) {
            // the c-th block that is not in the (ascending) list L is c + #{i : L[i] - i <= c}:
            // lanes compare two entries each (one vector load), a ballot counts -- instead of a
            // scalar scan with a dependent load per entry in front of every block
            const size_t c = b - slow.n;
            const int l = (int)(threadIdx.x & 63u);
            static_assert(kMaxSlowBlocks <= 128, "two entries per lane");
            const bool v0 = l < slow.n, v1 = l + 64 < slow.n;
            const unsigned e0 = slow.idx[v0 ? l : 0], e1 = slow.idx[v1 ? l + 64 : 0];
            const int k = __popcll(__ballot(v0 && (size_t)(e0 - (unsigned)l) <= c)) +
                          __popcll(__ballot(v1 && (size_t)(e1 - (unsigned)(l + 64)) <= c));
            b = c + (size_t)k;
        }
        // b is the same for the whole workgroup (one block per workgroup), but after the list
        // lookup the compiler no longer knows: say so, or everything derived from it -- the
        // block's clock-run window, the per-trip windows of the reference-order staging --
        // becomes per-lane loops of VECTOR loads from the kernel arguments, each behind
        // s_waitcnt vmcnt(0) (measured: a reference-order block of a launch whose clock table
        // has 25 runs took 45 us instead of 14)
        b = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((int)b);
    }
    const bool live = b < nblocks;
    const int64_t p0 = (int64_t)(b * hop) - (int64_t)off;
    bool mix_late = false;  // workgroup-uniform
    NcoWin run{0, 0};
    if constexpr (LATE) {
        mix_late = late_block(P, late, p0, N, off, n_in, &run, 64);
        run = NcoWin{__builtin_amdgcn_readfirstlane(run.lo), __builtin_amdgcn_readfirstlane(run.hi)};
    }
    const float2 *__restrict__ hf = mix_late ? late.h[run.lo] : hfreq;
    cf v[16];
    // polyphase form: this lane's entry of pass 2's twiddle rows, fetched before anything else
    // (it goes to LDS after pass 1; asked for there, its L2 round trip stood in front of the
    // first barrier of every block)
    cf4 p2_entry = cf4{0.f, 0.f, 0.f, 0.f};
    if constexpr (fold_poly(N, FOLD)) {
        if (lane < 16 * (N / 256)) p2_entry = ptabs.p2[lane];
    }
    const bool direct = LATE && mix_late;  // workgroup-uniform
    // register image of the first pass: radix-16 edge for the polyphase form, the N-point
    // plan's first radix otherwise
    constexpr int RIN = fold_poly(N, FOLD) ? 16 : R0;
    if (direct) {
        // a late block lies wholly inside the buffer and needs no arithmetic per input
        // sample: its samples go from global memory straight into the first pass's
        // register image (a wave reads 64 consecutive samples per load), no LDS staging
        using R = typename Raw<FMT>::t;
        const R *src = (const R *)in + p0 + lane;
        R raw[16];
#pragma unroll
        for (int q = 0; q < 16; q++) raw[q] = (EXP & 1) ? (R)(lane * 3 + q) : src[fv::edge_off<N, RIN>(q)];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = fv::from2(Raw<FMT>::cvt_late(raw[q]));
    } else {
        stage_block16<N, FMT, true>((float2 *)lds, in, P, p0, n_in, hist, off, lane, live, new_hist);
        __syncthreads();
        fv::load_lds<N, RIN>(v, lds, lane);
    }
    if constexpr (fold_poly(N, FOLD)) {
        // branch regions 32/D elements apart in bank space: neighbouring lanes belong to
        // different branches, and the regions' natural size is a multiple of all 64 banks
        constexpr int M = N / FOLD, RL = 16 / FOLD, R2 = N / 256, LE = fv::lds_elems(M) + 32 / FOLD;
        static_assert(FOLD * LE == fv::lds_elems(N) + 32 && FOLD * (M / 16) == TPT && R2 * 256 == N,
                      "polyphase geometry");
        const int br = lane % FOLD, bl = lane / FOLD;  // branch, lane inside the branch
        cf *ldb = lds + br * LE;
        // pass 2's twiddle rows (16 rows of R2 entries, a few KB) live in LDS behind the data,
        // rows one entry apart in bank space: 16-byte reads at immediate offsets
        cf4 *ltab = reinterpret_cast<cf4 *>(lds + fv::lds_elems(N) + 32);
        if (lane < 16 * R2) ltab[(lane / R2) * (R2 + 1) + lane % R2] = p2_entry;
        // pass 1 of every branch: radix 16 at Ns = 1 over u_r[bl + (M/16) q] = x[lane + TPT q]
        fv::butterflies<16, false>(v);
        if (!direct) __syncthreads();  // the staged block has been read by everybody
        if constexpr (!(EXP & 16)) fv::store_lds<M, 16, 1>(v, ldb, bl);
        {  // pass 2: radix N/256 at Ns = 16, inside the branch
            __syncthreads();
            if constexpr (!(EXP & 16)) fv::load_lds<M, R2>(v, ldb, bl);
            fv::twiddle_rows<M, R2, 16, R2 + 1>(v, (const cf4 *)ltab, bl);
            fv::butterflies<R2, false>(v);
            __syncthreads();
            if constexpr (!(EXP & 16)) fv::store_lds<M, R2, 16>(v, ldb, bl);
        }
        // last pass (radix RL at Ns = N/16) of ALL branches at bins lane + TPT i, times the
        // branch's filter spectrum, summed over the branches
        cf4 wl[RL > 1 ? RL - 1 : 1];
#pragma unroll
        for (int i = 1; i < RL; i++) wl[i - 1] = ptabs.p3[lane * RL + i];
        __syncthreads();
        cf zp[RL], zq[RL];
#pragma unroll
        for (int i = 0; i < RL; i++) zp[i] = zq[i] = cf{0.f, 0.f};
        const cf *pb = lds + fv::pad(lane);
        const cf *hl = (const cf *)hf + lane;
#pragma unroll
        for (int r2 = 0; r2 < FOLD; r2++) {
            cf u[RL];
#pragma unroll
            for (int i = 0; i < RL; i++) u[i] = (EXP & 8) ? v[(r2 * RL + i) & 15] : pb[r2 * LE + fv::pad(TPT * i)];
#pragma unroll
            for (int i = 1; i < RL; i++) u[i] = fv::cmul4(u[i], wl[i - 1]);
            if constexpr (RL > 1) fv::dft<RL, false>(u);
#pragma unroll
            for (int i = 0; i < RL; i++) fv::cacc(zp[i], zq[i], u[i], (EXP & 2) ? cf{0.5f, 0.25f} : hl[r2 * M + TPT * i]);
        }
        if (live) {
            cf *sp = (cf *)spec + b * M + lane;
#pragma unroll
            for (int i = 0; i < RL; i++) sp[TPT * i] = fv::cacc_finish(zp[i], zq[i]);
        }
        return;
    }
    if (direct) fv::forward<N>(v, lds, tabs.fwd, lane);
    else fv::forward<N, true>(v, lds, tabs.fwd, lane);
    {
        const cf *hl = (const cf *)hf + lane;
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = fv::cmul(v[q], (EXP & 2) ? cf{0.5f, 0.25f} : hl[fv::edge_off<N, 16>(q)]);
    }
    if constexpr (FOLD == 0) {
        fv::backward<N>(v, lds, tabs.bwd, lane);
        if (!direct) {
            if (live) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const unsigned idx = fv::edge_index<N, R0>(q, lane);
                    if (idx >= off && idx < off + hop && ((idx - off) % D) == 0) {
                        const size_t p = b * hop + (idx - off);
                        if (p < n_in) out[p / D] = fv::to2(v[q]);
                    }
                }
            }
        } else if (LATE && P.n == 0 && D == 1) {
            // a late block of a chain WITHOUT elementwise stages at D = 1 (BASELINE config 3 as a FIR): nothing to
            // mix and nothing to drop -- the kept outputs go from the backward transform's registers straight to
            // memory (a wave stores 64 consecutive outputs per instruction), no fifth trip through the LDS
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const unsigned idx = fv::edge_index<N, R0>(q, lane);
                if (idx >= off && idx < off + hop && !((EXP & 4) && v[q].x != 1234.5f)) out[b * hop + (idx - off)] = fv::to2(v[q]);
            }
        } else if constexpr (LATE) {
            // late block: the filtered block back to LDS, then the elementwise program over
            // the hop/D kept outputs (output t of the block sits at stream position b*hop + t*D)
            __syncthreads();  // the backward transform's last-pass reads are done
            fv::store_edge_lds<N, R0>(v, lds, lane);
            __syncthreads();
            const unsigned per = hop / D;
            constexpr int W = 4;
            const float2 *ldf = (const float2 *)lds;
#pragma unroll 1
            for (unsigned t0 = lane; t0 < per; t0 += W * TPT) {
                float2 y[W];
#pragma unroll
                for (int l = 0; l < W; l++) {
                    const unsigned tt = t0 + l * TPT < per ? t0 + l * TPT : t0;
                    y[l] = ldf[fv::pad(off + tt * D)];
                }
                ew_apply_n<W, true>(P, y, (uint64_t)b * hop + (uint64_t)t0 * D, run, (uint64_t)TPT * D);
#pragma unroll
                for (int l = 0; l < W; l++)
                    if (t0 + l * TPT < per) out[b * per + t0 + l * TPT] = y[l];
            }
        }
    } else {
        constexpr int M = N / FOLD, S = 16 / FOLD;
        static_assert(fv::ok(M) && S >= 1, "fold geometry");
        // slot q holds bin lane + q*TPT; folded bin lane + s*TPT sums the slots with q % S == s
        cf z[S];
#pragma unroll
        for (int s2 = 0; s2 < S; s2++) z[s2] = v[s2];
#pragma unroll
        for (int q = S; q < 16; q++) z[q % S] = z[q % S] + v[q];
        if (live) {
            cf *sp = (cf *)spec + b * M + lane;
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) sp[TPT * s2] = z[s2];
        }
    }
}

// (Round 4 measured a workgroup that WALKS consecutive blocks of BASELINE config 3 as a FIR -- c64 in, D = 1, the
// quarter two neighbours share passed on in registers, the next block's samples in flight under the transforms:
// 73.3 us against 75-78, the input is not what the kernel waits for.  The kernel, fir_walk_kernel16, was never
// launched by the library and is gone from this header; DESIGN.md section 8 keeps the measurement.)

// The other half of the folded FIR-decimate: the M-point inverse of every block's folded
// spectrum -- M/16 lanes per block, 256 / (M/16) blocks per workgroup, every lane busy --
// then, for blocks on the late-mixer path, the elementwise program over the hop/D
// outputs (output m sits at stream position D*m).
template <int N, int FOLD> struct SynthGeom {
    static constexpr int M = N / FOLD, TPTM = fv::tpt(M);
    static constexpr int BS = TPTM > 64 ? TPTM : 64;  // one wave where a block's group fits in it:
    static constexpr int XPB = BS / TPTM;              // many small workgroups, all resident at once
};

template <int N, int FOLD, bool LATE>
__global__ __launch_bounds__((SynthGeom<N, FOLD>::BS)) void fir_synth_kernel16(
    const float2 *__restrict__ spec, float2 *out, const cf4 *__restrict__ tabb, size_t nblocks,
    size_t n_in, unsigned hop, unsigned off, EwProgram P, LateFilters late) {
    using G = SynthGeom<N, FOLD>;
    constexpr int M = G::M, TPTM = G::TPTM, RM = fv::first_radix(M), XPB = G::XPB;
    static_assert(fv::ok(M) && XPB >= 1, "synthesis geometry");
    const int sub = threadIdx.x / TPTM, l2 = threadIdx.x % TPTM;
    cf *lds = fv_lds() + sub * fv::lds_elems(M);
    const size_t b = (size_t)blockIdx.x * XPB + sub;
    const bool live = b < nblocks;
    cf w[16];
    fv::BwdTwiddles<M> tw;
    tw.load(tabb, l2);  // every twiddle of the inverse up front: this kernel is one latency chain per wave
    {
        const cf *sp = (const cf *)spec + b * M + l2;
#pragma unroll
        for (int q = 0; q < 16; q++) w[q] = live ? sp[fv::edge_off<M, 16>(q)] : cf{0.f, 0.f};
    }
    fv::backward_regs<M>(w, lds, tw, l2);
    const unsigned i_lo = off / FOLD, i_hi = (off + hop) / FOLD, per = hop / FOLD;
    bool mix_late = false;  // uniform per block (= per group of TPTM lanes)
    NcoWin run{0, 0};
    if constexpr (LATE) {
        constexpr int GROUP = (TPTM == 32 || TPTM == 64) ? TPTM : TPTM > 64 ? 64 : 0;
        const bool lb = late_block(P, late, (int64_t)(b * hop) - (int64_t)off, N, off, live ? n_in : 0, &run, GROUP);
        mix_late = live && lb;
    }
    if (live && !mix_late) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const unsigned i = fv::edge_index<M, RM>(q, l2);
            if (i >= i_lo && i < i_hi) {
                const size_t m_out = b * per + (i - i_lo);
                if (m_out * FOLD < n_in) out[m_out] = fv::to2(w[q]);
            }
        }
    }
    if constexpr (LATE) {
        if (mix_late) {
            // The mixer straight from the last pass's registers.  In ascending time order the m-th
            // slot is q(m) = (m % (16/RM)) * RM + m / (16/RM), time index i = m*TPTM + l2, stream
            // position p0 + i*FOLD: a lane's outputs are equally spaced (TPTM*FOLD samples), so
            // the Shift stages run as a phase recurrence (ew_apply_seq) from the first slot that
            // can hold a valid output.
            const int64_t p0 = (int64_t)(b * hop) - (int64_t)off;
            float2 y[16];
#pragma unroll
            for (int m = 0; m < 16; m++) y[m] = fv::to2(w[(m % (16 / RM)) * RM + m / (16 / RM)]);
            const int m0 = (int)(i_lo / TPTM);  // uniform: slots below hold only the block's overlap
            ew_apply_seq<16>(P, y, m0, (uint64_t)(p0 + (int64_t)l2 * FOLD), (uint64_t)TPTM * FOLD, run);
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const unsigned i = m * TPTM + l2;
                if (i >= i_lo && i < i_hi) out[b * per + (i - i_lo)] = y[m];
            }
        }
    }
}

}  // namespace hz

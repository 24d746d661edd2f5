// hz_chain.hip -- fused operator chains and the FFT-convolution kernels.
//
// A chain is nested stream.* Readers collapsed into one launch per buffer:
// the "source functor" converts a raw input sample to complex64 and applies the
// elementwise stages in order (Shift NCO, Gain, Multiply), and the terminal
// stage decides how samples are consumed:
//
//   none / decimate / downsample  -> a streaming map kernel
//   convolution (reference: block-circular, stream/convolution.go:36-82)
//       -> per block: source functor -> forward FFT -> bins *= filter (Go
//          complex64 multiply) -> backward FFT -> (optional DecimateReader pick)
//   fir_decimate (north star: overlap-save FIR + decimate)
//       -> per block of N_fft: history/source -> forward FFT -> bins *= H ->
//          backward FFT -> keep the valid outputs on the decimation grid
//
// Every variant is one kernel on the context's stream; nothing is written to
// HBM between stages.
#include <math.h>

#include "hz_common.h"
#include "hz_device.h"
#include "hz_fft.h"
#include "hz_fft16.h"
#include "hz_fft_api.h"
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
            double s, c;
            go_sincos(ph, s, c);
            v = go_cmul(v, make_float2((float)c, (float)s));  // :82
        }
    }
    return v;
}

// The same program over W consecutive samples j0 .. j0+W-1 at once: the op loop is
// outside, the sample loop inside and unrolled, so the W independent Sincos /
// multiply chains interleave (instruction-level parallelism within a lane; the
// one-sample form serialises them behind the rolled op loop).
template <int W>
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
            double s[W], c[W];
#pragma unroll
            for (int l = 0; l < W; l++) go_sincos(__dmul_rn(o.tau_shift, ts[l]), s[l], c[l]);
#pragma unroll
            for (int l = 0; l < W; l++) v[l] = go_cmul(v[l], make_float2((float)c[l], (float)s[l]));
        }
    }
}

// The two most common programs -- Shift, and Shift then Gain (BASELINE config 2) -- spelt
// out, without the op loop: same operations in the same order, but straight-line code the
// scheduler can overlap with the loads and stores around it.  SHAPE 0 = interpret.
enum EwShape { SHAPE_ANY = 0, SHAPE_SHIFT = 1, SHAPE_SHIFT_GAIN = 2 };
template <int W, int SHAPE>
__device__ __forceinline__ void ew_apply_shape(const EwProgram &P, float2 (&v)[W], uint64_t j0, NcoWin w) {
    if constexpr (SHAPE == SHAPE_ANY) {
        ew_apply_n<W>(P, v, j0, w);
    } else {
        double s[W], c[W];
#pragma unroll
        for (int l = 0; l < W; l++) go_sincos(__dmul_rn(P.op[0].tau_shift, nco_ts(P.segs, w, j0 + l)), s[l], c[l]);
#pragma unroll
        for (int l = 0; l < W; l++) {
            v[l] = go_cmul(v[l], make_float2((float)c[l], (float)s[l]));
            if constexpr (SHAPE == SHAPE_SHIFT_GAIN)
                v[l] = make_float2(__fmul_rn(v[l].x, P.op[1].a), __fmul_rn(v[l].y, P.op[1].a));
        }
    }
}

template <int FMT> struct Raw;
template <> struct Raw<HZSDR_FMT_C64> {
    using t = float2;
    static __device__ __forceinline__ float2 cvt(float2 r) { return r; }
};
template <> struct Raw<HZSDR_FMT_U8> {
    using t = uint16_t;
    static __device__ __forceinline__ float2 cvt(uint16_t r) { return make_float2(u8_to_f32(r & 0xFF), u8_to_f32(r >> 8)); }
};
template <> struct Raw<HZSDR_FMT_I8> {
    using t = uint16_t;
    static __device__ __forceinline__ float2 cvt(uint16_t r) { return make_float2(i8_to_f32((int8_t)(r & 0xFF)), i8_to_f32((int8_t)(r >> 8))); }
};
template <> struct Raw<HZSDR_FMT_I16> {
    using t = uint32_t;
    static __device__ __forceinline__ float2 cvt(uint32_t r) { return make_float2(i16_to_f32((int16_t)(r & 0xFFFF)), i16_to_f32((int16_t)(r >> 16))); }
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
    constexpr int U = W >= 4 ? 1 : 2;
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

// ==== the same two kernels on the radix-16 core (hz_fft16.h), N = 256 .. 4096 ================

// stage_block for the padded LDS image of the radix-16 core
template <int N, int FMT>
__device__ __forceinline__ void stage_block16(float2 *lds, const void *in, const EwProgram &P, int64_t p0,
                                              size_t n_in, const float2 *hist, unsigned off, int lane,
                                              bool live, float2 *new_hist = nullptr) {
    using R = typename Raw<FMT>::t;
    constexpr int TPT = f16::tpt(N), STEP = TPT * 2;
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
            ew_apply_n<2>(P, ab, (uint64_t)p, w);  // (uint64)(-1) + 1 wraps to sample 0
            if (in0) a = ab[0]; else if (p < 0 && hist) a = hist[p + off];
            if (in1) b = ab[1]; else if (p + 1 < 0 && hist) b = hist[p + 1 + off];
        }
        keep_history(new_hist, p, n_in, off, live, a, b);
        const int q = f16::pad(i0);  // i0 even: i0 and i0+1 share a 16-element row
        lds[q] = a;
        lds[q + 1] = b;
    }
}

template <int N, int FMT, bool STAGED>
__global__ __launch_bounds__(f16::block(N), 4) void conv_blocks_kernel16(const void *in, float2 *out,
                                                                      const float2 *__restrict__ filt,
                                                                      const float2 *__restrict__ tw,
                                                                      size_t nblocks, unsigned dec,
                                                                      size_t per, EwProgram P) {
    constexpr int TPT = f16::tpt(N), XPB = f16::xpb(N), R0 = f16::first_radix(N);
    __shared__ float2 lds_all[XPB * f16::lds_elems(N)];
    const int sub = threadIdx.x / TPT, lane = threadIdx.x % TPT;
    float2 *lds = lds_all + sub * f16::lds_elems(N);
    const size_t b = (size_t)blockIdx.x * XPB + sub;
    const bool live = b < nblocks;
    float2 v[16];
    if constexpr (STAGED) {
        stage_block16<N, FMT>(lds, in, P, (int64_t)(b * N), ~(size_t)0, nullptr, 0, lane, live);
        __syncthreads();
        f16::load_lds<N, R0>(v, lds, lane);
        f16::forward<N, true>(v, lds, tw, lane);
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++)
            v[q] = live ? Raw<FMT>::cvt(((const typename Raw<FMT>::t *)in)[b * N + f16::edge_index<N, R0>(q, lane)])
                        : make_float2(0.f, 0.f);
        f16::forward<N>(v, lds, tw, lane);
    }
#pragma unroll
    for (int q = 0; q < 16; q++)  // freq1[i] = freq1[i] * freq[i], fft/convolution.go:187-189
        v[q] = go_cmul(v[q], filt[f16::edge_index<N, 16>(q, lane)]);
    f16::backward<N>(v, lds, tw, lane);
    if (live) {
#pragma unroll
        for (int q = 0; q < 16; q++) conv_store(out, b * N + f16::edge_index<N, R0>(q, lane), v[q], dec, per);
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
// by the D filter spectra and add up: 24 complex multiplies and 30 adds where the N-point
// form spent a twiddled radix-16 butterfly, 16 multiplies and the fold (~ 10 % fewer vector
// instructions per block).  `hfreq` and `late.h[]` hold G[r][k] (D x M) for these chains.
// (N = 2048 and 1024 -- 128 and 64 lanes -- work the same way with a second pass of radix
// N/256 = 8 or 4 and the fused last pass at Ns = N/16.)
constexpr bool fold_poly(int n, int fold) {
    return (n == 4096 || n == 2048 || n == 1024) && (fold == 2 || fold == 4 || fold == 8 || fold == 16) &&
           n / fold >= 256;
}

// the FFT sizes whose blocks can mix late (one overlap-save block per workgroup)
constexpr bool late_capable(unsigned nfft) { return nfft >= 1024 && nfft <= 4096; }

// workgroup-uniform: does block b (input span [p0, p0 + N)) take the late-mixer path?
__device__ __forceinline__ bool late_block(const EwProgram &P, const LateFilters &late, int64_t p0, int N,
                                           unsigned off, size_t n_in, NcoWin *run) {
    *run = NcoWin{0, 0};
    if (p0 < 0 || (uint64_t)p0 + (uint64_t)N + off > n_in) return false;
    *run = nco_window(P.segs, (uint64_t)p0, (uint64_t)p0 + N - 1);
    return run->lo == run->hi && late.h[run->lo] != nullptr;
}

template <int N, int FMT, int FOLD, bool LATE>
__global__ __launch_bounds__(f16::block(N), LATE ? (FOLD != 0 ? 4 : 3) : 1) void fir_decimate_kernel16(
    const void *in, float2 *out, const float2 *__restrict__ hist, float2 *__restrict__ new_hist,
    const float2 *__restrict__ hfreq,
    const float2 *__restrict__ tw, float2 *__restrict__ spec, size_t nblocks, size_t n_in,
    unsigned hop, unsigned off, unsigned D, EwProgram P, LateFilters late,
    const float2 *__restrict__ tw_sub) {
    constexpr int R0 = f16::first_radix(N), TPT = f16::tpt(N);
    static_assert(f16::xpb(N) == 1 || FOLD == 0, "fold path assumes one block per workgroup");
    static_assert(!LATE || f16::xpb(N) == 1, "the late mixer assumes one block per workgroup");
    // (+32: the polyphase form shifts each branch's region by 32/D elements, see below)
    __shared__ float2 lds_all[f16::xpb(N) * f16::lds_elems(N) + 32];
    const int sub = threadIdx.x / TPT, lane = threadIdx.x % TPT;
    float2 *lds = lds_all + sub * f16::lds_elems(N);
    size_t b = (size_t)blockIdx.x * f16::xpb(N) + sub;
    if constexpr (LATE) {
        // the stream's last two blocks mix in reference order (several times the work of a
        // late block): dispatched last they would be the kernel's tail, so they go first
        static_assert(f16::xpb(N) == 1, "block rotation assumes one block per workgroup");
        if (nblocks > 2) b = b < 2 ? nblocks - 2 + b : b - 2;
    }
    const bool live = b < nblocks;
    const int64_t p0 = (int64_t)(b * hop) - (int64_t)off;
    bool mix_late = false;  // workgroup-uniform
    NcoWin run{0, 0};
    if constexpr (LATE) mix_late = late_block(P, late, p0, N, off, n_in, &run);
    const float2 *__restrict__ hf = mix_late ? late.h[run.lo] : hfreq;
    float2 v[16];
    const bool direct = LATE && mix_late;  // workgroup-uniform
    // register image of the first pass: radix-16 edge for the polyphase form, the N-point
    // plan's first radix otherwise
    constexpr int RIN = fold_poly(N, FOLD) ? 16 : R0;
    if (direct) {
        // a late block lies wholly inside the buffer and needs no arithmetic per input
        // sample: its samples go from global memory straight into the first pass's
        // register image (a wave reads 64 consecutive samples per load), no LDS staging
        using R = typename Raw<FMT>::t;
        const R *src = (const R *)in + p0;
        R raw[16];
#pragma unroll
        for (int q = 0; q < 16; q++) raw[q] = src[f16::edge_index<N, RIN>(q, lane)];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = Raw<FMT>::cvt(raw[q]);
    } else {
        stage_block16<N, FMT>(lds, in, P, p0, n_in, hist, off, lane, live, new_hist);
        __syncthreads();
        f16::load_lds<N, RIN>(v, lds, lane);
    }
    if constexpr (fold_poly(N, FOLD)) {
        // branch regions 32/D elements apart in bank space: neighbouring lanes belong to
        // different branches, and the regions' natural size is a multiple of all 64 banks
        constexpr int M = N / FOLD, RL = 16 / FOLD, R2 = N / 256, LE = f16::lds_elems(M) + 32 / FOLD;
        static_assert(FOLD * LE == f16::lds_elems(N) + 32 && FOLD * (M / 16) == TPT && R2 * 256 == N,
                      "polyphase geometry");
        const int br = lane % FOLD, bl = lane / FOLD;  // branch, lane inside the branch
        float2 *ldb = lds + br * LE;
        // pass 1 of every branch: radix 16 at Ns = 1 over u_r[bl + (M/16) q] = x[lane + TPT q]
        f16::butterflies<16, false>(v);
        if (!direct) __syncthreads();  // the staged block has been read by everybody
        f16::store_lds<M, 16>(v, ldb, bl, 1);
        {  // pass 2: radix N/256 at Ns = 16, inside the branch
            f16::TwRegs<R2> t;
            f16::twiddle_load<M, R2>(t, tw_sub, bl, 16);
            __syncthreads();
            f16::load_lds<M, R2>(v, ldb, bl);
            f16::twiddle_apply<M, R2, false>(v, t);
            f16::butterflies<R2, false>(v);
            __syncthreads();
            f16::store_lds<M, R2>(v, ldb, bl, 16);
        }
        // last pass (radix RL at Ns = N/16) of ALL branches at bins lane + TPT i, times the
        // branch's filter spectrum, summed over the branches
        float2 wl[RL > 1 ? RL - 1 : 1];
#pragma unroll
        for (int i = 1; i < RL; i++) wl[i - 1] = tw_sub[i * lane];
        __syncthreads();
        float2 z[RL];
#pragma unroll
        for (int i = 0; i < RL; i++) z[i] = make_float2(0.f, 0.f);
#pragma unroll
        for (int r2 = 0; r2 < FOLD; r2++) {
            float2 u[RL];
#pragma unroll
            for (int i = 0; i < RL; i++) u[i] = lds[r2 * LE + f16::pad(lane + TPT * i)];
#pragma unroll
            for (int i = 1; i < RL; i++) u[i] = f16::cmul(u[i], wl[i - 1]);
            if constexpr (RL > 1) f16::dft<RL, false>(u);
#pragma unroll
            for (int i = 0; i < RL; i++) z[i] = f16::cadd(z[i], f16::cmul(u[i], hf[r2 * M + lane + TPT * i]));
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < RL; i++) spec[b * M + lane + TPT * i] = z[i];
        }
        return;
    }
    if (direct) f16::forward<N>(v, lds, tw, lane);
    else f16::forward<N, true>(v, lds, tw, lane);
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = f16::cmul(v[q], hf[f16::edge_index<N, 16>(q, lane)]);
    if constexpr (FOLD == 0) {
        f16::backward<N>(v, lds, tw, lane);
        if (!direct) {
            if (live) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const unsigned idx = f16::edge_index<N, R0>(q, lane);
                    if (idx >= off && idx < off + hop && ((idx - off) % D) == 0) {
                        const size_t p = b * hop + (idx - off);
                        if (p < n_in) out[p / D] = v[q];
                    }
                }
            }
        } else if constexpr (LATE) {
            // late block: the filtered block back to LDS, then the elementwise program over
            // the hop/D kept outputs (output t of the block sits at stream position b*hop + t*D)
            __syncthreads();  // the backward transform's last-pass reads are done
#pragma unroll
            for (int q = 0; q < 16; q++) lds[f16::pad(f16::edge_index<N, R0>(q, lane))] = v[q];
            __syncthreads();
            const unsigned per = hop / D;
            constexpr int W = 4;
#pragma unroll 1
            for (unsigned t0 = lane; t0 < per; t0 += W * TPT) {
                float2 y[W];
#pragma unroll
                for (int l = 0; l < W; l++) {
                    const unsigned tt = t0 + l * TPT < per ? t0 + l * TPT : t0;
                    y[l] = lds[f16::pad(off + tt * D)];
                }
                ew_apply_n<W>(P, y, (uint64_t)b * hop + (uint64_t)t0 * D, run, (uint64_t)TPT * D);
#pragma unroll
                for (int l = 0; l < W; l++)
                    if (t0 + l * TPT < per) out[b * per + t0 + l * TPT] = y[l];
            }
        }
    } else {
        constexpr int M = N / FOLD, S = 16 / FOLD;
        static_assert(f16::ok(M) && S >= 1, "fold geometry");
        // slot q holds bin lane + q*TPT; folded bin lane + s*TPT sums the slots with q % S == s
        float2 z[S];
#pragma unroll
        for (int s2 = 0; s2 < S; s2++) z[s2] = v[s2];
#pragma unroll
        for (int q = S; q < 16; q++) z[q % S] = f16::cadd(z[q % S], v[q]);
        if (live) {
#pragma unroll
            for (int s2 = 0; s2 < S; s2++) spec[b * M + lane + TPT * s2] = z[s2];
        }
    }
}

// The other half of the folded FIR-decimate: the M-point inverse of every block's folded
// spectrum -- M/16 lanes per block, 256 / (M/16) blocks per workgroup, every lane busy --
// then, for blocks on the late-mixer path, the elementwise program over the hop/D
// outputs (output m sits at stream position D*m).
template <int N, int FOLD> struct SynthGeom {
    static constexpr int M = N / FOLD, TPTM = f16::tpt(M);
    static constexpr int BS = TPTM > 64 ? TPTM : 64;  // one wave where a block's group fits in it:
    static constexpr int XPB = BS / TPTM;              // many small workgroups, all resident at once
};

template <int N, int FOLD, bool LATE>
__global__ __launch_bounds__((SynthGeom<N, FOLD>::BS)) void fir_synth_kernel16(
    const float2 *__restrict__ spec, float2 *out, const float2 *__restrict__ tw_small, size_t nblocks,
    size_t n_in, unsigned hop, unsigned off, EwProgram P, LateFilters late) {
    using G = SynthGeom<N, FOLD>;
    constexpr int M = G::M, TPTM = G::TPTM, RM = f16::first_radix(M), XPB = G::XPB;
    static_assert(f16::ok(M) && XPB >= 1, "synthesis geometry");
    __shared__ float2 lds_all[XPB * f16::lds_elems(M)];
    const int sub = threadIdx.x / TPTM, l2 = threadIdx.x % TPTM;
    float2 *lds = lds_all + sub * f16::lds_elems(M);
    const size_t b = (size_t)blockIdx.x * XPB + sub;
    const bool live = b < nblocks;
    float2 w[16];
#pragma unroll
    for (int q = 0; q < 16; q++)
        w[q] = live ? spec[b * M + f16::edge_index<M, 16>(q, l2)] : make_float2(0.f, 0.f);
    f16::backward<M>(w, lds, tw_small, l2);
    const unsigned i_lo = off / FOLD, i_hi = (off + hop) / FOLD, per = hop / FOLD;
    bool mix_late = false;  // uniform per block (= per group of TPTM lanes)
    NcoWin run{0, 0};
    if constexpr (LATE) mix_late = live && late_block(P, late, (int64_t)(b * hop) - (int64_t)off, N, off, n_in, &run);
    if (live && !mix_late) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const unsigned i = f16::edge_index<M, RM>(q, l2);
            if (i >= i_lo && i < i_hi) {
                const size_t m_out = b * per + (i - i_lo);
                if (m_out * FOLD < n_in) out[m_out] = w[q];
            }
        }
    }
    if constexpr (LATE) {
        __syncthreads();  // every group's last-pass reads are done: its region can take the outputs
        if (mix_late) {
#pragma unroll
            for (int q = 0; q < 16; q++) lds[f16::edge_index<M, RM>(q, l2)] = w[q];
        }
        __syncthreads();
        if (mix_late) {
            // four outputs per lane per trip: four independent Sincos chains in flight
            constexpr int W = 4;
#pragma unroll 1
            for (unsigned t0 = l2; t0 < per; t0 += W * TPTM) {
                float2 y[W];
#pragma unroll
                for (int l = 0; l < W; l++) y[l] = lds[i_lo + (t0 + l * TPTM < per ? t0 + l * TPTM : t0)];
                ew_apply_n<W>(P, y, (uint64_t)b * hop + (uint64_t)t0 * FOLD, run, (uint64_t)TPTM * FOLD);
#pragma unroll
                for (int l = 0; l < W; l++)
                    if (t0 + l * TPTM < per) out[b * per + t0 + l * TPTM] = y[l];
            }
        }
    }
}

__global__ void scale_c64_kernel(float2 *buf, size_t n, float r) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        buf[i] = make_float2(buf[i].x * r, buf[i].y * r);
}

}  // namespace hz

// =============================================================================

enum ChainTerm { TERM_NONE = 0, TERM_DECIMATE, TERM_DOWNSAMPLE, TERM_CONV, TERM_FIR };

struct hzsdr_chain {
    hzsdr_ctx *ctx;
    int src_fmt;
    uint64_t sample_rate;
    // elementwise stages
    int n_ops = 0;
    hz::EwOp ops[hz::kMaxEw];
    bool has_shift = false;
    double ts = 0.0;  // the shared NCO clock
    // terminal
    int term = TERM_NONE;
    unsigned factor = 1;
    // convolution
    void *filt = nullptr;  // device, flen bins
    size_t flen = 0;
    // fir-decimate
    void *hfreq = nullptr;  // device, nfft bins (FFT(taps)/nfft)
    void *hist[2] = {nullptr, nullptr};
    int hist_cur = 0;
    size_t ntaps = 0;
    unsigned nfft = 0, hop = 0, off = 0;
    // late mixer (see fir_decimate_kernel16): the taps, and FFT(taps * exp(-i*Omega*k*step))/N
    // per distinct clock step seen so far (one per binade of the NCO clock)
    std::vector<float> taps_host;
    std::map<uint64_t, void *> late_cache;
    bool mix_in_order = false;
    bool poly = false;  // hfreq / late_cache hold the polyphase layout (fold_poly)
};

struct hzsdr_conv {
    hzsdr_ctx *ctx;
    int kind;  // 0 = ConvolveFreq, 1 = Convolve, 2 = CrossCorrelate
    void *dst;
    const void *src1, *src2;
    size_t n;
    void *filt;  // device copy of the frequency-domain filter (kind 0)
};

namespace hz {

template <int N, int FMT>
static void launch_conv_n(hzsdr_ctx *ctx, const void *in, void *out, const void *filt, const float2 *tw,
                          size_t nblocks, unsigned dec, size_t per, const EwProgram &P) {
    const bool direct = FMT == HZSDR_FMT_C64 && P.n == 0;
    if constexpr (f16::ok(N)) {  // radix-16 core
        constexpr int XPB = f16::xpb(N);
        const dim3 grid((unsigned)((nblocks + XPB - 1) / XPB)), block(f16::block(N));
        if (direct)
            hipLaunchKernelGGL((conv_blocks_kernel16<N, FMT, false>), grid, block, 0, ctx->stream, in,
                               (float2 *)out, (const float2 *)filt, tw, nblocks, dec, per, P);
        else
            hipLaunchKernelGGL((conv_blocks_kernel16<N, FMT, true>), grid, block, 0, ctx->stream, in,
                               (float2 *)out, (const float2 *)filt, tw, nblocks, dec, per, P);
    } else {  // radix-4 core: N < 256 and N = 8192
        constexpr int XPB = fft_xpb(N);
        const dim3 grid((unsigned)((nblocks + XPB - 1) / XPB)), block(fft_block(N));
        if (direct)
            hipLaunchKernelGGL((conv_blocks_kernel<N, FMT, false>), grid, block, 0, ctx->stream, in,
                               (float2 *)out, (const float2 *)filt, tw, nblocks, dec, per, P);
        else
            hipLaunchKernelGGL((conv_blocks_kernel<N, FMT, true>), grid, block, 0, ctx->stream, in,
                               (float2 *)out, (const float2 *)filt, tw, nblocks, dec, per, P);
    }
}

template <int FMT>
static int launch_conv_fmt(hzsdr_ctx *ctx, size_t n, const void *in, void *out, const void *filt,
                           const float2 *tw, size_t nblocks, unsigned dec, size_t per, const EwProgram &P) {
    switch (n) {
    case 4: launch_conv_n<4, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    case 8: launch_conv_n<8, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    case 16: launch_conv_n<16, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    case 32: launch_conv_n<32, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    case 64: launch_conv_n<64, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    case 128: launch_conv_n<128, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    case 256: launch_conv_n<256, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    case 512: launch_conv_n<512, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    case 1024: launch_conv_n<1024, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    case 2048: launch_conv_n<2048, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    case 4096: launch_conv_n<4096, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    case 8192: launch_conv_n<8192, FMT>(ctx, in, out, filt, tw, nblocks, dec, per, P); break;
    default: return HZSDR_ERR_INVALID_ARGUMENT;
    }
    return HZSDR_OK;
}

// Block-circular convolution of nblocks blocks of length n from a source of
// format fmt with elementwise program P (device pointers).
static int conv_blocks_device(hzsdr_ctx *ctx, int fmt, size_t n, const void *in, void *out,
                              const void *filt, size_t nblocks, unsigned dec, size_t per,
                              const EwProgram &P) {
    if (nblocks == 0) return HZSDR_OK;
    const float2 *tw;
    HZ_TRY(get_twiddles(ctx, n, &tw));
    switch (fmt) {
    case HZSDR_FMT_C64: return launch_conv_fmt<HZSDR_FMT_C64>(ctx, n, in, out, filt, tw, nblocks, dec, per, P);
    case HZSDR_FMT_U8: return launch_conv_fmt<HZSDR_FMT_U8>(ctx, n, in, out, filt, tw, nblocks, dec, per, P);
    case HZSDR_FMT_I8: return launch_conv_fmt<HZSDR_FMT_I8>(ctx, n, in, out, filt, tw, nblocks, dec, per, P);
    default: return launch_conv_fmt<HZSDR_FMT_I16>(ctx, n, in, out, filt, tw, nblocks, dec, per, P);
    }
}

// Generic (any power of two) single-block path: three steps through scratch.
static int conv_generic_device(hzsdr_ctx *ctx, void *dst, const void *src1, const void *src2_or_filt,
                               size_t n, int kind) {
    HZ_TRY(ensure_slot(ctx, 6, n * 8));
    void *f1 = ctx->slots[6].ptr;
    HZ_TRY(fft_device(ctx, src1, f1, n, 1, true));
    if (kind == 0) {
        pointwise_mul_device(ctx, f1, src2_or_filt, n, false);
    } else {
        HZ_TRY(ensure_slot(ctx, 7, n * 8));
        void *f2 = ctx->slots[7].ptr;
        HZ_TRY(fft_device(ctx, src2_or_filt, f2, n, 1, true));
        pointwise_mul_device(ctx, f1, f2, n, kind == 2);
    }
    return fft_device(ctx, f1, dst, n, 1, false);
}

template <int FMT>
static void launch_map(hzsdr_ctx *ctx, const void *in, void *out, size_t n, const EwProgram &P) {
    using R = typename Raw<FMT>::t;
    // four samples per lane (four interleaved Sincos chains) when both pointers allow the
    // wider vectors, else two; the ragged end, or everything for a sample-aligned Go
    // sub-slice, one at a time
    const auto ok = [&](int w) { return ((uintptr_t)in % (sizeof(R) * w) == 0) && ((uintptr_t)out % (8 * w) == 0); };
    size_t done = 0;
    const int shape = P.n == 1 && P.op[0].kind == EW_SHIFT ? SHAPE_SHIFT
                      : P.n == 2 && P.op[0].kind == EW_SHIFT && P.op[1].kind == EW_SCALE ? SHAPE_SHIFT_GAIN
                                                                                         : SHAPE_ANY;
    if (ok(4) && n >= 4) {
        const size_t nvec = n / 4;
        const dim3 grid(blocks_for(ctx, nvec)), block(kThreads);
        if (shape == SHAPE_SHIFT_GAIN)
            hipLaunchKernelGGL((chain_map_kernel<FMT, 4, SHAPE_SHIFT_GAIN>), grid, block, 0, ctx->stream, in,
                               (float2 *)out, nvec, (uint64_t)0, P);
        else if (shape == SHAPE_SHIFT)
            hipLaunchKernelGGL((chain_map_kernel<FMT, 4, SHAPE_SHIFT>), grid, block, 0, ctx->stream, in,
                               (float2 *)out, nvec, (uint64_t)0, P);
        else
            hipLaunchKernelGGL((chain_map_kernel<FMT, 4>), grid, block, 0, ctx->stream, in, (float2 *)out, nvec,
                               (uint64_t)0, P);
        done = nvec * 4;
    } else if (ok(2) && n >= 2) {
        const size_t nvec = n / 2;
        hipLaunchKernelGGL((chain_map_kernel<FMT, 2>), dim3(blocks_for(ctx, (nvec + 1) / 2)), dim3(kThreads), 0,
                           ctx->stream, in, (float2 *)out, nvec, (uint64_t)0, P);
        done = nvec * 2;
    }
    if (done < n)
        hipLaunchKernelGGL((chain_map_kernel<FMT, 1>), dim3(blocks_for(ctx, n - done)), dim3(kThreads), 0,
                           ctx->stream, (const R *)in + done, (float2 *)out + done, n - done,
                           (uint64_t)done, P);
}

// The taps' spectrum in the layout the analysis kernel multiplies by: H[k] = FFT_N(h)[k] / N,
// or, for the polyphase form, G[r][k] = FFT_M(g_r)[k] / M with g_r[j] = h[D j - r] (see
// fold_poly).  `taps`: ntaps complex64 in host memory; `dst`: N complex64 of device memory.
static int filter_spectrum(hzsdr_chain *c, const float *taps, void *dst) {
    hzsdr_ctx *ctx = c->ctx;
    const unsigned nfft = c->nfft;
    std::vector<float> padded(2 * (size_t)nfft, 0.0f);
    size_t len = nfft, batch = 1;
    if (c->poly) {
        const unsigned F = c->factor, M = nfft / F;
        for (unsigned r = 0; r < F; r++)
            for (unsigned j = 0; j < M; j++) {
                const long idx = (long)F * j - (long)r;
                if (idx < 0 || (size_t)idx >= c->ntaps) continue;
                padded[2 * ((size_t)r * M + j)] = taps[2 * idx];
                padded[2 * ((size_t)r * M + j) + 1] = taps[2 * idx + 1];
            }
        len = M;
        batch = F;
    } else {
        memcpy(padded.data(), taps, c->ntaps * 8);
    }
    HZ_TRY(ensure_slot(ctx, 8, (size_t)nfft * 8));
    HZ_HIP(ctx, hipMemcpyAsync(ctx->slots[8].ptr, padded.data(), (size_t)nfft * 8, hipMemcpyHostToDevice, ctx->stream));
    HZ_TRY(fft_device(ctx, ctx->slots[8].ptr, dst, len, batch, true));
    hipLaunchKernelGGL(scale_c64_kernel, dim3(blocks_for(ctx, nfft)), dim3(kThreads), 0, ctx->stream,
                       (float2 *)dst, (size_t)nfft, 1.0f / (float)len);
    HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `padded` and slot 8 are reused
    return HZSDR_OK;
}

// The spectrum of taps[k] * exp(-i * omega * k * step) for one clock step.  A miss uploads,
// transforms and WAITS (filter_spectrum), so callers on the streaming path only look up
// (`make` = false: *dev = nullptr on a miss); hzsdr_chain_fir_decimate / _set_time prepare
// every step a stream can meet ahead of time (prepare_late_filters).  Entries are never
// evicted while the chain lives: pointers handed to enqueued kernels stay valid.
constexpr size_t kLateCacheMax = 64;
static int late_filter_for(hzsdr_chain *c, double step, double omega, void **dev, bool make) {
    hzsdr_ctx *ctx = c->ctx;
    *dev = nullptr;
    uint64_t key;
    memcpy(&key, &step, 8);
    auto it = c->late_cache.find(key);
    if (it != c->late_cache.end()) {
        *dev = it->second;
        return HZSDR_OK;
    }
    if (!make || c->late_cache.size() >= kLateCacheMax) return HZSDR_OK;  // that run mixes in reference order
    std::vector<float> mod(2 * c->ntaps);
    for (size_t k = 0; k < c->ntaps; k++) {
        const double ph = -omega * ((double)k * step);
        const double cr = cos(ph), ci = sin(ph);
        const double hr = c->taps_host[2 * k], hi = c->taps_host[2 * k + 1];
        mod[2 * k] = (float)(hr * cr - hi * ci);
        mod[2 * k + 1] = (float)(hr * ci + hi * cr);
    }
    void *h = nullptr;
    HZ_HIP(ctx, hipMalloc(&h, (size_t)c->nfft * 8));
    int rc = filter_spectrum(c, mod.data(), h);
    if (rc != HZSDR_OK) {
        (void)hipFree(h);
        return rc;
    }
    c->late_cache[key] = h;
    *dev = h;
    return HZSDR_OK;
}

static double chain_omega(const hzsdr_chain *c) {
    double omega = 0.0;
    for (int i = 0; i < c->n_ops; i++)
        if (c->ops[i].kind == EW_SHIFT) omega += c->ops[i].tau_shift;
    return omega;
}

// The late mixer's modulated spectra for every long run the clock can produce from `ts0`
// on: the rest of the current 2*pi period plus one whole period from 0 (the clock's step is
// a function of the binade alone, so after the first wrap the runs repeat).  Done at chain
// construction and whenever the clock is set, so hzsdr_chain_run never allocates or waits.
static int prepare_late_filters(hzsdr_chain *c, double ts0) {
    if (!c->has_shift || !late_capable(c->nfft) || c->taps_host.empty()) return HZSDR_OK;
    const double omega = chain_omega(c);
    const uint64_t period = (uint64_t)(6.283185307179586 * (double)c->sample_rate) + 2;
    for (int pass = 0; pass < 2; pass++) {
        std::vector<hzsdr_nco_segment> segs(96);
        size_t need = 0;
        double ts_end = 0.0;
        if (hzsdr_nco_segments(c->sample_rate, pass == 0 ? ts0 : 0.0, period, segs.data(), segs.size(), &need,
                               &ts_end) != HZSDR_OK)
            continue;
        const size_t have = need < segs.size() ? need : segs.size();
        for (size_t q = 0; q < have; q++) {
            if (segs[q].count < 2 * (uint64_t)c->nfft) continue;
            void *dev;
            HZ_TRY(late_filter_for(c, segs[q].step, omega, &dev, true));
        }
    }
    return HZSDR_OK;
}

// The modulated filter of every clock run long enough to hold a whole block (lookups only).
static int late_filters(hzsdr_chain *c, const EwProgram &P, size_t n, LateFilters *out, bool *any) {
    *any = false;
    memset(out, 0, sizeof *out);
    if (c->mix_in_order || P.segs.big_n != 0 || c->taps_host.empty()) return HZSDR_OK;
    if (!c->has_shift) {
        // no clock involved: Gain / Multiply commute with the filter everywhere, with the
        // taps as they are (the kernel sees an empty run table: run 0)
        out->h[0] = (const float2 *)c->hfreq;
        *any = true;
        return HZSDR_OK;
    }
    const double omega = chain_omega(c);
    for (int r = 0; r < P.segs.n; r++) {
        const uint64_t first = P.segs.first[r], end = r + 1 < P.segs.n ? P.segs.first[r + 1] : (uint64_t)n;
        if (end - first < 2 * (uint64_t)c->nfft) continue;
        void *dev;
        HZ_TRY(late_filter_for(c, P.segs.step[r], omega, &dev, false));
        out->h[r] = (const float2 *)dev;
        if (dev) *any = true;
    }
    return HZSDR_OK;
}

template <int FMT>
static int run_fmt(hzsdr_chain *c, const void *in, size_t n_cons, void *out, size_t n_out,
                   const EwProgram &P) {
    hzsdr_ctx *ctx = c->ctx;
    switch (c->term) {
    case TERM_NONE: launch_map<FMT>(ctx, in, out, n_cons, P); break;
    case TERM_DECIMATE:
        hipLaunchKernelGGL((chain_decimate_kernel<FMT>), dim3(blocks_for(ctx, n_out)), dim3(kThreads), 0,
                           ctx->stream, in, (float2 *)out, n_out, kReaderBlock / c->factor,
                           (size_t)c->factor, P);
        break;
    case TERM_DOWNSAMPLE:
        hipLaunchKernelGGL((chain_downsample_kernel<FMT>), dim3(blocks_for(ctx, n_out)), dim3(kThreads), 0,
                           ctx->stream, in, (float2 *)out, n_out, kReaderBlock / c->factor, c->factor, P);
        break;
    case TERM_CONV:
        return conv_blocks_device(ctx, FMT, c->flen, in, out, c->filt, n_cons / c->flen, c->factor,
                                  c->factor > 1 ? kReaderBlock / c->factor : 0, P);
    case TERM_FIR: {
        const float2 *tw, *tws = nullptr;
        HZ_TRY(get_twiddles(ctx, c->nfft, &tw));
        const size_t nblocks = (n_cons + c->hop - 1) / c->hop;
        const float2 *hist = (const float2 *)c->hist[c->hist_cur];
        float2 *nhist = (float2 *)c->hist[c->hist_cur ^ 1];
        unsigned grid = (unsigned)nblocks;
        const unsigned D = c->factor;
#define HZ_FIR(N, FOLD)                                                                                  \
    hipLaunchKernelGGL((fir_decimate_kernel<N, FMT, FOLD>), dim3(grid), dim3(fft_block(N)), 0,           \
                       ctx->stream, in, (float2 *)out, hist, nhist, (const float2 *)c->hfreq, tw, tws, nblocks, \
                       n_cons, c->hop, c->off, D, P)
#define HZ_FIR_N(N)                                                                      \
    do {                                                                                 \
        constexpr int CNT_ = N / fft_tpt(N);                                             \
        const bool pow2 = (D & (D - 1)) == 0;                                            \
        if (pow2 && D >= 2 && D <= 16 && (int)D <= CNT_ && N / D >= 4)                   \
            HZ_TRY(get_twiddles(ctx, N / D, &tws));                                      \
        if (pow2 && D == 2 && 2 <= CNT_) HZ_FIR(N, (2 <= CNT_ ? 2 : 0));                 \
        else if (pow2 && D == 4 && 4 <= CNT_) HZ_FIR(N, (4 <= CNT_ ? 4 : 0));            \
        else if (pow2 && D == 8 && 8 <= CNT_) HZ_FIR(N, (8 <= CNT_ ? 8 : 0));            \
        else if (pow2 && D == 16 && 16 <= CNT_) HZ_FIR(N, (16 <= CNT_ ? 16 : 0));        \
        else HZ_FIR(N, 0);                                                               \
    } while (0)
// radix-16 core: fold when D is a power of two <= 16 and N/D is itself a radix-16 size
#define HZ_FIR16_L(N, FOLD, LATE, SPEC)                                                                 \
    hipLaunchKernelGGL((fir_decimate_kernel16<N, FMT, FOLD, LATE>),                                     \
                       dim3((unsigned)((nblocks + f16::xpb(N) - 1) / f16::xpb(N))), dim3(f16::block(N)), \
                       0, ctx->stream, in, (float2 *)out, hist, nhist, (const float2 *)c->hfreq, tw, SPEC, \
                       nblocks, n_cons, c->hop, c->off, D, P, late, tws)
#define HZ_FIR16(N, FOLD) HZ_FIR16_L(N, FOLD, false, (float2 *)nullptr)
#define HZ_SYNTH16(N, F, LATE)                                                                           \
    hipLaunchKernelGGL((fir_synth_kernel16<N, F, LATE>),                                                 \
                       dim3((unsigned)((nblocks + SynthGeom<N, F>::XPB - 1) / SynthGeom<N, F>::XPB)),     \
                       dim3(SynthGeom<N, F>::BS), 0, ctx->stream, (const float2 *)spec, (float2 *)out, tws, nblocks, n_cons, \
                       c->hop, c->off, P, late)
#define HZ_FIR16_FOLD(N, F)                                                            \
    if (D == F) {                                                                      \
        if constexpr (f16::xpb(N) == 1 && f16::ok(N / F)) {                            \
            HZ_TRY(get_twiddles(ctx, N / F, &tws));                                    \
            bool any_late = false;                                                     \
            HZ_TRY(late_filters(c, P, n_cons, &late, &any_late));                      \
            HZ_TRY(ensure_slot(ctx, 11, nblocks * (size_t)(N / F) * 8));               \
            float2 *spec = (float2 *)ctx->slots[11].ptr;                               \
            if (any_late) {                                                            \
                HZ_FIR16_L(N, F, true, spec);                                          \
                HZ_SYNTH16(N, F, true);                                                \
            } else {                                                                   \
                HZ_FIR16_L(N, F, false, spec);                                         \
                HZ_SYNTH16(N, F, false);                                               \
            }                                                                          \
            break;                                                                     \
        }                                                                              \
    }
// any other factor (1, 3, 5, 10, ...): full backward transform in the analysis kernel; the
// late mixer applies there too when a workgroup is one block (N >= 1024)
#define HZ_FIR16_FULL(N)                                                               \
    if constexpr (f16::xpb(N) == 1) {                                                  \
        bool any_late = false;                                                         \
        HZ_TRY(late_filters(c, P, n_cons, &late, &any_late));                          \
        if (any_late) HZ_FIR16_L(N, 0, true, (float2 *)nullptr);                       \
        else HZ_FIR16(N, 0);                                                           \
    } else {                                                                           \
        HZ_FIR16(N, 0);                                                                \
    }
#define HZ_FIR16_N(N)            \
    do {                         \
        HZ_FIR16_FOLD(N, 2)      \
        HZ_FIR16_FOLD(N, 4)      \
        HZ_FIR16_FOLD(N, 8)      \
        HZ_FIR16_FOLD(N, 16)     \
        HZ_FIR16_FULL(N)         \
    } while (0)
        LateFilters late{};
        switch (c->nfft) {
        case 256: HZ_FIR16_N(256); break;
        case 512: HZ_FIR16_N(512); break;
        case 1024: HZ_FIR16_N(1024); break;
        case 2048: HZ_FIR16_N(2048); break;
        case 4096: HZ_FIR16_N(4096); break;
        case 8192: HZ_FIR_N(8192); break;
        default: return HZSDR_ERR_INVALID_ARGUMENT;
        }
#undef HZ_FIR16_N
#undef HZ_FIR16_FULL
#undef HZ_FIR16_FOLD
#undef HZ_FIR16
#undef HZ_FIR16_L
#undef HZ_SYNTH16
#undef HZ_FIR_N
#undef HZ_FIR
        c->hist_cur ^= 1;  // the kernel wrote the next run's history into nhist
        break;
    }
    }
    return HZSDR_OK;
}

// The chain's kernel(s) over device buffers, enqueued on the context's stream.
// *ts_after is the NCO clock after `cons` samples; the caller commits it.
static int chain_launch(hzsdr_chain *c, const void *din, size_t cons, void *dout, size_t outn, double *ts_after) {
    EwProgram P{};
    P.n = c->n_ops;
    for (int i = 0; i < c->n_ops; i++) P.op[i] = c->ops[i];
    double ts = c->ts;
    if (c->has_shift) HZ_TRY(nco_plan(c->ctx, c->sample_rate, &ts, cons, &P.segs));
    int rc;
    switch (c->src_fmt) {
    case HZSDR_FMT_C64: rc = run_fmt<HZSDR_FMT_C64>(c, din, cons, dout, outn, P); break;
    case HZSDR_FMT_U8: rc = run_fmt<HZSDR_FMT_U8>(c, din, cons, dout, outn, P); break;
    case HZSDR_FMT_I8: rc = run_fmt<HZSDR_FMT_I8>(c, din, cons, dout, outn, P); break;
    default: rc = run_fmt<HZSDR_FMT_I16>(c, din, cons, dout, outn, P); break;
    }
    HZ_TRY(rc);
    *ts_after = ts;
    return HZSDR_OK;
}

// Snapshot a caller's frequency-domain filter into library memory ON THE CONTEXT'S STREAM:
// in a DEVICE context the filter may still be being produced by work enqueued on that
// stream (an hzsdr_fft_transform, a torch kernel), which the legacy null stream does not
// order against.  HOST contexts wait, so the caller's slice may be reused on return.
static int upload_filter(hzsdr_ctx *ctx, void *dst, const void *src, size_t bytes) {
    const bool host = ctx->memspace == HZSDR_MEM_HOST;
    HZ_HIP(ctx, hipMemcpyAsync(dst, src, bytes, host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
    if (host) HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HZSDR_OK;
}

static int chain_terminal_set(hzsdr_chain *c) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    if (c->term != TERM_NONE) return fail(c->ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: terminal stage already set");
    return HZSDR_OK;
}

static int chain_push(hzsdr_chain *c, const EwOp &op) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    if (c->term != TERM_NONE) return fail(c->ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: stage after the terminal stage");
    if (c->n_ops >= kMaxEw) return fail(c->ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: too many elementwise stages");
    c->ops[c->n_ops++] = op;
    return HZSDR_OK;
}

}  // namespace hz

extern "C" {

int hzsdr_chain_create(hzsdr_ctx *ctx, int src_format, uint64_t sample_rate, hzsdr_chain **out) {
    if (!ctx || !out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (hz::format_size(src_format) == 0) return hz::fail(ctx, HZSDR_ERR_FORMAT_UNKNOWN, "chain: unknown source format");
    hzsdr_chain *c = new hzsdr_chain();
    c->ctx = ctx;
    c->src_fmt = src_format;
    c->sample_rate = sample_rate;
    *out = c;
    return HZSDR_OK;
}

int hzsdr_chain_shift(hzsdr_chain *c, double shift_hz) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    if (c->sample_rate == 0) return hz::fail(c->ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: shift needs a sample rate");
    hz::EwOp op{};
    op.kind = hz::EW_SHIFT;
    op.tau_shift = (M_PI * 2) * shift_hz;
    int rc = hz::chain_push(c, op);
    if (rc == HZSDR_OK) c->has_shift = true;
    return rc;
}

int hzsdr_chain_gain(hzsdr_chain *c, float r) {
    hz::EwOp op{};
    op.kind = hz::EW_SCALE;
    op.a = r;
    return hz::chain_push(c, op);
}

int hzsdr_chain_rotate(hzsdr_chain *c, float re, float im) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    if (re == 1.0f && im == 0.0f) return HZSDR_OK;  // stream/multiply.go:59-62
    hz::EwOp op{};
    op.kind = hz::EW_ROTATE;
    op.a = re;
    op.b = im;
    return hz::chain_push(c, op);
}

int hzsdr_chain_decimate(hzsdr_chain *c, unsigned factor) {
    HZ_TRY(hz::chain_terminal_set(c));
    if (factor == 0 || factor > hz::kReaderBlock) return hz::fail(c->ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: decimate factor");
    // DecimateBuffer has no i8 case (stream/decimate.go:85-97); after a convert
    // stage the stream is c64, so the restriction only bites a bare i8 chain.
    c->term = TERM_DECIMATE;
    c->factor = factor;
    return HZSDR_OK;
}

int hzsdr_chain_downsample(hzsdr_chain *c, unsigned factor) {
    HZ_TRY(hz::chain_terminal_set(c));
    if (factor == 0 || factor > hz::kReaderBlock) return hz::fail(c->ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: downsample factor");
    c->term = TERM_DOWNSAMPLE;
    c->factor = factor;
    return HZSDR_OK;
}

int hzsdr_chain_convolution(hzsdr_chain *c, const void *filter_freq, size_t filter_len,
                            unsigned decimate_factor) {
    using namespace hz;
    HZ_TRY(chain_terminal_set(c));
    hzsdr_ctx *ctx = c->ctx;
    if (!filter_freq || !fft_lds_ok(filter_len))
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: filter length must be a power of two in 4..8192");
    if (decimate_factor == 0 || decimate_factor > kReaderBlock) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: decimate factor");
    HZ_TRY(enter(ctx));
    void *filt = nullptr;
    HZ_HIP(ctx, hipMalloc(&filt, filter_len * 8));
    int rc = upload_filter(ctx, filt, filter_freq, filter_len * 8);
    if (rc != HZSDR_OK) {
        (void)hipFree(filt);
        return rc;
    }
    c->filt = filt;
    c->flen = filter_len;
    c->factor = decimate_factor;
    c->term = TERM_CONV;
    return HZSDR_OK;
}

int hzsdr_chain_fir_decimate(hzsdr_chain *c, const float *taps, size_t n_taps, unsigned factor) {
    using namespace hz;
    HZ_TRY(chain_terminal_set(c));
    hzsdr_ctx *ctx = c->ctx;
    if (!taps || n_taps == 0 || factor == 0) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: fir taps / factor");
    // N_fft = next power of two >= 4 * taps, in [256, 8192]
    unsigned nfft = 256;
    while (nfft < 4 * n_taps && nfft < 8192) nfft <<= 1;
    unsigned off = (unsigned)(n_taps - 1);
    off = (off + factor - 1) / factor * factor;  // first valid output on the decimation grid
    if (off + factor > nfft) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: too many taps for the 8192-point overlap-save block");
    unsigned hop = (nfft - off) / factor * factor;
    HZ_TRY(enter(ctx));
    const size_t hb = (size_t)(off ? off : 1) * 8;
    c->ntaps = n_taps;
    c->nfft = nfft;
    c->hop = hop;
    c->off = off;
    c->factor = factor;
    c->poly = fold_poly((int)nfft, (int)factor);
    c->taps_host.assign(taps, taps + 2 * n_taps);
    // every allocation is released again if a later step fails: the chain stays without a
    // terminal stage (a retry starts clean, nothing leaks)
    auto build = [&]() -> int {
        HZ_HIP(ctx, hipMalloc(&c->hfreq, (size_t)nfft * 8));
        HZ_HIP(ctx, hipMalloc(&c->hist[0], hb));
        HZ_HIP(ctx, hipMalloc(&c->hist[1], hb));
        HZ_HIP(ctx, hipMemsetAsync(c->hist[0], 0, hb, ctx->stream));
        HZ_HIP(ctx, hipMemsetAsync(c->hist[1], 0, hb, ctx->stream));
        HZ_TRY(filter_spectrum(c, taps, c->hfreq));
        return prepare_late_filters(c, c->ts);
    };
    const int rc = build();
    if (rc != HZSDR_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        for (void **p : {&c->hfreq, &c->hist[0], &c->hist[1]}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        for (auto &kv : c->late_cache) (void)hipFree(kv.second);
        c->late_cache.clear();
        c->taps_host.clear();
        return rc;
    }
    c->term = TERM_FIR;
    return HZSDR_OK;
}

int hzsdr_chain_plan(const hzsdr_chain *c, size_t n_in, size_t *n_consumed, size_t *n_out) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    size_t cons = n_in, outn = n_in;
    switch (c->term) {
    case TERM_NONE: break;
    case TERM_DECIMATE:
    case TERM_DOWNSAMPLE:
        cons = n_in / hz::kReaderBlock * hz::kReaderBlock;
        outn = cons / hz::kReaderBlock * (hz::kReaderBlock / c->factor);
        break;
    case TERM_CONV: {
        size_t blk = c->flen;
        if (c->factor > 1 && blk < hz::kReaderBlock) blk = hz::kReaderBlock;
        cons = n_in / blk * blk;
        outn = c->factor > 1 ? cons / hz::kReaderBlock * (hz::kReaderBlock / c->factor) : cons;
        break;
    }
    case TERM_FIR:
        cons = n_in / c->factor * c->factor;
        outn = cons / c->factor;
        break;
    }
    if (n_consumed) *n_consumed = cons;
    if (n_out) *n_out = outn;
    return HZSDR_OK;
}

int hzsdr_chain_run(hzsdr_chain *c, const void *in, size_t n_in, void *out, size_t out_cap,
                    size_t *n_consumed, size_t *n_out) {
    using namespace hz;
    if (n_consumed) *n_consumed = 0;
    if (n_out) *n_out = 0;
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = c->ctx;
    size_t cons, outn;
    hzsdr_chain_plan(c, n_in, &cons, &outn);
    if (out_cap < outn) return fail(ctx, HZSDR_ERR_DST_TOO_SMALL, "chain: output buffer too small");
    if (cons && (!in || !out)) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (cons == 0) return HZSDR_OK;
    Stage st(ctx);
    const void *din;
    void *dout;
    HZ_TRY(st.in(0, in, cons * format_size(c->src_fmt), &din));
    HZ_TRY(st.out(1, out, outn * 8, &dout));
    double ts;
    HZ_TRY(chain_launch(c, din, cons, dout, outn, &ts));
    HZ_TRY(st.finish());
    c->ts = ts;
    if (n_consumed) *n_consumed = cons;
    if (n_out) *n_out = outn;
    return HZSDR_OK;
}

int hzsdr_chain_reset(hzsdr_chain *c) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = c->ctx;
    HZ_TRY(hz::enter(ctx));
    c->ts = 0.0;
    if (c->term == TERM_FIR) {
        const size_t hb = (size_t)(c->off ? c->off : 1) * 8;
        HZ_HIP(ctx, hipMemsetAsync(c->hist[0], 0, hb, ctx->stream));
        HZ_HIP(ctx, hipMemsetAsync(c->hist[1], 0, hb, ctx->stream));
    }
    return HZSDR_OK;
}

int hzsdr_chain_set_time(hzsdr_chain *c, double ts) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    if (!(ts >= 0.0) || ts > 6.283185307179586476925286766559)  // the closure's clock lives in [0, 2*pi]
        return hz::fail(c->ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: clock outside [0, 2*pi]");
    HZ_TRY(hz::enter(c->ctx));
    c->ts = ts;
    if (c->term == TERM_FIR) HZ_TRY(hz::prepare_late_filters(c, ts));
    return HZSDR_OK;
}

int hzsdr_chain_time(const hzsdr_chain *c, double *ts) {
    if (!c || !ts) return HZSDR_ERR_INVALID_ARGUMENT;
    *ts = c->ts;
    return HZSDR_OK;
}

int hzsdr_chain_free(hzsdr_chain *c) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->filt) (void)hipFree(c->filt);
    if (c->hfreq) (void)hipFree(c->hfreq);
    if (c->hist[0]) (void)hipFree(c->hist[0]);
    if (c->hist[1]) (void)hipFree(c->hist[1]);
    for (auto &kv : c->late_cache) (void)hipFree(kv.second);
    delete c;
    return HZSDR_OK;
}

int hzsdr_chain_mix_in_order(hzsdr_chain *c, int in_order) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    c->mix_in_order = in_order != 0;
    return HZSDR_OK;
}

// ---- fft.ConvolveFreq / Convolve / CrossCorrelate closures --------------------------------

int hzsdr_convolve_freq_create(hzsdr_ctx *ctx, void *dst, size_t dst_len, const void *src,
                               size_t src_len, const void *freq, size_t freq_len, hzsdr_conv **out) {
    using namespace hz;
    if (!ctx || !out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (src_len != dst_len || src_len != freq_len)  // fft/convolution.go:156-158
        return fail(ctx, HZSDR_ERR_LENGTH_MISMATCH, "sdr/fft.Convolve: Lengths do not match exactly");
    const size_t n = src_len;
    if (n == 0 || (n & (n - 1)) || !dst || !src || !freq)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "convolve: power-of-two length and non-null buffers");
    HZ_TRY(enter(ctx));
    void *filt = nullptr;
    HZ_HIP(ctx, hipMalloc(&filt, n * 8));
    int rc = upload_filter(ctx, filt, freq, n * 8);
    if (rc != HZSDR_OK) {
        (void)hipFree(filt);
        return rc;
    }
    *out = new hzsdr_conv{ctx, 0, dst, src, nullptr, n, filt};
    return HZSDR_OK;
}

int hzsdr_convolve_create(hzsdr_ctx *ctx, void *dst, size_t dst_len, const void *iq1, size_t iq1_len,
                          const void *iq2, size_t iq2_len, int mode, hzsdr_conv **out) {
    using namespace hz;
    if (!ctx || !out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (iq1_len != iq2_len || iq1_len != dst_len)  // fft/convolution.go:37-39
        return fail(ctx, HZSDR_ERR_LENGTH_MISMATCH, "sdr/fft: IQ/Dest buffer lengths do not match exactly");
    const size_t n = iq1_len;
    if (n == 0 || (n & (n - 1)) || !dst || !iq1 || !iq2)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "convolve: power-of-two length and non-null buffers");
    if (mode != HZSDR_CONV_CONVOLVE && mode != HZSDR_CONV_CROSS_CORRELATE) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = new hzsdr_conv{ctx, mode == HZSDR_CONV_CONVOLVE ? 1 : 2, dst, iq1, iq2, n, nullptr};
    return HZSDR_OK;
}

int hzsdr_conv_exec(hzsdr_conv *cv) {
    using namespace hz;
    if (!cv) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = cv->ctx;
    HZ_TRY(enter(ctx));
    const size_t bytes = cv->n * 8;
    Stage st(ctx);
    const void *d1, *d2 = nullptr;
    void *dd;
    HZ_TRY(st.in(0, cv->src1, bytes, &d1));
    if (cv->kind != 0) HZ_TRY(st.in(2, cv->src2, bytes, &d2));
    HZ_TRY(st.out(1, cv->dst, bytes, &dd));
    if (cv->kind == 0 && fft_lds_ok(cv->n)) {
        EwProgram P{};
        HZ_TRY(conv_blocks_device(ctx, HZSDR_FMT_C64, cv->n, d1, dd, cv->filt, 1, 1, 0, P));
    } else {
        HZ_TRY(conv_generic_device(ctx, dd, d1, cv->kind == 0 ? cv->filt : d2, cv->n, cv->kind));
    }
    return st.finish();
}

int hzsdr_conv_set_filter(hzsdr_conv *cv, const void *freq, size_t freq_len) {
    using namespace hz;
    if (!cv) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = cv->ctx;
    if (cv->kind != 0 || !freq) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "conv: not a ConvolveFreq closure");
    if (freq_len != cv->n) return fail(ctx, HZSDR_ERR_LENGTH_MISMATCH, "sdr/fft.Convolve: Lengths do not match exactly");
    HZ_TRY(enter(ctx));
    return upload_filter(ctx, cv->filt, freq, cv->n * 8);  // stream-ordered after earlier execs
}

int hzsdr_conv_free(hzsdr_conv *cv) {
    if (!cv) return HZSDR_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(cv->ctx->device);
    (void)hipStreamSynchronize(cv->ctx->stream);
    if (cv->filt) (void)hipFree(cv->filt);
    delete cv;
    return HZSDR_OK;
}

int hzsdr_convolution_blocks(hzsdr_ctx *ctx, void *out, size_t out_len, const void *in, size_t in_len,
                             const void *filter_freq, size_t filter_len, size_t *n_out) {
    using namespace hz;
    if (n_out) *n_out = 0;
    if (!ctx) return HZSDR_ERR_INVALID_ARGUMENT;
    if (!fft_lds_ok(filter_len) || !filter_freq)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "convolution: filter length must be a power of two in 4..8192");
    const size_t nblocks = in_len / filter_len, n = nblocks * filter_len;
    if (out_len < n) return fail(ctx, HZSDR_ERR_DST_TOO_SMALL, "convolution: output buffer too small");
    if (n && (!in || !out)) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    Stage st(ctx);
    const void *din, *dfilt;
    void *dout;
    HZ_TRY(st.in(0, in, n * 8, &din));
    HZ_TRY(st.in(2, filter_freq, filter_len * 8, &dfilt));
    HZ_TRY(st.out(1, out, n * 8, &dout));
    EwProgram P{};
    HZ_TRY(conv_blocks_device(ctx, HZSDR_FMT_C64, filter_len, din, dout, dfilt, nblocks, 1, 0, P));
    HZ_TRY(st.finish());
    if (n_out) *n_out = n;
    return HZSDR_OK;
}

}  // extern "C"

// =============================================================================
// Pinned ring in front of a chain: the memory a
// stream.RingBuffer's IQBufferAllocator hands out (stream/ring.go:60-68) is ONE
// hipHostMalloc region of slots * slot_length samples; each submitted slot goes
// upload -> chain kernel -> download on three streams chained by events, so the
// PCIe transfers of neighbouring slots overlap the kernel.
// =============================================================================

struct hzsdr_ring {
    hzsdr_chain *chain = nullptr;
    hzsdr_ctx *ctx = nullptr;
    size_t slot_len = 0, out_len = 0;  // samples per slot in / out
    int nslots = 0;
    hipStream_t s_up = nullptr, s_down = nullptr;
    char *pin_in = nullptr, *pin_out = nullptr;  // nslots * slot bytes each, contiguous
    char *dev_in = nullptr, *dev_out = nullptr;
    struct Slot {
        hipEvent_t up = nullptr, done = nullptr, down = nullptr;
        size_t n_out = 0;
        int state = 0;  // 0 free, 1 acquired, 2 in flight
    };
    std::vector<Slot> slots;
    size_t widx = 0, ridx = 0;  // next slot to acquire / to pop
    int inflight = 0;
    size_t in_bytes() const { return slot_len * (size_t)hz::format_size(chain->src_fmt); }
    size_t out_bytes() const { return out_len * 8; }
};

extern "C" {

int hzsdr_ring_free(hzsdr_ring *r) {
    if (!r) return HZSDR_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(r->ctx->device);
    if (r->s_up) (void)hipStreamSynchronize(r->s_up);
    (void)hipStreamSynchronize(r->ctx->stream);
    if (r->s_down) (void)hipStreamSynchronize(r->s_down);
    for (auto &s : r->slots) {
        if (s.up) (void)hipEventDestroy(s.up);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.down) (void)hipEventDestroy(s.down);
    }
    if (r->pin_in) (void)hipHostFree(r->pin_in);
    if (r->pin_out) (void)hipHostFree(r->pin_out);
    if (r->dev_in) (void)hipFree(r->dev_in);
    if (r->dev_out) (void)hipFree(r->dev_out);
    if (r->s_up) (void)hipStreamDestroy(r->s_up);
    if (r->s_down) (void)hipStreamDestroy(r->s_down);
    delete r;
    return HZSDR_OK;
}

int hzsdr_ring_create(hzsdr_chain *c, size_t slot_length, int slots, hzsdr_ring **out) {
    using namespace hz;
    if (!c || !out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    hzsdr_ctx *ctx = c->ctx;
    if (slots < 2 || slots > 64 || slot_length == 0)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: needs 2..64 slots of at least one sample");
    size_t cons, outn;
    hzsdr_chain_plan(c, slot_length, &cons, &outn);
    if (cons != slot_length)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: slot length must be a whole number of the chain's blocks");
    HZ_TRY(enter(ctx));
    hzsdr_ring *r = new hzsdr_ring();
    r->chain = c;
    r->ctx = ctx;
    r->slot_len = slot_length;
    r->out_len = outn ? outn : 1;
    r->nslots = slots;
    r->slots.resize(slots);
#define HZ_RING(call)                                                         \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) {                                              \
            int rc__ = hip_fail(ctx, e__, #call, __FILE__, __LINE__);         \
            hzsdr_ring_free(r);                                               \
            return rc__;                                                      \
        }                                                                     \
    } while (0)
    HZ_RING(hipStreamCreateWithFlags(&r->s_up, hipStreamNonBlocking));
    HZ_RING(hipStreamCreateWithFlags(&r->s_down, hipStreamNonBlocking));
    HZ_RING(hipHostMalloc((void **)&r->pin_in, r->in_bytes() * slots, hipHostMallocDefault));
    HZ_RING(hipHostMalloc((void **)&r->pin_out, r->out_bytes() * slots, hipHostMallocDefault));
    HZ_RING(hipMalloc((void **)&r->dev_in, r->in_bytes() * slots));
    HZ_RING(hipMalloc((void **)&r->dev_out, r->out_bytes() * slots));
    for (auto &s : r->slots) {
        HZ_RING(hipEventCreateWithFlags(&s.up, hipEventDisableTiming));
        HZ_RING(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        HZ_RING(hipEventCreateWithFlags(&s.down, hipEventDisableTiming));
    }
#undef HZ_RING
    *out = r;
    return HZSDR_OK;
}

int hzsdr_ring_iq_buffer(const hzsdr_ring *r, void **base, size_t *n_samples, size_t *slot_length) {
    if (!r) return HZSDR_ERR_INVALID_ARGUMENT;
    if (base) *base = r->pin_in;
    if (n_samples) *n_samples = r->slot_len * (size_t)r->nslots;
    if (slot_length) *slot_length = r->slot_len;
    return HZSDR_OK;
}

int hzsdr_ring_acquire(hzsdr_ring *r, int *slot, void **iq) {
    using namespace hz;
    if (!r || !slot) return HZSDR_ERR_INVALID_ARGUMENT;
    const int i = (int)(r->widx % (size_t)r->nslots);
    if (r->slots[i].state != 0)
        return fail(r->ctx, HZSDR_ERR_DST_TOO_SMALL, "ring: every slot is in flight (pop first)");  // the overrun case
    r->slots[i].state = 1;
    *slot = i;
    if (iq) *iq = r->pin_in + (size_t)i * r->in_bytes();
    return HZSDR_OK;
}

int hzsdr_ring_submit(hzsdr_ring *r, int slot, size_t n) {
    using namespace hz;
    if (!r || slot < 0 || slot >= r->nslots) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = r->ctx;
    auto &s = r->slots[slot];
    if (s.state != 1 || slot != (int)(r->widx % (size_t)r->nslots))
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: submit of a slot that is not the acquired one");
    size_t cons, outn;
    hzsdr_chain_plan(r->chain, n, &cons, &outn);
    if (n == 0 || n > r->slot_len || cons != n)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: a slot must hold a whole number of the chain's blocks");
    HZ_TRY(enter(ctx));
    const size_t fs = (size_t)format_size(r->chain->src_fmt);
    char *hin = r->pin_in + (size_t)slot * r->in_bytes(), *din = r->dev_in + (size_t)slot * r->in_bytes();
    char *hout = r->pin_out + (size_t)slot * r->out_bytes(), *dout = r->dev_out + (size_t)slot * r->out_bytes();
    HZ_HIP(ctx, hipMemcpyAsync(din, hin, n * fs, hipMemcpyHostToDevice, r->s_up));
    HZ_HIP(ctx, hipEventRecord(s.up, r->s_up));
    HZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, s.up, 0));
    double ts;
    HZ_TRY(chain_launch(r->chain, din, n, dout, outn, &ts));
    HZ_HIP(ctx, hipGetLastError());
    r->chain->ts = ts;
    HZ_HIP(ctx, hipEventRecord(s.done, ctx->stream));
    HZ_HIP(ctx, hipStreamWaitEvent(r->s_down, s.done, 0));
    HZ_HIP(ctx, hipMemcpyAsync(hout, dout, outn * 8, hipMemcpyDeviceToHost, r->s_down));
    HZ_HIP(ctx, hipEventRecord(s.down, r->s_down));
    s.n_out = outn;
    s.state = 2;
    r->widx++;
    r->inflight++;
    return HZSDR_OK;
}

int hzsdr_ring_pop(hzsdr_ring *r, const void **out, size_t *n_out) {
    using namespace hz;
    if (!r) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = r->ctx;
    const int i = (int)(r->ridx % (size_t)r->nslots);
    auto &s = r->slots[i];
    if (s.state != 2) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "ring: nothing in flight");  // the underrun case
    HZ_TRY(enter(ctx));
    HZ_HIP(ctx, hipEventSynchronize(s.down));
    if (out) *out = r->pin_out + (size_t)i * r->out_bytes();
    if (n_out) *n_out = s.n_out;
    s.state = 0;
    r->ridx++;
    r->inflight--;
    return HZSDR_OK;
}

int hzsdr_ring_in_flight(const hzsdr_ring *r) { return r ? r->inflight : -1; }

}  // extern "C"

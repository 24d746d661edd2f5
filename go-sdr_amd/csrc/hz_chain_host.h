// hz_chain_host.h -- the host-side state of chains and convolution closures, shared by hz_chain.hip (the chain API,
// the streaming terminals), hz_chain_fir.hip (the FIR-decimate terminal: spectra, tables, planners, launches),
// hz_conv.hip (block convolution and the fft.Convolve closures) and hz_ring.hip (the pinned ring in front of a chain).
#pragma once
#include <math.h>

#include <algorithm>

#include "hz_chain_dev.h"
#include "hz_firmm2_plan.h"
#include "hz_firmm_plan.h"
#include "hz_fft_api.h"

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
    void *hfreq_late = nullptr;  // the same times late_scale() (chains without a Shift; see late_scale)
    // the histories of consecutive calls: a ring of kHist buffers, call k reads hist[hist_cur] and writes
    // hist[hist_next()] (four, not two: a pipelined chain has two calls in flight -- hzsdr_chain_pipeline)
    static constexpr int kHist = 4;
    void *hist[kHist] = {nullptr, nullptr, nullptr, nullptr};
    int hist_cur = 0;
    int hist_next() const { return (hist_cur + 1) % kHist; }
    size_t ntaps = 0;
    unsigned nfft = 0, hop = 0, off = 0;
    // late mixer (see fir_decimate_kernel16): the taps, and FFT(taps * exp(-i*Omega*k*step))/N
    // per distinct clock step seen so far (one per binade of the NCO clock)
    std::vector<double> taps_host;  // (re, im) pairs, exact copies of the caller's float32 taps
    std::map<uint64_t, void *> late_cache;
    bool mix_in_order = false;
    bool shift_ulp1 = false;  // hzsdr_chain_shift_ulp1
    bool poly = false;  // hfreq / late_cache hold the polyphase layout (fold_poly)
    // int8 matrix form (hz_firmm.h): geometry, the taps on the device (fix-up workgroups) and
    // one digit table per distinct clock step (key 0: no Shift stage)
    bool mm_ok = false;
    int mm_ver = 1;  // 1: hz_firmm.h (one round of chunk workgroups), 2: hz_firmm2.h (persistent passes; D = 8)
    hz::mm::Geom mmg{};
    void *taps_dev = nullptr;
    std::map<uint64_t, void *> mm_cache;
    // the last `off` RAW samples of the previous call (two buffers, flipped with hist[]): valid
    // after a call on the matrix path; rh_step / rh_len describe the clock run they end in
    void *rhist[kHist] = {nullptr, nullptr, nullptr, nullptr};
    bool rh_valid = false;
    double rh_step = 0.0, rh_next = 0.0;
    uint64_t rh_len = 0;
    bool debug_mm = false;  // HZ_DEBUG_MM (diag_env)
    // hzsdr_chain_fir_options: which implementation the FIR-decimate terminal takes (0: the library chooses),
    // the smallest overlap-save block, the matrix loop's form -- A/B measurements and tests
    int fir_impl = HZSDR_FIR_IMPL_AUTO;
    unsigned fir_nfft_min = 0;
    int fir_loop_form = 0;
    int last_path = HZSDR_FIR_PATH_NONE;
    // hzsdr_chain_pipeline: consecutive calls on the matrix path alternate between two streams of the chain's own
    // (the next launch's workgroups start as this one's finish), the history of the next call is formed by a small
    // kernel of its own; `pcall` counts the overlapped calls since the chain's streams were last re-seeded from the
    // context's stream.  ONLY a call that states what its buffers wait for overlaps (hzsdr_chain_run_after, the ring:
    // `relaxed` / `ready` below, set for the duration of that call); hzsdr_chain_run is ordered like any other call.
    bool pipelined = false;
    hipStream_t pstream[2] = {nullptr, nullptr};
    hipEvent_t ev_done[4] = {nullptr, nullptr, nullptr, nullptr}, ev_hist[4] = {nullptr, nullptr, nullptr, nullptr}, ev_in = nullptr;
    uint64_t pcall = 0;
    bool relaxed = false;        // this call: the caller says when its buffers are ready (not: behind the context's stream)
    hipEvent_t ready = nullptr;  // ... namely when this event has fired (null: now)
    // the buffers of the last two overlapped calls: a call that touches them is ordered behind those calls
    struct Span {
        const char *p;
        size_t n;
    };
    struct CallBufs {
        int nb = 0;
        Span in[hz::mm2::kMaxBatch], out[hz::mm2::kMaxBatch];
    };
    CallBufs pbufs[2];  // [0]: the previous call, [1]: the one before it
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

// The diagnostic environment switches (README.md), read ONCE per process -- getenv is not safe against a
// concurrent setenv, and a chain's behaviour must not depend on what the environment holds at the moment a stage is
// added -- and only in a library built with -DHZSDR_DIAG (csrc/Makefile: DIAG=1).  Programs select an
// implementation per chain with hzsdr_chain_fir_options instead.
struct DiagEnv {
    bool fir_fft = false, mm_v1 = false, no_slow_first = false, debug_late = false, debug_mm = false;
    unsigned nfft_min = 0;
    int rolled = 0;
};
inline const DiagEnv &diag_env() {
    static const DiagEnv e = [] {
        DiagEnv d;
#ifdef HZSDR_DIAG
        d.fir_fft = getenv("HZ_FIR_FFT") != nullptr;
        d.mm_v1 = getenv("HZ_MM_V1") != nullptr;
        d.no_slow_first = getenv("HZ_NO_SLOW_FIRST") != nullptr;
        d.debug_late = getenv("HZ_DEBUG_LATE") != nullptr;
        d.debug_mm = getenv("HZ_DEBUG_MM") != nullptr;
        if (const char *v = getenv("HZ_FIR_NFFT_MIN")) d.nfft_min = (unsigned)atoi(v);
        if (const char *v = getenv("HZ_MM_ROLLED")) d.rolled = atoi(v);
#endif
        return d;
    }();
    return e;
}

// Launch with `lds` bytes of dynamic LDS; above the 64 KiB default a kernel needs its limit
// raised once (160 KiB per CU on gfx950).
template <class K, class... A>
inline int launch_fv(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, A... args) {
    if (lds > 48 * 1024) HZ_TRY(raise_dynamic_lds((const void *)kernel));
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
    return HZSDR_OK;  // (the launch's own status: the caller's hipGetLastError)
}

// hz_conv.hip: `nblocks` blocks of n samples through forward FFT -> bins *= filter -> backward FFT (-> DecimateReader pick)
int conv_blocks_device(hzsdr_ctx *ctx, int fmt, size_t n, const void *in, void *out, const void *filt, size_t nblocks,
                       unsigned dec, size_t per, const EwProgram &P);
// One call's buffers: nbuf inputs of n_each samples and their outputs (hzsdr_chain_run_batch; nbuf = 1: an ordinary call)
struct CallBatch {
    const void *const *ins;
    void *const *outs;
    size_t nbuf, n_each, out_each;
};
// internal status of a batched call that the single-launch form cannot take: the caller runs the buffers one by one
constexpr int kBatchFallback = -1000;
// hz_chain_fir.hip: one call of a chain whose terminal is the FIR-decimate stage (n_cons samples in all; `cb` with
// nbuf > 1: the call's buffers, in / out are its first -- kBatchFallback unless the persistent-pass matrix kernel
// takes the whole call); the clock's filters / digit tables for a chain whose clock was set by hand
template <int FMT> int fir_run(hzsdr_chain *c, const void *in, size_t n_cons, void *out, const EwProgram &P, const CallBatch *cb = nullptr);
int prepare_late_filters(hzsdr_chain *c, double ts0);
int prepare_mm_tables(hzsdr_chain *c, double ts0);
// hz_chain.hip
int chain_launch(hzsdr_chain *c, const void *din, size_t cons, void *dout, size_t outn, double *ts_after, const CallBatch *cb = nullptr);
int upload_filter(hzsdr_ctx *ctx, void *dst, const void *src, size_t bytes);
int chain_terminal_set(hzsdr_chain *c);
// hz_chain_fir.hip: a pipelined chain's next overlapped call starts over from the context's stream (every call is
// joined to that stream as it is launched, so there is nothing to wait for: a call on the context's stream, a reset,
// a change of the clock need just this)
int pipeline_drain(hzsdr_chain *c);
// the stream and order an overlapped call launches on: `a` for the call's kernel, `b` for the kernel that forms the
// next call's history (null for chains without one); re-seeds both from the context's stream when the chain starts
// over or the call's buffers overlap the last two calls'.  pipeline_join: behind the launches.
int pipeline_begin(hzsdr_chain *c, const CallBatch &cb, int fmt_size, hipStream_t *a, hipStream_t *b);
int pipeline_join(hzsdr_chain *c, hipStream_t a, hipStream_t b);
// the chain's two streams and events, created on first use
int pipeline_streams(hzsdr_chain *c);

}  // namespace hz

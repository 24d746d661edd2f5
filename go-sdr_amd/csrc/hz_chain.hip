// hz_chain.hip -- fused operator chains: the chain API, the elementwise stages and the streaming terminals (the
// FIR-decimate terminal lives in hz_chain_fir.hip, the convolution launches in hz_conv.hip, the pinned ring in
// hz_ring.hip; hz_chain_host.h holds what they share).
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
#include "hz_chain_host.h"

namespace hz {

void nco_shift_ulp1_map4(hzsdr_ctx *ctx, void *buf, size_t nvec4, uint64_t base, double tau_shift, const NcoSegs &sg) {
    EwProgram P{};
    P.n = 1;
    P.op[0].kind = EW_SHIFT;
    P.op[0].tau_shift = tau_shift;
    P.segs = sg;
    hipLaunchKernelGGL((chain_map_kernel<HZSDR_FMT_C64, 4, SHAPE_SHIFT_ULP1>), dim3(blocks_for(ctx, nvec4)), dim3(kThreads), 0, ctx->stream,
                       (const void *)buf, (float2 *)buf, nvec4, base, P);
}

// A call of shift_exact_kernel that touches `bytes` (an in-place buffer counted once) uses non-temporal loads and
// stores from 96 MiB on: the next buffer of a stream then finds nothing of this one in the 256 MB memory-side cache
// anyway, and past the cache the map takes 47.8 us per 2^24 samples where plain accesses take 53.9 (from the cache
// the kernel is bound by its float64 instructions at ~46 us either way; tools/nco_ablate.hip, "rotation").
static bool shift_streams_past_cache(size_t bytes) { return bytes >= ((size_t)96 << 20); }

// shift_exact_kernel's launch: its workgroups walk the tiles (hz_chain_dev.h) -- as many workgroups as fit the chip at
// once (80 registers: six waves per SIMD, twelve two-wave workgroups per CU), fewer for a short call
#ifndef HZ_SHIFT_WGS
#define HZ_SHIFT_WGS 12
#endif
static unsigned shift_grid(const hzsdr_ctx *ctx, size_t nvec) {
    const size_t tiles = (nvec + (size_t)kShiftU * kShiftThreads - 1) / ((size_t)kShiftU * kShiftThreads);
    if (!kShiftPrefetch) return blocks_for(ctx, (nvec + kShiftU - 1) / kShiftU, kShiftThreads);
    const size_t cap = (size_t)ctx->num_cus * HZ_SHIFT_WGS;
    return (unsigned)(tiles < cap ? (tiles ? tiles : 1) : cap);
}

void nco_shift_exact_map2(hzsdr_ctx *ctx, void *buf, size_t nvec2, uint64_t base, double tau_shift, const NcoSegs &sg) {
    EwProgram P{};
    P.n = 1;
    P.op[0].kind = EW_SHIFT;
    P.op[0].tau_shift = tau_shift;
    P.segs = sg;
    const dim3 grid(shift_grid(ctx, nvec2)), block(kShiftThreads);
    if (shift_streams_past_cache(nvec2 * 16))
        hipLaunchKernelGGL((shift_exact_kernel<HZSDR_FMT_C64, false, true>), grid, block, 0, ctx->stream, (const void *)buf, (float4 *)buf, nvec2, base, P);
    else
        hipLaunchKernelGGL((shift_exact_kernel<HZSDR_FMT_C64, false>), grid, block, 0, ctx->stream, (const void *)buf, (float4 *)buf, nvec2, base, P);
}

template <int FMT>
static void launch_map(hzsdr_ctx *ctx, const void *in, void *out, size_t n, const EwProgram &P, bool ulp1) {
    using R = typename Raw<FMT>::t;
    // four samples per lane (four interleaved Sincos chains) when both pointers allow the
    // wider vectors, else two; the ragged end, or everything for a sample-aligned Go
    // sub-slice, one at a time
    const auto ok = [&](int w) { return ((uintptr_t)in % (sizeof(R) * w) == 0) && ((uintptr_t)out % (8 * w) == 0); };
    size_t done = 0;
    const int shape = P.n == 1 && P.op[0].kind == EW_SHIFT ? SHAPE_SHIFT
                      : P.n == 2 && P.op[0].kind == EW_SHIFT && P.op[1].kind == EW_SCALE ? SHAPE_SHIFT_GAIN
                                                                                         : SHAPE_ANY;
    if (!ulp1 && (shape == SHAPE_SHIFT || shape == SHAPE_SHIFT_GAIN) && ok(2) && n >= 2) {
        // the bit-exact Shift (+ Gain): two samples per vector, the factor by sincos_narrow (shift_exact_kernel)
        const size_t nvec = n / 2;
        const dim3 grid(shift_grid(ctx, nvec)), block(kShiftThreads);
        const bool nt = shift_streams_past_cache(in == out ? 8 * n : (sizeof(R) + 8) * n);
        if (shape == SHAPE_SHIFT_GAIN && nt)
            hipLaunchKernelGGL((shift_exact_kernel<FMT, true, true>), grid, block, 0, ctx->stream, in, (float4 *)out, nvec, (uint64_t)0, P);
        else if (shape == SHAPE_SHIFT_GAIN)
            hipLaunchKernelGGL((shift_exact_kernel<FMT, true>), grid, block, 0, ctx->stream, in, (float4 *)out, nvec, (uint64_t)0, P);
        else if (nt)
            hipLaunchKernelGGL((shift_exact_kernel<FMT, false, true>), grid, block, 0, ctx->stream, in, (float4 *)out, nvec, (uint64_t)0, P);
        else
            hipLaunchKernelGGL((shift_exact_kernel<FMT, false>), grid, block, 0, ctx->stream, in, (float4 *)out, nvec, (uint64_t)0, P);
        done = nvec * 2;
    } else if (ok(4) && n >= 4) {
        const size_t nvec = n / 4;
        const dim3 grid(blocks_for(ctx, nvec)), block(kThreads);
        if (shape == SHAPE_SHIFT_GAIN && ulp1)
            hipLaunchKernelGGL((chain_map_kernel<FMT, 4, SHAPE_SHIFT_GAIN_ULP1>), grid, block, 0, ctx->stream, in,
                               (float2 *)out, nvec, (uint64_t)0, P);
        else if (shape == SHAPE_SHIFT && ulp1)
            hipLaunchKernelGGL((chain_map_kernel<FMT, 4, SHAPE_SHIFT_ULP1>), grid, block, 0, ctx->stream, in,
                               (float2 *)out, nvec, (uint64_t)0, P);
        else if (shape == SHAPE_SHIFT_GAIN)
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

template <int FMT>
static int run_fmt(hzsdr_chain *c, const void *in, size_t n_cons, void *out, size_t n_out,
                   const EwProgram &P, const CallBatch *cb) {
    hzsdr_ctx *ctx = c->ctx;
    if (cb && cb->nbuf > 1 && c->term != TERM_FIR) return kBatchFallback;
    if (c->term != TERM_NONE && c->term != TERM_FIR) HZ_TRY(pipeline_drain(c));  // (these run on the context's stream)
    switch (c->term) {
    case TERM_NONE:
        if (c->pipelined && c->relaxed && P.segs.big_n == 0) {  // (the long clock table lives in ONE scratch slot)
            // hzsdr_chain_pipeline on a chain without a terminal, a call that says what its buffers wait for
            // (hzsdr_chain_run_after): nothing on the device carries over from call to call (the clock is the host's),
            // so consecutive calls alternate between the chain's two streams and the context's stream waits for each
            // -- two launches in flight keep the memory system busy through the tail of one and the head of the next
            // (Shift + Gain from HBM: 45.5 us per 2^24 samples against 50.4)
            const void *in1[1] = {in};
            void *out1[1] = {out};
            const CallBatch one{in1, out1, 1, n_cons, n_out};
            hipStream_t a, saved = ctx->stream;
            HZ_TRY(pipeline_begin(c, one, format_size(FMT), &a, nullptr));
            ctx->stream = a;
            launch_map<FMT>(ctx, in, out, n_cons, P, c->shift_ulp1);
            ctx->stream = saved;
            HZ_TRY(pipeline_join(c, a, nullptr));
            break;
        }
        HZ_TRY(pipeline_drain(c));
        launch_map<FMT>(ctx, in, out, n_cons, P, c->shift_ulp1);
        break;
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
    case TERM_FIR: return fir_run<FMT>(c, in, n_cons, out, P, cb);
    }
    return HZSDR_OK;
}

// The chain's kernel(s) over device buffers, enqueued on the context's stream.
// *ts_after is the NCO clock after `cons` samples; the caller commits it.
// (cb with nbuf > 1: `cons` / `outn` are the whole call's, din / dout its first buffers; kBatchFallback -- nothing
// launched, nothing committed -- when the chain cannot take the buffers in one launch)
int chain_launch(hzsdr_chain *c, const void *din, size_t cons, void *dout, size_t outn, double *ts_after, const CallBatch *cb) {
    if (cb && cb->nbuf > 1 && !(c->term == TERM_FIR && c->mm_ok && c->mm_ver == 2)) return kBatchFallback;
    EwProgram P{};
    P.n = c->n_ops;
    for (int i = 0; i < c->n_ops; i++) P.op[i] = c->ops[i];
    double ts = c->ts;
    if (c->has_shift) {
        // (a call over several buffers plans the clock over their concatenation; what fails for THAT length only -- a
        // clock table too long, no room for it -- is not the call's error: the buffers go one by one, as promised)
        const int rp = nco_plan(c->ctx, c->sample_rate, &ts, cons, &P.segs);
        if (rp != HZSDR_OK) return cb && cb->nbuf > 1 ? kBatchFallback : rp;
        if (cb && cb->nbuf > 1 && P.segs.big_n != 0) return kBatchFallback;  // (mm2_plan would refuse the long table anyway)
    }
    int rc;
    switch (c->src_fmt) {
    case HZSDR_FMT_C64: rc = run_fmt<HZSDR_FMT_C64>(c, din, cons, dout, outn, P, cb); break;
    case HZSDR_FMT_U8: rc = run_fmt<HZSDR_FMT_U8>(c, din, cons, dout, outn, P, cb); break;
    case HZSDR_FMT_I8: rc = run_fmt<HZSDR_FMT_I8>(c, din, cons, dout, outn, P, cb); break;
    default: rc = run_fmt<HZSDR_FMT_I16>(c, din, cons, dout, outn, P, cb); break;
    }
    if (rc != HZSDR_OK) return rc;
    *ts_after = ts;
    return HZSDR_OK;
}

// Snapshot a caller's frequency-domain filter into library memory ON THE CONTEXT'S STREAM:
// in a DEVICE context the filter may still be being produced by work enqueued on that
// stream (an hzsdr_fft_transform, a torch kernel), which the legacy null stream does not
// order against.  HOST contexts wait, so the caller's slice may be reused on return.
int upload_filter(hzsdr_ctx *ctx, void *dst, const void *src, size_t bytes) {
    const bool host = ctx->memspace == HZSDR_MEM_HOST;
    HZ_HIP(ctx, hipMemcpyAsync(dst, src, bytes, host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
    if (host) HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HZSDR_OK;
}

int chain_terminal_set(hzsdr_chain *c) {
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
    c->debug_mm = hz::diag_env().debug_mm;
    c->fir_nfft_min = hz::diag_env().nfft_min;
    c->fir_loop_form = hz::diag_env().rolled;
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
    if (!filter_freq || !fft_length_ok(filter_len))
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: filter length 1 ... 2^24 (a power of two) or 1 ... 2^23 (any other)");
    if (decimate_factor == 0 || decimate_factor > kReaderBlock) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: decimate factor");
    HZ_TRY(enter(ctx));
    HZ_TRY(fft_prepare(ctx, filter_len));  // (a length that is not a power of two: its chirp tables now, not inside the first run)
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
        if (c->factor > 1) {
            // (a DecimateReader behind the ConvolutionReader reads 32 Ki-sample blocks of its output: whole blocks of both,
            // i.e. their least common multiple -- 32 Ki itself for the power-of-two lengths up to it)
            size_t a = blk, b = hz::kReaderBlock;
            while (b) {
                const size_t t = a % b;
                a = b, b = t;
            }
            blk = blk / a * hz::kReaderBlock;
        }
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

// One call of the chain over nbuf buffers of n_in samples each (nbuf = 1: hzsdr_chain_run).  relaxed: the caller says
// what the buffers wait for (`ready`: an event, null: nothing) instead of "whatever the context's stream holds".
static int chain_run_impl(hzsdr_chain *c, const void *const *ins, void *const *outs, size_t nbuf, size_t n_in, size_t out_cap,
                          size_t *n_consumed, size_t *n_out, bool relaxed, hipEvent_t ready) {
    using namespace hz;
    if (n_consumed) *n_consumed = 0;
    if (n_out) *n_out = 0;
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = c->ctx;
    if (nbuf == 0 || nbuf > (size_t)mm2::kMaxBatch || !ins || !outs)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: a call takes 1..8 buffers");
    size_t cons, outn;
    hzsdr_chain_plan(c, n_in, &cons, &outn);
    if (out_cap < outn) return fail(ctx, HZSDR_ERR_DST_TOO_SMALL, "chain: output buffer too small");
    for (size_t j = 0; j < nbuf; j++)
        if (cons && (!ins[j] || !outs[j])) return HZSDR_ERR_INVALID_ARGUMENT;
    if (nbuf > 1 && cons != n_in) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: the buffers of a batch are consumed whole (a multiple of the chain's block)");
    HZ_TRY(enter(ctx));
    if (cons == 0) return HZSDR_OK;
    const bool device = ctx->memspace == HZSDR_MEM_DEVICE;
    if (ready) {
        // a HOST-space call (synchronous, staged on the context's stream) and a chain that does not overlap its calls
        // simply wait for the event where every call waits: on the host / on the context's stream
        if (!device) HZ_HIP(ctx, hipEventSynchronize(ready));
        else if (!(relaxed && c->pipelined)) HZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ready, 0));
    }
    struct Scope {  // (the call's ordering lives in the chain for the launches' duration)
        hzsdr_chain *c;
        ~Scope() { c->relaxed = false, c->ready = nullptr; }
    } scope{c};
    c->relaxed = relaxed && device && c->pipelined;
    c->ready = c->relaxed ? ready : nullptr;
    if (nbuf > 1 && device) {  // the buffers in ONE launch where the chain has such a form
        const CallBatch cb{ins, outs, nbuf, cons, outn};
        double ts;
        const int rc = chain_launch(c, ins[0], cons * nbuf, outs[0], outn * nbuf, &ts, &cb);
        if (rc == HZSDR_OK) {
            c->ts = ts;
            if (n_consumed) *n_consumed = cons;
            if (n_out) *n_out = outn;
            return HZSDR_OK;
        }
        if (rc != kBatchFallback) return rc;
    }
    for (size_t j = 0; j < nbuf; j++) {  // one by one: the same results as nbuf calls
        Stage st(ctx);
        const void *din;
        void *dout;
        HZ_TRY(st.in(0, ins[j], cons * format_size(c->src_fmt), &din));
        HZ_TRY(st.out(1, outs[j], outn * 8, &dout));
        double ts;
        HZ_TRY(chain_launch(c, din, cons, dout, outn, &ts));
        HZ_TRY(st.finish());
        c->ts = ts;
    }
    if (n_consumed) *n_consumed = cons;
    if (n_out) *n_out = outn;
    return HZSDR_OK;
}

int hzsdr_chain_run(hzsdr_chain *c, const void *in, size_t n_in, void *out, size_t out_cap,
                    size_t *n_consumed, size_t *n_out) {
    const void *ins[1] = {in};
    void *outs[1] = {out};
    return chain_run_impl(c, ins, outs, 1, n_in, out_cap, n_consumed, n_out, false, nullptr);
}

int hzsdr_chain_run_after(hzsdr_chain *c, const void *in, size_t n_in, void *out, size_t out_cap,
                          size_t *n_consumed, size_t *n_out, void *ready_event) {
    const void *ins[1] = {in};
    void *outs[1] = {out};
    return chain_run_impl(c, ins, outs, 1, n_in, out_cap, n_consumed, n_out, true, (hipEvent_t)ready_event);
}

int hzsdr_chain_run_batch(hzsdr_chain *c, const void *const *ins, void *const *outs, size_t n_buffers, size_t n_in,
                          size_t out_cap, size_t *n_consumed, size_t *n_out) {
    return chain_run_impl(c, ins, outs, n_buffers, n_in, out_cap, n_consumed, n_out, false, nullptr);
}

int hzsdr_chain_run_batch_after(hzsdr_chain *c, const void *const *ins, void *const *outs, size_t n_buffers, size_t n_in,
                                size_t out_cap, size_t *n_consumed, size_t *n_out, void *ready_event) {
    return chain_run_impl(c, ins, outs, n_buffers, n_in, out_cap, n_consumed, n_out, true, (hipEvent_t)ready_event);
}

int hzsdr_chain_reset(hzsdr_chain *c) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = c->ctx;
    HZ_TRY(hz::enter(ctx));
    c->ts = 0.0;
    c->rh_valid = false;
    if (c->term == TERM_FIR) {
        const size_t hb = (size_t)(c->off ? c->off : 1) * 8;
        HZ_TRY(hz::pipeline_drain(c));
        for (int k = 0; k < hzsdr_chain::kHist; k++) HZ_HIP(ctx, hipMemsetAsync(c->hist[k], 0, hb, ctx->stream));
    }
    return HZSDR_OK;
}

int hzsdr_chain_set_time(hzsdr_chain *c, double ts) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    if (!(ts >= 0.0) || ts > 6.283185307179586476925286766559)  // the closure's clock lives in [0, 2*pi]
        return hz::fail(c->ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: clock outside [0, 2*pi]");
    HZ_TRY(hz::enter(c->ctx));
    HZ_TRY(hz::pipeline_drain(c));
    c->ts = ts;
    c->rh_valid = false;  // the clock no longer continues the previous call's run
    if (c->term == TERM_FIR) {
        HZ_TRY(hz::prepare_late_filters(c, ts));
        HZ_TRY(hz::prepare_mm_tables(c, ts));
    }
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
    for (hipStream_t s : {c->pstream[0], c->pstream[1]})
        if (s) {
            (void)hipStreamSynchronize(s);
            auto &ss = c->ctx->side_streams;
            ss.erase(std::remove(ss.begin(), ss.end(), s), ss.end());
            (void)hipStreamDestroy(s);
        }
    for (int k = 0; k < 4; k++) {
        if (c->ev_done[k]) (void)hipEventDestroy(c->ev_done[k]);
        if (c->ev_hist[k]) (void)hipEventDestroy(c->ev_hist[k]);
    }
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->filt) (void)hipFree(c->filt);
    if (c->hfreq) (void)hipFree(c->hfreq);
    if (c->hfreq_late) (void)hipFree(c->hfreq_late);
    for (int k = 0; k < hzsdr_chain::kHist; k++)
        if (c->hist[k]) (void)hipFree(c->hist[k]);
    for (auto &kv : c->late_cache) (void)hipFree(kv.second);
    for (auto &kv : c->mm_cache) (void)hipFree(kv.second);
    if (c->taps_dev) (void)hipFree(c->taps_dev);
    for (int k = 0; k < hzsdr_chain::kHist; k++)
        if (c->rhist[k]) (void)hipFree(c->rhist[k]);
    delete c;
    return HZSDR_OK;
}

int hzsdr_chain_last_fir_path(const hzsdr_chain *c, int *path) {
    if (!c || !path) return HZSDR_ERR_INVALID_ARGUMENT;
    *path = c->last_path;
    return HZSDR_OK;
}

int hzsdr_chain_last_fir_kernel(const hzsdr_chain *c, int *kernel) {
    if (!c || !kernel) return HZSDR_ERR_INVALID_ARGUMENT;
    *kernel = c->last_path == HZSDR_FIR_PATH_MATRIX ? (c->mm_ver == 2 ? HZSDR_FIR_KERNEL_MATRIX_PASSES : HZSDR_FIR_KERNEL_MATRIX_CHUNKS)
                                                    : c->last_path;
    return HZSDR_OK;
}

int hzsdr_chain_shift_ulp1(hzsdr_chain *c, int on) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    c->shift_ulp1 = on != 0;
    return HZSDR_OK;
}

int hzsdr_chain_mix_in_order(hzsdr_chain *c, int in_order) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    c->mix_in_order = in_order != 0;
    return HZSDR_OK;
}

}  // extern "C"

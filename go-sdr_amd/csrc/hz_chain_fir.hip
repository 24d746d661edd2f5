// hz_chain_fir.hip -- the FIR-decimate terminal of a chain (north star: overlap-save FIR + decimate): the filter's
// spectra (one per clock step for the late mixer), the int8 matrix form's digit tables and planners
// (hz_firmm*.h), the lists of reference-order blocks, and the launches of one call.
#include "hz_chain_host.h"
#include "hz_firmm.h"
#include "hz_firmm2.h"

namespace hz {

// In-place radix-2 transform in float64 (host): the filter spectra are computed once per
// chain (and once per clock binade for the late mixer), so they are formed in double
// precision and rounded to float32 ONCE -- a float32 transform of the taps would put its
// own ~1e-7 relative error, the same in every block, into every output.
static void host_fft(std::vector<double> &re, std::vector<double> &im, size_t off, size_t n) {
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            std::swap(re[off + i], re[off + j]);
            std::swap(im[off + i], im[off + j]);
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len / 2;
        for (size_t k = 0; k < half; k++) {
            const double a = -2.0 * M_PI * (double)k / (double)len, wr = cos(a), wi = sin(a);
            for (size_t i = k; i < n; i += len) {
                const size_t u = off + i, v = off + i + half;
                const double tr = re[v] * wr - im[v] * wi, ti = re[v] * wi + im[v] * wr;
                re[v] = re[u] - tr;
                im[v] = im[u] - ti;
                re[u] += tr;
                im[u] += ti;
            }
        }
    }
}

// The taps' spectrum in the layout the analysis kernel multiplies by: H[k] = FFT_N(h)[k] / N,
// or, for the polyphase form, G[r][k] = FFT_M(g_r)[k] / M with g_r[j] = h[D j - r] (see
// fold_poly).  `taps`: ntaps complex values (real, imaginary) in host memory, float64;
// `dst`: N complex64 of device memory.  Uploaded through the context's stream; returns
// after the copy has completed (the staging vector is local).
static int filter_spectrum(hzsdr_chain *c, const double *taps, void *dst, double extra_scale = 1.0) {
    hzsdr_ctx *ctx = c->ctx;
    const unsigned nfft = c->nfft;
    std::vector<double> re(nfft, 0.0), im(nfft, 0.0);
    size_t len = nfft, batch = 1;
    if (c->poly) {
        const unsigned F = c->factor, M = nfft / F;
        for (unsigned r = 0; r < F; r++)
            for (unsigned j = 0; j < M; j++) {
                const long idx = (long)F * j - (long)r;
                if (idx < 0 || (size_t)idx >= c->ntaps) continue;
                re[(size_t)r * M + j] = taps[2 * idx];
                im[(size_t)r * M + j] = taps[2 * idx + 1];
            }
        len = M;
        batch = F;
    } else {
        for (size_t k = 0; k < c->ntaps; k++) {
            re[k] = taps[2 * k];
            im[k] = taps[2 * k + 1];
        }
    }
    for (size_t t = 0; t < batch; t++) host_fft(re, im, t * len, len);
    std::vector<float> h(2 * (size_t)nfft);
    const double scale = extra_scale / (double)len;
    for (size_t i = 0; i < nfft; i++) {
        h[2 * i] = (float)(re[i] * scale);
        h[2 * i + 1] = (float)(im[i] * scale);
    }
    HZ_HIP(ctx, hipMemcpyAsync(dst, h.data(), (size_t)nfft * 8, hipMemcpyHostToDevice, ctx->stream));
    HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HZSDR_OK;
}

// LATE blocks of a u8 source feed the transform b - 127.5 (exact in float32) instead of the
// converter's (b - 127.5) / 127.5: the division's scale moves into the late filters' spectra
// (formed in float64), three packed instructions per sample instead of five in the analysis
// kernel.  The converter's per-sample rounding (half an ulp, which the reference-order path
// and the oracle do perform) is thereby skipped: ~3e-8 relative, inside the FIR's error bound.
static double late_scale(const hzsdr_chain *c) { return c->src_fmt == HZSDR_FMT_U8 ? 1.0 / 127.5 : 1.0; }

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
    std::vector<double> mod(2 * c->ntaps);
    for (size_t k = 0; k < c->ntaps; k++) {
        const double ph = -omega * ((double)k * step);
        const double cr = cos(ph), ci = sin(ph);
        const double hr = c->taps_host[2 * k], hi = c->taps_host[2 * k + 1];
        mod[2 * k] = hr * cr - hi * ci;
        mod[2 * k + 1] = hr * ci + hi * cr;
    }
    void *h = nullptr;
    HZ_HIP(ctx, hipMalloc(&h, (size_t)c->nfft * 8));
    int rc = filter_spectrum(c, mod.data(), h, late_scale(c));
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
int prepare_late_filters(hzsdr_chain *c, double ts0) {
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

// ---- int8 matrix form (hz_firmm.h) -------------------------------------------------------------

static double mm_scale(const hzsdr_chain *c) { return c->src_fmt == HZSDR_FMT_U8 ? 1.0 / 127.5 : 1.0 / 128.0; }

// Byte sources with a decimation of 8 or 16, up to the tap count at which the direct form's work
// (proportional to the taps) passes the transforms' (tools/firmm_probe.py, 2^24 samples, us per
// call, matrix / transform: D = 8: 64 taps 33 / 152, 256: 35 / 73, 512: 41 / 50, 1024: 48 / 53,
// 1536: 70 / 70, 2048: 87 / 73; D = 16 (chunks of 1024 outputs): 256: 24 / 70, 512: 28 / 78, 1024: 36 / 48,
// 1536: 46 / 67, 2047: 54 / 69, 3000: 85 / 82; D = 32: 256: 31 / 71, 1024: 41 / 80, 4096: 99 / 151; D = 64: 1024:
// 39 / 79, 4096: 77 / 152; the D = 8 / 16 transform figures from before N_fft started at 256 D).
// hzsdr_chain_fir_options(HZSDR_FIR_IMPL_TRANSFORMS) keeps a chain on the transform kernels (A/B measurements, tests).
static bool mm_eligible(const hzsdr_chain *c) {
    if (c->fir_impl == HZSDR_FIR_IMPL_TRANSFORMS || (c->fir_impl == HZSDR_FIR_IMPL_AUTO && diag_env().fir_fft)) return false;
    if (c->src_fmt != HZSDR_FMT_U8 && c->src_fmt != HZSDR_FMT_I8) return false;
    if (!mm::factor_ok(c->factor)) return false;
    return c->ntaps >= 16 && c->ntaps <= (c->factor == 8 ? 1536u : c->factor <= 24 ? 2560u : 4096u);
}

// hz_firmm2.h (the persistent-pass form): D = 8, and a pass image / table that fit its fixed register counts.
// hzsdr_chain_fir_options(HZSDR_FIR_IMPL_MATRIX_CHUNKS) keeps the first form (A/B measurements).
static bool mm2_eligible(const hzsdr_chain *c) {
    if (c->fir_impl == HZSDR_FIR_IMPL_MATRIX_CHUNKS || (c->fir_impl == HZSDR_FIR_IMPL_AUTO && diag_env().mm_v1)) return false;
    if (!mm2::factor_ok(c->factor)) return false;
    // (the kernel's int32 sum of the two top digit planes must hold for every input: hz_firmm_plan.h)
    if (!mm::int32_combine_ok(c->taps_host.data(), c->ntaps, mm_scale(c), mm::digit_shift(c->taps_host.data(), c->ntaps, mm_scale(c)))) return false;
    const mm2::Geom g = mm2::make_geom((int)c->ntaps, (int)c->factor, c->off, 0);
    const int D = (int)c->factor;
    return mm2::image_bytes(D, g.ks) <= (size_t)mm2::kU * 64 * 16 && mm2::table_bytes(g.ne) <= (size_t)4 * mm2::kThreads * 16 &&
           mm2::lds_bytes(D, g.ks, g.ne, g.ntaps) <= 160 * 1024 && g.ntaps + D * (mm2::kFixOut - 1) <= 5 * 256;
}

static void mm_geometry(hzsdr_chain *c) {
    const int D = (int)c->factor;
    c->mm_ver = mm2_eligible(c) ? 2 : 1;
    // q = round(h' 2^S) with |q| <= 2^30 (hz_firmm_plan.h: digit_shift)
    const int S = mm::digit_shift(c->taps_host.data(), c->ntaps, mm_scale(c));
    if (c->mm_ver == 2) {
        const mm2::Geom g2 = mm2::make_geom((int)c->ntaps, D, c->off, S);
        mm::Geom &g = c->mmg;
        g.ntaps = g2.ntaps, g.w0 = g2.w0, g.ks = g2.ks, g.ne = g2.ne, g.e0 = g2.e0, g.shift = g2.shift, g.off = g2.off;
    } else {
        c->mmg = mm::make_geom((int)c->ntaps, D, c->off, S);
    }
}

// The digit table of taps[k] * exp(-i omega k step) * scale (hz_firmm.h: F[digit][E][part][16],
// then the constant term).  Like late_filter_for: a miss builds, uploads and WAITS unless
// `make` is false.
static int mm_table_for(hzsdr_chain *c, double step, double omega, void **dev, bool make) {
    hzsdr_ctx *ctx = c->ctx;
    *dev = nullptr;
    uint64_t key;
    memcpy(&key, &step, 8);
    auto it = c->mm_cache.find(key);
    if (it != c->mm_cache.end()) {
        *dev = it->second;
        return HZSDR_OK;
    }
    if (!make || c->mm_cache.size() >= kLateCacheMax) return HZSDR_OK;
    // (the table's contents: hz_firmm_plan.h, HIP-free and fuzzed on the host)
    const std::vector<uint8_t> tab = mm::digit_table(c->mmg, (int)c->factor, c->taps_host.data(), mm_scale(c), step, omega,
                                                     c->src_fmt == HZSDR_FMT_U8, c->mm_ver == 2);
    void *d = nullptr;
    HZ_HIP(ctx, hipMalloc(&d, tab.size()));
    hipError_t e = hipMemcpyAsync(d, tab.data(), tab.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(d);
        HZ_HIP(ctx, e);
    }
    c->mm_cache[key] = d;
    *dev = d;
    return HZSDR_OK;
}

// Tables for every long run the clock can produce from ts0 on (see prepare_late_filters).
int prepare_mm_tables(hzsdr_chain *c, double ts0) {
    if (!c->mm_ok) return HZSDR_OK;
    void *dev;
    if (!c->has_shift) return mm_table_for(c, 0.0, 0.0, &dev, true);
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
            if (segs[q].count < 8 * (uint64_t)c->ntaps) continue;
            HZ_TRY(mm_table_for(c, segs[q].step, omega, &dev, true));
        }
    }
    return HZSDR_OK;
}

// Splits the outputs of one call between the matrix path (per clock run) and the fix-up
// workgroups.  false: the call stays on the transform kernels.
static bool mm_plan(hzsdr_chain *c, const EwProgram &P, size_t n, const void *in, const void *out, mm::Runs *R,
                    mm::Fix *F) {
    memset(R, 0, sizeof *R);
    memset(F, 0, sizeof *F);
    if (!c->mm_ok || c->mix_in_order || P.segs.big_n != 0 || (((uintptr_t)in | (uintptr_t)out) & 15) != 0) return false;
    for (int i = 0; i < P.n; i++)
        if (P.op[i].kind == EW_SHIFT && !(fabs(P.op[i].tau_shift) * 6.2832 < 1073741824.0)) return false;
    const int nr = c->has_shift ? P.segs.n : 1;
    if (nr < 1 || nr > mm::kMaxRuns) return false;
    // does run 0 continue the run the previous call ended in (same step, no reset in between)?  Then
    // the clock is exactly linear across the call boundary and the first windows may reach back
    // into the raw history instead of going to the fix-up tasks.
    const bool cont = c->rh_valid && (!c->has_shift || (P.segs.step[0] == c->rh_step && c->rh_len >= c->ntaps && c->rh_next == P.segs.t0[0]));
    const void *tabs[mm::kMaxRuns];
    for (int r = 0; r < nr; r++) {
        void *dev = nullptr;
        (void)mm_table_for(c, c->has_shift ? P.segs.step[r] : 0.0, 0.0, &dev, false);
        tabs[r] = dev;
    }
    const uint64_t zero_first = 0;
    // the planner itself is host-only code (hz_firmm_plan.h: sanitizer-built and fuzzed in tests/host/)
    const mm::ChunkPlanIn pin{n, c->factor, (int)c->ntaps, cont, nr, c->has_shift ? P.segs.first : &zero_first, tabs};
    uint64_t fix_total = 0;
    const bool ok = mm::plan_chunks(pin, R, F, &fix_total);
    const uint64_t n_chunks = (uint64_t)R->n_wg;
    if (c->debug_mm) {
        fprintf(stderr, "hzsdr mm: %d runs, cont %d, %llu chunks, %d fix intervals (%d tasks, %llu outputs)\n", nr, R->cont,
                (unsigned long long)n_chunks, F->n, F->n_wg, (unsigned long long)fix_total);
        for (int r = 0; r < nr; r++)
            fprintf(stderr, "   run %d first %llu valid [%u, %u) first chunk %d table %p\n", r,
                    (unsigned long long)(c->has_shift ? P.segs.first[r] : 0), R->m_lo[r], R->m_hi[r], R->wg_first[r], R->tab[r]);
        for (int k = 0; k < F->n; k++) fprintf(stderr, "   fix [%u, %u) first task %d\n", F->m_a[k], F->m_b[k], F->wg_first[k]);
    }
    return ok;
}

static int mm_launch(hzsdr_chain *c, const void *in, void *out, size_t n, const EwProgram &P, const mm::Runs &R,
                     const mm::Fix &F) {
    HZ_TRY(pipeline_drain(c));
    return mm::launch_fir(c->ctx->stream, c->src_fmt, c->factor, in, (float2 *)out, (const float2 *)c->hist[c->hist_cur],
                          (float2 *)c->hist[c->hist_next()], (const uint8_t *)c->rhist[c->hist_cur],
                          (uint8_t *)c->rhist[c->hist_next()], (const float2 *)c->taps_dev, n, c->mmg, P, R, F);
}

// What a chain carries from call to call about the RAW history (the last `off` bytes pairs of the previous call): is it
// valid, how many of its last samples lie in the clock run that call ended in, and that call's last clock value.
struct RawCarry {
    bool valid;
    uint64_t len;
    double ts;
};
// ... after a call of n samples on the matrix path whose clock runs were (first, nr) and whose run 0 continued the
// carried run (`cont`)
static void carry_after(const hzsdr_chain *c, RawCarry *k, size_t n, const uint64_t *first, int nr, bool cont, double ts_after) {
    const bool was = k->valid;
    k->valid = n >= c->off || was;
    if (c->has_shift) {
        const int last = nr - 1;
        const uint64_t len = n - first[last];
        k->len = (last == 0 && was && cont) ? k->len + len : len;  // (the samples of the stream's current run so far: it may have begun calls ago)
    }
    k->ts = ts_after;
}

// hz_firmm2.h: the plan of ONE call of n samples whose clock runs are (first, t0, step)[nr] for the persistent-pass
// kernel.  Every clock run with a table gets the
// outputs whose whole window lies in it (tile-aligned) and the passes of the call's 512-output grid that hold
// them -- a pass that straddles a boundary is multiplied once per run, each time with that run's table and
// valid range (a second short pass costs one wave 6 us; the fix-up tasks it replaces cost ~10 us EACH).  What
// no run holds -- windows that cross a boundary, the stream's start, runs without a table -- are fix-up tasks
// of 16 outputs.  false: the call stays on the transform kernels.
static bool mm2_plan_piece(hzsdr_chain *c, const EwProgram &P, const RawCarry &k, size_t n, const uint64_t *first, const double *t0,
                           const double *step, int nr, mm2::Plan *L, mm2::Fix *F, uint64_t *fix_total) {
    if (nr < 1 || nr > kNcoMaxSegs) return false;
    // the planner itself is host-only code (hz_firmm2_plan.h: sanitizer-built and fuzzed in tests/host/)
    mm2::PlanIn pin{};
    pin.n_in = n, pin.D = c->factor, pin.ntaps = (int)c->ntaps, pin.has_shift = c->has_shift;
    // Does run 0 continue the run the previous call ended in?  The previous call's last sample had the clock k.ts
    // (committed by the caller when that call returned), this call's first sample has t0[0]: two consecutive points of
    // one line (hz_firmm2_plan.h: run_line) are in one binade and exactly one step apart.  (Not "equal steps and the
    // clock the last run predicted": a call that ends ONE sample into a run has no step to compare.)  The raw history
    // then holds min(len, off) samples of the run in front of the call.
    pin.back = 0;
    if (k.valid) {
        if (!c->has_shift) pin.back = c->off;
        else if (mm2::continues(k.ts, t0[0], step[0])) pin.back = k.len < c->off ? k.len : c->off;
    }
    pin.n_ops = P.n;
    pin.max_grid = c->ctx->num_cus;
    pin.shift_op = -1;
    int n_shift = 0;
    for (int i = 0; i < P.n; i++)
        if (P.op[i].kind == EW_SHIFT) {
            pin.shift_op = i;
            n_shift++;
        }
    if (n_shift != 1) pin.shift_op = -1;
    else pin.tau = P.op[pin.shift_op].tau_shift;
    const void *tabs[kNcoMaxSegs];
    for (int r = 0; r < nr; r++) {
        void *dev = nullptr;
        (void)mm_table_for(c, c->has_shift ? step[r] : 0.0, 0.0, &dev, false);
        tabs[r] = dev;
    }
    mm2::ClockRuns cr{nr, first, t0, step, tabs};
    return mm2::plan_call(pin, cr, L, F, fix_total);
}

static bool mm2_plan(hzsdr_chain *c, const EwProgram &P, size_t n, const void *in, const void *out, mm2::Plan *L,
                     mm2::Fix *F, const CallBatch *cb = nullptr) {
    memset(L, 0, sizeof *L);
    memset(F, 0, sizeof *F);
    if (!c->mm_ok || c->mix_in_order || P.segs.big_n != 0 || (((uintptr_t)in | (uintptr_t)out) & 15) != 0) return false;
    for (int i = 0; i < P.n; i++)
        if (P.op[i].kind == EW_SHIFT && !(fabs(P.op[i].tau_shift) * 6.2832 < 1073741824.0)) return false;
    const uint64_t zero_first = 0;
    const double zero = 0.0;
    const RawCarry k0{c->rh_valid, c->rh_len, c->ts};
    const bool batched = cb && cb->nbuf > 1;
    if (batched) {  // (a call over several buffers: every one aligned, whole passes per buffer, the history in the last)
        if (!mm2::batch_ok(cb->n_each, c->factor, cb->nbuf) || cb->n_each < c->mmg.off) return false;
        for (size_t j = 0; j < cb->nbuf; j++)
            if ((((uintptr_t)cb->ins[j] | (uintptr_t)cb->outs[j]) & 15) != 0) return false;
    }
    const int nr = c->has_shift ? P.segs.n : 1;
    uint64_t fix_total = 0;
    bool ok = mm2_plan_piece(c, P, k0, n, c->has_shift ? P.segs.first : &zero_first, c->has_shift ? P.segs.t0 : &zero,
                             c->has_shift ? P.segs.step : &zero, nr, L, F, &fix_total);
    if (ok && batched) {
        // hzsdr_chain_run_batch promises the results of n_buffers hzsdr_chain_run calls, bit for bit.  What the
        // matrix path computes for an output does not depend on how the stream is cut (hz_firmm2_plan.h: run_line) --
        // but WHETHER a call takes the matrix path does (a short call that is mostly clock boundaries keeps the
        // transform kernels, other arithmetic inside the same error bound).  So the one-launch form is taken only when
        // every buffer by itself would have taken the matrix path.  What a buffer's own plan holds is what the call's
        // plan holds of it (the same outputs are fix-up outputs either way; an interval that spans a buffer boundary
        // becomes two), so the call's counts bound every buffer's: if even ALL the call's fix-up outputs in one buffer
        // would pass that buffer's checks, every buffer passes -- the common case, decided here at no cost.
        const uint64_t out_each = cb->n_each / c->factor;
        const int grid_each = (int)std::min<uint64_t>((uint64_t)c->ctx->num_cus, out_each / (uint64_t)mm2::pass_out((int)c->factor));
        const bool surely = fix_total * 8 <= out_each && F->n + 1 <= mm2::kMaxFix && F->n_task + 2 <= 4 * grid_each;
        if (!surely) {
            // otherwise the single calls are planned here, on the host (clock runs and planner, a few microseconds per
            // buffer, no device work), and thrown away
            RawCarry k = k0;
            for (size_t j = 0; j < cb->nbuf && ok; j++) {
                uint64_t first[kNcoMaxSegs];
                double t0[kNcoMaxSegs], step[kNcoMaxSegs], ts_end = k.ts;
                int nrj = 1;
                if (c->has_shift) {
                    hzsdr_nco_segment sg[kNcoMaxSegs];
                    size_t need = 0;
                    if (hzsdr_nco_segments(c->sample_rate, k.ts, cb->n_each, sg, kNcoMaxSegs, &need, &ts_end) != HZSDR_OK || need > (size_t)kNcoMaxSegs) {
                        ok = false;
                        break;
                    }
                    nrj = (int)need;
                    for (int r = 0; r < nrj; r++) first[r] = sg[r].first, t0[r] = sg[r].t0, step[r] = sg[r].step;
                } else {
                    first[0] = 0, t0[0] = 0.0, step[0] = 0.0;
                }
                mm2::Plan Lj;
                mm2::Fix Fj;
                if (!mm2_plan_piece(c, P, k, cb->n_each, first, t0, step, nrj, &Lj, &Fj, nullptr)) ok = false;
                else carry_after(c, &k, cb->n_each, first, nrj, Lj.cont != 0, ts_end);
            }
        }
    }
    if (c->debug_mm) {
        fprintf(stderr, "hzsdr mm2: %s: %d of %d runs on the matrix path, cont %d, %d passes, %d fix intervals (%d tasks, %llu outputs)\n",
                ok ? "matrix path" : "transform kernels", L->n, nr, L->cont, L->n_pass, F->n, F->n_task, (unsigned long long)fix_total);
        for (int r = 0; r < L->n; r++)
            fprintf(stderr, "   run first %llu valid [%u, %u) passes [%d, %d) table %p\n", (unsigned long long)L->run[r].first, L->run[r].m_lo,
                    L->run[r].m_hi, L->run[r].pass_first, L->run[r].pass_end, L->run[r].tab);
        for (int k = 0; k < F->n; k++) fprintf(stderr, "   fix [%u, %u) first task %d\n", F->m_a[k], F->m_b[k], F->task_first[k]);
    }
    return ok;
}

// ---- hzsdr_chain_pipeline ----------------------------------------------------------------------------------------
// A stream runs its launches one behind the other: the next call's workgroups wait for the LAST workgroup of this
// one (they end over ~3 us) and then pay the kernel's head in full -- two independent chains on two streams take
// 32.7 us per call where one takes 37.5 (tools/two_streams.py).  Consecutive calls of ONE chain depend on each other
// through the history alone, and that is a function of the call's INPUT: a pipelined chain forms it with a small
// kernel of its own (mm2::launch_history) and alternates its calls between two streams A, B of its own:
//     call k:    history kernel k (writes the history call k+1 reads) on stream (k+1) mod 2,
//                the call's kernel on stream k mod 2
// so stream order alone puts kernel k+1 behind history kernel k, and history kernel k behind kernel k-3, the last
// reader of the buffer it writes (a ring of four histories).  BOTH kernels of a call are joined to the context's
// stream as they are launched (an event each): whatever the caller enqueues there afterwards -- a consumer of the
// output, a producer that refills the input -- is ordered behind them, as behind any other call.
//
// What cannot come from the context's stream is the call's START: that stream has just been made to wait for the
// previous call, so a launch ordered behind it cannot overlap that call.  Round 4 simply did not order the launch
// behind anything and said so in the header; a caller that filled its input on the context's stream raced.  Now the
// overlap is taken ONLY by a call that says what its buffers wait for (hzsdr_chain_run_after: an event, or "nothing");
// hzsdr_chain_run itself is an ordinary call on the context's stream whatever the mode.  What the library can see
// itself it orders itself: a call whose buffers overlap those of the two calls before it starts over behind them.
int pipeline_streams(hzsdr_chain *c) {
    hzsdr_ctx *ctx = c->ctx;
    if (c->ev_in) return HZSDR_OK;  // (created last: everything else exists)
    for (hipStream_t *s : {&c->pstream[0], &c->pstream[1]}) {
        if (*s) continue;  // (left by an earlier attempt that failed further down)
        HZ_HIP(ctx, hipStreamCreateWithFlags(s, hipStreamNonBlocking));
        ctx->side_streams.push_back(*s);
    }
    for (int k = 0; k < 4; k++) {
        if (!c->ev_done[k]) HZ_HIP(ctx, hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming));
        if (!c->ev_hist[k]) HZ_HIP(ctx, hipEventCreateWithFlags(&c->ev_hist[k], hipEventDisableTiming));
    }
    HZ_HIP(ctx, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    return HZSDR_OK;
}

int pipeline_drain(hzsdr_chain *c) {
    // (every overlapped call was joined to the context's stream when it was launched: work enqueued there now is
    // behind all of them; the chain's streams are re-seeded from it by the next overlapped call)
    if (c->ready) {  // a call that named an event and runs on the context's stream after all: that stream waits for it
        HZ_HIP(c->ctx, hipStreamWaitEvent(c->ctx->stream, c->ready, 0));
        c->ready = nullptr;
    }
    c->pcall = 0;
    c->pbufs[0].nb = c->pbufs[1].nb = 0;
    return HZSDR_OK;
}

static bool spans_overlap(const hzsdr_chain::Span &x, const hzsdr_chain::Span &y) {
    return x.n && y.n && x.p < y.p + y.n && y.p < x.p + x.n;
}

int pipeline_begin(hzsdr_chain *c, const CallBatch &cb, int fmt_size, hipStream_t *a, hipStream_t *b) {
    hzsdr_ctx *ctx = c->ctx;
    HZ_TRY(pipeline_streams(c));
    hzsdr_chain::CallBufs now;
    now.nb = (int)cb.nbuf;
    for (size_t j = 0; j < cb.nbuf; j++) {
        now.in[j] = {(const char *)cb.ins[j], cb.n_each * (size_t)fmt_size};
        now.out[j] = {(const char *)cb.outs[j], cb.out_each * 8};
    }
    // The call before this one runs on the other stream, unordered against this call's kernel: what this call
    // writes must not be what that one reads or writes, what this call reads not what that one writes.  The call
    // before that shares this call's stream, but its history kernel (on the other one) read its input.
    bool clash = false;
    const hzsdr_chain::CallBufs &p1 = c->pbufs[0], &p2 = c->pbufs[1];
    for (int i = 0; i < now.nb && !clash; i++) {
        for (int j = 0; j < p1.nb && !clash; j++)
            clash = spans_overlap(now.out[i], p1.out[j]) || spans_overlap(now.out[i], p1.in[j]) || spans_overlap(now.in[i], p1.out[j]);
        for (int j = 0; j < p2.nb && !clash; j++) clash = spans_overlap(now.out[i], p2.in[j]) || spans_overlap(now.in[i], p2.out[j]);
    }
    if (clash) HZ_TRY(pipeline_drain(c));
    const uint64_t k = c->pcall;
    *a = c->pstream[k & 1];
    if (b) *b = c->pstream[(k + 1) & 1];
    if (k == 0) {  // starting over: behind everything the context's stream holds (every earlier call of the chain is)
        HZ_HIP(ctx, hipEventRecord(c->ev_in, ctx->stream));
        HZ_HIP(ctx, hipStreamWaitEvent(c->pstream[0], c->ev_in, 0));
        HZ_HIP(ctx, hipStreamWaitEvent(c->pstream[1], c->ev_in, 0));
    }
    if (c->ready) {  // what the caller said the call's buffers wait for
        HZ_HIP(ctx, hipStreamWaitEvent(*a, c->ready, 0));
        if (b) HZ_HIP(ctx, hipStreamWaitEvent(*b, c->ready, 0));
    }
    // The two streams never wait for each other otherwise, so nothing would bound how far one may fall behind: call
    // k - 3's kernel ran on the OTHER stream, and a caller that rotates three buffer sets writes the set it wrote.
    // This call's stream waits for that kernel (long finished in any normal run: the wait costs the GPU nothing);
    // with the check above -- calls k - 1 and k - 2 -- every earlier call is then ordered in front of this one.
    if (k >= 3) HZ_HIP(ctx, hipStreamWaitEvent(*a, c->ev_done[(k - 3) & 3], 0));
    c->pbufs[1] = c->pbufs[0];
    c->pbufs[0] = now;
    return HZSDR_OK;
}

int pipeline_join(hzsdr_chain *c, hipStream_t a, hipStream_t b) {
    hzsdr_ctx *ctx = c->ctx;
    const uint64_t k = c->pcall;
    HZ_HIP(ctx, hipEventRecord(c->ev_done[k & 3], a));
    HZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_done[k & 3], 0));
    if (b) {
        HZ_HIP(ctx, hipEventRecord(c->ev_hist[k & 3], b));
        HZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_hist[k & 3], 0));
    }
    c->pcall = k + 1;
    return HZSDR_OK;
}

static int mm2_launch_pipelined(hzsdr_chain *c, const void *in, void *out, size_t n, const EwProgram &P, const mm2::Plan &L,
                                const mm2::Fix &F, const mm2::Geom &g2, const mm2::Batch &B, const CallBatch &cb) {
    hzsdr_ctx *ctx = c->ctx;
    hipStream_t a, b;
    HZ_TRY(pipeline_begin(c, cb, format_size(c->src_fmt), &a, &b));
    // (the history is the call's last `off` samples: of the last buffer, indexed like the whole call)
    const void *last = (const void *)(uintptr_t)B.vin[B.nbuf - 1];
    // (the call's kernel first: a sequence's first call starts on an idle GPU, and the history kernel is needed by the
    // NEXT call only)
    int rc = mm2::launch_fir(a, ctx->num_cus, c->src_fmt, c->factor, in, (float2 *)out, (const float2 *)c->hist[c->hist_cur], nullptr,
                             (const uint8_t *)c->rhist[c->hist_cur], nullptr, (const float2 *)c->taps_dev, n, g2, L, P, F, B, c->fir_loop_form);
    if (rc == HZSDR_OK)
        rc = mm2::launch_history(b, c->src_fmt, last, (float2 *)c->hist[c->hist_next()], (uint8_t *)c->rhist[c->hist_next()], n, c->mmg.off, P);
    // (whatever happened, what was launched is joined; after a failure the chain starts over: the history ring's
    // place in the streams' order is no longer what the next call would assume)
    const int rj = pipeline_join(c, a, b);
    if (rc != HZSDR_OK || rj != HZSDR_OK) (void)pipeline_drain(c);
    return rc != HZSDR_OK ? rc : rj;
}

static int mm2_launch(hzsdr_chain *c, const void *in, void *out, size_t n, const EwProgram &P, const mm2::Plan &L,
                      const mm2::Fix &F, const CallBatch *cbp) {
    mm2::Geom g2{};
    g2.ntaps = c->mmg.ntaps, g2.w0 = c->mmg.w0, g2.ks = c->mmg.ks, g2.ne = c->mmg.ne, g2.e0 = c->mmg.e0, g2.shift = c->mmg.shift,
    g2.off = c->mmg.off;
    const void *in1[1] = {in};
    void *out1[1] = {out};
    const CallBatch one{in1, out1, 1, n, n / c->factor};
    const CallBatch &cb = cbp && cbp->nbuf > 1 ? *cbp : one;
    const mm2::Batch B = cb.nbuf > 1 ? mm2::make_batch(cb.ins, cb.outs, cb.nbuf, cb.n_each, c->factor) : mm2::one_buffer(in, out, n, c->factor);
    // (`relaxed`: device buffers whose readiness the caller stated -- hzsdr_chain_run_after in a DEVICE-space context, the ring)
    if (c->pipelined && c->relaxed && n >= c->mmg.off && c->mmg.off > 0)
        return mm2_launch_pipelined(c, in, out, n, P, L, F, g2, B, cb);
    HZ_TRY(pipeline_drain(c));
    return mm2::launch_fir(c->ctx->stream, c->ctx->num_cus, c->src_fmt, c->factor, in, (float2 *)out,
                           (const float2 *)c->hist[c->hist_cur], (float2 *)c->hist[c->hist_next()],
                           (const uint8_t *)c->rhist[c->hist_cur], (uint8_t *)c->rhist[c->hist_next()],
                           (const float2 *)c->taps_dev, n, g2, L, P, F, B, c->fir_loop_form);
}

// The modulated filter of every clock run long enough to hold a whole block (lookups only).
static int late_filters(hzsdr_chain *c, const EwProgram &P, size_t n, LateFilters *out, bool *any) {
    *any = false;
    memset(out, 0, sizeof *out);
    if (c->mix_in_order || P.segs.big_n != 0 || c->taps_host.empty()) return HZSDR_OK;
    if (!c->has_shift) {
        // no clock involved: Gain / Multiply commute with the filter everywhere, with the
        // taps as they are (the kernel sees an empty run table: run 0)
        out->h[0] = (const float2 *)(c->hfreq_late ? c->hfreq_late : c->hfreq);
        *any = true;
        return HZSDR_OK;
    }
    // sincos_late (the late mixer's Sincos) takes |tau * ts| < 2^30, ts <= 2 pi
    for (int i = 0; i < P.n; i++)
        if (P.op[i].kind == EW_SHIFT && !(fabs(P.op[i].tau_shift) * 6.2832 < 1073741824.0)) return HZSDR_OK;
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

// The blocks of this run that mix in reference order, ascending, for the analysis kernel's
// dispatch order (device side: late_block).  Block b spans [b*hop - off, b*hop - off + N) and
// is late iff that span lies in [0, n - off] AND inside one clock run that has a filter.
// Walks the runs, not the blocks; more than kMaxSlowBlocks of them -> empty list (stream order).
static void slow_blocks(const hzsdr_chain *c, const EwProgram &P, const LateFilters &late, size_t n,
                        size_t nblocks, SlowBlocks *out) {
    out->n = 0;
    if (diag_env().no_slow_first) return;  // measurement aid (tools/wrap_probe.py): stream order
    const int64_t N = c->nfft, hop = c->hop, off = c->off;
    std::vector<unsigned> v;
    auto add_range = [&](int64_t lo, int64_t hi) {  // blocks lo .. hi inclusive, clipped
        if (lo < 0) lo = 0;
        if (hi >= (int64_t)nblocks) hi = (int64_t)nblocks - 1;
        for (int64_t b = lo; b <= hi && v.size() <= (size_t)kMaxSlowBlocks; b++) v.push_back((unsigned)b);
    };
    // every block whose span contains a sample of [s_lo, s_hi]: b*hop - off <= s_hi and b*hop - off + N > s_lo
    auto touching = [&](int64_t s_lo, int64_t s_hi) {
        int64_t lo = s_lo + off - N;            // b*hop > lo
        lo = lo < 0 ? 0 : lo / hop + 1;
        add_range(lo, (s_hi + off) / hop);
    };
    // stream edges: spans that start before sample 0 (b*hop < off) or end after n - off (b*hop > n - N)
    if (off > 0) add_range(0, (off - 1) / hop);
    add_range((int64_t)n >= N ? ((int64_t)n - N) / hop + 1 : 0, (int64_t)nblocks - 1);
    const int nr = P.segs.n > 0 ? P.segs.n : 1;
    for (int r = 0; r < nr; r++) {
        const int64_t first = P.segs.n > 0 ? (int64_t)P.segs.first[r] : 0;
        const int64_t end = (P.segs.n > 0 && r + 1 < P.segs.n) ? (int64_t)P.segs.first[r + 1] : (int64_t)n;
        if (late.h[r] == nullptr) touching(first, end - 1);      // a run without a filter: every block in it
        else if (r > 0) touching(first - 1, first);              // a boundary: the blocks that straddle it
        if (v.size() > (size_t)kMaxSlowBlocks) return;
    }
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    if (v.size() > (size_t)kMaxSlowBlocks) return;
    out->n = (int)v.size();
    for (size_t i = 0; i < v.size(); i++) out->idx[i] = v[i];
    if (diag_env().debug_late) {
        int nh = 0;
        for (int r = 0; r < nr; r++) nh += late.h[r] != nullptr;
        fprintf(stderr, "hzsdr: %d runs, %d with a late filter, %d of %zu blocks listed as reference-order:", nr, nh, out->n, nblocks);
        for (int i = 0; i < out->n; i++) fprintf(stderr, " %u", out->idx[i]);
        fprintf(stderr, "\n");
        for (int r = 0; r < nr; r++)
            fprintf(stderr, "   run %d first %llu step %.17g filter %p\n", r, (unsigned long long)P.segs.first[r], P.segs.step[r], (const void *)late.h[r]);
    }
}

// workgroups of fir_decimate_kernel16: with the late mixer, the listed blocks first, then the
// rest in eight contiguous runs (one per XCD), padded to whole rounds of eight
static unsigned fir_grid(size_t nblocks, int xpb, bool late, int n_slow) {
    if (!late) return (unsigned)((nblocks + xpb - 1) / xpb);
    const size_t rest = nblocks - (size_t)n_slow;
    return (unsigned)(n_slow + 8 * ((rest + 7) / 8));
}

template <int FMT>
int fir_run(hzsdr_chain *c, const void *in, size_t n_cons, void *out, const EwProgram &P, const CallBatch *cb) {
    hzsdr_ctx *ctx = c->ctx;
    const bool batched = cb && cb->nbuf > 1;
    if (batched && !(c->mm_ok && c->mm_ver == 2)) return kBatchFallback;
    do {
        const size_t nblocks = (n_cons + c->hop - 1) / c->hop;
        const float2 *hist = (const float2 *)c->hist[c->hist_cur];
        float2 *nhist = (float2 *)c->hist[c->hist_next()];
        const unsigned D = c->factor;
        if (c->mm_ok) {
            // byte source, D = 8, 16, ...: the int8 matrix form (hz_firmm2.h / hz_firmm.h), one launch
            mm::Runs R;
            mm::Fix F;
            mm2::Plan L2;
            mm2::Fix F2;
            bool ran = false;
            int cont = 0;
            if (c->mm_ver == 2) {
                if (mm2_plan(c, P, n_cons, in, out, &L2, &F2, cb)) {
                    const int rc = mm2_launch(c, in, out, n_cons, P, L2, F2, cb);
                    if (rc != HZSDR_OK) {  // (nothing was committed: the histories are the previous call's)
                        c->rh_valid = false;
                        return rc;
                    }
                    ran = true;
                    cont = L2.cont;
                }
            } else if (mm_plan(c, P, n_cons, in, out, &R, &F)) {
                const int rc = mm_launch(c, in, out, n_cons, P, R, F);
                if (rc != HZSDR_OK) {
                    c->rh_valid = false;
                    return rc;
                }
                ran = true;
                cont = R.cont;
            }
            if (ran) {
                c->hist_cur = c->hist_next();
                c->last_path = HZSDR_FIR_PATH_MATRIX;
                // the raw history now ends in this call's last clock run
                RawCarry k{c->rh_valid, c->rh_len, c->ts};
                const uint64_t zero_first = 0;
                carry_after(c, &k, n_cons, c->has_shift ? P.segs.first : &zero_first, c->has_shift ? P.segs.n : 1, cont != 0, c->ts);
                c->rh_valid = k.valid, c->rh_len = k.len;
                if (c->has_shift) {
                    const int last = P.segs.n - 1;
                    const uint64_t len = n_cons - P.segs.first[last];
                    c->rh_step = P.segs.step[last];
                    // the clock the run assigns to the NEXT sample: a call that continues the run starts there
                    // (equal steps alone do not say so: with a sample rate whose 1/fs is a power of two every
                    // binade has the same step, and a 2*pi wrap on a call boundary would pass unnoticed)
                    c->rh_next = fma((double)len, P.segs.step[last], P.segs.t0[last]);
                }
                break;
            }
        }
        if (batched) return kBatchFallback;  // (the transform kernels take one buffer at a time)
        if (!fv::ok((int)c->nfft)) return HZSDR_ERR_INVALID_ARGUMENT;
        // (the transform kernels run on the context's stream.  Round 6 built the overlapped form of the two-kernel calls --
        // call k + 1's analysis beside call k's synthesis, first with the calls alternating between the chain's two
        // streams, then with every analysis on one stream and every synthesis on the other -- bit-identical and
        // SLOWER both ways: 57-58 against 50-52 us per 2^24 i16 samples at 1024 taps, 47-49 against 38-42 at 256
        // (profiles/r06_transform_pipeline_attempt.txt): the events a 45 us call then carries between its kernels cost
        // more than the 12 us synthesis can hide.  Taken out; tests/test_gpu_firmm.py keeps the equality test.)
        HZ_TRY(pipeline_drain(c));
        FvTabs tabs{}, tabs_m{};
        PolyTabs ptabs{};
        HZ_TRY(get_fv_tables(ctx, c->nfft, &tabs));
        LateFilters late{};
        SlowBlocks slow{};
// packed-math core: fold when D is a power of two <= 16 and N/D is itself a core size
#define HZ_FIR16_L(N, FOLD, LATE, SPEC)                                                                        \
    HZ_TRY(launch_fv(fir_decimate_kernel16<N, FMT, FOLD, LATE>, dim3(fir_grid(nblocks, fv::xpb(N), LATE, slow.n)), \
              dim3(fv::block(N)), fir_lds_bytes(N, FOLD), ctx->stream, in,                                      \
              (float2 *)out, hist, nhist, (const float2 *)c->hfreq, tabs, SPEC, nblocks, n_cons, c->hop, c->off, D, \
              P, late, ptabs, slow))
#define HZ_FIR16(N, FOLD) HZ_FIR16_L(N, FOLD, false, (float2 *)nullptr)
#define HZ_SYNTH16(N, F, LATE)                                                                                 \
    HZ_TRY(launch_fv(fir_synth_kernel16<N, F, LATE>,                                                           \
              dim3((unsigned)((nblocks + SynthGeom<N, F>::XPB - 1) / SynthGeom<N, F>::XPB)),                    \
              dim3(SynthGeom<N, F>::BS), (size_t)SynthGeom<N, F>::XPB * fv::lds_elems(N / F) * sizeof(cf),      \
              ctx->stream, (const float2 *)spec, (float2 *)out, tabs_m.bwd, nblocks, n_cons, c->hop, c->off, P, late))
#define HZ_FIR16_FOLD(N, F)                                                            \
    if (D == F) {                                                                      \
        if constexpr (fv::xpb(N) == 1 && fv::ok(N / F)) {                              \
            HZ_TRY(get_fv_tables(ctx, N / F, &tabs_m));                                \
            if constexpr (fold_poly(N, F)) HZ_TRY(get_fv_poly_tables(ctx, N, F, &ptabs)); \
            bool any_late = false;                                                     \
            HZ_TRY(late_filters(c, P, n_cons, &late, &any_late));                      \
            if (any_late) slow_blocks(c, P, late, n_cons, nblocks, &slow);             \
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
    if constexpr (fv::xpb(N) == 1) {                                                   \
        bool any_late = false;                                                         \
        HZ_TRY(late_filters(c, P, n_cons, &late, &any_late));                          \
        if (any_late) slow_blocks(c, P, late, n_cons, nblocks, &slow);                 \
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
        switch (c->nfft) {
        case 256: HZ_FIR16_N(256); break;
        case 512: HZ_FIR16_N(512); break;
        case 1024: HZ_FIR16_N(1024); break;
        case 2048: HZ_FIR16_N(2048); break;
        case 4096: HZ_FIR16_N(4096); break;
        case 8192: HZ_FIR16_N(8192); break;
        default: return HZSDR_ERR_INVALID_ARGUMENT;
        }
#undef HZ_FIR16_N
#undef HZ_FIR16_FULL
#undef HZ_FIR16_FOLD
#undef HZ_FIR16
#undef HZ_FIR16_L
#undef HZ_SYNTH16
        c->hist_cur = c->hist_next();  // the kernel wrote the next run's history into nhist
        c->rh_valid = false;  // (the transform kernels keep no raw history)
        c->last_path = HZSDR_FIR_PATH_TRANSFORM;
        break;
    } while (0);
    return HZSDR_OK;
}
template int fir_run<HZSDR_FMT_C64>(hzsdr_chain *, const void *, size_t, void *, const EwProgram &, const CallBatch *);
template int fir_run<HZSDR_FMT_U8>(hzsdr_chain *, const void *, size_t, void *, const EwProgram &, const CallBatch *);
template int fir_run<HZSDR_FMT_I8>(hzsdr_chain *, const void *, size_t, void *, const EwProgram &, const CallBatch *);
template int fir_run<HZSDR_FMT_I16>(hzsdr_chain *, const void *, size_t, void *, const EwProgram &, const CallBatch *);

}  // namespace hz

extern "C" {

int hzsdr_chain_fir_options(hzsdr_chain *c, int impl, unsigned nfft_min, int loop_form) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    if (c->term != TERM_NONE) return hz::fail(c->ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: fir options go in front of the terminal stage");
    if (impl < HZSDR_FIR_IMPL_AUTO || impl > HZSDR_FIR_IMPL_MATRIX_CHUNKS || (nfft_min && (nfft_min < 256 || nfft_min > 8192 || (nfft_min & (nfft_min - 1)))))
        return hz::fail(c->ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: fir options");
    c->fir_impl = impl;
    c->fir_nfft_min = nfft_min;
    c->fir_loop_form = loop_form;
    return HZSDR_OK;
}

int hzsdr_chain_pipeline(hzsdr_chain *c, int on) {
    if (!c) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(hz::enter(c->ctx));
    HZ_TRY(hz::pipeline_drain(c));
    c->pipelined = on != 0;
    return HZSDR_OK;
}

int hzsdr_chain_fir_decimate(hzsdr_chain *c, const float *taps, size_t n_taps, unsigned factor) {
    using namespace hz;
    HZ_TRY(chain_terminal_set(c));
    hzsdr_ctx *ctx = c->ctx;
    if (!taps || n_taps == 0 || factor == 0) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "chain: fir taps / factor");
    // N_fft = next power of two >= 4 * taps, in [N_min, 8192].  N_min is the smallest size at which the
    // fast forms of the kernels apply: the late mixer needs one block per workgroup (N >= 1024) and
    // the polyphase analysis N / D >= 256 for D = 2, 4, 8, 16.  Measured on 2^24 u8 samples (us per
    // chain_run, N_min = 256 as in round 1 -> now): D = 8, 64 taps 152 -> 43, 256 taps 71 -> 45;
    // D = 16, 256 taps 69 -> 41; D = 4, 128 taps 167 -> 54; D = 2, 64 taps 157 -> 78.
    // (hzsdr_chain_fir_options' nfft_min overrides N_min: the measurement aid those numbers come from.)
    unsigned nfft = 1024;
    if (factor == 2 || factor == 4 || factor == 8 || factor == 16) nfft = std::max(1024u, 256u * factor);
    if (c->fir_nfft_min) nfft = c->fir_nfft_min;
    if (nfft < 256 || nfft > 8192 || (nfft & (nfft - 1))) nfft = 1024;
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
        for (int k = 0; k < hzsdr_chain::kHist; k++) {
            HZ_HIP(ctx, hipMalloc(&c->hist[k], hb));
            HZ_HIP(ctx, hipMemsetAsync(c->hist[k], 0, hb, ctx->stream));
        }
        HZ_TRY(filter_spectrum(c, c->taps_host.data(), c->hfreq));
        if (late_scale(c) != 1.0) {
            HZ_HIP(ctx, hipMalloc(&c->hfreq_late, (size_t)nfft * 8));
            HZ_TRY(filter_spectrum(c, c->taps_host.data(), c->hfreq_late, late_scale(c)));
        }
        HZ_TRY(prepare_late_filters(c, c->ts));
        c->mm_ok = mm_eligible(c);
        if (c->mm_ok) {
            mm_geometry(c);
            std::vector<float> tf(2 * n_taps);
            for (size_t i = 0; i < 2 * n_taps; i++) tf[i] = taps[i];
            for (int k = 0; k < hzsdr_chain::kHist; k++) HZ_HIP(ctx, hipMalloc(&c->rhist[k], hb));
            HZ_HIP(ctx, hipMalloc(&c->taps_dev, 8 * n_taps));
            HZ_HIP(ctx, hipMemcpyAsync(c->taps_dev, tf.data(), 8 * n_taps, hipMemcpyHostToDevice, ctx->stream));
            HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
            HZ_TRY(prepare_mm_tables(c, c->ts));
        }
        return HZSDR_OK;
    };
    const int rc = build();
    if (rc != HZSDR_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        for (void **p : {&c->hfreq, &c->hfreq_late, &c->hist[0], &c->hist[1], &c->hist[2], &c->hist[3], &c->taps_dev, &c->rhist[0], &c->rhist[1],
                         &c->rhist[2], &c->rhist[3]}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        for (auto &kv : c->late_cache) (void)hipFree(kv.second);
        c->late_cache.clear();
        for (auto &kv : c->mm_cache) (void)hipFree(kv.second);
        c->mm_cache.clear();
        c->mm_ok = false;
        c->taps_host.clear();
        return rc;
    }
    c->term = TERM_FIR;
    return HZSDR_OK;
}

}  // extern "C"

/*
 * oracle_parallel.c -- OpenMP drivers around the oracle, used ONLY by the
 * cpu_baseline leg of bench.py ("reference algorithm, parallelised; not the
 * reference": SURVEY.md section 8(d)).  TEST INFRASTRUCTURE, like hzsdr_oracle.c.
 * Chunks are independent, so results equal the serial oracle bit for bit
 * (Shift: each chunk re-derives its start time by running the serial time
 * recurrence up to its first sample, which is cheap next to Sincos).
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void orc_u8_to_c64(const uint8_t *src, float *dst, long n);
void orc_scale(float r, float *buf, long n);
void orc_shift(double *ts_state, unsigned long sample_rate, double freq_hz, float *buf, long n,
               int use_libm);
void orc_shift_ts(double *ts_state, unsigned long sample_rate, double *out, long n);
long orc_convolution_reader(float *out, const float *in, long n, const float *filter, long flen);

int orc_max_threads(void) { return omp_get_max_threads(); }

void orc_par_u8_to_c64(const uint8_t *src, float *dst, long n, int threads) {
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long c = 0; c < threads; c++) {
        long lo = n * c / threads, hi = n * (c + 1) / threads;
        orc_u8_to_c64(src + 2 * lo, dst + 2 * lo, hi - lo);
    }
}

/* Shift then Gain (BASELINE config 2) */
void orc_par_shift_gain(double *ts_state, unsigned long sample_rate, double freq_hz, float gain,
                        float *buf, long n, int threads) {
    double *starts = malloc(sizeof(double) * (threads + 1));
    /* serial pre-pass over the time recurrence only (adds, no Sincos) */
    double ts = *ts_state;
    const double inc = 1.0 / (double)sample_rate, tau = 6.283185307179586476925286766559;
    long next = 0;
    for (int c = 0; c <= threads; c++) {
        long lo = n * c / threads;
        for (; next < lo; next++) {
            ts += inc;
            if (ts > tau)
                ts -= tau;
        }
        starts[c] = ts;
    }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long c = 0; c < threads; c++) {
        long lo = n * c / threads, hi = n * (c + 1) / threads;
        double t = starts[c];
        orc_shift(&t, sample_rate, freq_hz, buf + 2 * lo, hi - lo, 0);
        orc_scale(gain, buf + 2 * lo, hi - lo);
    }
    *ts_state = starts[threads];
    free(starts);
}

void orc_fir_decimate_f64(float *out, const float *x, long n, const float *taps, long t,
                          unsigned d, const float *hist);

/* orc_fir_decimate_f64 with the OUTPUT cut into chunks: output m reads x[d*m - t + 1 .. d*m],
 * which lies in x (or in `hist`, the t-1 samples in front of x; NULL = zeros) whatever
 * chunk computes it, so the result equals the serial oracle bit for bit. */
void orc_par_fir_decimate_f64(float *out, const float *x, long n, const float *taps, long t, unsigned d,
                              const float *hist, int threads) {
    long cnt = n / (long)d, chunks = 8L * threads;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (long c = 0; c < chunks; c++) {
        long mlo = cnt * c / chunks, mhi = cnt * (c + 1) / chunks;
        if (mhi <= mlo)
            continue;
        long j0 = (long)d * mlo;
        if (j0 >= t - 1) { /* "history" = the t-1 samples in front of the chunk, earlier entries of x */
            orc_fir_decimate_f64(out + 2 * mlo, x + 2 * j0, (mhi - mlo) * (long)d, taps, t, d,
                                 t > 1 ? x + 2 * (j0 - (t - 1)) : 0);
        } else { /* chunk starts inside the first t-1 samples: private history = tail of hist ++ x[0..j0) */
            float *hz = calloc((size_t)(2 * (t - 1)) + 2, sizeof(float));
            if (hist)
                memcpy(hz, hist + 2 * j0, sizeof(float) * 2 * (size_t)(t - 1 - j0));
            memcpy(hz + 2 * (t - 1 - j0), x, sizeof(float) * 2 * (size_t)j0);
            orc_fir_decimate_f64(out + 2 * mlo, x + 2 * j0, (mhi - mlo) * (long)d, taps, t, d, hz);
            free(hz);
        }
    }
}

/* The whole north-star chain on every core: u8 -> c64 -> Shift -> t-tap direct-form FIR
 * -> decimate by d, from a fresh stream (clock 0, empty history).  `buf`: n complex64 of
 * scratch (the full-rate intermediate a CPU caller of the reference would hold).
 * Stage 1 (convert + Shift) is chunked like orc_par_shift_gain, stage 2 is
 * orc_par_fir_decimate_f64: both equal the serial oracle bit for bit. */
void orc_par_chain_fir(const uint8_t *src, float *buf, float *out, long n, unsigned long sample_rate,
                       double freq_hz, const float *taps, long t, unsigned d, int threads) {
    double *starts = malloc(sizeof(double) * (threads + 1));
    double ts = 0.0;
    const double inc = 1.0 / (double)sample_rate, tau = 6.283185307179586476925286766559;
    long next = 0;
    for (int c = 0; c <= threads; c++) {
        long lo = n * c / threads;
        for (; next < lo; next++) {
            ts += inc;
            if (ts > tau)
                ts -= tau;
        }
        starts[c] = ts;
    }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long c = 0; c < threads; c++) {
        long lo = n * c / threads, hi = n * (c + 1) / threads;
        double tt = starts[c];
        orc_u8_to_c64(src + 2 * lo, buf + 2 * lo, hi - lo);
        orc_shift(&tt, sample_rate, freq_hz, buf + 2 * lo, hi - lo, 0);
    }
    free(starts);
    orc_par_fir_decimate_f64(out, buf, n, taps, t, d, 0, threads);
}

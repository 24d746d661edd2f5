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

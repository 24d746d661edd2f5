// hz_host.cpp -- the pure-host part of the C-ABI: strings, the identity lookup
// table, the beamform steering-angle math and the NCO time-sequence planner.
// None of this touches the GPU; the arithmetic kernels live in the .hip files.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/hzsdr.h"

extern "C" {

const char *hzsdr_backend(void) { return "hip:gfx950"; }
const char *hzsdr_version(void) { return "hzsdr-hip 0.1.0"; }

const char *hzsdr_strerror(int status) {
    switch (status) {
    case HZSDR_OK: return "ok";
    case HZSDR_ERR_FORMAT_MISMATCH: return "sdr: iq sample formats do not match";           // iq.go:30
    case HZSDR_ERR_FORMAT_UNKNOWN: return "sdr: iq sample format is not understood";        // iq.go:34
    case HZSDR_ERR_DST_TOO_SMALL: return "sdr: destination sample buffer is too small";     // iq.go:38
    case HZSDR_ERR_CONVERSION_NOT_IMPLEMENTED: return "sdr: unknown conversion method";      // conv.go:30
    case HZSDR_ERR_LENGTH_MISMATCH: return "sdr: buffer lengths do not match exactly";
    case HZSDR_ERR_INVALID_ARGUMENT: return "hzsdr: invalid argument";
    case HZSDR_ERR_NO_DEVICE: return "hzsdr: no usable gfx950 (MI355X) device";
    case HZSDR_ERR_HIP: return "hzsdr: HIP runtime error";
    case HZSDR_ERR_OUT_OF_MEMORY: return "hzsdr: out of device memory";
    default: return "hzsdr: unknown status";
    }
}

int hzsdr_format_size(int format) {
    switch (format) {
    case HZSDR_FMT_U8:
    case HZSDR_FMT_I8: return 2;
    case HZSDR_FMT_I16: return 4;
    case HZSDR_FMT_C64: return 8;
    default: return 0;
    }
}

// iq_lookup_table.go:69-90: entry i holds the two bytes of uint16(i) in memory
// order (little-endian host: I = low byte, Q = high byte).
int hzsdr_lut_identity(void *table) {
    if (!table) return HZSDR_ERR_INVALID_ARGUMENT;
    uint8_t *t = (uint8_t *)table;
    for (uint32_t i = 0; i < 65536; i++) {
        t[2 * i] = (uint8_t)(i & 0xFF);
        t[2 * i + 1] = (uint8_t)(i >> 8);
    }
    return HZSDR_OK;
}

// ---- stream/beamform.go:42-128 ---------------------------------------------

// hz.tools/rf v0.0.7 Hz.Wavelength(): speed of light / frequency.
static double wavelength(double hz) { return 299792458.0 / hz; }

int hzsdr_beamform_angles_2d(double frequency_hz, double angle_deg, const double center[2],
                             const double *antennas_xy, int n, float *out) {
    if (n < 0 || (n > 0 && (!center || !antennas_xy || !out))) return HZSDR_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n; i++) {
        const double ax = antennas_xy[2 * i], ay = antennas_xy[2 * i + 1];
        const double xd = ax - center[0], yd = ay - center[1];
        const double dist = sqrt((xd * xd) + (yd * yd));  // computeDistance, :42-48
        if (dist == 0) {                                    // :73-76
            out[2 * i] = 1.0f;
            out[2 * i + 1] = 0.0f;
            continue;
        }
        const double angle_r = angle_deg * (M_PI / 180);    // :79
        const double opposite = ay - center[1];             // :81
        const double theta = asin(opposite / dist);         // :82
        const double p_theta = theta + angle_r;             // :85
        const double p_opposite = sin(p_theta) * dist;      // :86
        const double shift_deg = (p_opposite / wavelength(frequency_hz)) * 360;  // :96
        const double shift_r = shift_deg * (M_PI / 180);    // :97
        out[2 * i] = (float)cos(shift_r);                   // :100-103 conj(cos + i sin)
        out[2 * i + 1] = (float)(-sin(shift_r));
    }
    return HZSDR_OK;
}

int hzsdr_beamform_angles(double frequency_hz, double angle_deg, const double *distances, int n,
                          float *out) {
    if (n < 0 || (n > 0 && (!distances || !out))) return HZSDR_ERR_INVALID_ARGUMENT;
    if (n == 0) return HZSDR_OK;
    // antennas on the x axis, centre = the first one (:111-127)
    const double center[2] = {distances[0], 0.0};
    for (int i = 0; i < n; i++) {
        const double ant[2] = {distances[i], 0.0};
        int rc = hzsdr_beamform_angles_2d(frequency_hz, angle_deg, center, ant, 1, out + 2 * i);
        if (rc != HZSDR_OK) return rc;
    }
    return HZSDR_OK;
}

// ---- NCO time-sequence planner ----------------------------------------------
//
// stream/shifter.go:76-79 advances a float64 clock one rounded addition per
// sample: ts = fl(ts + inc); if ts > 2pi { ts = fl(ts - 2pi) }.  That recurrence
// is serial, but inside one binade [2^e, 2^(e+1)) every ts is a multiple of
// u = 2^(e-52), so fl(ts + inc) = ts + S*u with S = inc/u rounded to nearest
// (ties to even): the sequence is EXACTLY linear until it leaves the binade or
// wraps.  The planner walks binade by binade (a few dozen per 2pi seconds of
// signal) and emits (first, count, t0, step) runs; the kernel evaluates
// ts_i = fma(i, step, t0), which is exact because the true value is
// representable.  Boundary steps are taken with the real float64 additions.
int hzsdr_nco_segments(uint64_t sample_rate, double ts_start, uint64_t n,
                       hzsdr_nco_segment *segments, size_t cap, size_t *n_segments,
                       double *ts_end) {
    if (sample_rate == 0 || !n_segments) return HZSDR_ERR_INVALID_ARGUMENT;
    const double inc = 1.0 / (double)sample_rate;
    const double tau = M_PI * 2;
    const uint64_t top = ((uint64_t)1 << 53) - 1;
    double ts = ts_start;
    uint64_t j = 0;
    size_t count = 0;
    while (j < n) {
        volatile double tv = ts + inc;  // one rounding, never contracted
        double t = tv;
        if (t > tau) {
            volatile double w = t - tau;
            t = w;
        }
        uint64_t run = 1;
        double step = 0.0;
        if (isnormal(t) && t > 0) {
            int e2;
            (void)frexp(t, &e2);  // t = m * 2^e2, m in [0.5, 1)
            const int e = e2 - 1;  // t in [2^e, 2^(e+1))
            const double u = ldexp(1.0, e - 52);
            const double q = inc / u;  // exact: power-of-two scaling
            if (q < 9007199254740992.0) {
                const double ipart = floor(q);
                const double frac = q - ipart;
                const uint64_t T = (uint64_t)(t / u);
                const uint64_t I = (uint64_t)ipart;
                uint64_t S;
                bool ok = true;
                if (frac == 0.5) {
                    // round-half-even: from an even T the step is the even one of I, I+1
                    if (T & 1)
                        ok = false;  // one irregular step first
                    S = I + (I & 1);
                } else {
                    S = I + (frac > 0.5 ? 1 : 0);
                }
                if (ok) {
                    if (S == 0) {
                        run = n - j;  // clock no longer advances
                    } else {
                        if (T + I <= top) run = (top - I - T) / S + 2;
                        if (e == 2) {  // same binade as 2pi: stop before the wrap
                            const uint64_t tau_u = (uint64_t)(tau / u);
                            const uint64_t lim = T <= tau_u ? (tau_u - T) / S + 1 : 1;
                            if (run > lim) run = lim;
                        } else if (e > 2) {
                            run = 1;
                        }
                    }
                    step = (double)S * u;
                }
            }
        }
        if (run > n - j) run = n - j;
        if (run == 1) step = 0.0;
        if (segments && count < cap) {
            segments[count].first = j;
            segments[count].count = run;
            segments[count].t0 = t;
            segments[count].step = step;
        }
        count++;
        ts = fma((double)(run - 1), step, t);
        j += run;
    }
    *n_segments = count;
    if (ts_end) *ts_end = ts;
    return HZSDR_OK;
}

}  // extern "C"

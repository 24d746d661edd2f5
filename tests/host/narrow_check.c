/* narrow_check.c -- the claim of sincos_narrow (csrc/hz_device.h) over many more phases than the test suite's 9e6:
 * oracle/hzsdr_oracle.c's restatement of the device function against its restatement of math.Sincos, on every core.
 *   gcc -O2 -ffp-contract=off -fopenmp tests/host/narrow_check.c oracle/hzsdr_oracle.c oracle/oracle_parallel.c -lm -o /tmp/narrow_check
 *   /tmp/narrow_check 2000000000
 * Phases: a third uniform over +-10^u, u in [-17, 8.7]; a third within 3 ulp of multiples of pi/4 up to 2^29; a third
 * tau shift ts as the kernel forms it (ts = k / fs exactly linear, fs and shift random). */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_sincos_narrow(double x, float *sn, float *cs);
void orc_go_sincos(double x, double *sn, double *cs);

static uint64_t sm(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double u01(uint64_t *s) { return (double)(sm(s) >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 100000000LL;
    long long accepted = 0, wrong = 0, total = 0;
#pragma omp parallel reduction(+ : accepted, wrong, total)
    {
        uint64_t st = 0x1234567ull * (uint64_t)(omp_get_thread_num() + 1);
        const long long per = n / omp_get_num_threads();
        double fs = 20e6, tau = 6.283185307179586 * 2.5e6;
        for (long long i = 0; i < per; i++) {
            double x;
            switch (i % 3) {
            case 0: x = (u01(&st) * 2 - 1) * pow(10.0, u01(&st) * 25.7 - 17.0); break;
            case 1: {
                x = floor(u01(&st) * 683565275.0) * 0.78539816339744830962;
                const int d = (int)(sm(&st) % 7) - 3;
                for (int t = 0; t < abs(d); t++) x = nextafter(x, d > 0 ? INFINITY : -INFINITY);
                if (sm(&st) & 1) x = -x;
                break;
            }
            default:
                if ((i & 0xFFFFF) == 2) { fs = pow(10.0, 3 + 6 * u01(&st)); tau = 6.283185307179586 * (u01(&st) * 2 - 1) * fs; }
                x = tau * ((double)(sm(&st) % (uint64_t)(6.2 * fs + 1)) * (1.0 / fs));
            }
            const double ax = fabs(x);
            if (!(ax < 536870912.0) || (ax < 8.673617379884035e-19 && x != 0)) continue;
            total++;
            float sn, cs;
            if (!orc_sincos_narrow(x, &sn, &cs)) continue;
            accepted++;
            double s, c;
            orc_go_sincos(x, &s, &c);
            const float s32 = (float)s, c32 = (float)c;
            if (memcmp(&s32, &sn, 4) != 0 || memcmp(&c32, &cs, 4) != 0) wrong++;
        }
    }
    printf("%lld phases in the straight path's range: %lld accepted (%.3e refused), %lld of the accepted differ from complex64(math.Sincos)\n", total,
           accepted, (double)(total - accepted) / (double)total, wrong);
    return wrong != 0;
}

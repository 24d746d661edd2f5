/*
 * hzsdr_oracle.c -- CPU restatement of the hz.tools/sdr (hztools/go-sdr) sample
 * processing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py are the only callers.  Nothing under
 * go-sdr_amd/ may include, link or call it; the product path is the HIP
 * library and it fails loudly when that library or a GPU is missing.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout).  The reference is Go; the arithmetic restated here
 * is what the Go toolchain emits on amd64 at GOAMD64=v1 (the default):
 *
 *   - float32 / float64 real arithmetic is never fused (no FMA), each
 *     operation rounds once (IEEE-754 RN).  Compile this file with
 *     -ffp-contract=off -fno-fast-math.
 *   - complex64 * complex64 is NOT computed in float32: cmd/compile's SSA
 *     builder (ssagen/ssa.go, OMUL on complex types, "Compute in Float64 to
 *     minimize cancellation error") widens the four components to float64,
 *     forms (ar*br - ai*bi) and (ar*bi + ai*br) in float64 and narrows the two
 *     results to float32.  The products of two float32 values are exact in
 *     float64, so the result is RN32(RN64(exact difference / sum)) and is
 *     independent of whether the f64 multiply-add is fused.
 *   - complex64 + complex64 is two float32 additions.
 *   - float32 -> int8/uint8/int16 conversions go through a truncating
 *     float32 -> int32 conversion (CVTTSS2SL; out of range or NaN gives
 *     0x80000000) followed by truncation to the narrow type.
 *
 * PARITY PINNING.  The reference cannot be built here (Go toolchain absent,
 * hz.tools/rf v0.0.7 not vendored).  The integer / LUT / single-rounding
 * float paths are pinned by the reference's own known-answer tests, committed
 * as tests/golden/reference_kats.json and checked in tests/test_oracle.py.
 * Two pieces of arithmetic live in third-party code that is not under the
 * reference tree; for those this oracle says so here and in DESIGN.md:
 *
 *   - math.Sincos (Go standard library, version = whatever toolchain builds
 *     the module; go.mod says "go 1.14"): restated below from the published
 *     algorithm (Cephes sin/cos polynomials, 3-part Cody-Waite pi/4, and the
 *     Payne-Hanek reduction of src/math/trig_reduce.go for |x| >= 2^29).
 *     Last-bit parity with a real Go build is UNPINNED (no Go here); the
 *     restatement is validated against mpmath to < 1 f64 ulp in
 *     tests/test_oracle.py, and the reference's own test only pins 1e-4
 *     (stream/shifter_test.go:35-72).
 *   - the FFT behind fft.Planner (no implementation in the reference): values
 *     and normalisation are UNPINNED; the oracle is a float64 FFT rounded to
 *     complex64 at the Plan boundary, forward = exp(-j 2 pi k n / N),
 *     backward unnormalised (evidence: rtl/kerberos/internal/reader.go:54-64
 *     divides by N itself), conformance per testutils/fft.go:54-138.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_FMT_C64 1 /* iq.go:110-126 */
#define ORC_FMT_U8 2
#define ORC_FMT_I16 3
#define ORC_FMT_I8 4

#define ORC_OK 0
#define ORC_ERR_FORMAT_MISMATCH (-1) /* iq.go:27-39 sentinels */
#define ORC_ERR_FORMAT_UNKNOWN (-2)
#define ORC_ERR_DST_TOO_SMALL (-3)
#define ORC_ERR_NOT_IMPLEMENTED (-4) /* conv.go:30 */
#define ORC_ERR_LENGTH (-5)

/* SampleFormat.Size, iq.go:93-106 */
int orc_format_size(int fmt) {
    switch (fmt) {
    case ORC_FMT_U8:
    case ORC_FMT_I8:
        return 2;
    case ORC_FMT_I16:
        return 4;
    case ORC_FMT_C64:
        return 8;
    default:
        return 0;
    }
}

/* Go float32 -> int32 on amd64: CVTTSS2SL. */
static int32_t go_f32_to_i32(float x) {
    if (!(x > -2147483904.0f && x < 2147483648.0f)) /* also NaN */
        return INT32_MIN;
    return (int32_t)x;
}

/* Go complex64 multiply (see header). */
static void go_cmul(float ar, float ai, float br, float bi, float *re, float *im) {
    double a = ar, b = ai, c = br, d = bi;
    double xr = a * c - b * d;
    double xi = a * d + b * c;
    *re = (float)xr;
    *im = (float)xi;
}

/* ------------------------------------------------------------------ */
/* Converters                                                          */
/* ------------------------------------------------------------------ */

/* iq_u8.go:111-121 convU8ToC64Native; iq_u8_amd64.s:27-90 is bit-identical */
void orc_u8_to_c64(const uint8_t *src, float *dst, long n) {
    for (long i = 0; i < 2 * n; i++)
        dst[i] = ((float)src[i] - 127.5f) / 127.5f;
}

/* iq_u8.go:89-100 */
void orc_u8_to_i8(const uint8_t *src, int8_t *dst, long n) {
    for (long i = 0; i < 2 * n; i++)
        dst[i] = (int8_t)((int16_t)src[i] - 128);
}

/* iq_u8.go:75-86 */
void orc_u8_to_i16(const uint8_t *src, int16_t *dst, long n) {
    for (long i = 0; i < 2 * n; i++)
        dst[i] = (int16_t)(((int32_t)src[i] << 8) - 32768);
}

/* iq_i8.go:109-119 */
void orc_i8_to_c64(const int8_t *src, float *dst, long n) {
    for (long i = 0; i < 2 * n; i++)
        dst[i] = (float)src[i] / 128.0f;
}

/* iq_i8.go:85-96 */
void orc_i8_to_u8(const int8_t *src, uint8_t *dst, long n) {
    for (long i = 0; i < 2 * n; i++)
        dst[i] = (uint8_t)((int16_t)src[i] + 128);
}

/* iq_i8.go:71-82 */
void orc_i8_to_i16(const int8_t *src, int16_t *dst, long n) {
    for (long i = 0; i < 2 * n; i++)
        dst[i] = (int16_t)((uint16_t)(int16_t)src[i] << 8);
}

/* iq_i16.go:137-147 (math.MaxInt16 untyped constant -> float32 32767) */
void orc_i16_to_c64(const int16_t *src, float *dst, long n) {
    for (long i = 0; i < 2 * n; i++)
        dst[i] = (float)src[i] / 32767.0f;
}

/* iq_i16.go:116-128 */
void orc_i16_to_u8(const int16_t *src, uint8_t *dst, long n) {
    for (long i = 0; i < 2 * n; i++)
        dst[i] = (uint8_t)((uint16_t)((int32_t)src[i] + 32768) >> 8);
}

/* iq_i16.go:150-162 (arithmetic shift) */
void orc_i16_to_i8(const int16_t *src, int8_t *dst, long n) {
    for (long i = 0; i < 2 * n; i++)
        dst[i] = (int8_t)(src[i] >> 8);
}

/* iq_i16.go:103-111 ShiftLSBToMSBBits */
void orc_i16_shift_lsb_to_msb(int16_t *buf, long n, int bits) {
    int shift = 16 - bits;
    for (long i = 0; i < 2 * n; i++)
        buf[i] = (int16_t)((uint16_t)buf[i] << shift);
}

/* iq_c64.go:77-89; un-fused multiply then add, truncation, no clamp */
void orc_c64_to_u8(const float *src, uint8_t *dst, long n) {
    for (long i = 0; i < 2 * n; i++) {
        float v = src[i] * 127.5f;
        v = v + 127.5f;
        dst[i] = (uint8_t)go_f32_to_i32(v);
    }
}

/* iq_c64.go:92-103 */
void orc_c64_to_i16(const float *src, int16_t *dst, long n) {
    for (long i = 0; i < 2 * n; i++)
        dst[i] = (int16_t)go_f32_to_i32(src[i] * 32767.0f);
}

/* iq_c64.go:106-117 */
void orc_c64_to_i8(const float *src, int8_t *dst, long n) {
    for (long i = 0; i < 2 * n; i++)
        dst[i] = (int8_t)go_f32_to_i32(src[i] * 127.0f);
}

/* conv.go:55-93 ConvertBuffer (and copy.go:31-52 for equal formats).
 * Returns the number of samples converted or a negative error. */
long orc_convert(int dst_fmt, void *dst, long dst_len, int src_fmt, const void *src, long n) {
    if (orc_format_size(src_fmt) == 0 || orc_format_size(dst_fmt) == 0)
        return ORC_ERR_FORMAT_UNKNOWN;
    if (src_fmt == dst_fmt) { /* CopySamples: copy(dst, src) copies min(len) */
        long m = n < dst_len ? n : dst_len;
        memmove(dst, src, (size_t)m * orc_format_size(src_fmt));
        return m;
    }
    if (n > dst_len)
        return ORC_ERR_DST_TOO_SMALL;
    switch (src_fmt * 8 + dst_fmt) {
    case ORC_FMT_U8 * 8 + ORC_FMT_C64: orc_u8_to_c64(src, dst, n); break;
    case ORC_FMT_U8 * 8 + ORC_FMT_I8: orc_u8_to_i8(src, dst, n); break;
    case ORC_FMT_U8 * 8 + ORC_FMT_I16: orc_u8_to_i16(src, dst, n); break;
    case ORC_FMT_I8 * 8 + ORC_FMT_C64: orc_i8_to_c64(src, dst, n); break;
    case ORC_FMT_I8 * 8 + ORC_FMT_U8: orc_i8_to_u8(src, dst, n); break;
    case ORC_FMT_I8 * 8 + ORC_FMT_I16: orc_i8_to_i16(src, dst, n); break;
    case ORC_FMT_I16 * 8 + ORC_FMT_C64: orc_i16_to_c64(src, dst, n); break;
    case ORC_FMT_I16 * 8 + ORC_FMT_U8: orc_i16_to_u8(src, dst, n); break;
    case ORC_FMT_I16 * 8 + ORC_FMT_I8: orc_i16_to_i8(src, dst, n); break;
    case ORC_FMT_C64 * 8 + ORC_FMT_U8: orc_c64_to_u8(src, dst, n); break;
    case ORC_FMT_C64 * 8 + ORC_FMT_I16: orc_c64_to_i16(src, dst, n); break;
    case ORC_FMT_C64 * 8 + ORC_FMT_I8: orc_c64_to_i8(src, dst, n); break;
    default: return ORC_ERR_NOT_IMPLEMENTED;
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* c64 vector ops (internal/simd)                                      */
/* ------------------------------------------------------------------ */

/* internal/simd/mult.go:25-27,40; mult_simd_amd64.s:27-55 (MULPS) */
void orc_scale(float r, float *buf, long n) {
    for (long i = 0; i < 2 * n; i++)
        buf[i] = buf[i] * r;
}

/* internal/simd/mult.go:29-33 rotateComplexNative (amd64 has no asm rotate:
 * mult_simd_amd64.go:38) -> Go complex64 multiply */
void orc_rotate(float re, float im, float *buf, long n) {
    for (long i = 0; i < n; i++)
        go_cmul(buf[2 * i], buf[2 * i + 1], re, im, &buf[2 * i], &buf[2 * i + 1]);
}

/* internal/simd/add.go:33; add_nosimd.go:26-30; add_simd_amd64.s:27-71 (ADDPS).
 * c may alias a or b.  Returns ORC_ERR_LENGTH when lengths differ. */
int orc_add(const float *a, long na, const float *b, long nb, float *c, long nc) {
    if (na != nb || na != nc)
        return ORC_ERR_LENGTH;
    for (long i = 0; i < 2 * na; i++)
        c[i] = a[i] + b[i];
    return ORC_OK;
}

/* stream/add.go:121-185 addReader.Read data path: zero `out`, then
 * out += buf_k for k = 0..K-1 in order (c64: stream/add.go:115-119,165-169). */
void orc_sum_c64(float *out, const float *const *bufs, int k, long n) {
    for (long i = 0; i < 2 * n; i++)
        out[i] = 0.0f;
    for (int j = 0; j < k; j++)
        for (long i = 0; i < 2 * n; i++)
            out[i] = out[i] + bufs[j][i];
}

/* stream/add.go:105-113,170-175 (wrapping int16 add) */
void orc_sum_i16(int16_t *out, const int16_t *const *bufs, int k, long n) {
    for (long i = 0; i < 2 * n; i++)
        out[i] = 0;
    for (int j = 0; j < k; j++)
        for (long i = 0; i < 2 * n; i++)
            out[i] = (int16_t)((uint16_t)out[i] + (uint16_t)bufs[j][i]);
}

/* stream/add.go:95-103,176-181 (wrapping int8 add) */
void orc_sum_i8(int8_t *out, const int8_t *const *bufs, int k, long n) {
    for (long i = 0; i < 2 * n; i++)
        out[i] = 0;
    for (int j = 0; j < k; j++)
        for (long i = 0; i < 2 * n; i++)
            out[i] = (int8_t)((uint8_t)out[i] + (uint8_t)bufs[j][i]);
}

/* ------------------------------------------------------------------ */
/* LookupTable (iq_lookup_table.go)                                    */
/* ------------------------------------------------------------------ */

/* iq_lookup_table.go:56-64: index = the two raw bytes read as a native
 * (little-endian on amd64) uint16: I + (Q << 8).  Gather loops :177-251.
 * `tab` has 65536 entries of dst_fmt; src is u8 or i8 (same raw bytes). */
long orc_lut_apply(int dst_fmt, void *dst, long dst_len, const void *tab, const uint8_t *src,
                   long n) {
    int sz = orc_format_size(dst_fmt);
    if (sz == 0)
        return ORC_ERR_FORMAT_UNKNOWN;
    if (dst_len < n)
        return ORC_ERR_DST_TOO_SMALL;
    uint8_t *d = dst;
    const uint8_t *t = tab;
    for (long i = 0; i < n; i++) {
        uint32_t idx = (uint32_t)src[2 * i] | ((uint32_t)src[2 * i + 1] << 8);
        memcpy(d + (size_t)i * sz, t + (size_t)idx * sz, sz);
    }
    return n;
}

/* iq_lookup_table.go:69-78 / :81-90 identity tables (same raw bytes) */
void orc_lut_identity(uint8_t *tab /* 65536*2 bytes */) {
    for (uint32_t i = 0; i < 65536; i++) {
        tab[2 * i] = (uint8_t)(i & 0xFF);
        tab[2 * i + 1] = (uint8_t)(i >> 8);
    }
}

/* stream/multiply.go:143-172 uint8MultiplyReader.SetMultiplier: private
 * 65535-entry table indexed I*255 + Q (stream/multiply.go:106-108), fill loop
 * runs Q to 256 inclusive (:157-163), then u8 -> c64 -> Multiply(m) -> u8. */
void orc_rotate_table_u8(float re, float im, uint8_t *tab /* 65535*2 bytes */) {
    float *cbuf = malloc(sizeof(float) * 2 * 65535);
    memset(tab, 0, 2 * 65535);
    for (uint32_t realv = 0; realv < 256; realv++)
        for (uint32_t imagv = 0; imagv <= 256; imagv++) {
            uint8_t i8 = (uint8_t)realv, q8 = (uint8_t)imagv;
            uint32_t idx = (uint32_t)i8 * 255 + (uint32_t)q8;
            tab[2 * idx] = i8;
            tab[2 * idx + 1] = q8;
        }
    orc_u8_to_c64(tab, cbuf, 65535);
    orc_rotate(re, im, cbuf, 65535);
    orc_c64_to_u8(cbuf, tab, 65535);
    free(cbuf);
}

/* stream/multiply.go:118-140 uint8MultiplyReader.Read: in place gather */
void orc_rotate_u8_apply(const uint8_t *tab, uint8_t *buf, long n) {
    for (long i = 0; i < n; i++) {
        uint32_t idx = (uint32_t)buf[2 * i] * 255 + (uint32_t)buf[2 * i + 1];
        buf[2 * i] = tab[2 * idx];
        buf[2 * i + 1] = tab[2 * idx + 1];
    }
}

/* stream/multiply.go:212-238 int8MultiplyReader.SetMultiplier: identity i8
 * table -> c64 -> Multiply(m) -> i8, used through sdr.LookupTable. */
void orc_rotate_table_i8(float re, float im, int8_t *tab /* 65536*2 bytes */) {
    float *cbuf = malloc(sizeof(float) * 2 * 65536);
    orc_lut_identity((uint8_t *)tab);
    orc_i8_to_c64(tab, cbuf, 65536);
    orc_rotate(re, im, cbuf, 65536);
    orc_c64_to_i8(cbuf, tab, 65536);
    free(cbuf);
}

/* ------------------------------------------------------------------ */
/* math.Sincos restated (Go standard library, not under the reference) */
/* ------------------------------------------------------------------ */

/* src/math/sin.go: Cephes sin/cos minimax coefficients */
static const double go_sin_c[6] = {
    1.58962301576546568060e-10, -2.50507477628578072866e-8, 2.75573136213857245213e-6,
    -1.98412698295895385996e-4, 8.33333333332211858878e-3,  -1.66666666666666307295e-1,
};
static const double go_cos_c[6] = {
    -1.13585365213876817300e-11, 2.08757008419747316778e-9, -2.75573141792967388112e-7,
    2.48015872888517045348e-5,   -1.38888888888730564116e-3, 4.16666666666665929218e-2,
};

/* src/math/trig_reduce.go: mPi4 = the binary digits of 4/pi as 64-bit words
 * (1 integer bit then 1216 fractional bits).  The table is generated at
 * first use from the hex expansion below, which tests/test_oracle.py
 * recomputes with mpmath. */
static const uint64_t go_mpi4[20] = {
    0x0000000000000001ULL, 0x45f306dc9c882a53ULL, 0xf84eafa3ea69bb81ULL, 0xb6c52b3278872083ULL,
    0xfca2c757bd778ac3ULL, 0x6e48dc74849ba5c0ULL, 0x0c925dd413a32439ULL, 0xfc3bd63962534e7dULL,
    0xd1046bea5d768909ULL, 0xd338e04d68befc82ULL, 0x7323ac7306a673e9ULL, 0x3908bf177bf25076ULL,
    0x3ff12fffbc0b301fULL, 0xde5e2316b414da3eULL, 0xda6cfd9e4f96136eULL, 0x9e8c7ecd3cbfd45aULL,
    0xea4f758fd7cbe2f6ULL, 0x7a0e73ef14a525d4ULL, 0xd7f6bf623f1aba10ULL, 0xac06608df8f6d757ULL,
};

const uint64_t *orc_go_mpi4(void) { return go_mpi4; }

static void mul64(uint64_t a, uint64_t b, uint64_t *hi, uint64_t *lo) {
    unsigned __int128 p = (unsigned __int128)a * b;
    *hi = (uint64_t)(p >> 64);
    *lo = (uint64_t)p;
}

/* src/math/trig_reduce.go trigReduce: Payne-Hanek, x >= 2^29 here */
static void go_trig_reduce(double x, uint64_t *jout, double *zout) {
    const double PI4 = 0.78539816339744830961566084581987572; /* Pi/4 */
    if (x < PI4) {
        *jout = 0;
        *zout = x;
        return;
    }
    uint64_t ix;
    memcpy(&ix, &x, 8);
    int exp = (int)((ix >> 52) & 0x7FF) - 1023 - 52;
    ix &= ~((uint64_t)0x7FF << 52);
    ix |= (uint64_t)1 << 52;
    unsigned digit = (unsigned)(exp + 61) / 64, bitshift = (unsigned)(exp + 61) % 64;
    /* Go's shift semantics: x >> 64 == 0 */
#define SHR(v, s) ((s) >= 64 ? 0 : ((v) >> (s)))
    uint64_t z0 = (go_mpi4[digit] << bitshift) | SHR(go_mpi4[digit + 1], 64 - bitshift);
    uint64_t z1 = (go_mpi4[digit + 1] << bitshift) | SHR(go_mpi4[digit + 2], 64 - bitshift);
    uint64_t z2 = (go_mpi4[digit + 2] << bitshift) | SHR(go_mpi4[digit + 3], 64 - bitshift);
    uint64_t z2hi, z2lo, z1hi, z1lo;
    mul64(z2, ix, &z2hi, &z2lo);
    (void)z2lo;
    mul64(z1, ix, &z1hi, &z1lo);
    uint64_t z0lo = z0 * ix;
    uint64_t lo = z1lo + z2hi;
    uint64_t c = lo < z1lo ? 1 : 0;
    uint64_t hi = z0lo + z1hi + c;
    uint64_t j = hi >> 61;
    hi = (hi << 3) | (lo >> 61);
    unsigned lz = hi == 0 ? 64 : (unsigned)__builtin_clzll(hi);
    uint64_t e = (uint64_t)(1023 - (lz + 1));
    hi = (lz + 1 >= 64 ? 0 : (hi << (lz + 1))) | SHR(lo, 64 - (lz + 1));
    hi >>= 64 - 52;
    hi |= e << 52;
    double z;
    memcpy(&z, &hi, 8);
    if (j & 1) {
        j++;
        j &= 7;
        z = z - 1.0;
    }
    *jout = j;
    *zout = z * PI4;
#undef SHR
}

/* src/math/sincos.go Sincos */
void orc_go_sincos(double x, double *sn, double *cs) {
    const double PI4A = 7.85398125648498535156e-1;  /* 0x3fe921fb40000000 */
    const double PI4B = 3.77489470793079817668e-8;  /* 0x3e64442d00000000 */
    const double PI4C = 2.69515142907905952645e-15; /* 0x3ce8469898cc5170 */
    const double M4PI = 1.27323954473516268615107010698; /* 4/Pi */
    if (x == 0) {
        *sn = x;
        *cs = 1;
        return;
    }
    if (isnan(x) || isinf(x)) {
        *sn = NAN;
        *cs = NAN;
        return;
    }
    int sin_sign = 0, cos_sign = 0;
    if (x < 0) {
        x = -x;
        sin_sign = 1;
    }
    uint64_t j;
    double y, z;
    if (x >= (double)(1 << 29)) {
        go_trig_reduce(x, &j, &z);
    } else {
        j = (uint64_t)(x * M4PI);
        y = (double)j;
        if (j & 1) {
            j++;
            y = y + 1;
        }
        j &= 7;
        z = ((x - y * PI4A) - y * PI4B) - y * PI4C;
    }
    if (j > 3) {
        j -= 4;
        sin_sign = !sin_sign;
        cos_sign = !cos_sign;
    }
    if (j > 1)
        cos_sign = !cos_sign;
    double zz = z * z;
    double c = 1.0 - 0.5 * zz +
               zz * zz *
                   ((((((go_cos_c[0] * zz) + go_cos_c[1]) * zz + go_cos_c[2]) * zz + go_cos_c[3]) *
                         zz +
                     go_cos_c[4]) *
                        zz +
                    go_cos_c[5]);
    double s = z + z * zz *
                       ((((((go_sin_c[0] * zz) + go_sin_c[1]) * zz + go_sin_c[2]) * zz +
                          go_sin_c[3]) *
                             zz +
                         go_sin_c[4]) *
                            zz +
                        go_sin_c[5]);
    if (j == 1 || j == 2) {
        double t = s;
        s = c;
        c = t;
    }
    if (cos_sign)
        c = -c;
    if (sin_sign)
        s = -s;
    *sn = s;
    *cs = c;
}

/* ------------------------------------------------------------------ */
/* stream.ShiftBuffer (stream/shifter.go:66-85)                        */
/* ------------------------------------------------------------------ */

/* A restatement of the device function sincos_narrow (go-sdr_amd/csrc/hz_device.h), for the CPU check of its claim:
 * Go's range reduction (the first two steps fused: their products are exact), the Cephes polynomials with fused
 * Horner steps, the narrowing decision on the float64 bits.  Returns 1 where the float32 pair is claimed to equal
 * complex64(math.Sincos(x)).  Test infrastructure like everything here: tests/test_oracle.py compares it with
 * orc_go_sincos over many arguments (orc_sincos_narrow_check below). */
int orc_sincos_narrow(double x, float *sn, float *cs) {
    const double PI4A = 7.85398125648498535156e-1, PI4B = 3.77489470793079817668e-8, PI4C = 2.69515142907905952645e-15;
    const double M4PI = 1.27323954473516268615107010698;
    const double ax = fabs(x);
    int32_t ji = (int32_t)(ax * M4PI);
    ji += ji & 1;
    const double y = (double)ji;
    double z = fma(-y, PI4A, ax);
    z = fma(-y, PI4B, z);
    z = z - y * PI4C;
    const double zz = z * z;
    double ps = fma(go_sin_c[0], zz, go_sin_c[1]);
    ps = fma(ps, zz, go_sin_c[2]);
    ps = fma(ps, zz, go_sin_c[3]);
    ps = fma(ps, zz, go_sin_c[4]);
    ps = fma(ps, zz, go_sin_c[5]);
    const double s = fma(z * zz, ps, z);
    double pc = fma(go_cos_c[0], zz, go_cos_c[1]);
    pc = fma(pc, zz, go_cos_c[2]);
    pc = fma(pc, zz, go_cos_c[3]);
    pc = fma(pc, zz, go_cos_c[4]);
    pc = fma(pc, zz, go_cos_c[5]);
    const double c = fma(zz * zz, pc, fma(-0.5, zz, 1.0));
    uint64_t sb, cb, xb;
    memcpy(&sb, &s, 8);
    memcpy(&cb, &c, 8);
    memcpy(&xb, &x, 8);
    const uint32_t ds = ((uint32_t)sb + (128u - 0x10000000u)) & 0x1FFFFFFFu;
    const uint32_t dc = ((uint32_t)cb + (128u - 0x10000000u)) & 0x1FFFFFFFu;
    const float sf = (float)s, cf = (float)c;
    const int swap = ji & 2;
    float a = swap ? cf : sf, b = swap ? sf : cf;
    const uint32_t t = (uint32_t)ji << 29;
    uint32_t ab, bb;
    memcpy(&ab, &a, 4);
    memcpy(&bb, &b, 4);
    ab ^= ((uint32_t)(xb >> 32) ^ t) & 0x80000000u;
    bb ^= (t ^ (t << 1)) & 0x80000000u;
    memcpy(sn, &ab, 4);
    memcpy(cs, &bb, 4);
    return ds > 256u && dc > 256u;
}

/* Over xs[0..n): how many arguments the check accepts, and how many of THOSE differ from complex64(math.Sincos)
 * in any bit (must be 0); the caller keeps |x| inside [2^-60, 2^29) or x = +-0, as the kernel does. */
void orc_sincos_narrow_check(const double *xs, long n, long *accepted, long *wrong) {
    long acc = 0, bad = 0;
    for (long i = 0; i < n; i++) {
        float sn, cs;
        if (!orc_sincos_narrow(xs[i], &sn, &cs)) continue;
        acc++;
        double s, c;
        orc_go_sincos(xs[i], &s, &c);
        const float s32 = (float)s, c32 = (float)c;
        if (memcmp(&s32, &sn, 4) != 0 || memcmp(&c32, &cs, 4) != 0) bad++;
    }
    *accepted = acc;
    *wrong = bad;
}

/* The closure state `ts` persists across buffers: *ts is read and updated.
 * use_libm != 0 swaps math.Sincos for libm sincos (cross-check only). */
void orc_shift(double *ts_state, unsigned long sample_rate, double freq_hz, float *buf, long n,
               int use_libm) {
    double ts = *ts_state;
    const double inc = 1.0 / (double)sample_rate;
    const double tau = M_PI * 2;
    const double shift = freq_hz;
    for (long j = 0; j < n; j++) {
        ts += inc;
        if (ts > tau)
            ts -= tau;
        double ph = tau * shift * ts; /* (tau*shift)*ts, left to right */
        double im, rl;
        if (use_libm) {
            im = sin(ph);
            rl = cos(ph);
        } else {
            orc_go_sincos(ph, &im, &rl);
        }
        go_cmul(buf[2 * j], buf[2 * j + 1], (float)rl, (float)im, &buf[2 * j], &buf[2 * j + 1]);
    }
    *ts_state = ts;
}

/* The NCO time sequence alone (for testing the piecewise-linear table). */
void orc_shift_ts(double *ts_state, unsigned long sample_rate, double *out, long n) {
    double ts = *ts_state;
    const double inc = 1.0 / (double)sample_rate;
    const double tau = M_PI * 2;
    for (long j = 0; j < n; j++) {
        ts += inc;
        if (ts > tau)
            ts -= tau;
        out[j] = ts;
    }
    *ts_state = ts;
}

/* ------------------------------------------------------------------ */
/* Decimate / Downsample                                               */
/* ------------------------------------------------------------------ */

/* stream/decimate.go:59-101 DecimateBuffer.  `offset` is accepted and ignored
 * exactly as the reference does.  u8 / i16 / c64 only (:85-97). */
long orc_decimate(int to_fmt, void *to, long to_len, int from_fmt, const void *from, long from_len,
                  unsigned factor, long offset) {
    (void)offset;
    if (from_fmt != to_fmt)
        return ORC_ERR_FORMAT_MISMATCH;
    long d = (long)factor;
    long cnt = from_len / d;
    if (to_len < cnt)
        return ORC_ERR_DST_TOO_SMALL;
    if (cnt > 0 && from_fmt != ORC_FMT_U8 && from_fmt != ORC_FMT_I16 && from_fmt != ORC_FMT_C64)
        return ORC_ERR_FORMAT_UNKNOWN;
    int sz = orc_format_size(from_fmt);
    for (long i = 0; i < cnt; i++)
        memcpy((uint8_t *)to + (size_t)i * sz, (const uint8_t *)from + (size_t)(d * i) * sz, sz);
    return cnt;
}

/* stream/downsample.go:68-127 DownsampleBuffer: per output, convert a window
 * of `factor` samples to c64 (:104-114), sum in order from +0 (:116-118),
 * divide each component by float32(factor) (:120-123). */
long orc_downsample(float *to, long to_len, int to_fmt, int from_fmt, const void *from,
                    long from_len, unsigned factor, long offset) {
    (void)offset;
    if (to_fmt != ORC_FMT_C64)
        return ORC_ERR_FORMAT_MISMATCH;
    long d = (long)factor;
    long cnt = from_len / d;
    if (to_len < cnt)
        return ORC_ERR_DST_TOO_SMALL;
    if (cnt > 0 && from_fmt != ORC_FMT_U8 && from_fmt != ORC_FMT_I16 && from_fmt != ORC_FMT_C64)
        return ORC_ERR_FORMAT_UNKNOWN;
    float *win = malloc(sizeof(float) * 2 * (size_t)(d > 0 ? d : 1));
    for (long i = 0; i < cnt; i++) {
        long start = i * d;
        const float *w = win;
        if (from_fmt == ORC_FMT_U8)
            orc_u8_to_c64((const uint8_t *)from + 2 * start, win, d);
        else if (from_fmt == ORC_FMT_I16)
            orc_i16_to_c64((const int16_t *)from + 2 * start, win, d);
        else
            w = (const float *)from + 2 * start;
        float sr = 0.0f, si = 0.0f;
        for (long j = 0; j < d; j++) {
            sr = sr + w[2 * j];
            si = si + w[2 * j + 1];
        }
        to[2 * i] = sr / (float)d;
        to[2 * i + 1] = si / (float)d;
    }
    free(win);
    return cnt;
}

/* ------------------------------------------------------------------ */
/* FFT (behind fft.Planner; values UNPINNED, see header) + fft glue    */
/* ------------------------------------------------------------------ */

/* float64 iterative radix-2, any power of two; direction: 1 forward
 * (exp(-j...)), 0 backward (exp(+j...), unnormalised). */
static void fft64(double *re, double *im, long n, int forward) {
    for (long i = 1, j = 0; i < n; i++) {
        long bit = n >> 1;
        for (; j & bit; bit >>= 1)
            j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (long len = 2; len <= n; len <<= 1) {
        double ang = (forward ? -2.0 : 2.0) * M_PI / (double)len;
        for (long i = 0; i < n; i += len)
            for (long k = 0; k < len / 2; k++) {
                double wr = cos(ang * (double)k), wi = sin(ang * (double)k);
                long a = i + k, b = i + k + len / 2;
                double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

/* One fft.Plan.Transform(): c64 in -> c64 out (fft/fft.go:45-59).  Returns
 * ORC_ERR_DST_TOO_SMALL on a length mismatch (testutils/fft.go:127-137). */
int orc_fft(const float *in, long n_in, float *out, long n_out, int forward) {
    if (n_in != n_out)
        return ORC_ERR_DST_TOO_SMALL;
    long n = n_in;
    if (n <= 0)
        return ORC_ERR_LENGTH;
    double *re = malloc(sizeof(double) * n), *im = malloc(sizeof(double) * n);
    for (long i = 0; i < n; i++) {
        re[i] = in[2 * i];
        im[i] = in[2 * i + 1];
    }
    if (n & (n - 1)) {
        /* a length that is not a power of two (the reference's Planner takes any: fft/fft.go:45-48): the DFT
         * as it is written, O(n^2), float64, the twiddle of n k taken from a table at (n k) mod N -- exact
         * index arithmetic, so the only errors are the table's and the sums' roundings.  Test sizes only. */
        double *wr = malloc(sizeof(double) * n), *wi = malloc(sizeof(double) * n);
        double *xr = malloc(sizeof(double) * n), *xi = malloc(sizeof(double) * n);
        const double sgn = forward ? -1.0 : 1.0;
        for (long m = 0; m < n; m++) {
            const double a = sgn * 6.283185307179586476925286766559 * (double)m / (double)n;
            wr[m] = cos(a);
            wi[m] = sin(a);
            xr[m] = re[m];
            xi[m] = im[m];
        }
        for (long k = 0; k < n; k++) {
            double sr = 0, si = 0;
            long idx = 0;
            for (long m = 0; m < n; m++) {
                sr += xr[m] * wr[idx] - xi[m] * wi[idx];
                si += xr[m] * wi[idx] + xi[m] * wr[idx];
                idx += k;
                if (idx >= n)
                    idx -= n;
            }
            re[k] = sr;
            im[k] = si;
        }
        free(wr);
        free(wi);
        free(xr);
        free(xi);
    } else
    fft64(re, im, n, forward);
    for (long i = 0; i < n; i++) {
        out[2 * i] = (float)re[i];
        out[2 * i + 1] = (float)im[i];
    }
    free(re);
    free(im);
    return ORC_OK;
}

/* fft/convolution.go:150-192 ConvolveFreq closure: forward plan src->freq1,
 * freq1[i] *= freq[i] (Go complex64 multiply, :187-189), backward plan
 * freq1->dst.  dst may alias src. */
int orc_convolve_freq(float *dst, const float *src, const float *freq, long n) {
    float *f1 = malloc(sizeof(float) * 2 * n);
    int rc = orc_fft(src, n, f1, n, 1);
    if (rc == ORC_OK) {
        for (long i = 0; i < n; i++)
            go_cmul(f1[2 * i], f1[2 * i + 1], freq[2 * i], freq[2 * i + 1], &f1[2 * i],
                    &f1[2 * i + 1]);
        rc = orc_fft(f1, n, dst, n, 0);
    }
    free(f1);
    return rc;
}

/* fft/convolution.go:97-147 Convolve (conj == 0) / CrossCorrelate (conj != 0) */
int orc_convolve(float *dst, const float *iq1, const float *iq2, long n, int conj) {
    float *f1 = malloc(sizeof(float) * 2 * n), *f2 = malloc(sizeof(float) * 2 * n);
    int rc = orc_fft(iq1, n, f1, n, 1);
    if (rc == ORC_OK)
        rc = orc_fft(iq2, n, f2, n, 1);
    if (rc == ORC_OK) {
        for (long i = 0; i < n; i++) {
            float br = f2[2 * i], bi = conj ? -f2[2 * i + 1] : f2[2 * i + 1];
            go_cmul(f1[2 * i], f1[2 * i + 1], br, bi, &f1[2 * i], &f1[2 * i + 1]);
        }
        rc = orc_fft(f1, n, dst, n, 0);
    }
    free(f1);
    free(f2);
    return rc;
}

/* stream/convolution.go:36-82 ConvolutionReader: block = len(filter),
 * block-circular, no overlap, no scaling; a trailing partial block is never
 * produced (ReadFull of a whole block fails first: read_transformer.go:120). */
long orc_convolution_reader(float *out, const float *in, long n, const float *filter, long flen) {
    long blocks = n / flen;
    for (long b = 0; b < blocks; b++) {
        int rc = orc_convolve_freq(out + 2 * b * flen, in + 2 * b * flen, filter, flen);
        if (rc != ORC_OK)
            return rc;
    }
    return blocks * flen;
}

/* North-star FIR-decimate (not a reference function; SURVEY.md section 7 hard part
 * 5): y[m] = sum_k h[k] * x[D*m - k], x[j<0] taken from `hist` (hist[t-1-i] is
 * x[-1-i], i.e. hist holds the previous t-1 samples in time order) or zero.
 * float64 accumulation, complex taps; the truth the overlap-save kernel is
 * compared with under a tolerance. */
void orc_fir_decimate_f64(float *out, const float *x, long n, const float *taps, long t,
                          unsigned d, const float *hist) {
    long cnt = n / (long)d;
    for (long m = 0; m < cnt; m++) {
        double ar = 0, ai = 0;
        long j0 = (long)d * m;
        for (long k = 0; k < t; k++) {
            long j = j0 - k;
            double xr, xi;
            if (j >= 0) {
                xr = x[2 * j];
                xi = x[2 * j + 1];
            } else if (hist) {
                long h = (t - 1) + j; /* j = -1 -> t-2 */
                if (h < 0)
                    continue;
                xr = hist[2 * h];
                xi = hist[2 * h + 1];
            } else
                continue;
            double hr = taps[2 * k], hi = taps[2 * k + 1];
            ar += xr * hr - xi * hi;
            ai += xr * hi + xi * hr;
        }
        out[2 * m] = (float)ar;
        out[2 * m + 1] = (float)ai;
    }
}

/* ------------------------------------------------------------------ */
/* Beamform                                                            */
/* ------------------------------------------------------------------ */

/* hz.tools/rf v0.0.7 Hz.Wavelength(): c / f (pinned to 1e-4 by
 * stream/beamform_test.go:115-155). */
static double rf_wavelength(double hz) { return 299792458.0 / hz; }

/* stream/beamform.go:42-48 */
static double compute_distance(const double p1[2], const double p2[2]) {
    double xd = p1[0] - p2[0], xy = p1[1] - p2[1];
    return sqrt((xd * xd) + (xy * xy));
}

/* stream/beamform.go:57-107 BeamformAngles2D.  out: n complex64 (re,im). */
void orc_beamform_angles_2d(double frequency_hz, double angle_deg, const double center[2],
                            const double *antennas /* n x 2 */, int n, float *out) {
    for (int i = 0; i < n; i++) {
        const double *ant = antennas + 2 * i;
        double nd = compute_distance(ant, center);
        if (nd == 0) {
            out[2 * i] = 1.0f;
            out[2 * i + 1] = 0.0f;
            continue;
        }
        double angle_r = angle_deg * (M_PI / 180);
        double n_opp = ant[1] - center[1];
        double n_theta = asin(n_opp / nd);
        double p_theta = n_theta + angle_r;
        double p_opp = sin(p_theta) * nd;
        double phase_shift = (p_opp / rf_wavelength(frequency_hz)) * 360;
        double phase_r = phase_shift * (M_PI / 180);
        out[2 * i] = (float)cos(phase_r);
        out[2 * i + 1] = (float)(-sin(phase_r)); /* cmplx.Conj */
    }
}

/* stream/beamform.go:111-127 BeamformAngles: antennas on the x axis,
 * centre = the first antenna. */
void orc_beamform_angles(double frequency_hz, double angle_deg, const double *distances, int n,
                         float *out) {
    if (n == 0)
        return;
    double *ant = malloc(sizeof(double) * 2 * n);
    for (int i = 0; i < n; i++) {
        ant[2 * i] = distances[i];
        ant[2 * i + 1] = 0;
    }
    orc_beamform_angles_2d(frequency_hz, angle_deg, ant, ant, n, out);
    free(ant);
}

/* stream/beamform.go:148-171 ReadBeamform data path for c64 inputs: per
 * channel multiplyReader (skipped when m == 1: stream/multiply.go:59-62),
 * then the ordered sum of stream/add.go.  Inputs are not modified. */
void orc_beamform(float *out, const float *const *chans, const float *weights, int k, long n) {
    for (long i = 0; i < 2 * n; i++)
        out[i] = 0.0f;
    for (int c = 0; c < k; c++) {
        float wr = weights[2 * c], wi = weights[2 * c + 1];
        int identity = (wr == 1.0f && wi == 0.0f);
        for (long i = 0; i < n; i++) {
            float yr = chans[c][2 * i], yi = chans[c][2 * i + 1];
            if (!identity)
                go_cmul(yr, yi, wr, wi, &yr, &yi);
            out[2 * i] = out[2 * i] + yr;
            out[2 * i + 1] = out[2 * i + 1] + yi;
        }
    }
}

/* testutils/cw.go:31-46 CW tone generator (test input only, libm cos/sin) */
void orc_cw(float *buf, long n, double freq, long sample_rate, double phase) {
    const double tau = M_PI * 2;
    for (long i = 0; i < n; i++) {
        double now = (double)i / (double)sample_rate;
        buf[2 * i] = (float)cos(tau * freq * now + phase);
        buf[2 * i + 1] = (float)sin(tau * freq * now + phase);
    }
}

/* ------------------------------------------------------------------ */
/* rtl/kerberos coherent sync ("next" row)        */
/* ------------------------------------------------------------------ */

/* rtl/kerberos/internal/align.go:128-149 (checkAlignment): index of the largest
 * float64(re*re + im*im) (float32 products and sum, un-fused), exact zeros
 * skipped, first maximum wins; indices above n/2 fold to negative lags.
 * Returns 0 and *lag = -1 when every element is zero (maxPowI stays -1). */
int orc_peak_lag(const float *corr, long n, long *lag) {
    double max_pow = -INFINITY;
    long max_i = -1;
    for (long i = 0; i < n; i++) {
        float re = corr[2 * i], im = corr[2 * i + 1];
        if (re == 0.0f && im == 0.0f)
            continue;
        float rr = re * re, ii = im * im;
        double pow = (double)(rr + ii);
        if (pow > max_pow) {
            max_pow = pow;
            max_i = i;
        }
    }
    if (max_i > n / 2)
        max_i -= n;
    *lag = max_i;
    return ORC_OK;
}

/* rtl/kerberos/internal/align.go:257-262 (PhaseOffsets inner loop for one pair):
 * sum over i of cmplx.Phase(complex128(a[i] * conj(b[i]))) = atan2(im, re) in
 * float64, accumulated in order, divided by n.  math.Atan2 is Go's own (Cephes);
 * libm's atan2 is used here: last-bit parity UNPINNED, like Sincos. */
double orc_mean_phase(const float *a, const float *b, long n) {
    double acc = 0.0;
    for (long i = 0; i < n; i++) {
        float re, im;
        go_cmul(a[2 * i], a[2 * i + 1], b[2 * i], -b[2 * i + 1], &re, &im);
        acc += atan2((double)im, (double)re);
    }
    return acc / (double)n;
}

/* ------------------------------------------------------------------ */
/* rtl/kerberos graft + foreign-endian payloads */
/* ------------------------------------------------------------------ */

/* rtl/kerberos/internal/reader.go:47-64 FFTShiftAndScale: data[i], data[half+i] =
 * data[half+i]/scale, data[i]/scale component-wise in float32; an odd tail element
 * is left alone. */
void orc_fftshift_scale(float *data, long n, float scale) {
    long half = n / 2;
    for (long i = 0; i < half; i++) {
        float lr = data[2 * i], li = data[2 * i + 1];
        float hr = data[2 * (half + i)], hi = data[2 * (half + i) + 1];
        data[2 * i] = hr / scale;
        data[2 * i + 1] = hi / scale;
        data[2 * (half + i)] = lr / scale;
        data[2 * (half + i) + 1] = li / scale;
    }
}

/* One trip of graftReader.do's loop, rtl/kerberos/internal/graft.go:97-114: band i is
 * forward-transformed into freq[i*n:(i+1)*n], FFTShiftAndScale(.., float32(n)), then
 * one backward transform of all count*n bins into out. */
int orc_graft(float *out, const float *const *bands, int count, long n) {
    long total = n * count;
    float *freq = malloc(sizeof(float) * 2 * total);
    int rc = ORC_OK;
    for (int c = 0; c < count && rc == ORC_OK; c++) {
        rc = orc_fft(bands[c], n, freq + 2 * c * n, n, 1);
        if (rc == ORC_OK)
            orc_fftshift_scale(freq + 2 * c * n, n, (float)n);
    }
    if (rc == ORC_OK)
        rc = orc_fft(freq, total, out, total, 0);
    free(freq);
    return rc;
}

/* bytes_io.go:30-64 / :150-197: the foreign-order ByteReader / ByteWriter move every
 * int16 (width 2) or float32 (width 4) component through binary.Read/Write with the
 * other byte order, i.e. reverse each component's bytes. */
void orc_byteswap(unsigned char *buf, long ncomp, int width) {
    for (long i = 0; i < ncomp; i++)
        for (int b = 0; b < width / 2; b++) {
            unsigned char t = buf[i * width + b];
            buf[i * width + b] = buf[i * width + width - 1 - b];
            buf[i * width + width - 1 - b] = t;
        }
}

// hz_device.h -- device-side arithmetic shared by every kernel.
//
// The arithmetic contracts (what "bit-exact with the reference" means here)
// are the Go/amd64 ones: float32/float64 real ops round once and are never
// fused (the translation units are built with -ffp-contract=off and anything
// that must not fuse is written with explicit operations), complex64 products
// are formed in float64 and narrowed once, float->narrow-int conversions go
// through a truncating float->int32 conversion.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hz {

// Streaming (use-once) global accesses: non-temporal loads / stores of a trivially
// copyable T of 2, 4, 8 or 16 bytes, so that a buffer that is read or written exactly
// once does not displace the tables other kernels keep in L2 / the Infinity Cache.
template <int BYTES> struct NtWord;
template <> struct NtWord<2> { using t = uint16_t; };
template <> struct NtWord<4> { using t = uint32_t; };
template <> struct NtWord<8> { typedef uint32_t t __attribute__((ext_vector_type(2))); };
template <> struct NtWord<16> { typedef uint32_t t __attribute__((ext_vector_type(4))); };

template <class T> __device__ __forceinline__ T nt_load(const T *p) {
    using W = typename NtWord<sizeof(T)>::t;
    const W w = __builtin_nontemporal_load(reinterpret_cast<const W *>(p));
    return __builtin_bit_cast(T, w);
}
template <class T> __device__ __forceinline__ void nt_store(T *p, const T &v) {
    using W = typename NtWord<sizeof(T)>::t;
    __builtin_nontemporal_store(__builtin_bit_cast(W, v), reinterpret_cast<W *>(p));
}

// A kernel's accesses as a template switch: NT = the call streams past the memory-side cache.  A call that touches
// 96 MiB or more finds nothing of itself in the 256 MB Infinity Cache the next time anyway; measured over rotations of
// buffers (tools/stream_rate.hip, from HBM both ways, 2^24 samples): Scale in place 48.0 -> 43.2 us (6.2 TB/s), the
// boxcar Downsample / 8 from i16 17.3 -> 16.1 us (5.2 TB/s), u8 -> c64 36.1 -> 34.5 us (4.9 TB/s: four fifths of its
// bytes are writes) -- and 20 % SLOWER over one cached buffer pair, hence the threshold.
constexpr size_t kStreamPastCacheBytes = (size_t)96 << 20;
inline bool streams_past_cache(size_t bytes) { return bytes >= kStreamPastCacheBytes; }
template <bool NT, class T> __device__ __forceinline__ T ld_stream(const T *p) {
    if constexpr (NT) return nt_load(p);
    else return *p;
}
template <bool NT, class T> __device__ __forceinline__ void st_stream(T *p, const T &v) {
    if constexpr (NT) nt_store(p, v);
    else *p = v;
}

// Go complex64 * complex64: cmd/compile widens to float64 ("Compute in
// Float64 to minimize cancellation error"), forms ar*br - ai*bi and
// ar*bi + ai*br, narrows each once.  Products of two float32 values are exact
// in float64, so fma(a, c, -(b*d)) rounds the same exact value once and is
// bit-identical to the un-fused form while costing half the f64 issue slots.
// Call sites: internal/simd/mult.go:29-33, stream/shifter.go:82,
// fft/convolution.go:107-109,129-134,187-189.
__device__ __forceinline__ float2 go_cmul(float2 a, float2 b) {
    double ar = a.x, ai = a.y, br = b.x, bi = b.y;
    double re = __fma_rn(ar, br, -(ai * bi));
    double im = __fma_rn(ar, bi, ai * br);
    return make_float2((float)re, (float)im);
}

// Go float32 -> int32 on amd64 (CVTTSS2SL): truncation; NaN and out-of-range
// give 0x80000000 ("integer indefinite"), unlike v_cvt_i32_f32 which saturates.
__device__ __forceinline__ int32_t go_f32_to_i32(float x) {
    int32_t r = (int32_t)x;  // v_cvt_i32_f32: truncates, saturates, NaN -> 0
    bool in_range = (x > -2147483904.0f) && (x < 2147483648.0f);
    return in_range ? r : INT32_MIN;
}

// ---- per-component converters (one float / integer component each) --------

// Correctly rounded x / d for a CONSTANT divisor without the ~10-instruction
// IEEE divide expansion: q0 = RN(x * RN(1/d)); the remainder x - d*q0 is exact
// in one fma; one more fma applies the correction (Markstein).  The result is
// the IEEE quotient for every input the converters can produce -- the domains
// are 256 and 65 536 values and tests/test_gpu_parity.py checks all of them
// bit for bit against the oracle's real division.
__device__ __forceinline__ float div_const_rn(float x, float d, float rcp_d) {
    float q0 = __fmul_rn(x, rcp_d);
    float rem = __fmaf_rn(-d, q0, x);
    return __fmaf_rn(rem, rcp_d, q0);
}
// x / d for the two divisors the converters use, in two instructions: with c = RN(1/d) and
// c_lo = RN(1/d - c), fma(x, c, RN(x * c_lo)) is the correctly rounded quotient for EVERY
// input these converters can see -- checked exhaustively in exact rational arithmetic (all
// 256 values of b - 127.5 against d = 127.5, all 65 536 int16 values against d = 32767) and
// again on the GPU by tests/test_gpu_parity.py::test_convert_exhaustive_bit_exact.  It is
// NOT a general division (div_const_rn is).
__device__ __forceinline__ float div_by_const2(float x, float c, float c_lo) {
    return __fmaf_rn(x, c, __fmul_rn(x, c_lo));
}
// iq_u8.go:111-121 / iq_u8_amd64.s:71-89: (float32(b) - 127.5) / 127.5
__device__ __forceinline__ float u8_to_f32(uint32_t b) {
    return div_by_const2(__fsub_rn((float)b, 127.5f), 0x1.010102p-7f, -0x1.fdfdfep-32f);
}
// iq_i8.go:109-119: float32(b) / 128 (a power of two: the multiply is exact)
__device__ __forceinline__ float i8_to_f32(int32_t b) { return __fmul_rn((float)b, 0.0078125f); }
// iq_i16.go:137-147: float32(v) / 32767
__device__ __forceinline__ float i16_to_f32(int32_t v) {
    return div_by_const2((float)v, 0x1.0002p-15f, 0x1.0002p-45f);
}
// iq_c64.go:77-89: uint8(x*127.5 + 127.5), un-fused
__device__ __forceinline__ uint32_t f32_to_u8(float x) {
    float v = __fadd_rn(__fmul_rn(x, 127.5f), 127.5f);
    return (uint32_t)go_f32_to_i32(v) & 0xFFu;
}
// iq_c64.go:92-103: int16(x * 32767)
__device__ __forceinline__ uint32_t f32_to_i16(float x) {
    return (uint32_t)go_f32_to_i32(__fmul_rn(x, 32767.0f)) & 0xFFFFu;
}
// iq_c64.go:106-117: int8(x * 127)
__device__ __forceinline__ uint32_t f32_to_i8(float x) {
    return (uint32_t)go_f32_to_i32(__fmul_rn(x, 127.0f)) & 0xFFu;
}

// ---- math.Sincos (Go standard library) restated for the device -------------
// src/math/sin.go coefficients (Cephes), src/math/sincos.go Cody-Waite
// reduction for x < 2^29, src/math/trig_reduce.go Payne-Hanek above.  Every
// operation is an explicit un-fused IEEE double op so the result equals the
// host restatement in oracle/hzsdr_oracle.c bit for bit.

__device__ __constant__ static const uint64_t kMPi4[20] = {
    0x0000000000000001ULL, 0x45f306dc9c882a53ULL, 0xf84eafa3ea69bb81ULL, 0xb6c52b3278872083ULL,
    0xfca2c757bd778ac3ULL, 0x6e48dc74849ba5c0ULL, 0x0c925dd413a32439ULL, 0xfc3bd63962534e7dULL,
    0xd1046bea5d768909ULL, 0xd338e04d68befc82ULL, 0x7323ac7306a673e9ULL, 0x3908bf177bf25076ULL,
    0x3ff12fffbc0b301fULL, 0xde5e2316b414da3eULL, 0xda6cfd9e4f96136eULL, 0x9e8c7ecd3cbfd45aULL,
    0xea4f758fd7cbe2f6ULL, 0x7a0e73ef14a525d4ULL, 0xd7f6bf623f1aba10ULL, 0xac06608df8f6d757ULL,
};

__device__ __forceinline__ uint64_t shr64(uint64_t v, unsigned s) { return s >= 64 ? 0 : v >> s; }
__device__ __forceinline__ uint64_t shl64(uint64_t v, unsigned s) { return s >= 64 ? 0 : v << s; }

// trigReduce(x) for x >= 2^29 (never reached below pi/4 from Sincos)
__device__ __noinline__ void go_trig_reduce(double x, uint32_t &j_out, double &z_out) {
    uint64_t ix = (uint64_t)__double_as_longlong(x);
    int exp = (int)((ix >> 52) & 0x7FF) - 1023 - 52;
    ix &= ~(0x7FFULL << 52);
    ix |= 1ULL << 52;
    unsigned digit = (unsigned)(exp + 61) / 64, bitshift = (unsigned)(exp + 61) % 64;
    uint64_t d0 = kMPi4[digit], d1 = kMPi4[digit + 1], d2 = kMPi4[digit + 2], d3 = kMPi4[digit + 3];
    uint64_t z0 = (d0 << bitshift) | shr64(d1, 64 - bitshift);
    uint64_t z1 = (d1 << bitshift) | shr64(d2, 64 - bitshift);
    uint64_t z2 = (d2 << bitshift) | shr64(d3, 64 - bitshift);
    uint64_t z2hi = __umul64hi(z2, ix);
    uint64_t z1hi = __umul64hi(z1, ix), z1lo = z1 * ix;
    uint64_t z0lo = z0 * ix;
    uint64_t lo = z1lo + z2hi;
    uint64_t c = lo < z1lo ? 1 : 0;
    uint64_t hi = z0lo + z1hi + c;
    uint32_t j = (uint32_t)(hi >> 61);
    hi = (hi << 3) | (lo >> 61);
    unsigned lz = hi == 0 ? 64u : (unsigned)__clzll((long long)hi);
    uint64_t e = (uint64_t)(1023 - (lz + 1));
    hi = shl64(hi, lz + 1) | shr64(lo, 64 - (lz + 1));
    hi >>= 64 - 52;
    hi |= e << 52;
    double z = __longlong_as_double((long long)hi);
    if (j & 1) {
        j++;
        j &= 7;
        z = __dsub_rn(z, 1.0);
    }
    j_out = j;
    z_out = __dmul_rn(z, 0.78539816339744830961566084581987572);
}

// The shared tail of math.Sincos: octant bookkeeping and the two polynomials on
// the reduced argument z of octant j.
__device__ __forceinline__ void go_sincos_tail(uint32_t j, double z, bool sin_sign, double &sn, double &cs) {
    bool cos_sign = false;
    if (j > 3) {
        j -= 4;
        sin_sign = !sin_sign;
        cos_sign = !cos_sign;
    }
    if (j > 1) cos_sign = !cos_sign;
    double zz = __dmul_rn(z, z);
    // cos = 1.0 - 0.5*zz + zz*zz*((((((c0*zz)+c1)*zz+c2)*zz+c3)*zz+c4)*zz+c5)
    double pc = __dmul_rn(-1.13585365213876817300e-11, zz);
    pc = __dmul_rn(__dadd_rn(pc, 2.08757008419747316778e-9), zz);
    pc = __dmul_rn(__dadd_rn(pc, -2.75573141792967388112e-7), zz);
    pc = __dmul_rn(__dadd_rn(pc, 2.48015872888517045348e-5), zz);
    pc = __dmul_rn(__dadd_rn(pc, -1.38888888888730564116e-3), zz);
    pc = __dadd_rn(pc, 4.16666666666665929218e-2);
    double c = __dadd_rn(__dsub_rn(1.0, __dmul_rn(0.5, zz)), __dmul_rn(__dmul_rn(zz, zz), pc));
    // sin = z + z*zz*((((((s0*zz)+s1)*zz+s2)*zz+s3)*zz+s4)*zz+s5)
    double ps = __dmul_rn(1.58962301576546568060e-10, zz);
    ps = __dmul_rn(__dadd_rn(ps, -2.50507477628578072866e-8), zz);
    ps = __dmul_rn(__dadd_rn(ps, 2.75573136213857245213e-6), zz);
    ps = __dmul_rn(__dadd_rn(ps, -1.98412698295895385996e-4), zz);
    ps = __dmul_rn(__dadd_rn(ps, 8.33333333332211858878e-3), zz);
    ps = __dadd_rn(ps, -1.66666666666666307295e-1);
    double s = __dadd_rn(z, __dmul_rn(__dmul_rn(z, zz), ps));
    if (j == 1 || j == 2) {
        double t = s;
        s = c;
        c = t;
    }
    sn = sin_sign ? -s : s;
    cs = cos_sign ? -c : c;
}

// |x| >= 2^29, Inf or NaN.  Inlined on purpose: making THIS an out-of-line call
// (measured) slows the hot loops by 15-35 % -- a call in the loop body forces the
// register allocator to keep the whole live state call-safe; only the integer
// Payne-Hanek core (go_trig_reduce) stays out of line.
__device__ __forceinline__ void go_sincos_large(double x, double &sn, double &cs) {
    const double ax = fabs(x);
    if (!(ax <= 1.7976931348623157e308)) {  // NaN or Inf
        sn = __longlong_as_double(0x7FF8000000000001LL);
        cs = sn;
        return;
    }
    uint32_t j;
    double z;
    go_trig_reduce(ax, j, z);
    go_sincos_tail(j, z, x < 0, sn, cs);
}

// math.Sincos(x) -> (sin, cos)
__device__ __forceinline__ void go_sincos(double x, double &sn, double &cs) {
    const double PI4A = 7.85398125648498535156e-1;
    const double PI4B = 3.77489470793079817668e-8;
    const double PI4C = 2.69515142907905952645e-15;
    const double M4PI = 1.27323954473516268615107010698;
    // Straight-line common path (|x| < 2^29): x == 0 returns (x, 1) through a select
    // at the end, and j fits an int32 (|x| * 4/pi < 2^31), so the float64 <-> uint64
    // conversion sequences of the Go source become single instructions with
    // identical values.
    const double ax = fabs(x);
    if (!(ax < 536870912.0)) {  // large, Inf, NaN
        go_sincos_large(x, sn, cs);
        return;
    }
    int32_t ji = __double2int_rz(__dmul_rn(ax, M4PI));
    ji += ji & 1;  // map zeros to origin: j++, y++ (y + 1 is exact)
    const double y = (double)ji;
    const double z = __dsub_rn(__dsub_rn(__dsub_rn(ax, __dmul_rn(y, PI4A)), __dmul_rn(y, PI4B)),
                               __dmul_rn(y, PI4C));
    go_sincos_tail((uint32_t)ji & 7, z, x < 0, sn, cs);
    if (x == 0.0) sn = x;  // +-0 in, +-0 out (cs is already exactly 1)
}

// complex64(math.Sincos(x)) -- the float32 pair stream/shifter.go:82 narrows the float64 results to -- for
// 2^-60 <= |x| < 2^29 or x = +-0, and whether that pair is CERTAIN to be math.Sincos' bit for bit.
// The reduction is Go's own (j, z = ((|x| - y PI4A) - y PI4B) - y PI4C: y PI4A and y PI4B are exact products, so
// the two fused steps round as Go's unfused ones do; the reduction is not relatively accurate near multiples of
// pi/4 -- 6e-9 off the true sine there -- so a cheaper one cannot stand in for it); the two Cephes polynomials are
// evaluated with fused Horner steps (16 float64 instructions where Go's unfused form has 28), which moves a result
// by at most a few units in its last place: 2^-52 relative at worst over 2e8 arguments up to 2^29, random and
// within 4 ulp of multiples of pi/4.  float32 drops mantissa bits 28..0 and rounds at their half point; a result
// whose dropped bits lie more than 128 units (2^-46 .. 2^-45 of its value: 64 times the worst difference seen) from
// that point narrows the same way as every float64 that close to it, math.Sincos' among them.  Elsewhere -- 2^-21
// of the components, 2^-20 of the factors -- the function returns false and the caller evaluates go_sincos.  The
// quadrant's exchange and signs are applied to the float32 pair (exact either side of the narrowing).
__device__ __forceinline__ bool sincos_narrow(double x, float &sn, float &cs) {
    const double PI4A = 7.85398125648498535156e-1, PI4B = 3.77489470793079817668e-8, PI4C = 2.69515142907905952645e-15;
    const double M4PI = 1.27323954473516268615107010698;
    const double ax = fabs(x);
    int32_t ji = __double2int_rz(__dmul_rn(ax, M4PI));
    ji += ji & 1;
    const double y = (double)ji;
    double z = __fma_rn(-y, PI4A, ax);
    z = __fma_rn(-y, PI4B, z);
    z = __dsub_rn(z, __dmul_rn(y, PI4C));
    const double zz = __dmul_rn(z, z);
    double ps = __fma_rn(1.58962301576546568060e-10, zz, -2.50507477628578072866e-8);
    ps = __fma_rn(ps, zz, 2.75573136213857245213e-6);
    ps = __fma_rn(ps, zz, -1.98412698295895385996e-4);
    ps = __fma_rn(ps, zz, 8.33333333332211858878e-3);
    ps = __fma_rn(ps, zz, -1.66666666666666307295e-1);
    const double s = __fma_rn(__dmul_rn(z, zz), ps, z);
    double pc = __fma_rn(-1.13585365213876817300e-11, zz, 2.08757008419747316778e-9);
    pc = __fma_rn(pc, zz, -2.75573141792967388112e-7);
    pc = __fma_rn(pc, zz, 2.48015872888517045348e-5);
    pc = __fma_rn(pc, zz, -1.38888888888730564116e-3);
    pc = __fma_rn(pc, zz, 4.16666666666665929218e-2);
    const double c = __fma_rn(__dmul_rn(zz, zz), pc, __fma_rn(-0.5, zz, 1.0));
    const unsigned ds = ((unsigned)__double2loint(s) + (128u - 0x10000000u)) & 0x1FFFFFFFu;
    const unsigned dc = ((unsigned)__double2loint(c) + (128u - 0x10000000u)) & 0x1FFFFFFFu;
    const float sf = (float)s, cf = (float)c;
    const bool swap = ji & 2;
    const float a = swap ? cf : sf, b = swap ? sf : cf;
    const unsigned t = (unsigned)ji << 29;  // bit 31 = quadrant bit 1, bit 30 = quadrant bit 0
    sn = __uint_as_float(__float_as_uint(a) ^ (((unsigned)__double2hiint(x) ^ t) & 0x80000000u));
    cs = __uint_as_float(__float_as_uint(b) ^ ((t ^ (t << 1)) & 0x80000000u));
    return ds > 256u && dc > 256u;
}

// complex64(math.Sincos(x)) for ANY x, bit for bit: sincos_narrow where it is certain (all but 2^-20 of the
// factors inside its range), the operation-for-operation go_sincos OUT OF LINE for the rest -- a lane that cannot
// decide calls it, the others wait.  Every reference-order Shift that is not the streaming map (which queues its
// undecided vectors instead: shift_exact_kernel) takes this: the in-order FIR chain, the chains with Decimate /
// Downsample / convolution terminals, generic programs, fix-up and history tasks.  Round 5: the in-order north-star
// chain ran the 45-instruction go_sincos on every input sample, 127.6 us per 2^24 samples.
__device__ __attribute__((noinline)) void go_sincos_slow32(double x, float &sn, float &cs) {
    double s, c;
    go_sincos(x, s, c);
    sn = (float)s, cs = (float)c;
}
__device__ __forceinline__ void go_sincos32(double x, float &sn, float &cs) {
    const double ax = fabs(x);
    // (sincos_narrow's range: 2^-60 <= |x| < 2^29, or +-0; NaN and infinities fail the comparison)
    const bool in_range = ax == 0.0 || (ax >= 8.67361737988403547206e-19 && ax < 536870912.0);
    float s1 = 0.f, c1 = 0.f;
    bool ok = in_range;
    if (in_range) ok = sincos_narrow(x, s1, c1);
    if (!ok) go_sincos_slow32(x, s1, c1);
    sn = s1, cs = c1;
}

// sin and cos of x, |x| < 2^30, for the LATE mixer (hz_chain_dev.h): the same Cephes kernels
// as math.Sincos on the same float64 argument, but with a two-fma Cody-Waite reduction by
// pi/2, fused Horner steps and branch-free quadrant logic -- a third of the issue slots of
// the operation-for-operation restatement above.  The float64 results agree with
// math.Sincos to ~1e-16, so after narrowing to float32 the mixer's factors are the same
// numbers except within ~1e-8 ulp of a rounding boundary; they multiply a filter output
// that is held to an error bound, not to bits (reference-order blocks, and every
// bit-exact operator, keep go_sincos).
__device__ __forceinline__ void sincos_late(double x, double &sn, double &cs) {
    const double k = rint(__dmul_rn(x, 0.63661977236758134308));  // x * 2/pi
    double r = __fma_rn(-k, 1.57079632679489655800e+00, x);
    r = __fma_rn(-k, 6.12323399573676603587e-17, r);
    const int q = __double2int_rn(k);
    const double z = __dmul_rn(r, r);
    double ps = __fma_rn(1.58962301576546568060e-10, z, -2.50507477628578072866e-8);
    ps = __fma_rn(ps, z, 2.75573136213857245213e-6);
    ps = __fma_rn(ps, z, -1.98412698295895385996e-4);
    ps = __fma_rn(ps, z, 8.33333333332211858878e-3);
    ps = __fma_rn(ps, z, -1.66666666666666307295e-1);
    const double s = __fma_rn(__dmul_rn(r, z), ps, r);
    double pc = __fma_rn(-1.13585365213876817300e-11, z, 2.08757008419747316778e-9);
    pc = __fma_rn(pc, z, -2.75573141792967388112e-7);
    pc = __fma_rn(pc, z, 2.48015872888517045348e-5);
    pc = __fma_rn(pc, z, -1.38888888888730564116e-3);
    pc = __fma_rn(pc, z, 4.16666666666665929218e-2);
    const double c = __fma_rn(__dmul_rn(z, z), pc, __fma_rn(-0.5, z, 1.0));
    const bool swap = q & 1;
    const double a = swap ? c : s, b = swap ? s : c;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

// sin and cos of 2 pi t / 2^32 (t: a phase in 2^-32 turns), float32 arithmetic only: within 0.8 ulp (0.30 ulp
// rms: a correctly rounded result has 0.29).  The nearest quarter turn comes off the integer phase exactly; the
// rest r, |r| <= 1/8 turn, keeps 24 bits in r and its last six in e; odd / even polynomials in r with the leading
// coefficients split, one rounding at the end of each.
__device__ __forceinline__ void sincos_turns32(uint32_t t, float &sn, float &cs) {
    const uint32_t q = (t + 0x20000000u) >> 30;
    const int32_t rf = (int32_t)(t - (q << 30));
    const float r = (float)(rf & ~63) * 2.3283064365386963e-10f;  // 2^-32: exact
    const float e = (float)(rf & 63) * 1.4629180792671596e-09f;   // 2 pi 2^-32
    const float zh = r * r, zl = __fmaf_rn(r, r, -zh);
    float ps = __fmaf_rn(zh, 42.058692932128906f, -76.70585632324219f);
    ps = __fmaf_rn(ps, zh, 81.6052474975586f);
    ps = __fmaf_rn(ps, zh, -41.34170150756836f);
    float pc = __fmaf_rn(zh, -26.42625617980957f, 60.2446403503418f);
    pc = __fmaf_rn(pc, zh, -85.45681762695312f);
    pc = __fmaf_rn(pc, zh, 64.93939208984375f);
    const float rin = r * __fmaf_rn(zh, ps, -1.7484555314695172e-07f);
    const float small = __fmaf_rn(zh, zh * pc, __fmaf_rn(zl, -19.739208221435547f, zh * -5.807431762150372e-07f));
    const float s0 = __fmaf_rn(r, 6.2831854820251465f, rin), c0 = __fmaf_rn(zh, -19.739208221435547f, small) + 1.0f;
    const float s = __fmaf_rn(r, 6.2831854820251465f, __fmaf_rn(e, c0, rin));
    const float c = __fmaf_rn(zh, -19.739208221435547f, __fmaf_rn(-e, s0, small)) + 1.0f;
    const float a = (q & 1) ? c : s, b = (q & 1) ? s : c;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}
// a float64 phase in radians (|x| < 2^30) as a 32-bit fraction of a turn (absolute error <= 2^-53 |x| / 2 pi
// turns from the product, 2^-33 from the rounding)
__device__ __forceinline__ uint32_t turns32(double x) {
    const double t = __dmul_rn(x, 0.15915494309189534561);
    const double f = __builtin_amdgcn_fract(t);  // [0, 1): one instruction (t - floor(t) is three)
    // (one conversion instruction; a fraction that rounds up to a whole turn saturates to 2^32 - 1: 2^-32 turns off)
    return __double2uint_rn(__dmul_rn(f, 4294967296.0));
}

// The same for any float64 phase the reference can form (the Payne-Hanek range of math.Sincos included): 1 / 2 pi as a
// double-double, the product's low part by fma -- 2^-106 |x| turns from the product instead of 2^-53 |x|, i.e. the
// fraction of a turn is good to 2^-33 for |x| up to 2^70.  Three float64 instructions more than turns32.
__device__ __forceinline__ uint32_t turns32_wide(double x) {
    const double hi = 0.15915494309189535, lo = -9.839338337591243e-18;  // hi + lo = 1 / 2 pi to 2^-110
    const double t = __dmul_rn(x, hi);
    const double e = __fma_rn(x, lo, __fma_rn(x, hi, -t));  // x / 2 pi - t
    const double f = __builtin_amdgcn_fract(t) + e;         // [0, 1) up to e: the conversion below wraps mod 2^32
    return (uint32_t)(int64_t)__double2ll_rn(__dmul_rn(f, 4294967296.0));
}

}  // namespace hz

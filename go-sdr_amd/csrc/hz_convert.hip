// hz_convert.hip -- format converters, lookup tables, Decimate, Downsample.
//
// All of these are HBM-bound byte/element maps: the kernels are grid-stride
// loops whose wide side moves 16 B per lane per instruction (a wave covers
// 1 KiB contiguous), with a scalar path for ragged heads/tails and for
// pointers that are only sample-aligned (Go sub-slices).
#include "hz_common.h"
#include "hz_device.h"

namespace hz {

// ---- component maps ---------------------------------------------------------
// Every converter is a per-component map over the flat array of 2n components.

enum Conv {
    U8_C64, U8_I8, U8_I16, I8_C64, I8_U8, I8_I16, I16_C64, I16_U8, I16_I8, C64_U8, C64_I16, C64_I8,
    I16_SHL  // ShiftLSBToMSBBits
};

template <int C> struct ConvTraits;
#define HZ_TRAITS(C, S, D) \
    template <> struct ConvTraits<C> { using src_t = S; using dst_t = D; }
HZ_TRAITS(U8_C64, uint8_t, float);
HZ_TRAITS(U8_I8, uint8_t, int8_t);
HZ_TRAITS(U8_I16, uint8_t, int16_t);
HZ_TRAITS(I8_C64, int8_t, float);
HZ_TRAITS(I8_U8, int8_t, uint8_t);
HZ_TRAITS(I8_I16, int8_t, int16_t);
HZ_TRAITS(I16_C64, int16_t, float);
HZ_TRAITS(I16_U8, int16_t, uint8_t);
HZ_TRAITS(I16_I8, int16_t, int8_t);
HZ_TRAITS(C64_U8, float, uint8_t);
HZ_TRAITS(C64_I16, float, int16_t);
HZ_TRAITS(C64_I8, float, int8_t);
HZ_TRAITS(I16_SHL, int16_t, int16_t);

template <int C>
__device__ __forceinline__ typename ConvTraits<C>::dst_t conv1(typename ConvTraits<C>::src_t v,
                                                                int arg) {
    if constexpr (C == U8_C64) return u8_to_f32(v);
    if constexpr (C == U8_I8) return (int8_t)((int)v - 128);                       // iq_u8.go:93-96
    if constexpr (C == U8_I16) return (int16_t)(((int)v << 8) - 32768);            // iq_u8.go:79-82
    if constexpr (C == I8_C64) return i8_to_f32(v);
    if constexpr (C == I8_U8) return (uint8_t)((int)v + 128);                      // iq_i8.go:89-92
    if constexpr (C == I8_I16) return (int16_t)((uint32_t)(int)v << 8);            // iq_i8.go:75-78
    if constexpr (C == I16_C64) return i16_to_f32(v);
    if constexpr (C == I16_U8) return (uint8_t)((uint32_t)(((int)v + 32768) & 0xFFFF) >> 8);  // iq_i16.go:121-124
    if constexpr (C == I16_I8) return (int8_t)((int)v >> 8);                       // iq_i16.go:155-158
    if constexpr (C == C64_U8) return (uint8_t)f32_to_u8(v);
    if constexpr (C == C64_I16) return (int16_t)f32_to_i16(v);
    if constexpr (C == C64_I8) return (int8_t)f32_to_i8(v);
    if constexpr (C == I16_SHL) return (int16_t)((uint32_t)(uint16_t)v << arg);    // iq_i16.go:106-109
}

// Foreign byte order on either side (bytes_io.go:30-64, :150-197): SW bit 0 = the
// source components arrive byte-swapped, bit 1 = the destination wants them swapped.
template <class T> __device__ __forceinline__ T swap_bytes(T v) {
    if constexpr (sizeof(T) == 2) {
        const uint16_t u = (uint16_t)v;
        return (T)(uint16_t)((u << 8) | (u >> 8));
    } else if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_bswap32(__builtin_bit_cast(uint32_t, v)));
    } else {
        return v;
    }
}
template <int C, int SW>
__device__ __forceinline__ typename ConvTraits<C>::dst_t conv1s(typename ConvTraits<C>::src_t v, int arg) {
    if constexpr ((SW & 1) != 0) v = swap_bytes(v);
    auto r = conv1<C>(v, arg);
    if constexpr ((SW & 2) != 0) r = swap_bytes(r);
    return r;
}

template <class T, int N> struct alignas(sizeof(T) * N) Vec {
    T v[N];
};

// K components per lane per step, chosen so the wider side is 16 B.
template <int C> struct ConvGeom {
    using S = typename ConvTraits<C>::src_t;
    using D = typename ConvTraits<C>::dst_t;
    static constexpr int wide = sizeof(S) > sizeof(D) ? sizeof(S) : sizeof(D);
    static constexpr int K = 16 / wide;
};

template <int C, int SW, bool NT = false>
__global__ __launch_bounds__(kThreads) void convert_vec_kernel(
    const typename ConvTraits<C>::src_t *__restrict__ src,
    typename ConvTraits<C>::dst_t *__restrict__ dst, size_t nvec, int arg) {
    using G = ConvGeom<C>;
    using SV = Vec<typename G::S, G::K>;
    using DV = Vec<typename G::D, G::K>;
    const SV *s = reinterpret_cast<const SV *>(src);
    DV *d = reinterpret_cast<DV *>(dst);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // 4 independent vectors in flight per lane
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        SV a0 = ld_stream<NT>(s + i), a1 = ld_stream<NT>(s + i + stride), a2 = ld_stream<NT>(s + i + 2 * stride), a3 = ld_stream<NT>(s + i + 3 * stride);
        DV r0, r1, r2, r3;
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            r0.v[k] = conv1s<C, SW>(a0.v[k], arg);
            r1.v[k] = conv1s<C, SW>(a1.v[k], arg);
            r2.v[k] = conv1s<C, SW>(a2.v[k], arg);
            r3.v[k] = conv1s<C, SW>(a3.v[k], arg);
        }
        st_stream<NT>(d + i, r0);
        st_stream<NT>(d + i + stride, r1);
        st_stream<NT>(d + i + 2 * stride, r2);
        st_stream<NT>(d + i + 3 * stride, r3);
    }
    for (; i < nvec; i += stride) {
        SV a = ld_stream<NT>(s + i);
        DV r;
#pragma unroll
        for (int k = 0; k < G::K; k++) r.v[k] = conv1s<C, SW>(a.v[k], arg);
        st_stream<NT>(d + i, r);
    }
}

// The form for calls that stream past the cache: a workgroup's tile is U runs of 256 vectors, a lane's U loads in flight
// before its first store, non-temporal both ways, one tile per workgroup -- u8 -> c64 reads 4 bytes per lane and load:
// with ONE such load per wave in flight the chip's resident waves hold 2 MB of reads, the rate of a 2 us latency at
// 1 TB/s, which is what the kernel ran at (tools/stream_rate.hip over a rotation of buffers: 31.6 us; U = 2: 28.0 us =
// 6.0 TB/s, U = 4 28.5, U = 8 30.8).
template <int C, int SW, int U>
__global__ __launch_bounds__(kThreads) void convert_tile_kernel(const typename ConvTraits<C>::src_t *__restrict__ src,
                                                                typename ConvTraits<C>::dst_t *__restrict__ dst, int arg) {
    using G = ConvGeom<C>;
    using SV = Vec<typename G::S, G::K>;
    using DV = Vec<typename G::D, G::K>;
    const SV *s = reinterpret_cast<const SV *>(src) + (size_t)blockIdx.x * (kThreads * U) + threadIdx.x;
    DV *d = reinterpret_cast<DV *>(dst) + (size_t)blockIdx.x * (kThreads * U) + threadIdx.x;
    SV a[U];
#pragma unroll
    for (int u = 0; u < U; u++) a[u] = ld_stream<true>(s + u * kThreads);
#pragma unroll
    for (int u = 0; u < U; u++) {
        DV r;
#pragma unroll
        for (int k = 0; k < G::K; k++) r.v[k] = conv1s<C, SW>(a[u].v[k], arg);
        st_stream<true>(d + u * kThreads, r);
    }
}

template <int C, int SW>
__global__ __launch_bounds__(kThreads) void convert_scalar_kernel(
    const typename ConvTraits<C>::src_t *__restrict__ src,
    typename ConvTraits<C>::dst_t *__restrict__ dst, size_t ncomp, int arg) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncomp; i += stride)
        dst[i] = conv1s<C, SW>(src[i], arg);
}

// Launch one converter over ncomp components: vector body where both pointers
// are 16 B / K-aligned, scalar kernel for the tail or for unaligned slices.
template <int C, int SW = 0>
static void launch_convert(hzsdr_ctx *ctx, const void *src, void *dst, size_t ncomp, int arg = 0) {
    using G = ConvGeom<C>;
    using S = typename G::S;
    using D = typename G::D;
    if (ncomp == 0) return;
    const S *s = (const S *)src;
    D *d = (D *)dst;
    const bool aligned = ((uintptr_t)s % (sizeof(S) * G::K) == 0) && ((uintptr_t)d % (sizeof(D) * G::K) == 0);
    size_t nvec = aligned ? ncomp / G::K : 0;
    if (nvec) {
        // (one vector per lane up to the grid's cap of 128 workgroups per CU, the kernel's four-deep loop only beyond:
        // u8 -> c64 over 2^24 samples 26.7 us instead of 27.5 from the cache, 35.5 instead of 36.7 from HBM)
        // (a call of 96 MiB or more streams past the memory-side cache: hz_device.h)
        if (streams_past_cache(ncomp * (sizeof(S) + sizeof(D)))) {
            constexpr int U = 2;
            const size_t tiles = nvec / ((size_t)kThreads * U), rest = nvec - tiles * kThreads * U;
            hipLaunchKernelGGL((convert_tile_kernel<C, SW, U>), dim3((unsigned)tiles), dim3(kThreads), 0, ctx->stream, s, d, arg);
            if (rest)
                hipLaunchKernelGGL((convert_vec_kernel<C, SW, true>), dim3(blocks_for(ctx, rest)), dim3(kThreads), 0, ctx->stream,
                                   s + tiles * kThreads * U * G::K, d + tiles * kThreads * U * G::K, rest, arg);
        }
        else
            hipLaunchKernelGGL((convert_vec_kernel<C, SW>), dim3(blocks_for(ctx, nvec)), dim3(kThreads),
                               0, ctx->stream, s, d, nvec, arg);
    }
    size_t done = nvec * G::K;
    if (done < ncomp) {
        hipLaunchKernelGGL((convert_scalar_kernel<C, SW>), dim3(blocks_for(ctx, ncomp - done)),
                           dim3(kThreads), 0, ctx->stream, s + done, d + done, ncomp - done, arg);
    }
}

// Dispatch (src_format, dst_format) -> kernel; formats already validated.  `sw`: see conv1s.
template <int C>
static void launch_convert_sw(hzsdr_ctx *ctx, const void *src, void *dst, size_t nc, int sw) {
    switch (sw & 3) {
    case 0: launch_convert<C, 0>(ctx, src, dst, nc); break;
    case 1: launch_convert<C, 1>(ctx, src, dst, nc); break;
    case 2: launch_convert<C, 2>(ctx, src, dst, nc); break;
    default: launch_convert<C, 3>(ctx, src, dst, nc); break;
    }
}

int convert_device(hzsdr_ctx *ctx, int dst_fmt, void *dst, int src_fmt, const void *src, size_t n, int sw = 0) {
    const size_t nc = 2 * n;
    switch (src_fmt * 8 + dst_fmt) {
    case HZSDR_FMT_U8 * 8 + HZSDR_FMT_C64: launch_convert_sw<U8_C64>(ctx, src, dst, nc, sw); break;
    case HZSDR_FMT_U8 * 8 + HZSDR_FMT_I8: launch_convert_sw<U8_I8>(ctx, src, dst, nc, 0); break;
    case HZSDR_FMT_U8 * 8 + HZSDR_FMT_I16: launch_convert_sw<U8_I16>(ctx, src, dst, nc, sw); break;
    case HZSDR_FMT_I8 * 8 + HZSDR_FMT_C64: launch_convert_sw<I8_C64>(ctx, src, dst, nc, sw); break;
    case HZSDR_FMT_I8 * 8 + HZSDR_FMT_U8: launch_convert_sw<I8_U8>(ctx, src, dst, nc, 0); break;
    case HZSDR_FMT_I8 * 8 + HZSDR_FMT_I16: launch_convert_sw<I8_I16>(ctx, src, dst, nc, sw); break;
    case HZSDR_FMT_I16 * 8 + HZSDR_FMT_C64: launch_convert_sw<I16_C64>(ctx, src, dst, nc, sw); break;
    case HZSDR_FMT_I16 * 8 + HZSDR_FMT_U8: launch_convert_sw<I16_U8>(ctx, src, dst, nc, sw); break;
    case HZSDR_FMT_I16 * 8 + HZSDR_FMT_I8: launch_convert_sw<I16_I8>(ctx, src, dst, nc, sw); break;
    case HZSDR_FMT_C64 * 8 + HZSDR_FMT_U8: launch_convert_sw<C64_U8>(ctx, src, dst, nc, sw); break;
    case HZSDR_FMT_C64 * 8 + HZSDR_FMT_I16: launch_convert_sw<C64_I16>(ctx, src, dst, nc, sw); break;
    case HZSDR_FMT_C64 * 8 + HZSDR_FMT_I8: launch_convert_sw<C64_I8>(ctx, src, dst, nc, sw); break;
    default: return HZSDR_ERR_CONVERSION_NOT_IMPLEMENTED;
    }
    return HZSDR_OK;
}

// ---- CopySamples (copy.go:31-52) between device buffers ------------------------------------
// A call that streams past the memory-side cache is the library's own copy -- two 16-byte vectors per lane and trip,
// non-temporal both ways: 6.2 TB/s from HBM where hipMemcpyAsync and torch's copy_ reach 5.0-5.4 (tools/copy_rate.hip)
// -- and with that the yardstick bench.py quotes the HBM-bound kernels against.
__global__ __launch_bounds__(kThreads) void copy_stream_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t nvec) {
    constexpr int U = 2;
    const size_t tile = (size_t)kThreads * U;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < nvec; t0 += (size_t)gridDim.x * tile) {
        uint4 a[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * kThreads + threadIdx.x;
            if (i < nvec) a[u] = nt_load(src + i);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = t0 + (size_t)u * kThreads + threadIdx.x;
            if (i < nvec) nt_store(dst + i, a[u]);
        }
    }
}

// ---- lookup-table gather ------------------------------------------------------

// dst[i] = tab[idx(src[i])]; E = element type of one table entry / output sample.
// UMUL255 selects stream.Multiply's private u8 index I*255 + Q
// (stream/multiply.go:106-108) instead of the raw little-endian uint16.
template <class E, bool UMUL255>
__global__ __launch_bounds__(kThreads) void lut_kernel(const uint16_t *__restrict__ src,
                                                       const E *__restrict__ tab, E *dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t raw = src[i];
        uint32_t idx = UMUL255 ? (raw & 0xFF) * 255u + (raw >> 8) : raw;
        dst[i] = tab[idx];
    }
}

// 4 samples per lane: one 8-byte source load, four gathers, vector store
template <class E, bool UMUL255>
__global__ __launch_bounds__(kThreads) void lut_kernel_x4(const uint2 *__restrict__ src,
                                                          const E *__restrict__ tab,
                                                          Vec<E, 4> *dst, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        uint2 raw = src[i];
        uint32_t r[4] = {raw.x & 0xFFFF, raw.x >> 16, raw.y & 0xFFFF, raw.y >> 16};
        Vec<E, 4> o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t idx = UMUL255 ? (r[k] & 0xFF) * 255u + (r[k] >> 8) : r[k];
            o.v[k] = tab[idx];
        }
        dst[i] = o;
    }
}

template <class E, bool UMUL255>
static void launch_lut(hzsdr_ctx *ctx, const void *src, const void *tab, void *dst, size_t n) {
    if (n == 0) return;
    const bool aligned = ((uintptr_t)src % 8 == 0) && ((uintptr_t)dst % (sizeof(E) * 4) == 0);
    size_t nvec = aligned ? n / 4 : 0;
    if (nvec)
        hipLaunchKernelGGL((lut_kernel_x4<E, UMUL255>), dim3(blocks_for(ctx, nvec)), dim3(kThreads), 0,
                           ctx->stream, (const uint2 *)src, (const E *)tab, (Vec<E, 4> *)dst, nvec);
    size_t done = nvec * 4;
    if (done < n)
        hipLaunchKernelGGL((lut_kernel<E, UMUL255>), dim3(blocks_for(ctx, n - done)), dim3(kThreads), 0,
                           ctx->stream, (const uint16_t *)src + done, (const E *)tab, (E *)dst + done,
                           n - done);
}

static void lut_device(hzsdr_ctx *ctx, int dst_fmt, const void *src, const void *tab, void *dst,
                       size_t n) {
    switch (format_size(dst_fmt)) {
    case 2: launch_lut<uint16_t, false>(ctx, src, tab, dst, n); break;
    case 4: launch_lut<uint32_t, false>(ctx, src, tab, dst, n); break;
    default: launch_lut<uint2, false>(ctx, src, tab, dst, n); break;
    }
}

// ---- Decimate -----------------------------------------------------------------

// to[i] = from[factor * i]  (stream/decimate.go:84-98)
template <class E>
__global__ __launch_bounds__(kThreads) void decimate_kernel(const E *__restrict__ from, E *to,
                                                            size_t count, size_t factor) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
        to[i] = from[i * factor];
}

void decimate_device(hzsdr_ctx *ctx, int fmt, const void *from, void *to, size_t count,
                     size_t factor) {
    if (count == 0) return;
    dim3 g(blocks_for(ctx, count)), b(kThreads);
    switch (format_size(fmt)) {
    case 2: hipLaunchKernelGGL(decimate_kernel<uint16_t>, g, b, 0, ctx->stream, (const uint16_t *)from, (uint16_t *)to, count, factor); break;
    case 4: hipLaunchKernelGGL(decimate_kernel<uint32_t>, g, b, 0, ctx->stream, (const uint32_t *)from, (uint32_t *)to, count, factor); break;
    default: hipLaunchKernelGGL(decimate_kernel<uint2>, g, b, 0, ctx->stream, (const uint2 *)from, (uint2 *)to, count, factor); break;
    }
}

// ---- Downsample -----------------------------------------------------------------

template <int FMT> struct SrcSample;
template <> struct SrcSample<HZSDR_FMT_U8> {
    using raw_t = uint16_t;
    static __device__ __forceinline__ float2 cvt(uint16_t r) { return make_float2(u8_to_f32(r & 0xFF), u8_to_f32(r >> 8)); }
};
template <> struct SrcSample<HZSDR_FMT_I8> {
    using raw_t = uint16_t;
    static __device__ __forceinline__ float2 cvt(uint16_t r) { return make_float2(i8_to_f32((int8_t)(r & 0xFF)), i8_to_f32((int8_t)(r >> 8))); }
};
template <> struct SrcSample<HZSDR_FMT_I16> {
    using raw_t = uint32_t;
    static __device__ __forceinline__ float2 cvt(uint32_t r) { return make_float2(i16_to_f32((int16_t)(r & 0xFFFF)), i16_to_f32((int16_t)(r >> 16))); }
};
template <> struct SrcSample<HZSDR_FMT_C64> {
    using raw_t = float2;
    static __device__ __forceinline__ float2 cvt(float2 r) { return r; }
};

// One output per lane: convert `factor` consecutive samples, accumulate in
// order from +0, divide by float32(factor)  (stream/downsample.go:99-124).
// W raw samples are fetched per vector load (W * sizeof(raw) = 16 B) when the
// window geometry allows (factor % W == 0 and 16-B aligned source).
template <int FMT, int W, bool NT = false, int G = 1>
__global__ __launch_bounds__(kThreads) void downsample_kernel(
    const typename SrcSample<FMT>::raw_t *__restrict__ from, float2 *to, size_t count,
    unsigned factor) {
    using R = typename SrcSample<FMT>::raw_t;
    const float div = (float)factor;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const R *w = from + i * factor;
        float sr = 0.0f, si = 0.0f;
        if constexpr (W > 1) {
            // G vectors of a window are in flight together: a lane's window is one run of bytes, and a non-temporal
            // load leaves nothing of its cache line behind for the next one -- issued one by one the second vector of
            // a 32-byte window fetched its line AGAIN (FETCH_SIZE 1.39x the input, 25 us where the cached form took 19)
            const Vec<R, W> *wv = reinterpret_cast<const Vec<R, W> *>(w);
            for (unsigned j = 0; j < factor / W; j += G) {
                Vec<R, W> v[G];
#pragma unroll
                for (int g = 0; g < G; g++) v[g] = ld_stream<NT>(wv + j + g);
#pragma unroll
                for (int g = 0; g < G; g++)
#pragma unroll
                    for (int k = 0; k < W; k++) {
                        float2 c = SrcSample<FMT>::cvt(v[g].v[k]);
                        sr = __fadd_rn(sr, c.x);
                        si = __fadd_rn(si, c.y);
                    }
            }
        } else {
            for (unsigned j = 0; j < factor; j++) {
                float2 c = SrcSample<FMT>::cvt(w[j]);
                sr = __fadd_rn(sr, c.x);
                si = __fadd_rn(si, c.y);
            }
        }
        st_stream<NT>(to + i, make_float2(__fdiv_rn(sr, div), __fdiv_rn(si, div)));
    }
}

// Round 6, the large-call form for windows of exactly G vectors (i16 / 8: two; u8 / 8: one; i16 / 16: four): a lane
// holds M windows at a time -- i, i + stride, ... as the in-place maps do (hz_vector.hip) -- and ALL their vectors are in
// flight before the first sum: M G x 16 bytes per lane at once instead of G x 16, a quarter of the workgroups.  A
// 15 us kernel spends 2-3 us getting its first bytes; more of them per lane at that moment looked like what was left to
// gain.  MEASURED AND LEFT OFF (HZ_DOWNSAMPLE_M = 1 launches the one-window kernel below as before;
// tools/downsample_time.py over a rotation of six buffer pairs, profiles/r06_downsample_time.txt): i16 / 8 per call
// 17.6-17.8 us with one window per lane, 18.2-18.3 with two, 18.4-18.5 with four or eight (back to back 15.4-15.6 /
// 16.0 / 16.3).  The sums are the same additions in the same order.
#ifndef HZ_DOWNSAMPLE_M
#define HZ_DOWNSAMPLE_M 1
#endif
template <int FMT, int W, int G, int M>
__global__ __launch_bounds__(kThreads) void downsample_multi_kernel(
    const typename SrcSample<FMT>::raw_t *__restrict__ from, float2 *to, size_t count) {
    using R = typename SrcSample<FMT>::raw_t;
    constexpr unsigned factor = W * G;
    const float div = (float)factor;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto finish = [&](const Vec<R, W>(&v)[G], size_t o) {
        float sr = 0.0f, si = 0.0f;
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int k = 0; k < W; k++) {
                float2 c = SrcSample<FMT>::cvt(v[g].v[k]);
                sr = __fadd_rn(sr, c.x);
                si = __fadd_rn(si, c.y);
            }
        st_stream<true>(to + o, make_float2(__fdiv_rn(sr, div), __fdiv_rn(si, div)));
    };
    for (; i + (M - 1) * stride < count; i += M * stride) {
        Vec<R, W> v[M][G];
#pragma unroll
        for (int m = 0; m < M; m++)
#pragma unroll
            for (int g = 0; g < G; g++) v[m][g] = ld_stream<true>(reinterpret_cast<const Vec<R, W> *>(from + (i + m * stride) * factor) + g);
#pragma unroll
        for (int m = 0; m < M; m++) finish(v[m], i + m * stride);
    }
    for (; i < count; i += stride) {
        Vec<R, W> v[G];
#pragma unroll
        for (int g = 0; g < G; g++) v[g] = ld_stream<true>(reinterpret_cast<const Vec<R, W> *>(from + i * factor) + g);
        finish(v, i);
    }
}

template <int FMT>
static void launch_downsample(hzsdr_ctx *ctx, const void *from, void *to, size_t count,
                              unsigned factor) {
    using R = typename SrcSample<FMT>::raw_t;
    constexpr int W = 16 / sizeof(R);
    if (count == 0) return;
    dim3 g(blocks_for(ctx, count)), b(kThreads);
    // (from 64 MiB on here: a decimating kernel's call is mostly INPUT, and a stream's next buffer of that size has
    // pushed this one out of the 256 MB cache long before its turn comes again -- 2^24 i16 samples by 8 are 80 MiB)
    const bool vec = factor % W == 0 && (uintptr_t)from % 16 == 0;
    const unsigned nv = factor / W;  // 16-byte vectors per window
    const bool past = count * ((size_t)factor * sizeof(R) + 8) >= ((size_t)64 << 20);
    // non-temporal only where a window's vectors travel together (see the kernel): windows of 1, 2 or 4k vectors
    constexpr int M = HZ_DOWNSAMPLE_M;
    const dim3 gm(blocks_for(ctx, (count + M - 1) / M));
    if (M > 1 && vec && past && nv == 4)
        hipLaunchKernelGGL((downsample_multi_kernel<FMT, W, 4, (M > 2 ? M / 2 : 1)>), dim3(blocks_for(ctx, (count + M / 2 - 1) / (M > 2 ? M / 2 : 1))), b, 0, ctx->stream, (const R *)from, (float2 *)to, count);
    else if (M > 1 && vec && past && nv == 2)
        hipLaunchKernelGGL((downsample_multi_kernel<FMT, W, 2, M>), gm, b, 0, ctx->stream, (const R *)from, (float2 *)to, count);
    else if (M > 1 && vec && past && nv == 1)
        hipLaunchKernelGGL((downsample_multi_kernel<FMT, W, 1, M>), gm, b, 0, ctx->stream, (const R *)from, (float2 *)to, count);
    else if (vec && past && nv % 4 == 0)
        hipLaunchKernelGGL((downsample_kernel<FMT, W, true, 4>), g, b, 0, ctx->stream, (const R *)from, (float2 *)to, count, factor);
    else if (vec && past && nv == 2)
        hipLaunchKernelGGL((downsample_kernel<FMT, W, true, 2>), g, b, 0, ctx->stream, (const R *)from, (float2 *)to, count, factor);
    else if (vec && past && nv == 1)
        hipLaunchKernelGGL((downsample_kernel<FMT, W, true, 1>), g, b, 0, ctx->stream, (const R *)from, (float2 *)to, count, factor);
    else if (vec && nv % 2 == 0)
        hipLaunchKernelGGL((downsample_kernel<FMT, W, false, 2>), g, b, 0, ctx->stream, (const R *)from, (float2 *)to, count, factor);
    else if (vec)
        hipLaunchKernelGGL((downsample_kernel<FMT, W>), g, b, 0, ctx->stream, (const R *)from, (float2 *)to, count, factor);
    else
        hipLaunchKernelGGL((downsample_kernel<FMT, 1>), g, b, 0, ctx->stream, (const R *)from, (float2 *)to, count, factor);
}

void downsample_device(hzsdr_ctx *ctx, int from_fmt, const void *from, void *to, size_t count,
                       unsigned factor) {
    switch (from_fmt) {
    case HZSDR_FMT_U8: launch_downsample<HZSDR_FMT_U8>(ctx, from, to, count, factor); break;
    case HZSDR_FMT_I16: launch_downsample<HZSDR_FMT_I16>(ctx, from, to, count, factor); break;
    default: launch_downsample<HZSDR_FMT_C64>(ctx, from, to, count, factor); break;
    }
}

// used by hz_vector.hip (rotate) -- declared there
void rotate_device(hzsdr_ctx *ctx, void *buf, size_t n, float re, float im);

}  // namespace hz

// =============================================================================
// C ABI
// =============================================================================

struct hzsdr_lut {
    hzsdr_ctx *ctx;
    int src_fmt, dst_fmt;
    void *tab;  // device, 65536 entries
};

struct hzsdr_rotlut {
    hzsdr_ctx *ctx;
    int fmt;
    void *tab;     // device: u8 -> 65535 entries (I*255+Q), i8 -> 65536 entries
    float *cbuf;   // device scratch for the c64 round trip
};

extern "C" {

int hzsdr_convert(hzsdr_ctx *ctx, int dst_format, void *dst, size_t dst_len, int src_format,
                  const void *src, size_t src_len, size_t *n_out) {
    using namespace hz;
    if (n_out) *n_out = 0;
    if (!ctx) return HZSDR_ERR_INVALID_ARGUMENT;
    const int ss = format_size(src_format), ds = format_size(dst_format);
    if (ss == 0 || ds == 0) return fail(ctx, HZSDR_ERR_FORMAT_UNKNOWN, "convert: unknown format");
    size_t n = src_len;
    if (src_format == dst_format) {
        if (dst_len < n) n = dst_len;  // CopySamples -> copy(dst, src), copy.go:31-52
    } else if (src_len > dst_len) {
        return fail(ctx, HZSDR_ERR_DST_TOO_SMALL, "convert: dst shorter than src");  // conv.go:60-62
    }
    if (n && (!src || !dst)) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    if (src_format == dst_format) {
        hipMemcpyKind kind = ctx->memspace == HZSDR_MEM_HOST ? hipMemcpyHostToHost : hipMemcpyDeviceToDevice;
        if (ctx->memspace == HZSDR_MEM_HOST) {
            memmove(dst, src, n * ss);
        } else {
            const size_t bytes = n * (size_t)ss;
            const char *sb = (const char *)src;
            char *db = (char *)dst;
            const bool apart = sb + bytes <= db || db + bytes <= sb;  // (copy() allows overlap: the runtime's copy then)
            if (apart && streams_past_cache(2 * bytes) && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
                const size_t nvec = bytes / 16;
                const size_t tiles = (nvec + 2 * kThreads - 1) / (2 * kThreads);
                hipLaunchKernelGGL(copy_stream_kernel, dim3((unsigned)std::min<size_t>(tiles, (size_t)1 << 20)), dim3(kThreads), 0, ctx->stream,
                                   (const uint4 *)src, (uint4 *)dst, nvec);
                if (bytes % 16)
                    HZ_HIP(ctx, hipMemcpyAsync(db + nvec * 16, sb + nvec * 16, bytes % 16, kind, ctx->stream));
                HZ_HIP(ctx, hipGetLastError());
            } else {
                HZ_HIP(ctx, hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
            }
        }
        if (n_out) *n_out = n;
        return HZSDR_OK;
    }
    Stage st(ctx);
    const void *dsrc;
    void *ddst;
    HZ_TRY(st.in(0, src, n * ss, &dsrc));
    HZ_TRY(st.out(1, dst, n * ds, &ddst));
    HZ_TRY(convert_device(ctx, dst_format, ddst, src_format, dsrc, n));
    HZ_TRY(st.finish());
    if (n_out) *n_out = n;
    return HZSDR_OK;
}

int hzsdr_convert_foreign(hzsdr_ctx *ctx, int dst_format, void *dst, size_t dst_len, int dst_foreign,
                          int src_format, const void *src, size_t src_len, int src_foreign, size_t *n_out) {
    using namespace hz;
    if (n_out) *n_out = 0;
    if (!ctx) return HZSDR_ERR_INVALID_ARGUMENT;
    const int ss = format_size(src_format), ds = format_size(dst_format);
    if (ss == 0 || ds == 0) return fail(ctx, HZSDR_ERR_FORMAT_UNKNOWN, "convert: unknown format");
    if (src_format == dst_format)
        return fail(ctx, HZSDR_ERR_CONVERSION_NOT_IMPLEMENTED,
                    "convert_foreign: same format on both sides is a copy (+ hzsdr_byteswap)");
    if (src_len > dst_len) return fail(ctx, HZSDR_ERR_DST_TOO_SMALL, "convert: dst shorter than src");
    const size_t n = src_len;
    if (n && (!src || !dst)) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    Stage st(ctx);
    const void *dsrc;
    void *ddst;
    HZ_TRY(st.in(0, src, n * ss, &dsrc));
    HZ_TRY(st.out(1, dst, n * ds, &ddst));
    HZ_TRY(convert_device(ctx, dst_format, ddst, src_format, dsrc, n, (src_foreign ? 1 : 0) | (dst_foreign ? 2 : 0)));
    HZ_TRY(st.finish());
    if (n_out) *n_out = n;
    return HZSDR_OK;
}

int hzsdr_i16_shift_lsb_to_msb(hzsdr_ctx *ctx, void *buf, size_t n, int bits) {
    using namespace hz;
    if (!ctx || (n && !buf)) return HZSDR_ERR_INVALID_ARGUMENT;
    if (bits < 0 || bits > 16) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "shift: bits outside 0..16");
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    Stage st(ctx);
    void *d;
    HZ_TRY(st.inout(0, buf, n * 4, &d));
    launch_convert<I16_SHL>(ctx, d, d, 2 * n, 16 - bits);
    return st.finish();
}

int hzsdr_lut_create(hzsdr_ctx *ctx, int src_format, int dst_format, const void *table,
                     size_t table_len, hzsdr_lut **out) {
    using namespace hz;
    if (!ctx || !out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    const int ds = format_size(dst_format);
    if (ds == 0) return fail(ctx, HZSDR_ERR_FORMAT_UNKNOWN, "lut: unknown table format");
    if (table_len != 65536 || !table)  // iq_lookup_table.go:107-109
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "lut: table must be exactly 65536 samples");
    if (src_format != HZSDR_FMT_U8 && src_format != HZSDR_FMT_I8)  // :111-116
        return fail(ctx, HZSDR_ERR_FORMAT_UNKNOWN, "lut: input format must be u8 or i8");
    HZ_TRY(enter(ctx));
    void *tab = nullptr;
    HZ_HIP(ctx, hipMalloc(&tab, (size_t)65536 * ds));
    hipMemcpyKind kind = ctx->memspace == HZSDR_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    hipError_t e = hipMemcpyAsync(tab, table, (size_t)65536 * ds, kind, ctx->stream);
    if (e == hipSuccess && ctx->memspace == HZSDR_MEM_HOST) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(tab);
        return hip_fail(ctx, e, "lut table upload", __FILE__, __LINE__);
    }
    *out = new hzsdr_lut{ctx, src_format, dst_format, tab};
    return HZSDR_OK;
}

int hzsdr_lut_lookup(hzsdr_lut *lut, int dst_format, void *dst, size_t dst_len, int src_format,
                     const void *src, size_t src_len, size_t *n_out) {
    using namespace hz;
    if (n_out) *n_out = 0;
    if (!lut) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = lut->ctx;
    if (dst_format != lut->dst_fmt)  // iq_lookup_table.go:130-132
        return fail(ctx, HZSDR_ERR_FORMAT_MISMATCH, "lut: dst format differs from the table's");
    if (dst_len < src_len) return fail(ctx, HZSDR_ERR_DST_TOO_SMALL, "lut: dst shorter than src");  // :133-135
    if (src_format != lut->src_fmt) return fail(ctx, HZSDR_ERR_FORMAT_MISMATCH, "lut: src format differs");
    if (src_len && (!src || !dst)) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (src_len == 0) return HZSDR_OK;
    const int ds = format_size(dst_format);
    Stage st(ctx);
    const void *dsrc;
    void *ddst;
    HZ_TRY(st.in(0, src, src_len * 2, &dsrc));
    HZ_TRY(st.out(1, dst, src_len * ds, &ddst));
    lut_device(ctx, dst_format, dsrc, lut->tab, ddst, src_len);
    HZ_TRY(st.finish());
    if (n_out) *n_out = src_len;
    return HZSDR_OK;
}

int hzsdr_lut_free(hzsdr_lut *lut) {
    if (!lut) return HZSDR_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(lut->ctx->device);
    (void)hipStreamSynchronize(lut->ctx->stream);
    (void)hipFree(lut->tab);
    delete lut;
    return HZSDR_OK;
}

// ---- stream.Multiply for u8 / i8 ---------------------------------------------

int hzsdr_rotlut_set_multiplier(hzsdr_rotlut *t, float re, float im) {
    using namespace hz;
    if (!t) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = t->ctx;
    HZ_TRY(enter(ctx));
    // Host builds only the KEY table (which samples sit at which index, with the
    // reference's fill order); the arithmetic runs through the same GPU
    // converter / rotate kernels a caller would use (stream/multiply.go:165-171).
    const size_t entries = t->fmt == HZSDR_FMT_U8 ? 65535 : 65536;
    HZ_TRY(ensure_pinned(ctx, entries * 2));
    uint8_t *keys = (uint8_t *)ctx->pinned;
    if (t->fmt == HZSDR_FMT_U8) {
        memset(keys, 0, entries * 2);
        for (uint32_t realv = 0; realv < 256; realv++)        // stream/multiply.go:156-163
            for (uint32_t imagv = 0; imagv <= 256; imagv++) {
                uint8_t i8 = (uint8_t)realv, q8 = (uint8_t)imagv;
                uint32_t idx = (uint32_t)i8 * 255 + q8;        // :106-108
                keys[2 * idx] = i8;
                keys[2 * idx + 1] = q8;
            }
    } else {
        hzsdr_lut_identity(keys);                              // :222
    }
    HZ_HIP(ctx, hipMemcpyAsync(t->tab, keys, entries * 2, hipMemcpyHostToDevice, ctx->stream));
    HZ_TRY(convert_device(ctx, HZSDR_FMT_C64, t->cbuf, t->fmt, t->tab, entries));
    rotate_device(ctx, t->cbuf, entries, re, im);
    HZ_TRY(convert_device(ctx, t->fmt, t->tab, HZSDR_FMT_C64, t->cbuf, entries));
    HZ_HIP(ctx, hipGetLastError());
    // the pinned key buffer is reused by later calls: wait for the upload
    HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HZSDR_OK;
}

int hzsdr_rotlut_create(hzsdr_ctx *ctx, int format, float re, float im, hzsdr_rotlut **out) {
    using namespace hz;
    if (!ctx || !out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (format != HZSDR_FMT_U8 && format != HZSDR_FMT_I8)
        return fail(ctx, HZSDR_ERR_FORMAT_UNKNOWN, "rotlut: format must be u8 or i8");
    HZ_TRY(enter(ctx));
    hzsdr_rotlut *t = new hzsdr_rotlut{ctx, format, nullptr, nullptr};
    hipError_t e = hipMalloc(&t->tab, 65536 * 2);
    if (e == hipSuccess) e = hipMalloc((void **)&t->cbuf, 65536 * 8);
    if (e != hipSuccess) {
        if (t->tab) (void)hipFree(t->tab);
        delete t;
        return hip_fail(ctx, e, "rotlut alloc", __FILE__, __LINE__);
    }
    int rc = hzsdr_rotlut_set_multiplier(t, re, im);
    if (rc != HZSDR_OK) {
        hzsdr_rotlut_free(t);
        return rc;
    }
    *out = t;
    return HZSDR_OK;
}

int hzsdr_rotlut_apply(hzsdr_rotlut *t, void *buf, size_t n) {
    using namespace hz;
    if (!t || (n && !buf)) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = t->ctx;
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    Stage st(ctx);
    void *d;
    HZ_TRY(st.inout(0, buf, n * 2, &d));
    if (t->fmt == HZSDR_FMT_U8)
        launch_lut<uint16_t, true>(ctx, d, t->tab, d, n);
    else
        launch_lut<uint16_t, false>(ctx, d, t->tab, d, n);
    return st.finish();
}

int hzsdr_rotlut_free(hzsdr_rotlut *t) {
    if (!t) return HZSDR_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    if (t->tab) (void)hipFree(t->tab);
    if (t->cbuf) (void)hipFree(t->cbuf);
    delete t;
    return HZSDR_OK;
}

// ---- Decimate / Downsample -----------------------------------------------------

int hzsdr_decimate(hzsdr_ctx *ctx, int to_format, void *to, size_t to_len, int from_format,
                   const void *from, size_t from_len, unsigned factor, int64_t offset,
                   size_t *n_out) {
    using namespace hz;
    (void)offset;  // accepted and ignored, as stream/decimate.go:59-101 does
    if (n_out) *n_out = 0;
    if (!ctx) return HZSDR_ERR_INVALID_ARGUMENT;
    if (from_format != to_format)  // :60-62
        return fail(ctx, HZSDR_ERR_FORMAT_MISMATCH, "decimate: formats differ");
    if (factor == 0) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "decimate: factor 0");
    const size_t count = from_len / factor;
    if (to_len < count) return fail(ctx, HZSDR_ERR_DST_TOO_SMALL, "decimate: dst too small");  // :68-70
    if (count > 0 && from_format != HZSDR_FMT_U8 && from_format != HZSDR_FMT_I16 &&
        from_format != HZSDR_FMT_C64)  // :85-97 (i8 is not in the type switch)
        return fail(ctx, HZSDR_ERR_FORMAT_UNKNOWN, "decimate: format not handled");
    if (count && (!to || !from)) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (count == 0) return HZSDR_OK;
    const int sz = format_size(from_format);
    Stage st(ctx);
    const void *dfrom;
    void *dto;
    HZ_TRY(st.in(0, from, ((count - 1) * factor + 1) * sz, &dfrom));
    HZ_TRY(st.out(1, to, count * sz, &dto));
    decimate_device(ctx, from_format, dfrom, dto, count, factor);
    HZ_TRY(st.finish());
    if (n_out) *n_out = count;
    return HZSDR_OK;
}

int hzsdr_downsample(hzsdr_ctx *ctx, int to_format, void *to, size_t to_len, int from_format,
                     const void *from, size_t from_len, unsigned factor, int64_t offset,
                     size_t *n_out) {
    using namespace hz;
    (void)offset;
    if (n_out) *n_out = 0;
    if (!ctx) return HZSDR_ERR_INVALID_ARGUMENT;
    if (to_format != HZSDR_FMT_C64)  // stream/downsample.go:69-71
        return fail(ctx, HZSDR_ERR_FORMAT_MISMATCH, "downsample: dst must be c64");
    if (factor == 0) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "downsample: factor 0");
    const size_t count = from_len / factor;
    if (to_len < count) return fail(ctx, HZSDR_ERR_DST_TOO_SMALL, "downsample: dst too small");  // :77-79
    if (count > 0 && from_format != HZSDR_FMT_U8 && from_format != HZSDR_FMT_I16 &&
        from_format != HZSDR_FMT_C64)  // :104-114
        return fail(ctx, HZSDR_ERR_FORMAT_UNKNOWN, "downsample: format not handled");
    if (count && (!to || !from)) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (count == 0) return HZSDR_OK;
    Stage st(ctx);
    const void *dfrom;
    void *dto;
    HZ_TRY(st.in(0, from, count * factor * format_size(from_format), &dfrom));
    HZ_TRY(st.out(1, to, count * 8, &dto));
    downsample_device(ctx, from_format, dfrom, dto, count, factor);
    HZ_TRY(st.finish());
    if (n_out) *n_out = count;
    return HZSDR_OK;
}

}  // extern "C"

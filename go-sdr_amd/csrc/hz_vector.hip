// hz_vector.hip -- complex64 vector ops: Scale, Rotate, Add, the ordered K-way
// sum of stream.Add and the Beamform weighted sum.  All HBM-bound: 16 B per lane
// per access (two complex samples), four accesses in flight per lane.
#include "hz_common.h"
#include "hz_device.h"

namespace hz {

// ---- in-place elementwise over c64 ----------------------------------------------

struct OpScale {
    float r;
    __device__ __forceinline__ float2 operator()(float2 v) const {
        return make_float2(__fmul_rn(v.x, r), __fmul_rn(v.y, r));  // mult.go:25-27
    }
};
struct OpRotate {
    float2 m;
    __device__ __forceinline__ float2 operator()(float2 v) const { return go_cmul(v, m); }  // mult.go:29-33
};

template <class Op, bool NT = false>
__global__ __launch_bounds__(kThreads) void map_inplace_vec(float4 *buf, size_t nvec, Op op) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        float4 a0 = ld_stream<NT>(buf + i), a1 = ld_stream<NT>(buf + i + stride), a2 = ld_stream<NT>(buf + i + 2 * stride), a3 = ld_stream<NT>(buf + i + 3 * stride);
        float2 l, h;
        l = op(make_float2(a0.x, a0.y)); h = op(make_float2(a0.z, a0.w)); st_stream<NT>(buf + i, make_float4(l.x, l.y, h.x, h.y));
        l = op(make_float2(a1.x, a1.y)); h = op(make_float2(a1.z, a1.w)); st_stream<NT>(buf + i + stride, make_float4(l.x, l.y, h.x, h.y));
        l = op(make_float2(a2.x, a2.y)); h = op(make_float2(a2.z, a2.w)); st_stream<NT>(buf + i + 2 * stride, make_float4(l.x, l.y, h.x, h.y));
        l = op(make_float2(a3.x, a3.y)); h = op(make_float2(a3.z, a3.w)); st_stream<NT>(buf + i + 3 * stride, make_float4(l.x, l.y, h.x, h.y));
    }
    for (; i < nvec; i += stride) {
        float4 a = ld_stream<NT>(buf + i);
        float2 l = op(make_float2(a.x, a.y)), h = op(make_float2(a.z, a.w));
        st_stream<NT>(buf + i, make_float4(l.x, l.y, h.x, h.y));
    }
}

template <class Op>
__global__ void map_inplace_scalar(float2 *buf, size_t n, Op op) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) buf[i] = op(buf[i]);
}

// Split [buf, buf+n) into an unaligned head sample, a 16-B aligned body of
// sample pairs and a tail sample (sub-slices of c64 are only 8-B aligned).
template <class Op>
static void launch_map(hzsdr_ctx *ctx, void *buf, size_t n, Op op) {
    if (n == 0) return;
    float2 *p = (float2 *)buf;
    size_t head = ((uintptr_t)p % 16) ? 1 : 0;
    if (head > n) head = n;
    size_t nvec = (n - head) / 2;
    size_t tail = n - head - 2 * nvec;
    if (head) hipLaunchKernelGGL(map_inplace_scalar<Op>, dim3(1), dim3(64), 0, ctx->stream, p, head, op);
    if (nvec && streams_past_cache(nvec * 16))  // (past the memory-side cache: hz_device.h)
        hipLaunchKernelGGL((map_inplace_vec<Op, true>), dim3(blocks_for(ctx, nvec)), dim3(kThreads), 0, ctx->stream, (float4 *)(p + head), nvec, op);
    else if (nvec)
        hipLaunchKernelGGL(map_inplace_vec<Op>, dim3(blocks_for(ctx, nvec)), dim3(kThreads), 0,  // (one vector per lane up to the grid cap: hz_convert.hip)
                           ctx->stream, (float4 *)(p + head), nvec, op);
    if (tail)
        hipLaunchKernelGGL(map_inplace_scalar<Op>, dim3(1), dim3(64), 0, ctx->stream, p + head + 2 * nvec, tail, op);
}

void scale_device(hzsdr_ctx *ctx, void *buf, size_t n, float r) { launch_map(ctx, buf, n, OpScale{r}); }
void rotate_device(hzsdr_ctx *ctx, void *buf, size_t n, float re, float im) {
    launch_map(ctx, buf, n, OpRotate{make_float2(re, im)});
}

// ---- c = a + b (flat float add; c may alias a or b) --------------------------------

__global__ __launch_bounds__(kThreads) void add_vec_kernel(const float4 *a, const float4 *b, float4 *c,
                                                           size_t nvec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        float4 x = a[i], y = b[i];
        c[i] = make_float4(__fadd_rn(x.x, y.x), __fadd_rn(x.y, y.y), __fadd_rn(x.z, y.z), __fadd_rn(x.w, y.w));
    }
}
__global__ void add_scalar_kernel(const float *a, const float *b, float *c, size_t ncomp) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncomp; i += stride)
        c[i] = __fadd_rn(a[i], b[i]);
}

static void add_device(hzsdr_ctx *ctx, const void *a, const void *b, void *c, size_t n) {
    if (n == 0) return;
    const size_t ncomp = 2 * n;
    const bool aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) % 16) == 0;
    size_t nvec = aligned ? ncomp / 4 : 0;
    if (nvec)
        hipLaunchKernelGGL(add_vec_kernel, dim3(blocks_for(ctx, nvec)), dim3(kThreads), 0, ctx->stream,
                           (const float4 *)a, (const float4 *)b, (float4 *)c, nvec);
    size_t done = nvec * 4;
    if (done < ncomp)
        hipLaunchKernelGGL(add_scalar_kernel, dim3(blocks_for(ctx, ncomp - done)), dim3(kThreads), 0,
                           ctx->stream, (const float *)a + done, (const float *)b + done,
                           (float *)c + done, ncomp - done);
}

// ---- ordered K-way sums ----------------------------------------------------------

constexpr int kMaxChannels = 16;
struct PtrList {
    const void *p[kMaxChannels];
};
struct WeightList {
    float2 w[kMaxChannels];
    unsigned char identity[kMaxChannels];  // w == 1+0i: multiply skipped (stream/multiply.go:59-62)
};

// out = ((+0 + b0) + b1) + ...  flat over 16-B vectors of components.
// T: float (c64), int16_t, int8_t lanes inside a 16-B vector.
template <class T> struct alignas(16) V16 {
    T v[16 / sizeof(T)];
};
template <class T> __device__ __forceinline__ T add1(T a, T b);
template <> __device__ __forceinline__ float add1(float a, float b) { return __fadd_rn(a, b); }
template <> __device__ __forceinline__ int16_t add1(int16_t a, int16_t b) { return (int16_t)((uint16_t)a + (uint16_t)b); }
template <> __device__ __forceinline__ int8_t add1(int8_t a, int8_t b) { return (int8_t)((uint8_t)a + (uint8_t)b); }

template <class T, bool NT>  // NT: see beamform_kernel
__global__ __launch_bounds__(kThreads) void sum_kernel(V16<T> *out, PtrList bufs, int k, size_t nvec,
                                                       bool accumulate) {
    constexpr int L = 16 / sizeof(T);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        V16<T> acc;
        if (accumulate) {
            acc = out[i];
        } else {
#pragma unroll
            for (int l = 0; l < L; l++) acc.v[l] = (T)0;  // stream/add.go:165-167 zeroes first
        }
        for (int c = 0; c < k; c++) {
            V16<T> x = NT ? nt_load((const V16<T> *)bufs.p[c] + i) : ((const V16<T> *)bufs.p[c])[i];
#pragma unroll
            for (int l = 0; l < L; l++) acc.v[l] = add1<T>(acc.v[l], x.v[l]);
        }
        if (NT) nt_store(out + i, acc); else out[i] = acc;
    }
}
template <class T>
__global__ void sum_scalar_kernel(T *out, PtrList bufs, int k, size_t off, size_t ncomp,
                                  bool accumulate) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = off + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncomp; i += stride) {
        T acc = accumulate ? out[i] : (T)0;
        for (int c = 0; c < k; c++) acc = add1<T>(acc, ((const T *)bufs.p[c])[i]);
        out[i] = acc;
    }
}

template <class T>
static void launch_sum(hzsdr_ctx *ctx, void *out, const PtrList &bufs, int k, size_t n, bool accumulate) {
    const size_t ncomp = 2 * n;
    if (ncomp == 0) return;
    uintptr_t all = (uintptr_t)out;
    for (int c = 0; c < k; c++) all |= (uintptr_t)bufs.p[c];
    constexpr int L = 16 / sizeof(T);
    size_t nvec = (all % 16 == 0) ? ncomp / L : 0;
    const bool stream_past_cache = (size_t)(k + 1) * ncomp * sizeof(T) > ((size_t)192 << 20);
    if (nvec) {
        if (stream_past_cache)
            hipLaunchKernelGGL((sum_kernel<T, true>), dim3(blocks_for(ctx, nvec)), dim3(kThreads), 0, ctx->stream,
                               (V16<T> *)out, bufs, k, nvec, accumulate);
        else
            hipLaunchKernelGGL((sum_kernel<T, false>), dim3(blocks_for(ctx, nvec)), dim3(kThreads), 0, ctx->stream,
                               (V16<T> *)out, bufs, k, nvec, accumulate);
    }
    size_t done = nvec * L;
    if (done < ncomp)
        hipLaunchKernelGGL(sum_scalar_kernel<T>, dim3(blocks_for(ctx, ncomp - done)), dim3(kThreads), 0,
                           ctx->stream, (T *)out, bufs, k, done, ncomp, accumulate);
}

// ---- Beamform: out = ((acc0 + w0*x0) + w1*x1) + ...  ----------------------------
// FMT = channel format: the per-channel ConvertReader of stream/beamform.go:151
// is fused into the load.  Two samples per lane per step when FMT is c64.

template <int FMT> struct Chan;
template <> struct Chan<HZSDR_FMT_C64> {
    using raw_t = float2;
    static __device__ __forceinline__ float2 cvt(float2 r) { return r; }
};
template <> struct Chan<HZSDR_FMT_U8> {
    using raw_t = uint16_t;
    static __device__ __forceinline__ float2 cvt(uint16_t r) { return make_float2(u8_to_f32(r & 0xFF), u8_to_f32(r >> 8)); }
};
template <> struct Chan<HZSDR_FMT_I8> {
    using raw_t = uint16_t;
    static __device__ __forceinline__ float2 cvt(uint16_t r) { return make_float2(i8_to_f32((int8_t)(r & 0xFF)), i8_to_f32((int8_t)(r >> 8))); }
};
template <> struct Chan<HZSDR_FMT_I16> {
    using raw_t = uint32_t;
    static __device__ __forceinline__ float2 cvt(uint32_t r) { return make_float2(i16_to_f32((int16_t)(r & 0xFFFF)), i16_to_f32((int16_t)(r >> 16))); }
};

// NT: the channels and the output are touched once and together exceed the Infinity
// Cache, so they stream past it (non-temporal); a set that fits stays on plain accesses,
// which is faster when the same buffers come round again.
template <int FMT, int W, bool NT>
__global__ __launch_bounds__(kThreads) void beamform_kernel(float2 *out, PtrList chans, WeightList wl,
                                                            int k, size_t nvec, bool accumulate) {
    using R = typename Chan<FMT>::raw_t;
    struct alignas(sizeof(R) * W) RV { R v[W]; };
    struct alignas(8 * W) OV { float2 v[W]; };
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        OV acc;
        if (accumulate) {
            acc = ((const OV *)out)[i];
        } else {
#pragma unroll
            for (int l = 0; l < W; l++) acc.v[l] = make_float2(0.0f, 0.0f);
        }
        // channels in groups of four: the group's loads are issued together (four
        // 16-byte requests in flight per lane), then consumed in channel order
        for (int c0 = 0; c0 < k; c0 += 4) {
            RV x[4];
#pragma unroll
            for (int g = 0; g < 4; g++)
                if (c0 + g < k) x[g] = NT ? nt_load((const RV *)chans.p[c0 + g] + i) : ((const RV *)chans.p[c0 + g])[i];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (c0 + g < k) {
                    const float2 w = wl.w[c0 + g];
                    const bool ident = wl.identity[c0 + g];
#pragma unroll
                    for (int l = 0; l < W; l++) {
                        float2 y = Chan<FMT>::cvt(x[g].v[l]);
                        if (!ident) y = go_cmul(y, w);
                        acc.v[l].x = __fadd_rn(acc.v[l].x, y.x);
                        acc.v[l].y = __fadd_rn(acc.v[l].y, y.y);
                    }
                }
            }
        }
        if (NT) nt_store((OV *)out + i, acc); else ((OV *)out)[i] = acc;
    }
}

template <int FMT>
static void launch_beamform(hzsdr_ctx *ctx, void *out, const PtrList &ch, const WeightList &wl, int k,
                            size_t n, bool accumulate) {
    using R = typename Chan<FMT>::raw_t;
    if (n == 0) return;
    constexpr int W = 2;
    uintptr_t mis = (uintptr_t)out % (8 * W);
    for (int c = 0; c < k; c++) mis |= (uintptr_t)ch.p[c] % (sizeof(R) * W);
    size_t nvec = mis ? 0 : n / W;
    const bool stream_past_cache = ((size_t)k * sizeof(R) + 8) * n > ((size_t)192 << 20);
    if (nvec) {
        if (stream_past_cache)
            hipLaunchKernelGGL((beamform_kernel<FMT, W, true>), dim3(blocks_for(ctx, nvec)), dim3(kThreads), 0,
                               ctx->stream, (float2 *)out, ch, wl, k, nvec, accumulate);
        else
            hipLaunchKernelGGL((beamform_kernel<FMT, W, false>), dim3(blocks_for(ctx, nvec)), dim3(kThreads), 0,
                               ctx->stream, (float2 *)out, ch, wl, k, nvec, accumulate);
    }
    size_t done = nvec * W;
    if (done < n) {
        PtrList t = ch;
        for (int c = 0; c < k; c++) t.p[c] = (const R *)ch.p[c] + done;
        hipLaunchKernelGGL((beamform_kernel<FMT, 1, false>), dim3(blocks_for(ctx, n - done)), dim3(kThreads), 0,
                           ctx->stream, (float2 *)out + done, t, wl, k, n - done, accumulate);
    }
}

void beamform_device(hzsdr_ctx *ctx, void *out, int fmt, const PtrList &ch, const WeightList &wl, int k,
                     size_t n, bool accumulate) {
    switch (fmt) {
    case HZSDR_FMT_C64: launch_beamform<HZSDR_FMT_C64>(ctx, out, ch, wl, k, n, accumulate); break;
    case HZSDR_FMT_U8: launch_beamform<HZSDR_FMT_U8>(ctx, out, ch, wl, k, n, accumulate); break;
    case HZSDR_FMT_I8: launch_beamform<HZSDR_FMT_I8>(ctx, out, ch, wl, k, n, accumulate); break;
    default: launch_beamform<HZSDR_FMT_I16>(ctx, out, ch, wl, k, n, accumulate); break;
    }
}

}  // namespace hz

extern "C" {

int hzsdr_scale(hzsdr_ctx *ctx, void *buf, size_t n, float r) {
    using namespace hz;
    if (!ctx || (n && !buf)) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    Stage st(ctx);
    void *d;
    HZ_TRY(st.inout(0, buf, n * 8, &d));
    scale_device(ctx, d, n, r);
    return st.finish();
}

int hzsdr_rotate(hzsdr_ctx *ctx, void *buf, size_t n, float re, float im) {
    using namespace hz;
    if (!ctx || (n && !buf)) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    Stage st(ctx);
    void *d;
    HZ_TRY(st.inout(0, buf, n * 8, &d));
    rotate_device(ctx, d, n, re, im);
    return st.finish();
}

int hzsdr_add(hzsdr_ctx *ctx, const void *a, size_t na, const void *b, size_t nb, void *c, size_t nc) {
    using namespace hz;
    if (!ctx) return HZSDR_ERR_INVALID_ARGUMENT;
    if (na != nb || na != nc)  // internal/simd/add.go:34-36
        return fail(ctx, HZSDR_ERR_LENGTH_MISMATCH, "simd.AddComplex: a, b, and c are not the same length");
    if (na && (!a || !b || !c)) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (na == 0) return HZSDR_OK;
    Stage st(ctx);
    const void *da, *db;
    void *dc;
    HZ_TRY(st.in(0, a, na * 8, &da));
    HZ_TRY(st.in(1, b, na * 8, &db));
    HZ_TRY(st.out(2, c, na * 8, &dc));
    add_device(ctx, da, db, dc, na);
    return st.finish();
}

int hzsdr_sum(hzsdr_ctx *ctx, int format, void *out, const void *const *bufs, int count, size_t n) {
    using namespace hz;
    if (!ctx) return HZSDR_ERR_INVALID_ARGUMENT;
    if (format != HZSDR_FMT_C64 && format != HZSDR_FMT_I16 && format != HZSDR_FMT_I8)
        return fail(ctx, HZSDR_ERR_FORMAT_UNKNOWN, "sum: format must be c64, i16 or i8");  // stream/add.go:55-61
    if (count < 1 || count > kMaxChannels || !bufs)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "sum: 1..16 buffers");
    if (n && !out) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    const size_t bytes = n * format_size(format);
    Stage st(ctx);
    PtrList pl{};
    for (int c = 0; c < count; c++) {
        if (!bufs[c]) return HZSDR_ERR_INVALID_ARGUMENT;
        HZ_TRY(st.in(1 + c, bufs[c], bytes, &pl.p[c]));
    }
    void *dout;
    HZ_TRY(st.out(0, out, bytes, &dout));
    switch (format) {
    case HZSDR_FMT_C64: launch_sum<float>(ctx, dout, pl, count, n, false); break;
    case HZSDR_FMT_I16: launch_sum<int16_t>(ctx, dout, pl, count, n, false); break;
    default: launch_sum<int8_t>(ctx, dout, pl, count, n, false); break;
    }
    return st.finish();
}

int hzsdr_beamform_partial(hzsdr_ctx *ctx, void *out, int format, const void *const *channels,
                           const float *weights, int count, size_t n, int accumulate) {
    using namespace hz;
    if (!ctx) return HZSDR_ERR_INVALID_ARGUMENT;
    const int sz = format_size(format);
    if (sz == 0) return fail(ctx, HZSDR_ERR_FORMAT_UNKNOWN, "beamform: unknown channel format");
    if (count < 1 || count > kMaxChannels || !channels || !weights)
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "beamform: 1..16 channels with weights");
    if (n && !out) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    Stage st(ctx);
    PtrList pl{};
    WeightList wl{};
    for (int c = 0; c < count; c++) {
        if (!channels[c]) return HZSDR_ERR_INVALID_ARGUMENT;
        HZ_TRY(st.in(1 + c, channels[c], n * sz, &pl.p[c]));
        wl.w[c] = make_float2(weights[2 * c], weights[2 * c + 1]);
        wl.identity[c] = (weights[2 * c] == 1.0f && weights[2 * c + 1] == 0.0f) ? 1 : 0;
    }
    void *dout;
    if (accumulate)
        HZ_TRY(st.inout(0, out, n * 8, &dout));
    else
        HZ_TRY(st.out(0, out, n * 8, &dout));
    beamform_device(ctx, dout, format, pl, wl, count, n, accumulate != 0);
    return st.finish();
}

int hzsdr_beamform(hzsdr_ctx *ctx, void *out, int format, const void *const *channels,
                   const float *weights, int count, size_t n) {
    return hzsdr_beamform_partial(ctx, out, format, channels, weights, count, n, 0);
}

}  // extern "C"

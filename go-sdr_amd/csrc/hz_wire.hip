// hz_wire.hip -- "next" rows of SURVEY.md 8f: foreign-endian wire/disk formats
// (rank 4) and the kerberos graft band stitcher (rank 3).
#include "hz_common.h"
#include "hz_device.h"
#include "hz_fft_api.h"

namespace hz {

// bytes_io.go:30-64 / :150-197 (byteWriterForeign / byteReaderForeign): every int16
// (i16) or float32 (c64) component is written / read with the other byte order.
template <class T> __device__ __forceinline__ T bswap(T v);
template <> __device__ __forceinline__ uint16_t bswap(uint16_t v) { return (uint16_t)((v << 8) | (v >> 8)); }
template <> __device__ __forceinline__ uint32_t bswap(uint32_t v) { return __builtin_bswap32(v); }

template <class T>
__global__ __launch_bounds__(kThreads) void byteswap_kernel(T *buf, size_t ncomp) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncomp; i += stride) buf[i] = bswap<T>(buf[i]);
}

// 16 bytes per lane when the pointer allows
__global__ __launch_bounds__(kThreads) void byteswap16_vec_kernel(uint4 *buf, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        uint4 v = buf[i];
        auto sw = [](uint32_t w) { return ((w & 0x00FF00FFu) << 8) | ((w >> 8) & 0x00FF00FFu); };
        buf[i] = make_uint4(sw(v.x), sw(v.y), sw(v.z), sw(v.w));
    }
}
__global__ __launch_bounds__(kThreads) void byteswap32_vec_kernel(uint4 *buf, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        uint4 v = buf[i];
        buf[i] = make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w));
    }
}

// rtl/kerberos/internal/reader.go:47-64 FFTShiftAndScale: swap the halves and
// divide both components by `scale` (float32 IEEE division).
__global__ __launch_bounds__(kThreads) void fftshift_scale_kernel(float2 *data, size_t half, float scale) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += stride) {
        const float2 lo = data[i], hi = data[half + i];
        data[i] = make_float2(__fdiv_rn(hi.x, scale), __fdiv_rn(hi.y, scale));
        data[half + i] = make_float2(__fdiv_rn(lo.x, scale), __fdiv_rn(lo.y, scale));
    }
}

static void fftshift_scale_device(hzsdr_ctx *ctx, void *data, size_t n, float scale) {
    if (n / 2 == 0) return;
    hipLaunchKernelGGL(fftshift_scale_kernel, dim3(blocks_for(ctx, n / 2)), dim3(kThreads), 0, ctx->stream,
                       (float2 *)data, n / 2, scale);
}

}  // namespace hz

extern "C" {

int hzsdr_byteswap(hzsdr_ctx *ctx, int format, void *buf, size_t n) {
    using namespace hz;
    if (!ctx || (n && !buf)) return HZSDR_ERR_INVALID_ARGUMENT;
    const int sz = format_size(format);
    if (sz == 0) return fail(ctx, HZSDR_ERR_FORMAT_UNKNOWN, "byteswap: unknown format");
    HZ_TRY(enter(ctx));
    if (n == 0 || sz == 2) return HZSDR_OK;  // u8 / i8: bytes have no order (bytes_io.go:44-50)
    Stage st(ctx);
    void *d;
    HZ_TRY(st.inout(0, buf, n * sz, &d));
    const size_t bytes = n * sz;
    size_t nvec = ((uintptr_t)d % 16 == 0) ? bytes / 16 : 0;
    if (format == HZSDR_FMT_I16) {
        if (nvec) hipLaunchKernelGGL(byteswap16_vec_kernel, dim3(blocks_for(ctx, nvec)), dim3(kThreads), 0, ctx->stream, (uint4 *)d, nvec);
        const size_t done = nvec * 8, ncomp = 2 * n;
        if (done < ncomp)
            hipLaunchKernelGGL(byteswap_kernel<uint16_t>, dim3(blocks_for(ctx, ncomp - done)), dim3(kThreads), 0, ctx->stream, (uint16_t *)d + done, ncomp - done);
    } else {
        if (nvec) hipLaunchKernelGGL(byteswap32_vec_kernel, dim3(blocks_for(ctx, nvec)), dim3(kThreads), 0, ctx->stream, (uint4 *)d, nvec);
        const size_t done = nvec * 4, ncomp = 2 * n;
        if (done < ncomp)
            hipLaunchKernelGGL(byteswap_kernel<uint32_t>, dim3(blocks_for(ctx, ncomp - done)), dim3(kThreads), 0, ctx->stream, (uint32_t *)d + done, ncomp - done);
    }
    return st.finish();
}

int hzsdr_fftshift_scale(hzsdr_ctx *ctx, void *data, size_t n, float scale) {
    using namespace hz;
    if (!ctx || (n && !data)) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (n < 2) return HZSDR_OK;
    Stage st(ctx);
    void *d;
    HZ_TRY(st.inout(0, data, n * 8, &d));
    fftshift_scale_device(ctx, d, n, scale);
    return st.finish();
}

int hzsdr_graft(hzsdr_ctx *ctx, void *out, size_t out_len, const void *const *channels, int count, size_t n) {
    using namespace hz;
    if (!ctx || !channels || count < 1 || count > 8) return HZSDR_ERR_INVALID_ARGUMENT;
    const size_t total = n * (size_t)count;
    if (out_len < total) return fail(ctx, HZSDR_ERR_DST_TOO_SMALL, "graft: output shorter than count * n");
    if (n == 0 || (n & (n - 1)) || (total & (total - 1)) || total > ((size_t)1 << 24))
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "graft: n and count * n must be powers of two (<= 2^24)");
    if (!out) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    Stage st(ctx);
    void *dout;
    HZ_TRY(st.out(0, out, total * 8, &dout));
    HZ_TRY(ensure_slot(ctx, 9, total * 8));
    float2 *freq = (float2 *)ctx->slots[9].ptr;
    for (int c = 0; c < count; c++) {
        if (!channels[c]) return HZSDR_ERR_INVALID_ARGUMENT;
        const void *dc;
        HZ_TRY(st.in(1 + c, channels[c], n * 8, &dc));
        // graft.go:96-106: forward plan i into its slice of the shared spectrum, then
        // FFTShiftAndScale(slice, float32(fftSize))
        HZ_TRY(fft_device(ctx, dc, freq + (size_t)c * n, n, 1, true));
        fftshift_scale_device(ctx, freq + (size_t)c * n, n, (float)n);
    }
    HZ_TRY(fft_device(ctx, freq, dout, total, 1, false));  // graft.go:108: one backward plan over all bands
    return st.finish();
}

}  // extern "C"

// hz_nco.hip -- stream.ShiftBuffer (stream/shifter.go:66-85) on the GPU.
//
// The reference advances a float64 clock serially; hzsdr_nco_segments
// (hz_host.cpp) turns the next n clock values into a handful of exactly linear
// runs, so every lane can rebuild its own ts with one exact fma and the kernel
// is an ordinary in-place streaming map: load 16 B (two samples), Go's
// math.Sincos in float64 per sample, complex64 multiply, store 16 B.
#include "hz_common.h"
#include "hz_device.h"
#include "hz_nco.h"

namespace hz {

// buf[first + i] *= complex64(cos(ph), sin(ph)),  ph = (tau*shift) * ts_i
// ULP1 (hzsdr_nco_set_ulp1): the factor from the phase in turns and float32 polynomials (sincos_turns32,
// hz_device.h) -- within one float32 ulp of the reference's factor, a third of the vector instructions; the
// default is math.Sincos operation for operation (bit-identical to the oracle).
template <int ULP1>
__device__ __forceinline__ float2 nco_rotate(float2 v, double ts, double tau_shift) {
    double ph = __dmul_rn(tau_shift, ts);  // stream/shifter.go:81, (tau*shift)*ts left to right
    if constexpr (ULP1) {
        float sf, cf;
        // (ULP1 = 1: |tau shift| 2 pi < 1e8 rad, where the plain product is good to a thirtieth of an ulp; 2: any
        // phase the reference can form, by a double-double 1 / 2 pi -- three float64 instructions more)
        sincos_turns32(ULP1 == 1 ? turns32(ph) : turns32_wide(ph), sf, cf);
        return go_cmul(v, make_float2(cf, sf));
    } else {
        double s, c;
        go_sincos(ph, s, c);
        return go_cmul(v, make_float2((float)c, (float)s));  // :82 complex64(complex(rl, im))
    }
}

// Each workgroup owns a contiguous tile of kNcoTile vectors per trip; a lane
// issues its kNcoUnroll 16-byte loads back to back before any arithmetic, so a
// CU keeps ~4x the bytes in flight that one load per trip gave (that version
// sat at 4.4 TB/s, latency-bound, not f64-bound).
constexpr int kNcoUnroll = 4;

template <int ULP1>
__global__ __launch_bounds__(kThreads) void nco_shift_vec_kernel(float4 *buf, uint64_t base,
                                                                 size_t nvec, double tau_shift,
                                                                 NcoSegs sg) {
    const size_t tile = (size_t)kThreads * kNcoUnroll;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < nvec; t0 += (size_t)gridDim.x * tile) {
        const uint64_t j_lo = base + 2 * t0;
        const NcoWin w = nco_window(sg, j_lo, j_lo + 2 * tile - 1);
        float4 a[kNcoUnroll];
#pragma unroll
        for (int u = 0; u < kNcoUnroll; u++) {
            const size_t i = t0 + (size_t)u * kThreads + threadIdx.x;
            if (i < nvec) a[u] = buf[i];
        }
#pragma unroll
        for (int u = 0; u < kNcoUnroll; u++) {
            const size_t i = t0 + (size_t)u * kThreads + threadIdx.x;
            if (i < nvec) {
                const uint64_t j = base + 2 * i;
                float2 l = nco_rotate<ULP1>(make_float2(a[u].x, a[u].y), nco_ts(sg, w, j), tau_shift);
                float2 h = nco_rotate<ULP1>(make_float2(a[u].z, a[u].w), nco_ts(sg, w, j + 1), tau_shift);
                buf[i] = make_float4(l.x, l.y, h.x, h.y);
            }
        }
    }
}

template <int ULP1>
__global__ void nco_shift_scalar_kernel(float2 *buf, uint64_t base, size_t n, double tau_shift,
                                        NcoSegs sg) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const NcoWin w = nco_window_all(sg);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        buf[i] = nco_rotate<ULP1>(buf[i], nco_ts(sg, w, base + i), tau_shift);
}

// Plans the next n clock values from *ts into one table.
int nco_plan(hzsdr_ctx *ctx, uint64_t sample_rate, double *ts, size_t n, NcoSegs *sg) {
    memset(sg, 0, sizeof *sg);
    if (n == 0) return HZSDR_OK;
    std::vector<hzsdr_nco_segment> segs(kNcoMaxSegs);
    size_t need = 0;
    double ts_end = *ts;
    int rc = hzsdr_nco_segments(sample_rate, *ts, n, segs.data(), segs.size(), &need, &ts_end);
    if (rc != HZSDR_OK) return rc;
    if (need <= (size_t)kNcoMaxSegs) {
        sg->n = (int)need;
        for (size_t q = 0; q < need; q++) {
            sg->first[q] = segs[q].first;
            sg->t0[q] = segs[q].t0;
            sg->step[q] = segs[q].step;
        }
    } else {
        if (need > (size_t)INT32_MAX) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "nco: clock table too long");
        const size_t bytes = need * sizeof(hzsdr_nco_segment);
        HZ_TRY(ensure_pinned(ctx, bytes));
        hzsdr_nco_segment *h = (hzsdr_nco_segment *)ctx->pinned;
        rc = hzsdr_nco_segments(sample_rate, *ts, n, h, need, &need, &ts_end);
        if (rc != HZSDR_OK) return rc;
        HZ_TRY(ensure_slot(ctx, 9, bytes));
        HZ_HIP(ctx, hipMemcpyAsync(ctx->slots[9].ptr, h, bytes, hipMemcpyHostToDevice, ctx->stream));
        // the pinned staging buffer is reused by later calls
        HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        sg->big = (const hzsdr_nco_segment *)ctx->slots[9].ptr;
        sg->big_n = (int)need;
    }
    *ts = ts_end;
    return HZSDR_OK;
}

// In-place shift of n samples at device pointer buf, advancing *ts.
template <int ULP1>
static int nco_shift_launch(hzsdr_ctx *ctx, uint64_t sample_rate, double *ts, double shift_hz, void *buf, size_t n) {
    if (n == 0) return HZSDR_OK;
    NcoSegs sg;
    HZ_TRY(nco_plan(ctx, sample_rate, ts, n, &sg));
    const double tau_shift = (M_PI * 2) * shift_hz;  // tau * shift, one rounding (stream/shifter.go:81)
    float2 *q = (float2 *)buf;
    if constexpr (ULP1 == 1) {
        // the opt-in factor where the plain range reduction holds: four samples per lane on the chain's kernel
        // (44-46 us per 2^24 samples against 51 with two per lane); what does not fit 32-byte groups one at a time
        const size_t lead = std::min<size_t>(n, (32 - (uintptr_t)q % 32) % 32 / 8), nvec4 = (n - lead) / 4, rest = n - lead - 4 * nvec4;
        if (lead)
            hipLaunchKernelGGL(nco_shift_scalar_kernel<1>, dim3(1), dim3(64), 0, ctx->stream, q, (uint64_t)0, lead, tau_shift, sg);
        if (nvec4) nco_shift_ulp1_map4(ctx, q + lead, nvec4, (uint64_t)lead, tau_shift, sg);
        if (rest)
            hipLaunchKernelGGL(nco_shift_scalar_kernel<1>, dim3(1), dim3(64), 0, ctx->stream, q + lead + 4 * nvec4, (uint64_t)(lead + 4 * nvec4), rest,
                               tau_shift, sg);
        return HZSDR_OK;
    }
    size_t head = ((uintptr_t)q % 16) ? 1 : 0;
    size_t nvec = (n - head) / 2, tail = n - head - 2 * nvec;
    if (head)
        hipLaunchKernelGGL(nco_shift_scalar_kernel<ULP1>, dim3(1), dim3(64), 0, ctx->stream, q, (uint64_t)0, head, tau_shift, sg);
    if (nvec) {
        if constexpr (ULP1 == 0)  // the reference's factor bit for bit: the straight path of shift_exact_kernel
            nco_shift_exact_map2(ctx, q + head, nvec, (uint64_t)head, tau_shift, sg);
        else
            hipLaunchKernelGGL(nco_shift_vec_kernel<ULP1>, dim3(blocks_for(ctx, (nvec + kNcoUnroll - 1) / kNcoUnroll)), dim3(kThreads), 0,
                               ctx->stream, (float4 *)(q + head), (uint64_t)head, nvec, tau_shift, sg);
    }
    if (tail)
        hipLaunchKernelGGL(nco_shift_scalar_kernel<ULP1>, dim3(1), dim3(64), 0, ctx->stream,
                           q + head + 2 * nvec, (uint64_t)(head + 2 * nvec), tail, tau_shift, sg);
    return HZSDR_OK;
}

int nco_shift_device(hzsdr_ctx *ctx, uint64_t sample_rate, double *ts, double shift_hz, void *buf, size_t n, bool ulp1) {
    if (!ulp1) return nco_shift_launch<0>(ctx, sample_rate, ts, shift_hz, buf, n);
    // (ts < 2 pi: the largest phase of the call is |2 pi shift| 2 pi)
    return fabs((M_PI * 2) * shift_hz) * 6.2832 < 1.0e8 ? nco_shift_launch<1>(ctx, sample_rate, ts, shift_hz, buf, n)
                                                         : nco_shift_launch<2>(ctx, sample_rate, ts, shift_hz, buf, n);
}

}  // namespace hz

struct hzsdr_nco {
    hzsdr_ctx *ctx;
    uint64_t sample_rate;
    double ts;
    bool ulp1;
};

extern "C" {

int hzsdr_nco_create(hzsdr_ctx *ctx, uint64_t sample_rate, hzsdr_nco **out) {
    if (!ctx || !out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (sample_rate == 0) return hz::fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "nco: sample rate 0");
    *out = new hzsdr_nco{ctx, sample_rate, 0.0, false};
    return HZSDR_OK;
}

int hzsdr_nco_shift(hzsdr_nco *nco, double shift_hz, void *buf, size_t n) {
    using namespace hz;
    if (!nco || (n && !buf)) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = nco->ctx;
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    Stage st(ctx);
    void *d;
    HZ_TRY(st.inout(0, buf, n * 8, &d));
    double ts = nco->ts;
    HZ_TRY(nco_shift_device(ctx, nco->sample_rate, &ts, shift_hz, d, n, nco->ulp1));
    HZ_TRY(st.finish());
    nco->ts = ts;
    return HZSDR_OK;
}

int hzsdr_nco_get_time(const hzsdr_nco *nco, double *ts) {
    if (!nco || !ts) return HZSDR_ERR_INVALID_ARGUMENT;
    *ts = nco->ts;
    return HZSDR_OK;
}

int hzsdr_nco_set_time(hzsdr_nco *nco, double ts) {
    if (!nco) return HZSDR_ERR_INVALID_ARGUMENT;
    nco->ts = ts;
    return HZSDR_OK;
}

int hzsdr_nco_set_ulp1(hzsdr_nco *nco, int on) {
    if (!nco) return HZSDR_ERR_INVALID_ARGUMENT;
    nco->ulp1 = on != 0;
    return HZSDR_OK;
}

int hzsdr_nco_free(hzsdr_nco *nco) {
    if (!nco) return HZSDR_ERR_INVALID_ARGUMENT;
    delete nco;
    return HZSDR_OK;
}

}  // extern "C"

// hz_kerberos.hip -- the two reductions of rtl/kerberos coherent sync (SURVEY.md
// 8f rank 2, a "next" row): the cross-correlation peak search of checkAlignment
// and the mean phase of PhaseOffsets.  The correlation itself is
// hzsdr_convolve_create(HZSDR_CONV_CROSS_CORRELATE) on the FFT kernels.
#include <math.h>

#include "hz_common.h"
#include "hz_device.h"

namespace hz {

constexpr int kRedBlocks = 1024;

struct Peak {
    double pow;
    uint64_t idx;
};

// lexicographic "better": larger power, then smaller index (= the first maximum
// of the reference's sequential scan with a strict >)
__device__ __forceinline__ bool better(double p, uint64_t i, double q, uint64_t j) {
    return p > q || (p == q && i < j);
}

// rtl/kerberos/internal/align.go:134-143: pow = float64(re*re + im*im), float32
// products and sum un-fused; exact zeros are skipped.
__global__ __launch_bounds__(kThreads) void peak_kernel(const float2 *__restrict__ c, size_t n,
                                                        Peak *__restrict__ part) {
    double best = -INFINITY;
    uint64_t best_i = UINT64_MAX;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 v = c[i];
        if (v.x == 0.0f && v.y == 0.0f) continue;
        const double p = (double)__fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y));
        if (better(p, i, best, best_i)) {
            best = p;
            best_i = i;
        }
    }
    __shared__ double sp[kThreads];
    __shared__ uint64_t si[kThreads];
    sp[threadIdx.x] = best;
    si[threadIdx.x] = best_i;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && better(sp[threadIdx.x + s], si[threadIdx.x + s], sp[threadIdx.x], si[threadIdx.x])) {
            sp[threadIdx.x] = sp[threadIdx.x + s];
            si[threadIdx.x] = si[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = Peak{sp[0], si[0]};
}

// rtl/kerberos/internal/align.go:257-262: Phase(complex128(a * conj(b))), i.e.
// atan2(im, re) in float64 of the complex64 product.  Partial sums per
// workgroup (fixed tree), combined in workgroup order on the host: deterministic,
// not the reference's strictly sequential float64 sum (documented tolerance).
__global__ __launch_bounds__(kThreads) void phase_kernel(const float2 *__restrict__ a,
                                                         const float2 *__restrict__ b, size_t n,
                                                         double *__restrict__ part) {
    double acc = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 y = b[i];
        const float2 p = go_cmul(a[i], make_float2(y.x, -y.y));
        acc += atan2((double)p.y, (double)p.x);
    }
    __shared__ double s[kThreads];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int k = kThreads / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}

}  // namespace hz

extern "C" {

int hzsdr_peak_lag(hzsdr_ctx *ctx, const void *corr, size_t n, int64_t *lag) {
    using namespace hz;
    if (!ctx || !lag || (n && !corr)) return HZSDR_ERR_INVALID_ARGUMENT;
    *lag = -1;
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    Stage st(ctx);
    const void *d;
    HZ_TRY(st.in(0, corr, n * 8, &d));
    const unsigned blocks = blocks_for(ctx, n) < (unsigned)kRedBlocks ? blocks_for(ctx, n) : kRedBlocks;
    HZ_TRY(ensure_slot(ctx, 8, sizeof(Peak) * kRedBlocks));
    HZ_TRY(ensure_pinned(ctx, sizeof(Peak) * kRedBlocks));
    Peak *dpart = (Peak *)ctx->slots[8].ptr, *hpart = (Peak *)ctx->pinned;
    hipLaunchKernelGGL(peak_kernel, dim3(blocks), dim3(kThreads), 0, ctx->stream, (const float2 *)d, n, dpart);
    HZ_HIP(ctx, hipGetLastError());
    HZ_HIP(ctx, hipMemcpyAsync(hpart, dpart, sizeof(Peak) * blocks, hipMemcpyDeviceToHost, ctx->stream));
    HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the result is a host scalar
    double best = -INFINITY;
    uint64_t best_i = UINT64_MAX;
    for (unsigned b = 0; b < blocks; b++)
        if (hpart[b].pow > best || (hpart[b].pow == best && hpart[b].idx < best_i)) {
            best = hpart[b].pow;
            best_i = hpart[b].idx;
        }
    int64_t r = best_i == UINT64_MAX ? -1 : (int64_t)best_i;
    if (r > (int64_t)(n / 2)) r -= (int64_t)n;  // align.go:145-147
    *lag = r;
    return HZSDR_OK;
}

int hzsdr_mean_phase(hzsdr_ctx *ctx, const void *a, const void *b, size_t n, double *mean_phase) {
    using namespace hz;
    if (!ctx || !mean_phase || (n && (!a || !b))) return HZSDR_ERR_INVALID_ARGUMENT;
    *mean_phase = 0.0;
    HZ_TRY(enter(ctx));
    if (n == 0) return HZSDR_OK;
    Stage st(ctx);
    const void *da, *db;
    HZ_TRY(st.in(0, a, n * 8, &da));
    HZ_TRY(st.in(1, b, n * 8, &db));
    const unsigned blocks = blocks_for(ctx, n) < (unsigned)kRedBlocks ? blocks_for(ctx, n) : kRedBlocks;
    HZ_TRY(ensure_slot(ctx, 8, sizeof(double) * kRedBlocks));
    HZ_TRY(ensure_pinned(ctx, sizeof(double) * kRedBlocks));
    double *dpart = (double *)ctx->slots[8].ptr, *hpart = (double *)ctx->pinned;
    hipLaunchKernelGGL(phase_kernel, dim3(blocks), dim3(kThreads), 0, ctx->stream, (const float2 *)da,
                       (const float2 *)db, n, dpart);
    HZ_HIP(ctx, hipGetLastError());
    HZ_HIP(ctx, hipMemcpyAsync(hpart, dpart, sizeof(double) * blocks, hipMemcpyDeviceToHost, ctx->stream));
    HZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double acc = 0.0;
    for (unsigned k = 0; k < blocks; k++) acc += hpart[k];
    *mean_phase = acc / (double)n;  // align.go:266
    return HZSDR_OK;
}

}  // extern "C"

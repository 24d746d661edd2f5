// hz_fft.hip -- fft.Planner / fft.Plan (fft/fft.go:45-59) and the two-input
// fft.Convolve / fft.CrossCorrelate closures (fft/convolution.go:30-138).
//
// N in [4, 8192]: one LDS-resident kernel per Transform (hz_fft.h below 256 points, the
// packed-math core hz_fftv.h from 256 up); 2^14 .. 2^24 (kerberos: 64 Ki, graft: 256 Ki) in
// two steps, N = N1 * N2, two passes over HBM; N = 1, 2 by a plain global-memory radix-2 pass.
#include "hz_fft.h"
#include "hz_fftv.h"

#include <math.h>

#include "hz_common.h"
#include "hz_device.h"
#include "hz_fft_api.h"

namespace hz {

// ---- twiddle tables ---------------------------------------------------------------

int get_twiddles(hzsdr_ctx *ctx, size_t n, const float2 **out) {
    auto it = ctx->twiddles.find(n);
    if (it != ctx->twiddles.end()) {
        *out = (const float2 *)it->second;
        return HZSDR_OK;
    }
    std::vector<float2> h(n);
    for (size_t m = 0; m < n; m++) {
        double a = -2.0 * M_PI * (double)m / (double)n;
        h[m] = make_float2((float)cos(a), (float)sin(a));
    }
    void *d = nullptr;
    HZ_HIP(ctx, hipMalloc(&d, n * sizeof(float2)));
    hipError_t e = hipMemcpy(d, h.data(), n * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hip_fail(ctx, e, "twiddle upload", __FILE__, __LINE__);
    }
    ctx->twiddles[n] = d;
    *out = (const float2 *)d;
    return HZSDR_OK;
}

// ---- per-pass twiddle tables of the packed-math core (hz_fftv.h) ---------------------------
// Entry = (w.re, w.im, -w.im, w.re) = (w, i w), computed in float64.  Cached per context under
// keys that cannot collide with the plain tables' (n) and the big-twiddle tables' (2n + 1).

static void fv_entry(std::vector<float> &t, double angle) {
    const float c = (float)cos(angle), s = (float)sin(angle);
    t.push_back(c);
    t.push_back(s);
    t.push_back(-s);
    t.push_back(c);
}

// radix-16 pass at Ns = ns: rows k < ns of the six entries w^(m k), m = 1, 2, 3, 4, 8, 12,
// w = exp(sign * 2 pi i / (16 ns))
static void fv_pass16_table(std::vector<float> &t, int ns, double sign) {
    static const int m[6] = {1, 2, 3, 4, 8, 12};
    for (int k = 0; k < ns; k++)
        for (int e = 0; e < 6; e++) fv_entry(t, sign * 2.0 * M_PI * (double)m[e] * (double)k / (16.0 * ns));
}
// full rows (r < radix entries w^(r k)) for tables a kernel keeps in LDS
static void fv_rows_table(std::vector<float> &t, int radix, int ns, double sign) {
    for (int k = 0; k < ns; k++)
        for (int r = 0; r < radix; r++) fv_entry(t, sign * 2.0 * M_PI * (double)r * (double)k / ((double)ns * radix));
}
// pass of radix < 16 at Ns = ns: rows k < ns of radix-1 plain (re, im) twiddles w^(r k), r >= 1
static void fv_small_table(std::vector<float> &t, int radix, int ns, double sign) {
    for (int k = 0; k < ns; k++)
        for (int r = 1; r < radix; r++) {
            const double a = sign * 2.0 * M_PI * (double)r * (double)k / ((double)ns * radix);
            t.push_back((float)cos(a));
            t.push_back((float)sin(a));
        }
    while (t.size() % 4) t.push_back(0.f);
}

static int fv_upload(hzsdr_ctx *ctx, size_t key, const std::vector<float> &t, const void **out) {
    auto it = ctx->twiddles.find(key);
    if (it != ctx->twiddles.end()) {
        *out = it->second;
        return HZSDR_OK;
    }
    void *d = nullptr;
    HZ_HIP(ctx, hipMalloc(&d, t.size() * sizeof(float) + 16));
    hipError_t e = hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hip_fail(ctx, e, "twiddle upload", __FILE__, __LINE__);
    }
    ctx->twiddles[key] = d;
    *out = d;
    return HZSDR_OK;
}

int get_fv_tables(hzsdr_ctx *ctx, size_t n, fv::FvTabs *out) {
    if (!fv::ok((int)n)) return HZSDR_ERR_INVALID_ARGUMENT;
    const size_t kf = (n << 8) | 0x10, kb = (n << 8) | 0x11;
    const void *f = nullptr, *b = nullptr;
    auto itf = ctx->twiddles.find(kf), itb = ctx->twiddles.find(kb);
    if (itf != ctx->twiddles.end() && itb != ctx->twiddles.end()) {
        out->fwd = (const fv::cf4 *)itf->second;
        out->bwd = (const fv::cf4 *)itb->second;
        return HZSDR_OK;
    }
    const int N = (int)n, R0 = fv::first_radix(N);
    std::vector<float> tf, tb;
    for (int ns = R0; ns < N; ns *= 16) fv_pass16_table(tf, ns, -1.0);  // forward: (16, R0), (16, 16 R0), ...
    int ns = 16;
    for (; ns * R0 < N; ns *= 16) fv_pass16_table(tb, ns, +1.0);       // backward: (16, 16), (16, 256), ...
    if (R0 == 16) fv_pass16_table(tb, N / 16, +1.0);                    // ... then (R0, N / R0)
    else fv_small_table(tb, R0, N / R0, +1.0);
    if ((int)(tf.size() / 4) != fv::fwd_tab_len(N) || (int)(tb.size() / 4) != fv::bwd_tab_len(N))
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "fft: twiddle table geometry");
    HZ_TRY(fv_upload(ctx, kf, tf, &f));
    HZ_TRY(fv_upload(ctx, kb, tb, &b));
    out->fwd = (const fv::cf4 *)f;
    out->bwd = (const fv::cf4 *)b;
    return HZSDR_OK;
}

// the polyphase analysis (hz_chain_dev.h, fold_poly): branch transforms of M = n / fold points
// with plan 16, R2 = n/256, RL = 16/fold
int get_fv_poly_tables(hzsdr_ctx *ctx, size_t n, unsigned fold, fv::PolyTabs *out) {
    const size_t k2 = (n << 8) | 0x20, k3 = (n << 8) | 0x40 | fold;
    const int N = (int)n, R2 = N / 256, RL = 16 / (int)fold, TPT = N / 16;
    std::vector<float> t2, t3;
    fv_rows_table(t2, R2, 16, -1.0);
    for (int lane = 0; lane < TPT; lane++)
        for (int i = 0; i < RL; i++) fv_entry(t3, -2.0 * M_PI * (double)i * (double)lane / ((double)RL * TPT));
    const void *p2 = nullptr, *p3 = nullptr;
    HZ_TRY(fv_upload(ctx, k2, t2, &p2));
    HZ_TRY(fv_upload(ctx, k3, t3, &p3));
    out->p2 = (const fv::cf4 *)p2;
    out->p3 = (const fv::cf4 *)p3;
    return HZSDR_OK;
}

// ---- LDS kernels ---------------------------------------------------------------------

template <int N, bool FWD>
__global__ __launch_bounds__(fft_block(N), fft_waves(N)) void fft_plan_kernel(const float2 *__restrict__ in,
                                                                float2 *__restrict__ out,
                                                                const float2 *__restrict__ tw,
                                                                size_t batch) {
    constexpr int TPT = fft_tpt(N), XPB = fft_xpb(N), CNT = N / TPT;
    __shared__ float2 lds_all[XPB * N];
    const int sub = threadIdx.x / TPT, lane = threadIdx.x % TPT;
    float2 *lds = lds_all + sub * N;
    {   // one workgroup per XPB transforms, no grid-stride loop: a loop lets LICM hoist every
        // pass's twiddles into registers (254 VGPRs at N = 4096) and collapses occupancy
        const size_t t0 = (size_t)blockIdx.x * XPB;
        const size_t t = t0 + sub;
        const bool live = t < batch;
        const float2 *src = in + t * N;
        float2 *dst = out + t * N;
        FftRegs<N> R;
        __syncthreads();  // previous iteration's LDS reads are done
        if constexpr (FWD) {
#pragma unroll
            for (int q = 0; q < CNT; q++) {
                const int idx = fft_odd(N) ? edge2_index<N>(q, lane) : edge4_index<N>(q, lane);
                R.v[q] = live ? src[idx] : make_float2(0.f, 0.f);
            }
            fft_forward_regs<N>(R, lds, tw, lane);
            if (live) {
#pragma unroll
                for (int q = 0; q < CNT; q++) dst[edge4_index<N>(q, lane)] = R.v[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < CNT; q++) R.v[q] = live ? src[edge4_index<N>(q, lane)] : make_float2(0.f, 0.f);
            fft_backward_regs<N>(R, lds, tw, lane);
            if (live) {
#pragma unroll
                for (int q = 0; q < CNT; q++) {
                    const int idx = fft_odd(N) ? edge2_index<N>(q, lane) : edge4_index<N>(q, lane);
                    dst[idx] = R.v[q];
                }
            }
        }
    }
}

// packed-math core (hz_fftv.h), N = 256 .. 8192; dynamic LDS (N = 8192 needs 68 KiB)
template <int N, bool FWD>
__global__ __launch_bounds__(fv::block(N)) void fft_plan_kernel16(const float2 *__restrict__ in,
                                                                  float2 *__restrict__ out, fv::FvTabs tabs,
                                                                  size_t batch) {
    using fv::cf;
    constexpr int TPT = fv::tpt(N), XPB = fv::xpb(N), R0 = fv::first_radix(N);
    extern __shared__ __attribute__((aligned(16))) unsigned char hz_dyn_lds_fft[];
    const int sub = threadIdx.x / TPT, lane = threadIdx.x % TPT;
    cf *lds = reinterpret_cast<cf *>(hz_dyn_lds_fft) + sub * fv::lds_elems(N);
    const size_t t = (size_t)blockIdx.x * XPB + sub;
    const bool live = t < batch;
    const cf *src = (const cf *)in + t * N + lane;
    cf *dst = (cf *)out + t * N + lane;
    cf v[16];
    // time side = radix-R0 edge layout, frequency side = radix-16 edge layout
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = live ? src[FWD ? fv::edge_off<N, R0>(q) : fv::edge_off<N, 16>(q)] : cf{0.f, 0.f};
    if constexpr (FWD) fv::forward<N>(v, lds, tabs.fwd, lane); else fv::backward<N>(v, lds, tabs.bwd, lane);
    if (live) {
#pragma unroll
        for (int q = 0; q < 16; q++) dst[FWD ? fv::edge_off<N, 16>(q) : fv::edge_off<N, R0>(q)] = v[q];
    }
}

template <class K, class... A>
static int launch_dyn(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, A... args) {
    if (lds > 48 * 1024) HZ_TRY(raise_dynamic_lds((const void *)kernel));
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
    return HZSDR_OK;  // (the launch's own status: the caller's hipGetLastError)
}

template <int N>
static int launch_plan_n(hzsdr_ctx *ctx, const float2 *in, float2 *out, const float2 *tw, size_t batch,
                         bool fwd) {
    if constexpr (fv::ok(N)) {
        fv::FvTabs tabs{};
        HZ_TRY(get_fv_tables(ctx, N, &tabs));
        constexpr int XPB16 = fv::xpb(N);
        const dim3 grid16((unsigned)((batch + XPB16 - 1) / XPB16)), block16(fv::block(N));
        const size_t lds = (size_t)XPB16 * fv::lds_elems(N) * sizeof(fv::cf);
        if (fwd) HZ_TRY(launch_dyn(fft_plan_kernel16<N, true>, grid16, block16, lds, ctx->stream, in, out, tabs, batch));
        else HZ_TRY(launch_dyn(fft_plan_kernel16<N, false>, grid16, block16, lds, ctx->stream, in, out, tabs, batch));
        return HZSDR_OK;
    } else {
        constexpr int XPB = fft_xpb(N);
        size_t groups = (batch + XPB - 1) / XPB;
        unsigned grid = (unsigned)groups;
        if (fwd)
            hipLaunchKernelGGL((fft_plan_kernel<N, true>), dim3(grid), dim3(fft_block(N)), 0, ctx->stream, in, out, tw, batch);
        else
            hipLaunchKernelGGL((fft_plan_kernel<N, false>), dim3(grid), dim3(fft_block(N)), 0, ctx->stream, in, out, tw, batch);
        return HZSDR_OK;
    }
}

// ---- generic global-memory path ---------------------------------------------------------

__global__ void fft_global_r2_pass(const float2 *__restrict__ in, float2 *__restrict__ out,
                                   const float2 *__restrict__ tw, size_t n, size_t ns, size_t batch,
                                   bool inv) {
    const size_t half = n / 2, total = half * batch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const size_t t = g / half, j = g % half;
        const size_t k = j & (ns - 1);
        float2 a = in[t * n + j], b = in[t * n + j + half];
        if (ns > 1) {
            float2 w = tw[k * (n / (2 * ns))];
            if (inv) w.y = -w.y;
            b = cmulf(b, w);
        }
        const size_t j0 = ((j - k) << 1) + k;
        out[t * n + j0] = cadd(a, b);
        out[t * n + j0 + ns] = csub(a, b);
    }
}

static int fft_global(hzsdr_ctx *ctx, const float2 *in, float2 *out, size_t n, size_t batch, bool fwd) {
    const size_t bytes = n * batch * sizeof(float2);
    if (n == 1) {
        if (in != out) HZ_HIP(ctx, hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        return HZSDR_OK;
    }
    const float2 *tw;
    HZ_TRY(get_twiddles(ctx, n, &tw));
    HZ_TRY(ensure_slot(ctx, 10, bytes));
    HZ_TRY(ensure_slot(ctx, 11, bytes));
    float2 *ta = (float2 *)ctx->slots[10].ptr, *tb = (float2 *)ctx->slots[11].ptr;
    int passes = 0;
    for (size_t s = 1; s < n; s <<= 1) passes++;
    const float2 *src = in;
    size_t ns = 1;
    for (int p = 0; p < passes; p++, ns <<= 1) {
        float2 *dst = (p == passes - 1) ? out : ((p & 1) ? tb : ta);
        hipLaunchKernelGGL(fft_global_r2_pass, dim3(blocks_for(ctx, n / 2 * batch)), dim3(kThreads), 0,
                           ctx->stream, src, dst, tw, n, ns, batch, !fwd);
        src = dst;
    }
    return HZSDR_OK;
}

// ---- two-step path for N = N1 * N2 = 2^16 .. 2^24 (kerberos: 64 Ki, graft: 256 Ki) ----------
//
//   n = N2*n1 + n2,  k = k1 + N1*k2:
//   X[k1 + N1*k2] = sum_n2 W_N2^(n2 k2) * [ W_N^(n2 k1) * sum_n1 x[N2*n1 + n2] W_N1^(n1 k1) ]
//
// Kernel 1 ("columns"): a workgroup takes C = 4096/N1 adjacent columns n2, runs their
// N1-point transforms on the packed-math core, applies the W_N^(n2 k1) twiddle and writes
// A[k1][n2].  Lanes are mapped column-fastest, so both the strided input rows and the
// A rows are touched in C*8-byte contiguous pieces.
// Kernel 2 ("rows"): a workgroup takes C = 4096/N2 adjacent rows k1 of A (contiguous
// reads), runs their N2-point transforms, transposes the C x N2 result through LDS and
// writes X[k1 + N1*k2] in C*8-byte pieces.
// The big twiddle W_N^m (m < N) is the product of two table entries, m = a*2^s + b:
// two tables of 2^ceil(L/2) entries instead of one of N.

struct BigTw {
    const float2 *hi, *lo;  // hi[a] = exp(-2 pi i a 2^s / N), lo[b] = exp(-2 pi i b / N)
    int s;
};

__device__ __forceinline__ float2 big_twiddle(const BigTw &t, uint32_t m, bool inv) {
    const float2 a = t.hi[m >> t.s], b = t.lo[m & ((1u << t.s) - 1)];
    float2 w = cmulf(a, b);
    if (inv) w.y = -w.y;
    return w;
}

// (Round 4, tried: the forward pass's sixteen twiddles per lane from SEVEN table twiddles -- k1 = lane + q T, so
// W^(n2 k1) = W^(n2 lane) (W^(4 n2 T))^a (W^(n2 T))^b, q = 4 a + b: 14 gathers instead of 32 -- 74.5 us against 72.9:
// the gathers are not what the pass waits for; its 128-byte pieces at a power-of-two stride are.  Also tried, each
// measured per 2^24 points at N = 2^16 (tools/fft_pairs.py over a rocprofv3 kernel trace): N1 = 16 with a lane per
// column and no LDS (fully coalesced: the column pass 56.8 us -- but the row pass at N2 = 4096 then writes single
// 8-byte elements at a 128-byte stride: 244 us); the scratch between the passes in tiles of 16 k1 x 16 n2 so that
// both passes move 2 KB pieces (the column pass 84.5 us instead of 72.9: worse -- every workgroup's tiles of one
// register slot then sit 32 KB apart and the slots of all workgroups hit the same channels together); two adjacent
// columns per lane, 16-byte loads and stores, 256-byte pieces, the two transforms one behind the other through the
// same LDS (tools/fftbig_time.py: 120.8 us for both passes against 111.3 -- worse as well).)
// Which tile a workgroup takes.  The dispatcher deals a launch's workgroups to the eight XCDs in turn (linear index
// mod 8), and a tile's global pieces sit at (tile index) * 128 bytes modulo the row pitch: taken in launch order, the
// workgroups of ONE XCD would touch only the pieces whose index is congruent to that XCD's number -- the same few
// address bits, that XCD's same few L2 channels and fabric links, for the whole launch.  Instead every XCD walks its
// own CONTIGUOUS eighth of the launch's tiles (transforms and all).  A launch whose tile count is no multiple of eight
// keeps the launch order.
#ifndef HZ_FFT2_XCD
#define HZ_FFT2_XCD 1
#endif
#ifndef HZ_FFT2_ABL
#define HZ_FFT2_ABL 0  // timing ablations of the column pass (wrong results): 1 no big twiddle, 2 contiguous reads, 4 contiguous writes, 8 no transform
#endif
__device__ __forceinline__ void fft2_tile(uint32_t &bx, uint32_t &by) {
    bx = blockIdx.x, by = blockIdx.y;
    if constexpr (HZ_FFT2_XCD != 0) {
        const uint32_t total = gridDim.x * gridDim.y;
        if (total & 7u) return;
        const uint32_t w = blockIdx.x + gridDim.x * blockIdx.y, g = (w & 7u) * (total >> 3) + (w >> 3);
        bx = g % gridDim.x, by = g / gridDim.x;
    }
}

// (Round 6, measured and not kept: the column pass at N1 = 256 as PERSISTENT workgroups that walk their XCD's eighth of
// the tiles, the next tile's sixteen loads and four twiddle gathers in flight under this tile's transform, the
// transform's own twiddle rows in LDS -- what round 3's walk lost to, the gathers behind the prefetch, no longer exists.
// Correct (the FFT tests pass on it) and slower: 58-65 us per 2^24 points with three workgroups per CU, 57-63 with two,
// 65-69 with four, against 52-56 (profiles/r06_fft2_walk.txt; 140 registers).  As with the Shift map: what the pass lacks
// is not overlap inside a workgroup but workgroups in flight, and a walk has fewer.)
// THREADS: 256 lanes hold C = 4096 / N1 columns -- 128-byte pieces at N1 = 256, 8-byte ones at N1 = 4096 (N = 2^24);
// the long columns take wider workgroups, like the long rows below (dynamic LDS: up to 140 KB).  Measured (round 6,
// the column pass per 2^24 points, 256 | 512 | 1024 lanes): N1 = 512: 72 | 70 | 76 us; 1024: 96 | 86 | 93;
// 2048: 153 | 113 | 115; 4096: 278 | 161 | 121 (profiles/r06_fft2_colthreads.txt).
template <int N1> constexpr int cols_threads() { return N1 >= 4096 ? 1024 : N1 >= 1024 ? 512 : 256; }
#ifndef HZ_FFT2_LDS_SKEW
#define HZ_FFT2_LDS_SKEW 1
#endif
#ifndef HZ_FFT2_SKEW_UNIT
#define HZ_FFT2_SKEW_UNIT 32
#endif
// elements from one column's LDS region to the next (the kernel's comment says why), for the kernel and its launcher
template <int N1, int C> constexpr int cols_region() {
    return fv::lds_elems(N1) + (HZ_FFT2_LDS_SKEW && C > 1 ? ((HZ_FFT2_SKEW_UNIT / C) % 32 - fv::lds_elems(N1) % 32 + 32) % 32 : 0);
}
template <int N1, bool FWD, int THREADS>
__global__ __launch_bounds__(THREADS) void fft2_cols_kernel(const float2 *__restrict__ in, float2 *__restrict__ a_out,
                                                            fv::FvTabs tabs, BigTw bt, uint32_t n2_total) {
    using fv::cf;
    constexpr int TPT = fv::tpt(N1), C = THREADS / TPT, R0 = fv::first_radix(N1);
    [[maybe_unused]] constexpr int TILE = C * N1;  // (the ablation builds' contiguous tile)
    static_assert(C * TPT == THREADS && C >= 1, "whole columns per workgroup");
    // An 8-byte LDS access is served in two groups of 32 lanes over 64 banks: a group holds 32 / C lanes of each of
    // the C columns, so the columns' regions sit 32 / C elements (mod 32) apart and a group's 64 dwords fall on 64
    // banks -- both for the exchange's loads (a column's lanes an element apart) and for its stores at NS = 1 (17
    // elements apart).  lds_elems(256) = 272 = 16 mod 32 alone put all sixteen columns on two bank groups, eight lanes
    // to a bank: SQ_LDS_BANK_CONFLICT 31.5 M of 34.7 M LDS cycles per launch and XCD, waves waiting for LDS 21 % of their
    // time; now 2.1 M of 5.3 M, 0.4 % (profiles/r06_fft2_lds_conflicts*.txt; regions 64 / C apart, the first attempt:
    // 6.3 M of 9.5 M -- the two columns that then share a bank group still took turns).
    constexpr int REGION = cols_region<N1, C>();
    extern __shared__ __attribute__((aligned(16))) unsigned char fft2_cols_lds[];
    cf *lds_all = reinterpret_cast<cf *>(fft2_cols_lds);
    const int sub = threadIdx.x % C, lane = threadIdx.x / C;  // column-fastest
    cf *lds = lds_all + sub * REGION;
    uint32_t bx, by;
    fft2_tile(bx, by);
    const uint32_t n2 = bx * C + sub;
    const size_t base = (size_t)by * N1 * n2_total;  // batch
    // The twiddle of output k1 = lane + E(q) (E(q) = edge_off, the same for every lane) is W^(n2 lane) W^(n2 E(q)): the
    // second factor belongs to the COLUMN, sixteen values each -- one per lane of the workgroup, formed here from the
    // two tables and left in LDS (the transform's barriers order it); the first is one more table product per lane.
    // Four gathers per lane, issued beside the sixteen loads of the samples, where there were thirty-two behind the
    // transform (HZ_FFT2_TW=0): with those the pass took 68.5 us per 2^24 points at N = 2^16, without any twiddle 50.6
    // (tools/fft_ab.sh over -DHZ_FFT2_ABL builds; contiguous instead of strided pieces changed nothing).
#ifndef HZ_FFT2_TW
#define HZ_FFT2_TW 1
#endif
    cf *col_tw = lds_all + C * REGION;  // 16 * C entries behind the columns' regions
    cf w_lane;
    if constexpr (HZ_FFT2_TW != 0) {
        if (threadIdx.x < 16 * C) {
            const int q = threadIdx.x / C;
            constexpr int R = FWD ? 16 : R0;  // (edge_off<N1, R>(q) for a run-time q)
            const uint32_t e = (q / R) * TPT + (q % R) * (N1 / R);
            col_tw[threadIdx.x] = fv::from2(big_twiddle(bt, (bx * C + threadIdx.x % C) * e, !FWD));
        }
        w_lane = fv::from2(big_twiddle(bt, n2 * lane, !FWD));
    }
    cf v[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int n1 = FWD ? fv::edge_index<N1, R0>(q, lane) : fv::edge_index<N1, 16>(q, lane);
#if HZ_FFT2_ABL & 2
        v[q] = fv::from2(in[base + (size_t)bx * TILE + q * THREADS + threadIdx.x]);
#else
        v[q] = fv::from2(in[base + (size_t)n1 * n2_total + n2]);
#endif
    }
#if !(HZ_FFT2_ABL & 8)
    if constexpr (FWD) fv::forward<N1>(v, lds, tabs.fwd, lane); else fv::backward<N1>(v, lds, tabs.bwd, lane);
#endif
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const uint32_t k1 = FWD ? fv::edge_index<N1, 16>(q, lane) : fv::edge_index<N1, R0>(q, lane);
#if HZ_FFT2_ABL & 1
        const cf w = fv::from2(make_float2(1.0f, (float)bt.s));
#else
        cf w;
        if constexpr (HZ_FFT2_TW != 0) w = fv::cmul(w_lane, col_tw[q * C + sub]);
        else w = fv::from2(big_twiddle(bt, n2 * k1, !FWD));
#endif
#if HZ_FFT2_ABL & 4
        a_out[base + (size_t)bx * TILE + q * THREADS + threadIdx.x] = fv::to2(fv::cmul(v[q], w));
#else
        a_out[base + (size_t)k1 * n2_total + n2] = fv::to2(fv::cmul(v[q], w));
#endif
    }
}

// THREADS: 256, or more for the long rows -- C = 16 THREADS / N2 adjacent rows per tile, so that the output pieces (C
// adjacent k1 per k2) do not shrink to 32 and 8 bytes at N2 = 1024 and 4096 (dynamic LDS: above 64 KiB there).
// Measured (round 6, the row pass per 2^24 points, 256 | 512 | 1024 threads): N2 = 1024: 50 | 48 | 68 us;
// 2048: 72 | 59 | 65; 4096: 103 | 75 | 67.5 (profiles/r06_fft2_rowthreads.txt).
template <int N2> constexpr int rows_threads() { return N2 >= 4096 ? 1024 : N2 >= 1024 ? 512 : 256; }
template <int N2, bool FWD, int THREADS>
__global__ __launch_bounds__(THREADS) void fft2_rows_kernel(const float2 *__restrict__ a_in, float2 *__restrict__ out,
                                                            fv::FvTabs tabs, uint32_t n1_total) {
    using fv::cf;
    constexpr int TPT = fv::tpt(N2), C = THREADS / TPT, R0 = fv::first_radix(N2);
    static_assert(C * TPT == THREADS && C >= 1, "whole rows per workgroup");
    // (the transposed read-out below has C adjacent rows' lanes eight to a bank at lds_elems(256) = 272; regions set
    // apart to spread them, as in the column pass, were measured in round 6 and changed nothing: this pass does not
    // wait for its LDS)
    constexpr int REGION = fv::lds_elems(N2);
    extern __shared__ __attribute__((aligned(16))) unsigned char fft2_rows_lds[];
    cf *lds_all = reinterpret_cast<cf *>(fft2_rows_lds);
    const int sub = threadIdx.x / TPT, lane = threadIdx.x % TPT;  // lane-fastest: rows are contiguous
    cf *lds = lds_all + sub * REGION;
    uint32_t bx, by;
    fft2_tile(bx, by);
    const uint32_t k1 = bx * C + sub;
    const size_t base = (size_t)by * N2 * n1_total;
    cf v[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int n2 = FWD ? fv::edge_index<N2, R0>(q, lane) : fv::edge_index<N2, 16>(q, lane);
        v[q] = fv::from2(a_in[base + (size_t)k1 * N2 + n2]);
    }
    if constexpr (FWD) fv::forward<N2>(v, lds, tabs.fwd, lane); else fv::backward<N2>(v, lds, tabs.bwd, lane);
    __syncthreads();  // the transform's last LDS reads are done
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int k2 = FWD ? fv::edge_index<N2, 16>(q, lane) : fv::edge_index<N2, R0>(q, lane);
        lds[fv::pad(k2)] = v[q];
    }
    __syncthreads();
    // X[k1 + N1*k2]: C adjacent k1 per k2
    for (int e = threadIdx.x; e < C * N2; e += THREADS) {
        const int s2 = e % C, k2 = e / C;
        out[base + (size_t)k2 * n1_total + bx * C + s2] = fv::to2(lds_all[s2 * REGION + fv::pad(k2)]);
    }
}

// rows of 64 or 128 points (N = 2^14, 2^15: N1 = 256 columns on the packed-math core, the
// short rows on the radix-4 core of hz_fft.h, four points per lane)
// (round 6, measured: sixteen rows of 128 points per tile on 512 lanes -- 128-byte output pieces instead of 64 --
// 62-65 us per 2^24 points against 54-59 on 256 lanes: not the pieces; profiles/r06_fft2_smallrows.txt)
template <int N2, bool FWD>
__global__ __launch_bounds__(256) void fft2_rows_small_kernel(const float2 *__restrict__ a_in, float2 *__restrict__ out,
                                                              const float2 *__restrict__ tw2, uint32_t n1_total) {
    constexpr int TPT = fft_tpt(N2), C = 256 / TPT, CNT = N2 / TPT;
    constexpr int REGION = N2;
    __shared__ float2 lds_all[C * REGION];
    const int sub = threadIdx.x / TPT, lane = threadIdx.x % TPT;
    float2 *lds = lds_all + sub * REGION;
    uint32_t bx, by;
    fft2_tile(bx, by);
    const uint32_t k1 = bx * C + sub;
    const size_t base = (size_t)by * N2 * n1_total;
    FftRegs<N2> R;
#pragma unroll
    for (int q = 0; q < CNT; q++) {
        const int n2 = FWD ? (fft_odd(N2) ? edge2_index<N2>(q, lane) : edge4_index<N2>(q, lane)) : edge4_index<N2>(q, lane);
        R.v[q] = a_in[base + (size_t)k1 * N2 + n2];
    }
    if constexpr (FWD) fft_forward_regs<N2>(R, lds, tw2, lane); else fft_backward_regs<N2>(R, lds, tw2, lane);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CNT; q++) {
        const int k2 = FWD ? edge4_index<N2>(q, lane) : (fft_odd(N2) ? edge2_index<N2>(q, lane) : edge4_index<N2>(q, lane));
        lds[k2] = R.v[q];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * N2; e += 256) {
        const int s2 = e % C, k2 = e / C;
        out[base + (size_t)k2 * n1_total + bx * C + s2] = lds_all[s2 * REGION + k2];
    }
}

template <int N2> static void launch_rows_small(hzsdr_ctx *ctx, const float2 *a, float2 *out, const float2 *tw2,
                                                size_t n1, size_t batch, bool fwd) {
    constexpr int C = 256 / fft_tpt(N2);
    dim3 grid((unsigned)(n1 / C), (unsigned)batch);
    if (fwd) hipLaunchKernelGGL((fft2_rows_small_kernel<N2, true>), grid, dim3(256), 0, ctx->stream, a, out, tw2, (uint32_t)n1);
    else hipLaunchKernelGGL((fft2_rows_small_kernel<N2, false>), grid, dim3(256), 0, ctx->stream, a, out, tw2, (uint32_t)n1);
}

static int get_big_twiddles(hzsdr_ctx *ctx, size_t n, BigTw *bt) {
    int L = 0;
    while (((size_t)1 << L) < n) L++;
    const int s = L / 2;
    const size_t n_lo = (size_t)1 << s, n_hi = (size_t)1 << (L - s);
    const size_t key = (n << 1) | 1;  // shares the cache with the per-N tables (even keys)
    auto it = ctx->twiddles.find(key);
    float2 *d = nullptr;
    if (it != ctx->twiddles.end()) {
        d = (float2 *)it->second;
    } else {
        std::vector<float2> h(n_hi + n_lo);
        for (size_t a = 0; a < n_hi; a++) {
            double ang = -2.0 * M_PI * (double)(a << s) / (double)n;
            h[a] = make_float2((float)cos(ang), (float)sin(ang));
        }
        for (size_t b = 0; b < n_lo; b++) {
            double ang = -2.0 * M_PI * (double)b / (double)n;
            h[n_hi + b] = make_float2((float)cos(ang), (float)sin(ang));
        }
        HZ_HIP(ctx, hipMalloc((void **)&d, h.size() * sizeof(float2)));
        hipError_t e = hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return hip_fail(ctx, e, "twiddle upload", __FILE__, __LINE__);
        }
        ctx->twiddles[key] = d;
    }
    bt->hi = d;
    bt->lo = d + n_hi;
    bt->s = s;
    return HZSDR_OK;
}

template <int N1> static int launch_cols(hzsdr_ctx *ctx, const float2 *in, float2 *a, const fv::FvTabs &tw1,
                                         const BigTw &bt, size_t n2, size_t batch, bool fwd) {
    constexpr int T = cols_threads<N1>(), C = T / fv::tpt(N1);
    const dim3 grid((unsigned)(n2 / C), (unsigned)batch);
    const size_t lds = ((size_t)C * cols_region<N1, C>() + 16 * C) * sizeof(fv::cf);
    if (fwd) return launch_dyn(fft2_cols_kernel<N1, true, T>, grid, dim3(T), lds, ctx->stream, in, a, tw1, bt, (uint32_t)n2);
    return launch_dyn(fft2_cols_kernel<N1, false, T>, grid, dim3(T), lds, ctx->stream, in, a, tw1, bt, (uint32_t)n2);
}
template <int N2> static int launch_rows(hzsdr_ctx *ctx, const float2 *a, float2 *out, const fv::FvTabs &tw2,
                                         size_t n1, size_t batch, bool fwd) {
    constexpr int T = rows_threads<N2>(), C = T / fv::tpt(N2);
    const dim3 grid((unsigned)(n1 / C), (unsigned)batch);
    const size_t lds = (size_t)C * fv::lds_elems(N2) * sizeof(fv::cf);
    if (fwd) return launch_dyn(fft2_rows_kernel<N2, true, T>, grid, dim3(T), lds, ctx->stream, a, out, tw2, (uint32_t)n1);
    return launch_dyn(fft2_rows_kernel<N2, false, T>, grid, dim3(T), lds, ctx->stream, a, out, tw2, (uint32_t)n1);
}

static bool fft_two_step_ok(size_t n) { return n >= ((size_t)1 << 14) && n <= ((size_t)1 << 24) && (n & (n - 1)) == 0; }

// N1 as small as the packed-math core allows (best column coalescing), N2 = N / N1 <= 4096
static size_t two_step_n1(size_t n) {
    int L = 0, l1 = 8;
    while (((size_t)1 << L) < n) L++;
    while (L - l1 > 12) l1++;
    // (measured, round 6: N1 = N2 = 512 at 2^18 and 1024 at 2^20 -- 64- and 32-byte pieces on both sides instead of
    // 128-byte columns and short row pieces -- 120 against 117 us and 160 against 152 per 2^24 points: not better)
    return (size_t)1 << l1;
}

static int fft_two_step(hzsdr_ctx *ctx, const float2 *in, float2 *out, size_t n, size_t batch, bool fwd) {
    const size_t n1 = two_step_n1(n), n2 = n / n1;
    fv::FvTabs tw1{}, tw2{};
    const float2 *tw2s = nullptr;
    BigTw bt;
    HZ_TRY(get_fv_tables(ctx, n1, &tw1));
    if (n2 >= 256) HZ_TRY(get_fv_tables(ctx, n2, &tw2));
    else HZ_TRY(get_twiddles(ctx, n2, &tw2s));
    HZ_TRY(get_big_twiddles(ctx, n, &bt));
    HZ_TRY(ensure_slot(ctx, 10, n * batch * sizeof(float2)));
    float2 *a = (float2 *)ctx->slots[10].ptr;
    switch (n1) {
    case 256: HZ_TRY(launch_cols<256>(ctx, in, a, tw1, bt, n2, batch, fwd)); break;
    case 512: HZ_TRY(launch_cols<512>(ctx, in, a, tw1, bt, n2, batch, fwd)); break;
    case 1024: HZ_TRY(launch_cols<1024>(ctx, in, a, tw1, bt, n2, batch, fwd)); break;
    case 2048: HZ_TRY(launch_cols<2048>(ctx, in, a, tw1, bt, n2, batch, fwd)); break;
    default: HZ_TRY(launch_cols<4096>(ctx, in, a, tw1, bt, n2, batch, fwd)); break;
    }
    switch (n2) {
    case 64: launch_rows_small<64>(ctx, a, out, tw2s, n1, batch, fwd); break;
    case 128: launch_rows_small<128>(ctx, a, out, tw2s, n1, batch, fwd); break;
    case 256: HZ_TRY(launch_rows<256>(ctx, a, out, tw2, n1, batch, fwd)); break;
    case 512: HZ_TRY(launch_rows<512>(ctx, a, out, tw2, n1, batch, fwd)); break;
    case 1024: HZ_TRY(launch_rows<1024>(ctx, a, out, tw2, n1, batch, fwd)); break;
    case 2048: HZ_TRY(launch_rows<2048>(ctx, a, out, tw2, n1, batch, fwd)); break;
    default: HZ_TRY(launch_rows<4096>(ctx, a, out, tw2, n1, batch, fwd)); break;
    }
    return HZSDR_OK;
}

// ---- any length: Bluestein's chirp transform over the power-of-two kernels -----------------------------------
// The reference's Planner takes whatever length its buffers have (fft/fft.go:45-48), and stream.ConvolutionReader
// blocks on len(filter), whatever it is (stream/convolution.go:57-61): a 1000-tap or 1200-bin filter is ordinary.
// With n k = (n^2 + k^2 - (k - n)^2) / 2:
//     X[k] = c[k] sum_n (x[n] c[n]) b[k - n],   c[n] = exp(-+ i pi n^2 / N),  b = conj(c)
// -- a circular convolution of length M = the power of two >= 2 N - 1: two M-point transforms of the library's own
// plus three elementwise passes.  The chirp is formed from n^2 mod 2 N (exact integers) in float64; the chirp filter's
// spectrum B = FFT_M(b) / M is computed in float64 on the host once per (context, N) and rounded once.  Backward
// (unnormalised, like the power-of-two plans) = the same with every chirp conjugated; b is even, so its spectrum
// then is conj(B).  Accuracy: about that of two M-point transforms in a row (tests: <= 3e-7 log2 N relative L2).
constexpr size_t kBluesteinMax = (size_t)1 << 23;  // (M = 2^24, the largest power-of-two plan)
struct Bluestein {
    const float2 *c;  // N chirp values (forward)
    const float2 *B;  // M spectrum values of the chirp filter, / M
    size_t m;
};

static void host_fft64(std::vector<double> &re, std::vector<double> &im, size_t n) {  // radix-2, forward, in place
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(re[i], re[j]), std::swap(im[i], im[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len / 2;
        for (size_t k = 0; k < half; k++) {
            const double a = -2.0 * M_PI * (double)k / (double)len, wr = cos(a), wi = sin(a);
            for (size_t i = k; i < n; i += len) {
                const size_t u = i, v = i + half;
                const double tr = re[v] * wr - im[v] * wi, ti = re[v] * wi + im[v] * wr;
                re[v] = re[u] - tr, im[v] = im[u] - ti;
                re[u] += tr, im[u] += ti;
            }
        }
    }
}

static int get_bluestein(hzsdr_ctx *ctx, size_t n, Bluestein *out) {
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    out->m = m;
    // (keys of ctx->twiddles: the plain tables use n, the big twiddles 2 n + 1 -- both below 2^26 --, the packed core's
    // tables and this one a tag in bits 48 and up: disjoint by construction)
    static_assert(kBluesteinMax < ((size_t)1 << 40), "the length must stay below the key's tag");
    const size_t key = ((size_t)0xB1 << 48) | n;
    auto it = ctx->twiddles.find(key);
    if (it != ctx->twiddles.end()) {
        out->c = (const float2 *)it->second;
        out->B = out->c + n;
        return HZSDR_OK;
    }
    std::vector<double> br(m, 0.0), bi(m, 0.0);
    std::vector<float2> h(n + m);
    for (size_t i = 0; i < n; i++) {
        const unsigned long long q = ((unsigned long long)i * (unsigned long long)i) % (2ull * n);  // i^2 mod 2N: exact
        const double ang = M_PI * (double)q / (double)n;
        const double cr = cos(ang), ci = sin(ang);
        h[i] = make_float2((float)cr, (float)-ci);  // c[i] = exp(-i pi i^2 / N)
        br[i] = cr, bi[i] = ci;                     // b[i] = conj(c[i]), and b[M - i] = b[i]
        if (i) br[m - i] = cr, bi[m - i] = ci;
    }
    host_fft64(br, bi, m);
    const double inv = 1.0 / (double)m;
    for (size_t k = 0; k < m; k++) h[n + k] = make_float2((float)(br[k] * inv), (float)(bi[k] * inv));
    void *d = nullptr;
    HZ_HIP(ctx, hipMalloc(&d, h.size() * sizeof(float2)));
    hipError_t e = hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hip_fail(ctx, e, "chirp upload", __FILE__, __LINE__);
    }
    ctx->twiddles[key] = d;
    out->c = (const float2 *)d;
    out->B = out->c + n;
    return HZSDR_OK;
}

// a[b m + i] = x[b n + i] c[i] (i < n), 0 (n <= i < m); CONJ: the backward transform's chirp
template <bool CONJ>
__global__ __launch_bounds__(kThreads) void blue_pre_kernel(const float2 *__restrict__ x, float2 *__restrict__ a, const float2 *__restrict__ c,
                                                            uint32_t n, uint32_t m, size_t total) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const size_t b = g / m;
        const uint32_t i = (uint32_t)(g - b * m);
        float2 v = make_float2(0.f, 0.f);
        if (i < n) {
            float2 w = c[i];
            if (CONJ) w.y = -w.y;
            v = cmulf(x[b * n + i], w);
        }
        a[g] = v;
    }
}
// A[b m + k] *= B[k] (or conj(B[k]))
template <bool CONJ>
__global__ __launch_bounds__(kThreads) void blue_mul_kernel(float2 *__restrict__ a, const float2 *__restrict__ B, uint32_t m, size_t total) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        float2 w = B[g & (m - 1)];
        if (CONJ) w.y = -w.y;
        a[g] = cmulf(a[g], w);
    }
}
// X[b n + k] = p[b m + k] c[k]
template <bool CONJ>
__global__ __launch_bounds__(kThreads) void blue_post_kernel(const float2 *__restrict__ p, float2 *__restrict__ out, const float2 *__restrict__ c,
                                                             uint32_t n, uint32_t m, size_t total) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const size_t b = g / n;
        const uint32_t k = (uint32_t)(g - b * n);
        float2 w = c[k];
        if (CONJ) w.y = -w.y;
        out[g] = cmulf(p[b * m + k], w);
    }
}

int fft_device(hzsdr_ctx *ctx, const void *in, void *out, size_t n, size_t batch, bool fwd);

int fft_prepare(hzsdr_ctx *ctx, size_t n) {
    if (n <= 1 || !fft_length_ok(n)) return HZSDR_OK;
    if ((n & (n - 1)) == 0) {
        // every table the length's kernels read, so that a first transform inside somebody's timed or captured region
        // finds them (each getter uploads with a blocking copy the first time)
        const float2 *tw;
        fv::FvTabs tabs{};
        if (fft_two_step_ok(n)) {
            const size_t n1 = two_step_n1(n), n2 = n / n1;
            BigTw bt;
            HZ_TRY(get_fv_tables(ctx, n1, &tabs));
            if (n2 >= 256) HZ_TRY(get_fv_tables(ctx, n2, &tabs));
            else HZ_TRY(get_twiddles(ctx, n2, &tw));
            return get_big_twiddles(ctx, n, &bt);
        }
        if (fv::ok((int)n)) return get_fv_tables(ctx, n, &tabs);
        return get_twiddles(ctx, n, &tw);
    }
    Bluestein bl{};
    HZ_TRY(get_bluestein(ctx, n, &bl));
    return fft_prepare(ctx, bl.m);
}

static int fft_bluestein(hzsdr_ctx *ctx, const float2 *in, float2 *out, size_t n, size_t batch, bool fwd) {
    if (n > kBluesteinMax) return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "fft: a length that is not a power of two must be <= 2^23");
    Bluestein bl{};
    HZ_TRY(get_bluestein(ctx, n, &bl));
    const size_t m = bl.m;
    // (in pieces of at most 2^25 points of scratch per buffer: a large batch does not need batch * M of it)
    const size_t per = std::max<size_t>(1, ((size_t)1 << 25) / m);
    HZ_TRY(ensure_slot(ctx, 12, std::min(batch, per) * m * sizeof(float2)));
    HZ_TRY(ensure_slot(ctx, 13, std::min(batch, per) * m * sizeof(float2)));
    float2 *a = (float2 *)ctx->slots[12].ptr, *A = (float2 *)ctx->slots[13].ptr;
    for (size_t b0 = 0; b0 < batch; b0 += per) {
        const size_t nb = std::min(per, batch - b0);
        const float2 *x = in + b0 * n;
        float2 *y = out + b0 * n;
        const dim3 gm(blocks_for(ctx, nb * m)), gn(blocks_for(ctx, nb * n)), blk(kThreads);
        if (fwd) hipLaunchKernelGGL(blue_pre_kernel<false>, gm, blk, 0, ctx->stream, x, a, bl.c, (uint32_t)n, (uint32_t)m, nb * m);
        else hipLaunchKernelGGL(blue_pre_kernel<true>, gm, blk, 0, ctx->stream, x, a, bl.c, (uint32_t)n, (uint32_t)m, nb * m);
        HZ_TRY(fft_device(ctx, a, A, m, nb, true));
        if (fwd) hipLaunchKernelGGL(blue_mul_kernel<false>, gm, blk, 0, ctx->stream, A, bl.B, (uint32_t)m, nb * m);
        else hipLaunchKernelGGL(blue_mul_kernel<true>, gm, blk, 0, ctx->stream, A, bl.B, (uint32_t)m, nb * m);
        HZ_TRY(fft_device(ctx, A, a, m, nb, false));
        if (fwd) hipLaunchKernelGGL(blue_post_kernel<false>, gn, blk, 0, ctx->stream, (const float2 *)a, y, bl.c, (uint32_t)n, (uint32_t)m, nb * n);
        else hipLaunchKernelGGL(blue_post_kernel<true>, gn, blk, 0, ctx->stream, (const float2 *)a, y, bl.c, (uint32_t)n, (uint32_t)m, nb * n);
    }
    HZ_HIP(ctx, hipGetLastError());
    return HZSDR_OK;
}

// Transform `batch` consecutive length-n blocks of device memory (any length: see above).
int fft_device(hzsdr_ctx *ctx, const void *in, void *out, size_t n, size_t batch, bool fwd) {
    if (batch == 0 || n == 0) return HZSDR_OK;
    const float2 *i = (const float2 *)in;
    float2 *o = (float2 *)out;
    if ((n & (n - 1)) != 0) return fft_bluestein(ctx, i, o, n, batch, fwd);
    if (fft_two_step_ok(n)) return fft_two_step(ctx, i, o, n, batch, fwd);
    if (!fft_lds_ok(n)) return fft_global(ctx, i, o, n, batch, fwd);
    const float2 *tw = nullptr;
    if (!fv::ok((int)n)) HZ_TRY(get_twiddles(ctx, n, &tw));
    switch (n) {
    case 4: return launch_plan_n<4>(ctx, i, o, tw, batch, fwd);
    case 8: return launch_plan_n<8>(ctx, i, o, tw, batch, fwd);
    case 16: return launch_plan_n<16>(ctx, i, o, tw, batch, fwd);
    case 32: return launch_plan_n<32>(ctx, i, o, tw, batch, fwd);
    case 64: return launch_plan_n<64>(ctx, i, o, tw, batch, fwd);
    case 128: return launch_plan_n<128>(ctx, i, o, tw, batch, fwd);
    case 256: return launch_plan_n<256>(ctx, i, o, tw, batch, fwd);
    case 512: return launch_plan_n<512>(ctx, i, o, tw, batch, fwd);
    case 1024: return launch_plan_n<1024>(ctx, i, o, tw, batch, fwd);
    case 2048: return launch_plan_n<2048>(ctx, i, o, tw, batch, fwd);
    case 4096: return launch_plan_n<4096>(ctx, i, o, tw, batch, fwd);
    case 8192: return launch_plan_n<8192>(ctx, i, o, tw, batch, fwd);
    default: return HZSDR_ERR_INVALID_ARGUMENT;
    }
}

// f1[i] = f1[i] * f2[i] or f1[i] * conj(f2[i]) with Go complex64 semantics
// (fft/convolution.go:107-109, :129-134).
__global__ void pointwise_mul_kernel(float2 *f1, const float2 *__restrict__ f2, size_t n, bool conj) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float2 b = f2[i];
        if (conj) b.y = -b.y;
        f1[i] = go_cmul(f1[i], b);
    }
}

__global__ void pointwise_mul_blocks_kernel(float2 *f1, const float2 *__restrict__ f2, size_t period, size_t total) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) f1[i] = go_cmul(f1[i], f2[i % period]);
}
void pointwise_mul_blocks_device(hzsdr_ctx *ctx, void *f1, const void *f2, size_t period, size_t nblocks) {
    if (period == 0 || nblocks == 0) return;
    hipLaunchKernelGGL(pointwise_mul_blocks_kernel, dim3(blocks_for(ctx, period * nblocks)), dim3(kThreads), 0, ctx->stream, (float2 *)f1,
                       (const float2 *)f2, period, period * nblocks);
}

void pointwise_mul_device(hzsdr_ctx *ctx, void *f1, const void *f2, size_t n, bool conj) {
    if (n == 0) return;
    hipLaunchKernelGGL(pointwise_mul_kernel, dim3(blocks_for(ctx, n)), dim3(kThreads), 0, ctx->stream,
                       (float2 *)f1, (const float2 *)f2, n, conj);
}

}  // namespace hz

struct hzsdr_fft {
    hzsdr_ctx *ctx;
    void *iq, *freq;
    size_t n, batch;
    bool fwd;
};

extern "C" {

int hzsdr_fft_plan_batch(hzsdr_ctx *ctx, void *iq, void *freq, size_t n, size_t batch, int direction,
                         hzsdr_fft **out) {
    using namespace hz;
    if (!ctx || !out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    const bool pow2 = (n & (n - 1)) == 0;
    if (n == 0 || (pow2 && n > ((size_t)1 << 24)) || (!pow2 && n > kBluesteinMax))
        return fail(ctx, HZSDR_ERR_INVALID_ARGUMENT, "fft: length must be 1 ... 2^24 (a power of two) or 1 ... 2^23 (any other)");
    if (batch == 0 || !iq || !freq) return HZSDR_ERR_INVALID_ARGUMENT;
    if (direction != HZSDR_FFT_FORWARD && direction != HZSDR_FFT_BACKWARD) return HZSDR_ERR_INVALID_ARGUMENT;
    HZ_TRY(enter(ctx));
    if (n > 1) {  // plan-time cost, like any planner: the tables (a chirp transform's: formed once per context and length)
        HZ_TRY(fft_prepare(ctx, n));
        // ... and the two-step lengths' scratch between the passes
        if (pow2 && fft_two_step_ok(n)) HZ_TRY(ensure_slot(ctx, 10, n * batch * sizeof(float2)));
    }
    *out = new hzsdr_fft{ctx, iq, freq, n, batch, direction == HZSDR_FFT_FORWARD};
    return HZSDR_OK;
}

int hzsdr_fft_plan(hzsdr_ctx *ctx, void *iq, size_t iq_len, void *freq, size_t freq_len, int direction,
                   hzsdr_fft **out) {
    if (!ctx || !out) return HZSDR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (iq_len != freq_len)  // testutils/fft.go:127-137
        return hz::fail(ctx, HZSDR_ERR_DST_TOO_SMALL, "fft: iq and frequency lengths differ");
    return hzsdr_fft_plan_batch(ctx, iq, freq, iq_len, 1, direction, out);
}

int hzsdr_fft_transform(hzsdr_fft *p) {
    using namespace hz;
    if (!p) return HZSDR_ERR_INVALID_ARGUMENT;
    hzsdr_ctx *ctx = p->ctx;
    HZ_TRY(enter(ctx));
    const size_t bytes = p->n * p->batch * 8;
    Stage st(ctx);
    const void *din;
    void *dout;
    if (p->fwd) {
        HZ_TRY(st.in(0, p->iq, bytes, &din));
        HZ_TRY(st.out(1, p->freq, bytes, &dout));
    } else {
        HZ_TRY(st.in(0, p->freq, bytes, &din));
        HZ_TRY(st.out(1, p->iq, bytes, &dout));
    }
    HZ_TRY(fft_device(ctx, din, dout, p->n, p->batch, p->fwd));
    return st.finish();
}

int hzsdr_fft_free(hzsdr_fft *p) {
    if (!p) return HZSDR_ERR_INVALID_ARGUMENT;
    delete p;
    return HZSDR_OK;
}

}  // extern "C"

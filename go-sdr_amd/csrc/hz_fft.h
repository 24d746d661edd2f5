// hz_fft.h -- the workgroup-level FFT core shared by the Plan kernels
// (hz_fft.hip) and the fused convolution / FIR-decimate kernels (hz_conv.hip, hz_chain_fir.hip).
//
// Stockham autosort, power-of-two N, 4 <= N <= 8192, one transform per group of
// TPT = min(256, N/4) lanes, data resident in LDS (N * 8 bytes) between passes:
//
//   pass with radix R and Ns = product of earlier radices, butterfly j < N/R:
//     k = j mod Ns;  in[r] = src[j + r*N/R] * w^(r*k),  w = exp(-+2 pi i/(Ns*R))
//     out = DFT_R(in);  dst[(j-k)*R + k + r*Ns] = out[r]
//
// The FIRST pass takes its inputs straight from a caller-supplied loader into
// registers and the LAST pass hands its outputs to a caller-supplied sink from
// registers, both at indices j + r*N/4: coalesced 8-byte-per-lane global
// accesses, and -- the point of the layout -- the register image the forward
// transform ends with is exactly the one the backward transform starts from,
// so forward -> pointwise multiply -> backward (fft/convolution.go:183-191)
// runs in one kernel with no LDS or HBM round trip in between.
//
// Radix plan: all radix-4; when log2 N is odd one radix-2 pass is placed FIRST
// in forward plans and LAST in backward plans (so both ends facing the
// "frequency side" are radix-4).  No MFMA: butterflies are f32 VALU work.
//
// Numerics: float32 butterflies with FMA contraction allowed (the reference
// has no FFT of its own -- fft/fft.go:45-59 is an interface -- so there is no
// bit pattern to match; tests hold the result to an error bound against a
// float64 transform).  Twiddles come from a table computed in float64.
#pragma once
#include <hip/hip_runtime.h>

namespace hz {

constexpr int kFftMaxLds = 8192;

constexpr int fft_tpt(int n) { return n / 4 < 256 ? (n / 4 < 1 ? 1 : n / 4) : 256; }  // lanes per transform
constexpr int fft_block(int n) { return fft_tpt(n) < 64 ? 64 : fft_tpt(n); }          // lanes per workgroup
constexpr int fft_xpb(int n) { return fft_block(n) / fft_tpt(n); }                    // transforms per workgroup
// __launch_bounds__ second argument (minimum waves per SIMD): what the LDS
// footprint lets a CU hold anyway, so the register allocator does not trade
// occupancy for a few hoisted loads (N*8 bytes of LDS per 256-lane workgroup).
constexpr int fft_waves(int n) { return n <= 2048 ? 8 : (n <= 4096 ? 4 : 2); }
constexpr bool fft_odd(int n) {
    int l = 0;
    while ((1 << l) < n) l++;
    return l & 1;
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
#pragma clang fp contract(fast)
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// Registers of one lane: N/TPT complex values.  For a radix-R pass they are
// viewed as v[b*R + r], butterfly j = b*TPT + lane, b < N/(R*TPT).
template <int N> struct FftRegs {
    static constexpr int TPT = fft_tpt(N);
    static constexpr int CNT = N / TPT;
    float2 v[CNT];
};

// ---- radix-4 pass pieces ------------------------------------------------------

template <int N, bool INV>
__device__ __forceinline__ void r4_twiddle_butterfly(FftRegs<N> &R, const float2 *__restrict__ tw,
                                                     int lane, int Ns) {
    constexpr int TPT = fft_tpt(N), B = N / (4 * TPT);
#pragma unroll
    for (int b = 0; b < B; b++) {
        float2 *v = &R.v[4 * b];
        if (Ns > 1) {
            const int j = b * TPT + lane;
            const int k = j & (Ns - 1);
            const int stride = N / (Ns * 4);
#pragma unroll
            for (int r = 1; r < 4; r++) {
                float2 w = tw[k * r * stride];
                if (INV) w.y = -w.y;
                v[r] = cmulf(v[r], w);
            }
        }
        float2 a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]);
        float2 a2 = cadd(v[1], v[3]), d = csub(v[1], v[3]);
        float2 a3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);  // *(+i) / *(-i)
        v[0] = cadd(a0, a2);
        v[1] = cadd(a1, a3);
        v[2] = csub(a0, a2);
        v[3] = csub(a1, a3);
    }
}

template <int N>
__device__ __forceinline__ void r4_load_lds(FftRegs<N> &R, const float2 *lds, int lane) {
    constexpr int TPT = fft_tpt(N), B = N / (4 * TPT);
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) R.v[4 * b + r] = lds[b * TPT + lane + r * (N / 4)];
}

template <int N>
__device__ __forceinline__ void r4_store_lds(const FftRegs<N> &R, float2 *lds, int lane, int Ns) {
    constexpr int TPT = fft_tpt(N), B = N / (4 * TPT);
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int j = b * TPT + lane;
        const int k = j & (Ns - 1);
        const int j0 = ((j - k) << 2) + k;
#pragma unroll
        for (int r = 0; r < 4; r++) lds[j0 + r * Ns] = R.v[4 * b + r];
    }
}

// ---- radix-2 pass pieces (only when log2 N is odd) --------------------------------
// view: v[b*2 + r], butterfly j = b*TPT + lane, b < N/(2*TPT)

template <int N, bool INV>
__device__ __forceinline__ void r2_twiddle_butterfly(FftRegs<N> &R, const float2 *__restrict__ tw,
                                                     int lane, int Ns) {
    constexpr int TPT = fft_tpt(N), B = N / (2 * TPT);
#pragma unroll
    for (int b = 0; b < B; b++) {
        float2 *v = &R.v[2 * b];
        if (Ns > 1) {
            const int j = b * TPT + lane;
            const int k = j & (Ns - 1);
            float2 w = tw[k * (N / (Ns * 2))];
            if (INV) w.y = -w.y;
            v[1] = cmulf(v[1], w);
        }
        float2 s = cadd(v[0], v[1]), d = csub(v[0], v[1]);
        v[0] = s;
        v[1] = d;
    }
}

template <int N>
__device__ __forceinline__ void r2_load_lds(FftRegs<N> &R, const float2 *lds, int lane) {
    constexpr int TPT = fft_tpt(N), B = N / (2 * TPT);
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
        for (int r = 0; r < 2; r++) R.v[2 * b + r] = lds[b * TPT + lane + r * (N / 2)];
}

template <int N>
__device__ __forceinline__ void r2_store_lds(const FftRegs<N> &R, float2 *lds, int lane, int Ns) {
    constexpr int TPT = fft_tpt(N), B = N / (2 * TPT);
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int j = b * TPT + lane;
        const int k = j & (Ns - 1);
        const int j0 = ((j - k) << 1) + k;
#pragma unroll
        for (int r = 0; r < 2; r++) lds[j0 + r * Ns] = R.v[2 * b + r];
    }
}

// ---- whole transforms ---------------------------------------------------------------
// `lds` points at this transform's N-element LDS region, `lane` < TPT.  All
// lanes of the workgroup must call together (the passes contain barriers).

// Index of register slot q in the "radix-4 edge" layout: element j + r*N/4.
template <int N> __device__ __forceinline__ int edge4_index(int q, int lane) {
    constexpr int TPT = fft_tpt(N);
    return (q >> 2) * TPT + lane + (q & 3) * (N / 4);
}
// Index of register slot q in the "radix-2 edge" layout: element j + r*N/2.
template <int N> __device__ __forceinline__ int edge2_index(int q, int lane) {
    constexpr int TPT = fft_tpt(N);
    return (q >> 1) * TPT + lane + (q & 1) * (N / 2);
}

// FORWARD transform.  On entry R holds the inputs in the edge layout of the
// first pass (edge2 when log2 N is odd, else edge4); on exit R holds the
// spectrum in edge4 layout.
// FROM_LDS: the caller filled R from `lds` itself (staged inputs), so other
// lanes may still be reading it: barrier before the first in-place store.
template <int N, bool FROM_LDS = false>
__device__ __forceinline__ void fft_forward_regs(FftRegs<N> &R, float2 *lds, const float2 *tw, int lane) {
    int Ns = 1;
    if constexpr (fft_odd(N)) {
        r2_twiddle_butterfly<N, false>(R, tw, lane, 1);
        if constexpr (FROM_LDS) __syncthreads();
        r2_store_lds<N>(R, lds, lane, 1);
        Ns = 2;
    } else {
        r4_twiddle_butterfly<N, false>(R, tw, lane, 1);
        if constexpr (N == 4) return;
        if constexpr (FROM_LDS) __syncthreads();
        r4_store_lds<N>(R, lds, lane, 1);
        Ns = 4;
    }
    for (;;) {
        __syncthreads();
        r4_load_lds<N>(R, lds, lane);
        r4_twiddle_butterfly<N, false>(R, tw, lane, Ns);
        if (Ns * 4 >= N) break;  // last pass: results stay in registers
        __syncthreads();
        r4_store_lds<N>(R, lds, lane, Ns);
        Ns *= 4;
    }
    // last-pass outputs sit at j0 + r*Ns with Ns = N/4, j0 = j: the edge4 layout
}

// BACKWARD (unnormalised) transform.  On entry R holds the spectrum in edge4
// layout; on exit R holds the time samples in the edge layout of the last
// pass (edge2 when log2 N is odd, else edge4).
template <int N>
__device__ __forceinline__ void fft_backward_regs(FftRegs<N> &R, float2 *lds, const float2 *tw, int lane) {
    constexpr int LAST4 = fft_odd(N) ? N / 2 : N;  // radix-4 passes cover Ns = 1 .. LAST4/4
    r4_twiddle_butterfly<N, true>(R, tw, lane, 1);
    if constexpr (N == 4) return;
    // caller may have other lanes still reading lds from a previous transform
    __syncthreads();
    r4_store_lds<N>(R, lds, lane, 1);
    int Ns = 4;
    while (Ns < LAST4) {
        __syncthreads();
        r4_load_lds<N>(R, lds, lane);
        r4_twiddle_butterfly<N, true>(R, tw, lane, Ns);
        if (!fft_odd(N) && Ns * 4 >= N) return;  // last pass (even log2 N)
        __syncthreads();
        r4_store_lds<N>(R, lds, lane, Ns);
        Ns *= 4;
    }
    if constexpr (fft_odd(N)) {
        __syncthreads();
        r2_load_lds<N>(R, lds, lane);
        r2_twiddle_butterfly<N, true>(R, tw, lane, N / 2);
    }
}

// Host: N must be a power of two within the LDS kernels' range.
inline bool fft_lds_ok(size_t n) { return n >= 4 && n <= (size_t)kFftMaxLds && (n & (n - 1)) == 0; }

}  // namespace hz

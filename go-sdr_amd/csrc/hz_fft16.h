// hz_fft16.h -- workgroup FFT core, second generation, for N = 256 .. 8192.
//
// Sixteen points per lane, N/16 lanes per transform.  Each pass is a register
// DFT of radix R in {2, 4, 8, 16} (radix 16 = 4x4 with constant W16 twiddles),
// so a lane runs 16/R butterflies per pass:
//
//   N:     256      512        1024       2048       4096         8192
//   plan:  16,16    2,16,16    4,16,16    8,16,16    16,16,16     2,16,16,16   (forward)
//   lanes: 16       32         64         128        256          512
//
// i.e. THREE passes and TWO LDS exchanges for 4096 points (the radix-4 core in
// hz_fft.h needs six and ten), and a 1024-point transform lives in ONE wave, so
// its exchanges need no cross-wave barrier at all.  Backward plans are the
// forward plans reversed, which makes the register image at the frequency side
// the same for both directions (radix-16 edge: element j + r*N/16) -- forward ->
// bins * filter -> backward stays in registers -- and likewise at the time side.
//
// Stockham indexing, pass (R, Ns): butterfly j = b*TPT + lane, k = j mod Ns,
//   in[r]  = src[j + r*N/R] * w^(r*k),   w = exp(-+2 pi i / (Ns*R))
//   dst[(j-k)*R + k + r*Ns] = DFT_R(in)[r]
// LDS is padded by one element per 16 (index i lives at i + i/16): the
// stride-R stores of the first pass become conflict-free, reads stay contiguous.
//
// Inter-pass twiddles of a radix-16 pass: six table loads (w^1,2,3,4,8,12) and
// nine products give w^1..15.  Table: exp(-2 pi i m / N) computed in float64.
// float32 butterflies, FMA contraction allowed (no reference bit pattern exists
// for an FFT: fft/fft.go:45-59 is an interface; tests hold an error bound).
#pragma once
#include <hip/hip_runtime.h>

namespace hz {
namespace f16 {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
#pragma clang fp contract(fast)
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward) or +i (inverse)
template <bool INV> __device__ __forceinline__ float2 mul_mi(float2 a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
template <bool INV> __device__ __forceinline__ float2 cj(float2 w) { return INV ? make_float2(w.x, -w.y) : w; }

constexpr int first_radix(int n) {
    return n == 256 ? 16 : n == 512 ? 2 : n == 1024 ? 4 : n == 2048 ? 8 : n == 4096 ? 16 : 2;
}
constexpr int tpt(int n) { return n / 16; }
constexpr int block(int n) { return tpt(n) < 64 ? 64 : tpt(n); }
constexpr int xpb(int n) { return block(n) / tpt(n); }
constexpr int lds_elems(int n) { return n + n / 16; }
// 8192 would need 68 KiB of padded LDS (> the 64 KiB static limit): it stays on the radix-4 core
constexpr bool ok(int n) { return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096; }
__host__ __device__ constexpr int pad(int i) { return i + (i >> 4); }

// ---- register DFTs, natural-order in and out ------------------------------------------

template <bool INV> __device__ __forceinline__ void dft2(float2 *v) {
    float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}

template <bool INV> __device__ __forceinline__ void dft4(float2 &x0, float2 &x1, float2 &x2, float2 &x3) {
    float2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = mul_mi<INV>(csub(x1, x3));
    x0 = cadd(a0, a2);
    x1 = cadd(a1, a3);
    x2 = csub(a0, a2);
    x3 = csub(a1, a3);
}
template <bool INV> __device__ __forceinline__ void dft4(float2 *v) { dft4<INV>(v[0], v[1], v[2], v[3]); }

template <bool INV> __device__ __forceinline__ void dft8(float2 *v) {
    // n = c + 2d, K = b + 4a: two DFT4 over d, twiddle W8^(c b), radix-2 over c
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4<INV>(e0, e1, e2, e3);
    dft4<INV>(o0, o1, o2, o3);
    const float h = 0.70710678118654752440f;
    // W8^1 = h(1 -+ i), W8^2 = -+i, W8^3 = h(-1 -+ i)
    float2 t1 = INV ? make_float2(h * (o1.x - o1.y), h * (o1.x + o1.y)) : make_float2(h * (o1.x + o1.y), h * (o1.y - o1.x));
    float2 t2 = mul_mi<INV>(o2);
    float2 t3 = INV ? make_float2(-h * (o3.x + o3.y), h * (o3.x - o3.y)) : make_float2(h * (o3.y - o3.x), -h * (o3.x + o3.y));
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, t1); v[5] = csub(e1, t1);
    v[2] = cadd(e2, t2); v[6] = csub(e2, t2);
    v[3] = cadd(e3, t3); v[7] = csub(e3, t3);
}

template <bool INV> __device__ __forceinline__ void dft16(float2 *v) {
    // n = c + 4d, K = b + 4a:  X[b+4a] = sum_c W4^(c a) W16^(c b) sum_d x[c+4d] W4^(d b)
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    // stage A: DFT4 over d for each c (in place: v[c + 4d] -> t[c][b] stored at v[c + 4b])
#pragma unroll
    for (int c = 0; c < 4; c++) dft4<INV>(v[c], v[c + 4], v[c + 8], v[c + 12]);
    // twiddle t[c][b] *= W16^(c b), W16 = exp(-+ 2 pi i / 16)
    const float2 W1 = cj<INV>(make_float2(c1, -s1)), W2 = cj<INV>(make_float2(h, -h)),
                 W3 = cj<INV>(make_float2(s1, -c1)), W6 = cj<INV>(make_float2(-h, -h)),
                 W9 = cj<INV>(make_float2(-c1, s1));
    v[1 + 4 * 1] = cmul(v[1 + 4 * 1], W1);
    v[1 + 4 * 2] = cmul(v[1 + 4 * 2], W2);
    v[1 + 4 * 3] = cmul(v[1 + 4 * 3], W3);
    v[2 + 4 * 1] = cmul(v[2 + 4 * 1], W2);
    v[2 + 4 * 2] = mul_mi<INV>(v[2 + 4 * 2]);  // W16^4
    v[2 + 4 * 3] = cmul(v[2 + 4 * 3], W6);
    v[3 + 4 * 1] = cmul(v[3 + 4 * 1], W3);
    v[3 + 4 * 2] = cmul(v[3 + 4 * 2], W6);
    v[3 + 4 * 3] = cmul(v[3 + 4 * 3], W9);
    // stage B: DFT4 over c for each b: inputs v[c + 4b], outputs X[b + 4a] -> v[b + 4a]
    float2 out[16];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        float2 y0 = v[0 + 4 * b], y1 = v[1 + 4 * b], y2 = v[2 + 4 * b], y3 = v[3 + 4 * b];
        dft4<INV>(y0, y1, y2, y3);
        out[b] = y0;
        out[b + 4] = y1;
        out[b + 8] = y2;
        out[b + 12] = y3;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = out[i];
}

template <int R, bool INV> __device__ __forceinline__ void dft(float2 *v) {
    if constexpr (R == 2) dft2<INV>(v);
    else if constexpr (R == 4) dft4<INV>(v);
    else if constexpr (R == 8) dft8<INV>(v);
    else dft16<INV>(v);
}

// ---- one pass, in pieces ------------------------------------------------------------------

// register slot q of a radix-R pass <-> element index at the pass input
template <int N, int R> __device__ __forceinline__ int edge_index(int q, int lane) {
    return (q / R) * tpt(N) + lane + (q % R) * (N / R);
}

template <int N, int R> __device__ __forceinline__ void load_lds(float2 *v, const float2 *lds, int lane) {
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = lds[pad(edge_index<N, R>(q, lane))];
}

template <int N, int R> __device__ __forceinline__ void store_lds(const float2 *v, float2 *lds, int lane, int Ns) {
#pragma unroll
    for (int b = 0; b < 16 / R; b++) {
        const int j = b * tpt(N) + lane;
        const int k = j & (Ns - 1);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; r++) lds[pad(j0 + r * Ns)] = v[b * R + r];
    }
}

// Inter-pass twiddles in two phases so the table loads (L2 latency) can be issued
// BEFORE the workgroup barrier and the LDS reads of the pass they belong to.
template <int R> struct TwRegs {
    static constexpr int CNT = R == 16 ? 6 : (R - 1) * (16 / R);
    float2 w[CNT];
};

template <int N, int R>
__device__ __forceinline__ void twiddle_load(TwRegs<R> &t, const float2 *__restrict__ tw, int lane, int Ns) {
#pragma unroll
    for (int b = 0; b < 16 / R; b++) {
        const int j = b * tpt(N) + lane;
        const int k = j & (Ns - 1);
        const int base = k * (N / (Ns * R));
        if constexpr (R == 16) {
            t.w[0] = tw[base]; t.w[1] = tw[2 * base]; t.w[2] = tw[3 * base];
            t.w[3] = tw[4 * base]; t.w[4] = tw[8 * base]; t.w[5] = tw[12 * base];
        } else {
#pragma unroll
            for (int r = 1; r < R; r++) t.w[b * (R - 1) + r - 1] = tw[base * r];
        }
    }
}

// multiply inputs by w^(r k); Ns > 1
template <int N, int R, bool INV>
__device__ __forceinline__ void twiddle_apply(float2 *v, const TwRegs<R> &t) {
#pragma unroll
    for (int b = 0; b < 16 / R; b++) {
        float2 *x = v + b * R;
        if constexpr (R == 16) {
            const float2 w1 = cj<INV>(t.w[0]), w2 = cj<INV>(t.w[1]), w3 = cj<INV>(t.w[2]);
            const float2 w4 = cj<INV>(t.w[3]), w8 = cj<INV>(t.w[4]), w12 = cj<INV>(t.w[5]);
            x[1] = cmul(x[1], w1);
            x[2] = cmul(x[2], w2);
            x[3] = cmul(x[3], w3);
            x[4] = cmul(x[4], w4);
            x[5] = cmul(x[5], cmul(w4, w1));
            x[6] = cmul(x[6], cmul(w4, w2));
            x[7] = cmul(x[7], cmul(w4, w3));
            x[8] = cmul(x[8], w8);
            x[9] = cmul(x[9], cmul(w8, w1));
            x[10] = cmul(x[10], cmul(w8, w2));
            x[11] = cmul(x[11], cmul(w8, w3));
            x[12] = cmul(x[12], w12);
            x[13] = cmul(x[13], cmul(w12, w1));
            x[14] = cmul(x[14], cmul(w12, w2));
            x[15] = cmul(x[15], cmul(w12, w3));
        } else {
#pragma unroll
            for (int r = 1; r < R; r++) x[r] = cmul(x[r], cj<INV>(t.w[b * (R - 1) + r - 1]));
        }
    }
}

template <int R, bool INV> __device__ __forceinline__ void butterflies(float2 *v) {
#pragma unroll
    for (int b = 0; b < 16 / R; b++) dft<R, INV>(v + b * R);
}

// ---- whole transforms on registers -----------------------------------------------------------
// v[16]: on entry the inputs in the edge layout of the first pass's radix, on exit
// the outputs in the edge layout of the last pass's radix.  `lds` = this
// transform's padded region (lds_elems(N)).  Every lane of the workgroup calls.
//
// FROM_LDS: the caller loaded v from `lds` itself, so a barrier precedes the
// first store.

template <int N, bool FROM_LDS = false>
__device__ __forceinline__ void forward(float2 *v, float2 *lds, const float2 *tw, int lane) {
    constexpr int R0 = first_radix(N);
    butterflies<R0, false>(v);  // Ns = 1: no twiddles
    if constexpr (FROM_LDS) __syncthreads();
    store_lds<N, R0>(v, lds, lane, 1);
    int Ns = R0;
    for (;;) {
        TwRegs<16> t;
        twiddle_load<N, 16>(t, tw, lane, Ns);  // in flight across the barrier and the LDS reads
        __syncthreads();
        load_lds<N, 16>(v, lds, lane);
        twiddle_apply<N, 16, false>(v, t);
        butterflies<16, false>(v);
        if (Ns * 16 >= N) break;  // results stay in registers: radix-16 edge layout
        __syncthreads();
        store_lds<N, 16>(v, lds, lane, Ns);
        Ns *= 16;
    }
}

// `active` (wave-uniform where it matters): lanes with active == false only
// keep the workgroup barriers company -- used when a transform smaller than the
// workgroup (the folded N/D-point inverse) runs on its first N/16 lanes.
template <int N>
__device__ __forceinline__ void backward(float2 *v, float2 *lds, const float2 *tw, int lane,
                                         bool active = true) {
    constexpr int R0 = first_radix(N);
    if (active) butterflies<16, true>(v);  // first radix-16 pass, Ns = 1
    __syncthreads();                       // other lanes may still be reading lds (forward's last pass)
    if (active) store_lds<N, 16>(v, lds, lane, 1);
    int Ns = 16;
    while (Ns * R0 < N) {
        TwRegs<16> t;
        if (active) twiddle_load<N, 16>(t, tw, lane, Ns);
        __syncthreads();
        if (active) {
            load_lds<N, 16>(v, lds, lane);
            twiddle_apply<N, 16, true>(v, t);
            butterflies<16, true>(v);
        }
        __syncthreads();
        if (active) store_lds<N, 16>(v, lds, lane, Ns);
        Ns *= 16;
    }
    TwRegs<R0> t0;
    if (active) twiddle_load<N, R0>(t0, tw, lane, Ns);  // Ns = N / R0
    __syncthreads();
    if (active) {
        load_lds<N, R0>(v, lds, lane);
        twiddle_apply<N, R0, true>(v, t0);
        butterflies<R0, true>(v);
    }
}

}  // namespace f16
}  // namespace hz

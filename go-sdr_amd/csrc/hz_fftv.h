// hz_fftv.h -- workgroup FFT core, third generation, N = 256 .. 8192.
//
// Sixteen points per lane, N/16 lanes per transform,
// Stockham passes of radix 16 (4 x 4) with one smaller first (forward) / last (backward)
// pass, LDS padded one element per 16:
//
//   N:     256     512       1024      2048      4096        8192
//   plan:  16,16   2,16,16   4,16,16   8,16,16   16,16,16    2,16,16,16     (forward)
//   lanes: 16      32        64        128       256         512
//
// -- rewritten around what the gfx950 packed-float pipe can do WITHOUT moves.  A complex
// value is one native 2-vector (re, im) = one 64-bit register pair, so a complex add is ONE
// v_pk_add_f32.  v_pk_* sources may swap or broadcast their halves and negate BOTH halves
// for free (op_sel / neg_lo+neg_hi), but a per-half sign costs the compiler a negate plus a
// v_pk_mov (the second-generation core spent 230 of its 880 vector instructions per lane on
// such moves, measured on the north-star kernel).  So no operation here needs one:
//
//   * multiply by -+i:  swap(u) * J,  J = (1,-1) / (-1,1) a constant register pair; every use
//     is "a +- that", which contracts to ONE v_pk_fma_f32 (swap folded into op_sel);
//   * multiply by a constant h(1-+i), h(-1-+i) (W8, W16):  h * (x + rot(x)), h * (rot(x) - x),
//     the scalar h folding into the fma of the consuming add;
//   * multiply by a general twiddle w:  x.re * w + x.im * (i w) -- v_pk_mul + v_pk_fma with
//     broadcasts only -- with BOTH w and i*w = (-w.im, w.re) stored in the table (16-byte
//     entries), or as two constants for the butterflies' own W16 factors;
//   * inverse transforms use conjugate tables / constants (template parameter), not a
//     conjugation of the data.
//
// Twiddle tables are per pass and row-major in the butterfly position k, so a lane's
// twiddles are a few loads at immediate offsets from ONE address.  A radix-16 pass needs
// w^(n k), n = c + 4d < 16: its row holds SIX entries, w^k, w^2k, w^3k and w^4k, w^8k, w^12k,
// applied in two steps around the butterfly's first stage -- inputs *= (w^4k)^d before the
// DFT4 over d, its outputs *= w^(c k) after (a factor common to a DFT's inputs commutes
// with it) -- 24 products of 2 instructions instead of 15 products + 9 more to derive the
// missing twiddles (the older core), with 96 bytes per lane and pass from L2.  (Full
// 15-entry rows cost 240 bytes per lane and pass: measured, the L1/L2 traffic then binds.)
// Where a pass's table is small (the polyphase analysis: 16 rows) the caller keeps full
// rows in LDS instead (twiddle_rows).  Passes of radix R < 16 (the last backward pass of
// N = 512, 1024, 2048, 8192) take R-1 plain 8-byte entries per butterfly.
// LDS addresses likewise: every load / store of a pass is pad(lane-dependent base) plus a
// compile-time constant, i.e. one address register and immediate offsets.
//
// Stockham indexing, pass (R, Ns): butterfly j = b*TPT + lane, k = j mod Ns,
//   in[r]  = src[j + r*N/R] * w^(r*k)
//   dst[(j-k)*R + k + r*Ns] = DFT_R(in)[r]
//
// float32 butterflies with FMA contraction (no reference bit pattern exists for an FFT:
// fft/fft.go:45-59 is an interface; the tests hold an error bound).
#pragma once
#include <hip/hip_runtime.h>

namespace hz {
namespace fv {

typedef float cf __attribute__((ext_vector_type(2)));   // (re, im)
typedef float cf4 __attribute__((ext_vector_type(4)));  // twiddle entry (w.re, w.im, -w.im, w.re) = (w, i*w)

#define HZ_FV_FAST _Pragma("clang fp contract(fast)")

// device pointers to a plan's twiddle tables (hz_fft_api.h: get_fv_tables, get_fv_poly_tables)
struct FvTabs {
    const cf4 *fwd, *bwd;
};
struct PolyTabs {
    const cf4 *p2, *p3;
};

__host__ __device__ constexpr int log2i(int n) { int l = 0; while (n > 1) { n >>= 1; l++; } return l; }
__host__ __device__ constexpr int first_radix(int n) { return (log2i(n) & 3) == 0 ? 16 : 1 << (log2i(n) & 3); }
__host__ __device__ constexpr int tpt(int n) { return n / 16; }
__host__ __device__ constexpr int block(int n) { return tpt(n) < 64 ? 64 : tpt(n); }
__host__ __device__ constexpr int xpb(int n) { return block(n) / tpt(n); }
__host__ __device__ constexpr int lds_elems(int n) { return n + n / 16; }
__host__ __device__ constexpr bool ok(int n) { return n >= 256 && n <= 8192 && (n & (n - 1)) == 0; }
__host__ __device__ constexpr int pad(int i) { return i + (i >> 4); }

// ---- twiddle table geometry (in units of cf4 = 16 bytes) -----------------------------------
// forward plan: passes (R0, Ns 1), (16, R0), (16, 16 R0), ...; a 6-entry row per k < Ns for
// every pass with Ns > 1
constexpr int kRow16 = 6;
__host__ __device__ constexpr int fwd_tab_off(int n, int ns) {
    int off = 0;
    for (int s = first_radix(n); s < ns; s *= 16) off += s * kRow16;
    return off;
}
__host__ __device__ constexpr int fwd_tab_len(int n) { return fwd_tab_off(n, n); }
// backward plan: passes (16, 1), (16, 16), (16, 256), ... while Ns*R0 < N, then (R0, N/R0):
// 6-entry rows again for radix 16; for R0 < 16 the last table is N/R0 rows of R0-1 plain
// 8-byte twiddles (row k = butterfly position)
__host__ __device__ constexpr int bwd_tab_off(int n, int ns) {
    int off = 0;
    for (int s = 16; s < ns && s * first_radix(n) < n; s *= 16) off += s * kRow16;
    return off;
}
__host__ __device__ constexpr int bwd_last_len(int n) {
    return first_radix(n) == 16 ? (n / 16) * kRow16 : ((n / first_radix(n)) * (first_radix(n) - 1) + 1) / 2;
}
__host__ __device__ constexpr int bwd_tab_len(int n) { return bwd_tab_off(n, n) + bwd_last_len(n); }

// ---- complex helpers ------------------------------------------------------------------------

__device__ __forceinline__ cf swp(cf a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ cf bre(cf a) { return __builtin_shufflevector(a, a, 0, 0); }
__device__ __forceinline__ cf bim(cf a) { return __builtin_shufflevector(a, a, 1, 1); }
__device__ __forceinline__ cf lo(cf4 t) { return __builtin_shufflevector(t, t, 0, 1); }
__device__ __forceinline__ cf hi(cf4 t) { return __builtin_shufflevector(t, t, 2, 3); }
__device__ __forceinline__ cf from2(float2 a) { return cf{a.x, a.y}; }
__device__ __forceinline__ float2 to2(cf a) { return make_float2(a.x, a.y); }

// a * (-i) (forward) or a * (+i) (inverse), as a product with a constant pair: meant to be
// consumed by an add, which contracts it into one fma
template <bool INV> __device__ __forceinline__ cf rot(cf a) {
    HZ_FV_FAST
    const cf J = INV ? cf{-1.f, 1.f} : cf{1.f, -1.f};
    return swp(a) * J;
}
// a * w given the table entry (w, i w)
__device__ __forceinline__ cf cmul4(cf a, cf4 t) {
    HZ_FV_FAST
    return bre(a) * lo(t) + bim(a) * hi(t);
}
// a * w, w a compile-time constant (both pairs become literals / scalar registers)
__device__ __forceinline__ cf cmulc(cf a, float wr, float wi) {
    HZ_FV_FAST
    const cf w = {wr, wi}, iw = {-wi, wr};
    return bre(a) * w + bim(a) * iw;
}
// a * w for a run-time w held as one pair (three instructions: no i*w at hand)
__device__ __forceinline__ cf cmul(cf a, cf w) {
    HZ_FV_FAST
    const cf J = {-1.f, 1.f};
    return (swp(a) * bim(w)) * J + a * bre(w);
}
// acc += a * w with two accumulators and no per-half sign: re-part products into `p`,
// im-part products into `q`; the caller finishes with cacc_finish.  Two fma per product.
__device__ __forceinline__ void cacc(cf &p, cf &q, cf a, cf w) {
    HZ_FV_FAST
    p = bre(a) * w + p;
    q = bim(a) * swp(w) + q;
}
__device__ __forceinline__ cf cacc_finish(cf p, cf q) {
    HZ_FV_FAST
    const cf J = {-1.f, 1.f};
    return q * J + p;
}

// ---- register DFTs, natural order in and out ---------------------------------------------------

template <bool INV> __device__ __forceinline__ void dft2(cf &a, cf &b) {
    HZ_FV_FAST
    const cf t = a;
    a = t + b;
    b = t - b;
}

template <bool INV> __device__ __forceinline__ void dft4(cf &x0, cf &x1, cf &x2, cf &x3) {
    HZ_FV_FAST
    const cf a0 = x0 + x2, a1 = x0 - x2, a2 = x1 + x3, u = rot<INV>(x1 - x3);
    x0 = a0 + a2;
    x1 = a1 + u;
    x2 = a0 - a2;
    x3 = a1 - u;
}

template <bool INV> __device__ __forceinline__ void dft8(cf *v) {
    HZ_FV_FAST
    // n = c + 2d, K = b + 4a: two DFT4 over d, twiddle W8^(c b), radix 2 over c
    cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4<INV>(e0, e1, e2, e3);
    dft4<INV>(o0, o1, o2, o3);
    const float h = 0.70710678118654752440f;
    // W8^1 = h(1 -+ i), W8^2 = -+i, W8^3 = h(-1 -+ i)
    const cf t1 = h * (o1 + rot<INV>(o1));
    const cf t2 = rot<INV>(o2);
    const cf t3 = h * (rot<INV>(o3) - o3);
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + t1; v[5] = e1 - t1;
    v[2] = e2 + t2; v[6] = e2 - t2;
    v[3] = e3 + t3; v[7] = e3 - t3;
}

template <bool INV> __device__ __forceinline__ void dft16(cf *v) {
    HZ_FV_FAST
    // n = c + 4d, K = b + 4a:  X[b+4a] = sum_c W4^(c a) W16^(c b) sum_d x[c+4d] W4^(d b)
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    constexpr float sg = INV ? 1.f : -1.f;  // sign of the imaginary part of W16^m, 0 < m < 8
    // stage A: DFT4 over d for each c (v[c + 4d] -> t[c][b] stored at v[c + 4b])
#pragma unroll
    for (int c = 0; c < 4; c++) dft4<INV>(v[c], v[c + 4], v[c + 8], v[c + 12]);
    // t[c][b] *= W16^(c b):  W^1 = (c1, sg s1), W^2 = h(1, sg), W^3 = (s1, sg c1), W^4 = (0, sg),
    // W^6 = h(-1, sg), W^9 = -W^1
    v[1 + 4 * 1] = cmulc(v[1 + 4 * 1], c1, sg * s1);
    v[1 + 4 * 2] = h * (v[1 + 4 * 2] + rot<INV>(v[1 + 4 * 2]));
    v[1 + 4 * 3] = cmulc(v[1 + 4 * 3], s1, sg * c1);
    v[2 + 4 * 1] = h * (v[2 + 4 * 1] + rot<INV>(v[2 + 4 * 1]));
    v[2 + 4 * 2] = rot<INV>(v[2 + 4 * 2]);
    v[2 + 4 * 3] = h * (rot<INV>(v[2 + 4 * 3]) - v[2 + 4 * 3]);
    v[3 + 4 * 1] = cmulc(v[3 + 4 * 1], s1, sg * c1);
    v[3 + 4 * 2] = h * (rot<INV>(v[3 + 4 * 2]) - v[3 + 4 * 2]);
    v[3 + 4 * 3] = cmulc(v[3 + 4 * 3], -c1, -sg * s1);
    // stage B: DFT4 over c for each b: inputs v[c + 4b], outputs X[b + 4a] -> v[b + 4a]
    cf out[16];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        cf y0 = v[0 + 4 * b], y1 = v[1 + 4 * b], y2 = v[2 + 4 * b], y3 = v[3 + 4 * b];
        dft4<INV>(y0, y1, y2, y3);
        out[b] = y0;
        out[b + 4] = y1;
        out[b + 8] = y2;
        out[b + 12] = y3;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = out[i];
}

template <int R, bool INV> __device__ __forceinline__ void dft(cf *v) {
    if constexpr (R == 2) dft2<INV>(v[0], v[1]);
    else if constexpr (R == 4) dft4<INV>(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 8) dft8<INV>(v);
    else dft16<INV>(v);
}

template <int R, bool INV> __device__ __forceinline__ void butterflies(cf *v) {
#pragma unroll
    for (int b = 0; b < 16 / R; b++) dft<R, INV>(v + b * R);
}

// ---- one pass, in pieces -----------------------------------------------------------------------

// register slot q of a radix-R pass <-> element index at the pass input
template <int N, int R> __host__ __device__ constexpr int edge_off(int q) { return (q / R) * tpt(N) + (q % R) * (N / R); }
template <int N, int R> __device__ __forceinline__ int edge_index(int q, int lane) { return edge_off<N, R>(q) + lane; }

// edge_off is a multiple of 16 (N >= 256), so pad(lane + off) = pad(lane) + pad(off): one
// address register, immediate offsets
template <int N, int R> __device__ __forceinline__ void load_lds(cf *v, const cf *lds, int lane) {
    const cf *p = lds + pad(lane);
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = p[pad(edge_off<N, R>(q))];
}
template <int N, int R> __device__ __forceinline__ void store_edge_lds(const cf *v, cf *lds, int lane) {
    cf *p = lds + pad(lane);
#pragma unroll
    for (int q = 0; q < 16; q++) p[pad(edge_off<N, R>(q))] = v[q];
}

// Outputs of pass (R, NS).  With e = b*TPT*R + r*NS the low four bits of the lane-dependent
// part (j0 of butterfly 0) and of e never carry into each other, so again
// pad(j0 + e) = pad(j0) + pad(e).
template <int N, int R, int NS> __device__ __forceinline__ void store_lds(const cf *v, cf *lds, int lane) {
    static_assert(NS * R <= N && (NS == 1 || R == 16 || NS % 16 == 0), "store of a pass that is not the last");
    const int k = lane & (NS - 1);
    cf *p = lds + pad((lane - k) * R + k);
#pragma unroll
    for (int b = 0; b < 16 / R; b++)
#pragma unroll
        for (int r = 0; r < R; r++) p[pad(b * tpt(N) * R + r * NS)] = v[b * R + r];
}

// a radix-16 pass's inter-pass twiddles, first half: inputs v[c + 4d] *= (w^4k)^d
// (row = the lane's six entries w^k, w^2k, w^3k, w^4k, w^8k, w^12k)
template <class Row> __device__ __forceinline__ void twiddle16_pre(cf *v, Row row) {
    const cf4 t4 = row[3], t8 = row[4], t12 = row[5];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        v[c + 4] = cmul4(v[c + 4], t4);
        v[c + 8] = cmul4(v[c + 8], t8);
        v[c + 12] = cmul4(v[c + 12], t12);
    }
}

// radix-16 butterfly of a pass with Ns > 1: dft16 with the second half of the inter-pass
// twiddles, t[c][b] *= w^(c k), between its two stages
template <bool INV, class Row> __device__ __forceinline__ void dft16_tw(cf *v, Row row) {
    HZ_FV_FAST
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    constexpr float sg = INV ? 1.f : -1.f;
    twiddle16_pre(v, row);
#pragma unroll
    for (int c = 0; c < 4; c++) dft4<INV>(v[c], v[c + 4], v[c + 8], v[c + 12]);
    v[1 + 4 * 1] = cmulc(v[1 + 4 * 1], c1, sg * s1);
    v[1 + 4 * 2] = h * (v[1 + 4 * 2] + rot<INV>(v[1 + 4 * 2]));
    v[1 + 4 * 3] = cmulc(v[1 + 4 * 3], s1, sg * c1);
    v[2 + 4 * 1] = h * (v[2 + 4 * 1] + rot<INV>(v[2 + 4 * 1]));
    v[2 + 4 * 2] = rot<INV>(v[2 + 4 * 2]);
    v[2 + 4 * 3] = h * (rot<INV>(v[2 + 4 * 3]) - v[2 + 4 * 3]);
    v[3 + 4 * 1] = cmulc(v[3 + 4 * 1], s1, sg * c1);
    v[3 + 4 * 2] = h * (rot<INV>(v[3 + 4 * 2]) - v[3 + 4 * 2]);
    v[3 + 4 * 3] = cmulc(v[3 + 4 * 3], -c1, -sg * s1);
    const cf4 t1 = row[0], t2 = row[1], t3 = row[2];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        v[1 + 4 * b] = cmul4(v[1 + 4 * b], t1);
        v[2 + 4 * b] = cmul4(v[2 + 4 * b], t2);
        v[3 + 4 * b] = cmul4(v[3 + 4 * b], t3);
    }
    cf out[16];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        cf y0 = v[0 + 4 * b], y1 = v[1 + 4 * b], y2 = v[2 + 4 * b], y3 = v[3 + 4 * b];
        dft4<INV>(y0, y1, y2, y3);
        out[b] = y0;
        out[b + 4] = y1;
        out[b + 8] = y2;
        out[b + 12] = y3;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = out[i];
}

// one radix-16 pass at Ns = NS > 1 on registers: twiddles + butterfly (tab = the pass's table)
template <int N, int NS, bool INV>
__device__ __forceinline__ void pass16(cf *v, const cf4 *tab, int lane) {
    dft16_tw<INV>(v, tab + (lane & (NS - 1)) * kRow16);
}

// a pass of radix R < 16 at Ns = NS > 1: R-1 plain twiddles per butterfly, rows of a `cf` table
template <int N, int R, int NS, bool INV>
__device__ __forceinline__ void pass_small(cf *v, const cf *tab, int lane) {
#pragma unroll
    for (int b = 0; b < 16 / R; b++) {
        const cf *row = tab + ((b * tpt(N) + lane) & (NS - 1)) * (R - 1);
#pragma unroll
        for (int r = 1; r < R; r++) v[b * R + r] = cmul(v[b * R + r], row[r - 1]);
        dft<R, INV>(v + b * R);
    }
}

// inputs *= w^(r k) from FULL rows (R entries per k; r = 0 unused) -- for tables the caller
// keeps in LDS: R-1 16-byte reads at immediate offsets, two instructions per product
template <int N, int R, int NS, int STRIDE, class TabPtr>
__device__ __forceinline__ void twiddle_rows(cf *v, TabPtr tab, int lane) {
#pragma unroll
    for (int b = 0; b < 16 / R; b++) {
        const int k = (b * tpt(N) + lane) & (NS - 1);
        auto row = tab + k * STRIDE;
#pragma unroll
        for (int r = 1; r < R; r++) v[b * R + r] = cmul4(v[b * R + r], row[r]);
    }
}

// Between a pass's LDS stores and the next pass's loads: a workgroup barrier -- or, WAVE, nothing but the compiler's
// order: for a caller whose transforms each live inside ONE wave (64 lanes or fewer, the lanes of a transform adjacent
// in the wave, its LDS region its own).  A wave's LDS instructions execute in the order it issues them, so such a
// workgroup may hold many waves that never meet (conv_blocks_shared_kernel: sixteen, with one copy of the tables).
// (Not a property of N: fft2_cols_kernel spreads a 16-lane transform over four waves.)
template <bool WAVE> __device__ __forceinline__ void sync() {
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// ---- whole transforms on registers -------------------------------------------------------------
// v[16]: on entry the inputs in the edge layout of the first pass's radix, on exit the
// outputs in the edge layout of the last pass's radix.  `lds` = this transform's padded
// region (lds_elems(N)); `tab` = the plan's twiddle tables (make_fwd_table / make_bwd_table
// in hz_fft.hip).  Every lane of the workgroup calls.
// FROM_LDS: the caller loaded v from `lds` itself, so a barrier precedes the first store.

template <int N, int NS, bool WAVE = false> __device__ __forceinline__ void forward_from(cf *v, cf *lds, const cf4 *tab, int lane) {
    sync<WAVE>();
    load_lds<N, 16>(v, lds, lane);
    pass16<N, NS, false>(v, tab + fwd_tab_off(N, NS), lane);
    if constexpr (NS * 16 < N) {
        sync<WAVE>();
        store_lds<N, 16, NS>(v, lds, lane);
        forward_from<N, NS * 16, WAVE>(v, lds, tab, lane);
    }
}

template <int N, bool FROM_LDS = false, bool WAVE = false>
__device__ __forceinline__ void forward(cf *v, cf *lds, const cf4 *tab, int lane) {
    constexpr int R0 = first_radix(N);
    butterflies<R0, false>(v);  // Ns = 1: no twiddles
    if constexpr (FROM_LDS) sync<WAVE>();
    store_lds<N, R0, 1>(v, lds, lane);
    forward_from<N, R0, WAVE>(v, lds, tab, lane);
}

template <int N, int NS, bool WAVE = false> __device__ __forceinline__ void backward_from(cf *v, cf *lds, const cf4 *tab, int lane, bool active) {
    constexpr int R0 = first_radix(N);
    sync<WAVE>();
    if constexpr (NS * R0 < N) {
        if (active) {
            load_lds<N, 16>(v, lds, lane);
            pass16<N, NS, true>(v, tab + bwd_tab_off(N, NS), lane);
        }
        sync<WAVE>();
        if (active) store_lds<N, 16, NS>(v, lds, lane);
        backward_from<N, NS * 16, WAVE>(v, lds, tab, lane, active);
    } else {
        if (active) {
            load_lds<N, R0>(v, lds, lane);
            if constexpr (R0 == 16) pass16<N, NS, true>(v, tab + bwd_tab_off(N, NS), lane);
            else pass_small<N, R0, NS, true>(v, (const cf *)(tab + bwd_tab_off(N, NS)), lane);
        }
    }
}

// Every twiddle of a backward transform, loaded up front: for kernels that are one latency
// chain per wave (the synthesis kernel), so that the table reads (L2) overlap the first pass
// instead of standing between two passes each.
template <int N> struct BwdTwiddles {
    static constexpr int R0 = first_radix(N);
    static constexpr int count16() {
        int c = 0;
        for (int s = 16; s * R0 < N; s *= 16) c++;
        return c + (R0 == 16 ? 1 : 0);
    }
    static constexpr int NP16 = count16(), NSMALL = R0 < 16 ? (R0 - 1) * (16 / R0) : 0;
    cf4 r16[NP16 > 0 ? NP16 : 1][kRow16];
    cf small[NSMALL > 0 ? NSMALL : 1];
    __device__ __forceinline__ void load(const cf4 *tab, int lane) {
        int p = 0, ns = 16;
#pragma unroll
        for (; p < NP16; p++, ns *= 16) {
            const cf4 *row = tab + bwd_tab_off(N, ns) + (lane & (ns - 1)) * kRow16;
#pragma unroll
            for (int e = 0; e < kRow16; e++) r16[p][e] = row[e];
        }
        if constexpr (R0 < 16) {
            const cf *t = (const cf *)(tab + bwd_tab_off(N, N / R0));
#pragma unroll
            for (int b = 0; b < 16 / R0; b++)
#pragma unroll
                for (int r = 1; r < R0; r++) small[b * (R0 - 1) + r - 1] = t[(b * tpt(N) + lane) * (R0 - 1) + r - 1];
        }
    }
};

template <int N, int NS, int P> __device__ __forceinline__ void backward_from_regs(cf *v, cf *lds, const BwdTwiddles<N> &tw, int lane) {
    constexpr int R0 = first_radix(N);
    __syncthreads();
    if constexpr (NS * R0 < N) {
        load_lds<N, 16>(v, lds, lane);
        dft16_tw<true>(v, tw.r16[P]);
        __syncthreads();
        store_lds<N, 16, NS>(v, lds, lane);
        backward_from_regs<N, NS * 16, P + 1>(v, lds, tw, lane);
    } else {
        load_lds<N, R0>(v, lds, lane);
        if constexpr (R0 == 16) {
            dft16_tw<true>(v, tw.r16[P]);
        } else {
#pragma unroll
            for (int b = 0; b < 16 / R0; b++) {
#pragma unroll
                for (int r = 1; r < R0; r++) v[b * R0 + r] = cmul(v[b * R0 + r], tw.small[b * (R0 - 1) + r - 1]);
                dft<R0, true>(v + b * R0);
            }
        }
    }
}

// backward with the twiddles already in registers (BwdTwiddles::load before the input loads)
template <int N> __device__ __forceinline__ void backward_regs(cf *v, cf *lds, const BwdTwiddles<N> &tw, int lane) {
    butterflies<16, true>(v);
    __syncthreads();
    store_lds<N, 16, 1>(v, lds, lane);
    backward_from_regs<N, 16, 0>(v, lds, tw, lane);
}

// `active` (wave-uniform where it matters): lanes with active == false only keep the
// workgroup barriers company (a transform smaller than the workgroup on its first lanes).
template <int N, bool WAVE = false>
__device__ __forceinline__ void backward(cf *v, cf *lds, const cf4 *tab, int lane, bool active = true) {
    if (active) butterflies<16, true>(v);  // first radix-16 pass, Ns = 1
    sync<WAVE>();                       // other lanes may still be reading lds (forward's last pass)
    if (active) store_lds<N, 16, 1>(v, lds, lane);
    backward_from<N, 16, WAVE>(v, lds, tab, lane, active);
}

}  // namespace fv
}  // namespace hz

// hz_firmm.h -- the FIR-decimate terminal of a chain over RAW 8-bit IQ as an int8 MFMA product.
//
// For a u8 / i8 source and a decimation D of 8, 16, 24, 32, 40, 48 or 64 the filter runs in the time
// domain on the matrix cores instead of through overlap-save transforms (hz_chain_dev.h):
//
//   * the late mixer (see fir_decimate_kernel16) already filters the CONVERTED samples with the
//     run's modulated taps h'[k] = h[k] exp(-i Omega k step) / scale and mixes afterwards.  The
//     converted samples of a byte source are integers: x = (b - 128) + 0.5 (1 + i) for u8, b for
//     i8.  The taps become 32-bit fixed point q[k] = round(h'[k] 2^S) (|q| <= 2^30: 2^-30 of the
//     largest tap, far below float32), split into four balanced base-256 digits.  Then
//         y[m] 2^S = sum_d 256^(3-d) sum_k digit_d(q[k]) xi8[D m - k]  +  0.5 (1 + i) sum_k q[k]
//     is FOUR int8 matrix products with int32 accumulation -- EXACT integer arithmetic -- one
//     float64 combination and one rounding to float32: the filter output is the correctly
//     rounded value for the quantised taps (measured 2.6e-8 relative L2 against the float64
//     direct form; the transform path has 1.7e-7).
//   * the product: a tile = 16 consecutive outputs x (re, im) = 32 rows of a Toeplitz matrix
//     A[(i, part), k] over the window's bytes k = (sample a, part), tap D i + w0 - a; the other
//     operand is the raw byte stream itself (u8: XOR 0x80), column n = the window of tile n,
//     256 bytes further on (D = 8).  v_mfma_i32_32x32x32_i8: 32 rows x 32 tiles x 32 bytes per
//     instruction.  Every A fragment is an entry of ONE small table F[digit][E][part] (16 bytes:
//     eight taps x (coefficient of re, of im)), E = (D/8) i - h - 2 s: 21 KB per clock run,
//     read coalesced (a wave's 64 lanes read ~544 consecutive bytes) and L1-resident.
//   * a workgroup = two waves over the same 2048 outputs at D = 8, 1024 at D = 16 (34 KB of input in LDS,
//     16-byte pieces XOR-swizzled by the tile index so that a fragment read is conflict-free
//     without padding): wave 0 multiplies by digits 0-1, wave 1 by digits 2-3 (128 accumulator
//     registers each, two waves per SIMD); they exchange halves through LDS and each finishes
//     1024 outputs: float64 combination, the elementwise program as a phase recurrence (the
//     clock is exactly linear inside the run), store.
//   * the call's outputs are cut into chunks of 2048 (1024) on ONE grid; a chunk belongs to the clock run
//     that holds most of it.  Outputs whose window crosses a boundary of the clock's runs, the start
//     of the stream or a run without a table -- and what a boundary chunk's run does not hold -- are
//     FIX-UP TASKS (16 outputs in reference order: direct form, float64 accumulation); the history
//     for the next call (the last `off` samples after the elementwise program, the format the
//     transform kernels share, and the same samples as raw bytes) are HISTORY TASKS.  Chunk
//     workgroup b takes task b, b + grid, ... while its first input loads are in flight: no second
//     launch, no extra workgroups (1024 chunk workgroups fill the chip exactly at 2^21 outputs).
//     When the clock run continues across a call boundary the first windows reach back into the raw
//     history: a steady-state call has no fix-up at all.
//
// Measured (MI355X, 2^24 u8 samples, 1024 taps, D = 8; tools/mfma_fir.hip, tools/firmm_probe.py,
// profiles/r02_*): 43-46 us per call inside one clock run, 1.5 Pop/s = 0.30 of the dense int8 peak,
// the matrix pipe busy 94 % of the matrix loop and 42 % of the kernel (the chip holds ~1.55 GHz
// under this load); HBM traffic 1.07x the algorithmic bytes.  DESIGN.md section 4 has the history.
#pragma once
// A HAZARD THE COMPILER DOES NOT KNOW (gfx950, measured: tools/pk_glitch.hip, profiles/r04_pk_glitch.txt): a
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 whose op_sel takes the HIGH register of the src1 pair for the LOW result
// while src0 is taken straight (op_sel:[0,1]: what the compiler emits for (a, b) x (d, c)) reads that operand as ZERO in
// lanes 48-63 when the OTHER wave of the SIMD issues an MFMA at the wrong cycle -- 2e-3 of the executions while the
// partner wakes up into a first MFMA, 1e-7 beside a steady stream, never beside anything else (LDS, float64, float32
// vector work, an idle partner), never for the unswizzled forms, op_sel on src0 or src2, scalar float32 or float64
// instructions.  That was the matrix FIR's non-repeatable pass (round 3's "known issue"): the mixer's first step
// factor lost its term -s0 w.y in a wave's first pass (tools/mm2_glitch.hip follows it from the outputs down to the
// instruction).  The epilogues of these kernels run beside the partner wave's matrix loop by design.  src0 and src1
// commute, so the build rewrites every such instruction of the linked library with its first two sources exchanged
// (tools/fix_pk_opsel.py, run by the Makefile behind the link: same arithmetic, same size, same schedule; compiling
// the kernels without packed float32 altogether cost 2 us of 38), and tests/test_capi_cpu.py holds the library to it.
#include "hz_chain_dev.h"
#include "hz_firmm_plan.h"

namespace hz {
namespace mm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));


static_assert(kMaxRuns == kNcoMaxSegs, "one run per clock-table entry");
// How the two waves of a workgroup share the chunk's 4 digit planes x 4 column blocks:
// HZ_MM_SPLIT_BLOCKS = 0: by digits (wave 0: planes 0-1, wave 1: planes 2-3, all four blocks; the halves
// meet through LDS); 1: by blocks (each wave all four planes of two blocks: no exchange, twice the
// table loads, half the LDS reads).
#ifndef HZ_MM_SPLIT_BLOCKS
#define HZ_MM_SPLIT_BLOCKS 0
#endif
constexpr bool kSplitBlocks = HZ_MM_SPLIT_BLOCKS != 0;
constexpr int kND = kSplitBlocks ? 4 : 2;       // digit planes per wave

// LDS image of a chunk: 16-byte piece p of tile t at TB t + 16 (p ^ (t & 15)).  A fragment read
// (same piece, 32 consecutive tiles; ds_read_b128 serves 16 lanes per cycle) then touches 16
// different 16-byte bank groups in every lane group.
template <int D> __device__ __forceinline__ int sw(int tile, int piece) {
    return tile_bytes(D) * tile + 16 * (piece ^ (tile & 15));
}

// the index of the last entry <= key in an ascending table of n <= 64 ints held in the kernel
// arguments: one vector load + a ballot (a scalar scan waits for a dependent load per entry)
__device__ __forceinline__ int find_le(const int *tab, int n, int key) {
    const int l = (int)(threadIdx.x & 63u);
    const bool valid = l < n;
    const int v = tab[valid ? l : 0];
    return __popcll(__ballot(valid && v <= key)) - 1;
}

// One post-elementwise sample in reference order: position p of the buffer (p < 0: history).
// LATE: the Shift's Sincos flavour (ew_apply_n: 1 = sincos_late, 2 = sincos_turns32).  The persistent-pass kernel's
// history (hz_firmm2.h) is formed with 2, the flavour its fix-up tasks stage their windows with: a task that reads a
// sample from the history then gets the bits it would have computed from the previous buffer's bytes itself -- which is
// what a call over several buffers does (round 6: a batch equals single calls bit for bit; with 1 the first forty
// outputs of a buffer that starts less than a window behind a clock boundary differed in the last place).
template <int FMT, int LATE = 1>
__device__ __forceinline__ float2 ordered_sample(const void *in, const EwProgram &P, int64_t p, const float2 *hist,
                                                 unsigned off, NcoWin w) {
    using R = typename Raw<FMT>::t;
    if (p < 0) return hist ? hist[p + (int64_t)off] : make_float2(0.f, 0.f);
    float2 v[1] = {Raw<FMT>::cvt(((const R *)in)[p])};
    ew_apply_n<1, LATE>(P, v, (uint64_t)p, w);
    return v[0];
}

// the clock runs that hold samples [p_lo, p_hi] (clamped to the buffer), found by the wave together
// (nco_window_ballot; the inline table only -- the matrix form does not run with the device table).
// A task that scanned the whole table per sample (nco_window_all) waited for ~25 dependent scalar
// loads per sample on a call in which the clock wraps: +30 us on that call.
__device__ __forceinline__ NcoWin task_window(const EwProgram &P, int64_t p_lo, int64_t p_hi) {
    if (P.segs.n <= 1) return NcoWin{0, 0};
    const NcoWin w = nco_window_ballot(P.segs, (uint64_t)(p_lo < 0 ? 0 : p_lo), (uint64_t)(p_hi < 0 ? 0 : p_hi), 64);
    return NcoWin{__builtin_amdgcn_readfirstlane(w.lo), __builtin_amdgcn_readfirstlane(w.hi)};
}

// EXP: ablation switches for tools/mfma_fir.hip (0 in the library): 1 = no input loads, 2 = no
// matrix loop, 4 = no elementwise program, 8 = no stores (results are wrong with any of these set);
// 64 = s_memtime stamps at the phase boundaries of every workgroup (+ 256: s_memrealtime, 10 ns
// ticks: wall time instead of cycles), 128 = plain instead of
// non-temporal input loads (input phase 16.8 k -> 18.0 k cycles).
// (Switching the loop's operand loads off does not time the MFMAs alone: with undefined operands the
// compiler deletes the loop.  tools/mfma_rate.hip has the loop's MFMA pattern with and without loads.)
template <int FMT, int D, int EXP = 0>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void fir_mm_kernel(
    const void *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ hist,
    float2 *__restrict__ new_hist, const uint8_t *__restrict__ rhist, uint8_t *__restrict__ new_rhist,
    const float2 *__restrict__ taps, size_t n_in, Geom G, EwProgram P, Runs R, Fix F,
    unsigned long long *stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t mm_lds[];
    static_assert(D % 8 == 0, "windows start on 16-byte pieces");
    constexpr int TB = tile_bytes(D), PPT = pieces_per_tile(D);
    constexpr int NB = blocks_for(D), kNBW = kSplitBlocks ? NB / 2 : NB, BW = NB >= 2 ? NB / 2 : 1;  // blocks: per workgroup, per wave in the loop, per wave in the epilogue
    constexpr int kChunkOut = chunk_out(NB);
    static_assert(!kSplitBlocks || NB == 4, "the block-split variant is the D = 8 experiment");
    const int tid = threadIdx.x;
    const int wb = blockIdx.x;
    // (EXP & 64, tools/mfma_fir.hip: s_memtime at the phase boundaries of every workgroup)
    auto stamp = [&](int k) {
        if constexpr ((EXP & 64) != 0) {
            if (tid == 0)
                stamps[(size_t)wb * 8 + k] = (EXP & 256) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
        }
    };
    stamp(0);
    // ---- the small tasks: outputs in reference order (fix-up) and the next call's history --------
    // They have no workgroups of their own: as workgroups at the END of the grid each was a ~8 us
    // latency chain (load, Sincos, dot product, store) behind the chunk workgroups, and ahead of them
    // they push chunk workgroups out of the single round the chip holds (1024 at 2^21 outputs).
    // Chunk workgroup b takes task b, b + grid, ...
    const int n_hist_tasks = new_hist ? (int)((G.off + kThreads - 1) / kThreads) : 0;
    const int n_tasks = F.n_wg + n_hist_tasks;
    // A few tasks (a clock boundary in the call: ~40) hide best under the input burst of the workgroups
    // that take them; hundreds (the call in which the clock wraps) are cheaper behind the stores, where
    // their own loads do not queue behind the burst (measured: 48 / 53 us and 61 / 57 us per call).
    const bool tasks_late = n_tasks > 200;
    auto small_task = [&](int task) {
        using RW = typename Raw<FMT>::t;
        if (task < F.n_wg) {
            // Up to 16 outputs: their window's samples in reference order (load, convert, elementwise
            // program: three Sincos chains side by side per lane and trip) and the taps go to LDS (the
            // chunk image is dead by now), then 8 lanes per output run the direct form in float64.
            const int k = find_le(F.wg_first, F.n, task);
            const uint32_t m0 = F.m_a[k] + (uint32_t)(task - F.wg_first[k]) * kFixOut;
            const int cnt = (int)min((uint32_t)kFixOut, F.m_b[k] - m0);
            float2 *xs = reinterpret_cast<float2 *>(mm_lds);
            float2 *tl = xs + (G.ntaps + D * (kFixOut - 1));
            const int64_t p_lo = (int64_t)D * m0 - (G.ntaps - 1);
            const int n_s = G.ntaps + D * (cnt - 1);
            for (int idx = tid; idx < G.ntaps; idx += kThreads) tl[idx] = taps[idx];
            const NcoWin tw = task_window(P, p_lo, p_lo + n_s - 1);
            constexpr int W = 3;
#pragma unroll 1
            for (int i0 = tid; i0 < n_s; i0 += W * kThreads) {
                float2 v[W];
#pragma unroll
                for (int u = 0; u < W; u++) {
                    const int64_t pu = p_lo + i0 + u * kThreads;
                    const bool ok = i0 + u * kThreads < n_s && pu >= 0;
                    v[u] = Raw<FMT>::cvt(ok ? ((const RW *)in)[pu] : RW{});
                }
                // (positions before the buffer wrap around in uint64 and come back: their values are discarded)
                ew_apply_n<W, true>(P, v, (uint64_t)(p_lo + i0), tw, (uint64_t)kThreads);
#pragma unroll
                for (int u = 0; u < W; u++) {
                    const int idx = i0 + u * kThreads;
                    const int64_t pu = p_lo + idx;
                    if (idx < n_s)
                        xs[idx] = pu >= 0 ? v[u] : ((hist && pu + (int64_t)G.off >= 0) ? hist[pu + (int64_t)G.off] : make_float2(0.f, 0.f));
                }
            }
            __syncthreads();
            const int o = tid >> 3, sl = tid & 7;  // output, tap slice (k = sl mod 8)
            double ar = 0.0, ai = 0.0;
            if (o < cnt) {
                const float2 *xo = xs + D * o + (G.ntaps - 1);
#pragma unroll 4
                for (int kk = sl; kk < G.ntaps; kk += 8) {
                    const float2 hk = tl[kk], x = xo[-kk];
                    const double xr = x.x, xi = x.y, hr = hk.x, hi = hk.y;
                    ar = __fma_rn(xr, hr, ar);
                    ar = __fma_rn(-xi, hi, ar);
                    ai = __fma_rn(xr, hi, ai);
                    ai = __fma_rn(xi, hr, ai);
                }
            }
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) {
                ar += __shfl_xor(ar, d);
                ai += __shfl_xor(ai, d);
            }
            if (o < cnt && sl == 0) out[m0 + o] = make_float2((float)ar, (float)ai);
        } else {
            // 128 samples of the history: the last `off` samples after the elementwise program, and the
            // same samples as raw bytes (the next call's windows reach back into them when the clock
            // run continues across the call boundary, Runs::cont)
            const unsigned idx = (unsigned)(task - F.n_wg) * kThreads + tid;
            const int64_t h_lo = (int64_t)n_in - (int64_t)G.off + (int64_t)(task - F.n_wg) * kThreads;
            const NcoWin tw = task_window(P, h_lo, h_lo + kThreads - 1);
            if (idx < G.off) {
                const int64_t p = (int64_t)n_in - (int64_t)G.off + idx;
                new_hist[idx] = ordered_sample<FMT>(in, P, p, hist, G.off, tw);
                reinterpret_cast<RW *>(new_rhist)[idx] = p >= 0 ? ((const RW *)in)[p] : reinterpret_cast<const RW *>(rhist)[p + (int64_t)G.off];
            }
        }
        __syncthreads();  // (LDS: the reduction's slots are the chunk image's first bytes)
    };

    // ---- the chunk: outputs [2048 b, 2048 b + 2048) of the call, filtered with the taps of ONE clock
    // run (the host gave the chunk to the run that holds most of it); the outputs [v_lo, v_hi) of it
    // lie wholly in that run and are stored, the others belong to fix-up tasks.  (Chunks anchored at
    // the runs' starts instead made 1025 workgroups out of a 2^21-output call with one clock boundary
    // -- one more than the chip holds at once, a second round for a single workgroup: +10 us.)
    const int r = __builtin_amdgcn_readfirstlane(find_le(R.wg_first, R.n, wb));
    const uint32_t m_start = (uint32_t)wb * kChunkOut;
    const uint32_t v_lo = max(R.m_lo[r], m_start), v_hi = min(R.m_hi[r], m_start + (uint32_t)kChunkOut);
    const bool active = v_lo < v_hi;  // uniform
    const int wave = tid >> 6, l = tid & 63, n = l & 31, h = l >> 5, i = n >> 1, part = n & 1;
    const int64_t n_bytes = 2 * (int64_t)n_in;
    if (active) {
        // the chunk's input bytes: from the window start of its first tile
        const int64_t p0 = 2 * ((int64_t)D * m_start - G.w0);
        const int pieces = (int)(chunk_bytes(D, G.ks) / 16);
        const uint8_t *src = (const uint8_t *)in;
        constexpr int U = 17;  // loads in flight per lane and trip: the whole chunk at D = 8, 1024 taps (one HBM round trip)
        auto issue = [&](v4i(&x)[U], int q0) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q = q0 + u * kThreads;
                const int64_t p = p0 + (int64_t)q * 16;
                x[u] = v4i{0, 0, 0, 0};
                if (q < pieces) {
                    if ((EXP & 1) == 0 && p >= 0 && p + 16 <= n_bytes) {
                        if constexpr ((EXP & 128) != 0) x[u] = *reinterpret_cast<const v4i *>(src + p);
                        else x[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(src + p));
                    } else if (p < 0 && R.cont && p + 2 * (int64_t)G.off >= 0) {
                        // before the buffer: the previous call's last samples (2 off is a multiple of 16)
                        x[u] = *reinterpret_cast<const v4i *>(rhist + (p + 2 * (int64_t)G.off));
                    } else if (p + 16 > 0 && p < n_bytes) {  // straddles the buffer's end: byte by byte
                        union { v4i v; uint8_t b[16]; } t;
                        t.v = v4i{0, 0, 0, 0};
                        for (int e = 0; e < 16; e++)
                            if (p + e >= 0 && p + e < n_bytes) t.b[e] = src[p + e];
                        x[u] = t.v;
                    }
                }
            }
        };
        auto land = [&](v4i(&x)[U], int q0) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q = q0 + u * kThreads;
                if (q < pieces) {
                    if constexpr (FMT == HZSDR_FMT_U8) x[u] ^= (int)0x80808080;  // b - 128 as int8
                    *reinterpret_cast<v4i *>(mm_lds + sw<D>(q / PPT, q % PPT)) = x[u];
                }
            }
        };
        v4i x[U];
        issue(x, tid);
        if (!tasks_late)
            for (int task = wb; task < n_tasks; task += (int)gridDim.x) small_task(task);  // (uniform)
        land(x, tid);
#pragma unroll 1
        for (int q0 = tid + U * kThreads; q0 < pieces; q0 += U * kThreads) {
            issue(x, q0);
            land(x, q0);
        }
    }
    stamp(1);
    if (active) {
        __syncthreads();
        stamp(2);
        // A fragments: entry (D/8) i - h - 2 s + e0 of this wave's two digits, parts interleaved
        const v4i *ftab = (const v4i *)R.tab[r];
        const v4i *fa = ftab + (size_t)(kSplitBlocks ? 0 : 2 * wave) * 2 * G.ne + 2 * ((D / 8) * i - h + G.e0) + part;
        const int dstride = 2 * G.ne;
        v16i acc[kND][kNBW];
#pragma unroll
        for (int d = 0; d < kND; d++)
#pragma unroll
            for (int j = 0; j < kNBW; j++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[d][j][q] = 0;
        // Step s of the window reads piece 2 s + h of tile n (+ 32 j): with GS = PPT / 2 steps per group
        // the tile is n + g and the piece 2 j + h for step j of group g -- the swizzled address is
        // 16 ((2 j) ^ xh) past the group's base, xh = ((n + g) & 15) ^ h: two instructions per step.
        constexpr int GS = PPT / 2;
        auto load_b = [&](v4i(&b)[kNBW], int base, int xh, int j) {
            const uint8_t *bp = mm_lds + base + 16 * ((2 * j) ^ xh) + (kSplitBlocks ? wave * BW * 32 * TB : 0);
#pragma unroll
            for (int q = 0; q < kNBW; q++) b[q] = *reinterpret_cast<const v4i *>(bp + q * 32 * TB);
        };
        // (a running pointer per digit, stepped once per group: the loads inside a group take immediate
        // offsets instead of 64-bit address arithmetic per load)
        const v4i *pa[kND];
#pragma unroll
        for (int d = 0; d < kND; d++) pa[d] = fa + d * dstride;
        auto load_a = [&](v4i(&a)[kND], int js) {  // step js of the current group (may run into the next)
#pragma unroll
            for (int d = 0; d < kND; d++) a[d] = pa[d][-4 * js];
        };
        auto mma = [&](const v4i(&a)[kND], const v4i(&b)[kNBW]) {
#pragma unroll
            for (int d = 0; d < kND; d++)
#pragma unroll
                for (int q = 0; q < kNBW; q++) acc[d][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[d], b[q], acc[d][q], 0, 0, 0);
        };
        {
            // Operands ahead of the MFMAs that use them: the bytes two steps (LDS), the taps FOUR steps
            // (the table shares the L1 with the chunk's streaming loads and often comes from L2, 500+
            // cycles against the 256 of a step's MFMAs: one step ahead left the matrix pipe idle half
            // of the loop).  Inside a step the loads go between the MFMAs (sched_group_barrier): issued
            // as a clump after them, the wave's own MFMA stream had a ~100-cycle hole per step.
            // ks is a multiple of GS and the table is padded by four steps in front.
            v4i a[4][kND], b[4][kNBW];
#pragma unroll
            for (int q = 0; q < 4; q++) load_a(a[q], q);
            int t = n, base = TB * t, xh = (t & 15) ^ h;
            load_b(b[0], base, xh, 0);
            load_b(b[1], base, xh, 1);
            const int groups = ((EXP & 2) ? 0 : G.ks) / GS;
            auto group = [&](int) {
                const int tn = t + 1, base_n = TB * tn, xh_n = (tn & 15) ^ h;
#pragma unroll
                for (int j = 0; j < GS; j++) {
                    if (j + 2 < GS) load_b(b[(j + 2) & 3], base, xh, j + 2);
                    else load_b(b[(j + 2) & 3], base_n, xh_n, j + 2 - GS);  // (after the last group: one tile past the image, unused -- kLookAhead)
                    mma(a[j & 3], b[j & 3]);
                    load_a(a[j & 3], j + 4);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // the address of the step's reads
                    if constexpr (kND * kNBW >= kNBW + kND) {
#pragma unroll
                        for (int q = 0; q < kNBW; q++) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // LDS read
                        }
#pragma unroll
                        for (int q = 0; q < kND; q++) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // table load
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, kND * kNBW - kNBW - kND, 0);
                    } else {  // one column block: two MFMAs, one LDS read, two table loads per step
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                t = tn;
                base = base_n;
                xh = xh_n;
#pragma unroll
                for (int d = 0; d < kND; d++) pa[d] -= 4 * GS;
            };
            // The compiler drains every outstanding load at a loop header (s_waitcnt vmcnt(0) lgkmcnt(0):
            // it cannot count across the back edge), which empties the operand pipeline once per trip:
            // the common window (1024 taps at D = 8: nine groups) runs unrolled, without a header.
            if (groups == 9) {
#pragma unroll
                for (int g = 0; g < 9; g++) group(g);
            } else {
#pragma unroll 1
                for (int g = 0; g < groups; g++) group(g);
            }
        }
        stamp(3);
        // The planes meet.  Split by digits: wave 0 (planes 0-1) finishes column blocks 0-1, wave 1 (planes
        // 2-3) blocks 2-3; hi = acc0 * 256 + acc1 is exact in float64, the low pair's weight is 2^-16:
        // float32 is enough.  Split by blocks: a wave has all four planes of its two blocks.
        __syncthreads();  // both waves are done with the chunk's bytes
        const double *dc = reinterpret_cast<const double *>((const uint8_t *)R.tab[r] + (size_t)4 * G.ne * 32);
        const double dcr = dc[0], dci = dc[1];
        const double scale = __hiloint2double((1023 - G.shift) << 20, 0);  // 2^-S
        float2 y[BW][8];  // [own block][4 q4 + ... ]: outputs 4 q4 + 2 h + e of tile 32 (2 wave + j) + n
        if constexpr (kSplitBlocks) {
#pragma unroll
            for (int j = 0; j < BW; j++)
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++)
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        float c2[2];
#pragma unroll
                        for (int pt = 0; pt < 2; pt++) {
                            const int q = 4 * q4 + 2 * e + pt;
                            const double hi = __fma_rn((double)acc[0][j][q], 256.0, (double)acc[1][j][q]);
                            const double lo = __fma_rn((double)acc[kND - 2][j][q], 256.0, (double)acc[kND - 1][j][q]);
                            const double v = __fma_rn(hi, 65536.0, lo) + (pt ? dci : dcr);
                            c2[pt] = (float)(v * scale);
                        }
                        y[j][2 * q4 + e] = make_float2(c2[0], c2[1]);
                    }
        } else {
            double *xhi = reinterpret_cast<double *>(mm_lds);              // [BW][16][64]
            float *xlo = reinterpret_cast<float *>(mm_lds + BW * 16 * 64 * 8);  // [BW][16][64]
            if (wave == 0) {
#pragma unroll
                for (int j = 0; j < BW; j++)
#pragma unroll
                    for (int q = 0; q < 16; q++)
                        xhi[(j * 16 + q) * 64 + l] = __fma_rn((double)acc[0][(BW + j) % kNBW][q], 256.0, (double)acc[1][(BW + j) % kNBW][q]);
            } else {
#pragma unroll
                for (int j = 0; j < BW; j++)
#pragma unroll
                    for (int q = 0; q < 16; q++)
                        xlo[(j * 16 + q) * 64 + l] = __fmaf_rn((float)acc[0][j][q], 256.0f, (float)acc[1][j][q]);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < BW; j++)
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++)
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        float c2[2];
#pragma unroll
                        for (int pt = 0; pt < 2; pt++) {
                            const int q = 4 * q4 + 2 * e + pt;
                            double hi, lo;
                            if (wave == 0) {
                                hi = __fma_rn((double)acc[0][j][q], 256.0, (double)acc[1][j][q]);
                                lo = (double)xlo[(j * 16 + q) * 64 + l];
                            } else {
                                hi = xhi[(j * 16 + q) * 64 + l];
                                lo = (double)__fmaf_rn((float)acc[0][(BW + j) % kNBW][q], 256.0f, (float)acc[1][(BW + j) % kNBW][q]);
                            }
                            const double v = __fma_rn(hi, 65536.0, lo) + (pt ? dci : dcr);
                            c2[pt] = (float)(v * scale);
                        }
                        y[j][2 * q4 + e] = make_float2(c2[0], c2[1]);
                    }
        }
        // The elementwise program over this lane's 16 outputs m = mb + 512 j + 4 q4 + e: equally spaced
        // in three directions inside one exactly-linear clock run, so a Shift stage is
        // z0 * wB^j * wA^q4 * wE^e (see ew_apply_seq for the error argument).
        // (one column block per workgroup: wave 0 finishes it, wave 1 only hands its planes over)
        const bool fin = NB >= 2 || wave == 0;
        const uint32_t mb = m_start + (uint32_t)(32 * BW * wave + n) * kT + 2 * h;
#pragma unroll 1
        for (int oi = 0; oi < ((EXP & 4) || !fin ? 0 : P.n); oi++) {  // uniform per wave
            const EwOp &o = P.op[oi];
            if (o.kind == EW_SCALE) {
#pragma unroll
                for (int j = 0; j < BW; j++)
#pragma unroll
                    for (int q = 0; q < 8; q++) y[j][q] = make_float2(__fmul_rn(y[j][q].x, o.a), __fmul_rn(y[j][q].y, o.a));
            } else if (o.kind == EW_ROTATE) {
#pragma unroll
                for (int j = 0; j < BW; j++)
#pragma unroll
                    for (int q = 0; q < 8; q++) y[j][q] = go_cmul(y[j][q], make_float2(o.a, o.b));
            } else {
                // the clock at the lane's first output, by the run's line -- also for a lane whose first
                // outputs lie BEFORE the run (a chunk may start in the previous run; those outputs are
                // not stored, but the recurrence to the lane's valid ones starts there)
                const double step = P.segs.step[r];
                const int64_t dj = (int64_t)((uint64_t)D * mb) - (int64_t)P.segs.first[r];
                const double ts0 = __fma_rn((double)dj, step, P.segs.t0[r]);
                // The three step factors are the same for every lane: lanes 0 / 1 / 2 evaluate one each
                // and the wave reads them back (two Sincos per lane instead of four).  The products with
                // the filter outputs are float32 (the factors rounded once, as the reference does; the
                // late path is held to an error bound and float64 products cost a third of the epilogue).
                double z0s, z0c, fs, fc;
                sincos_late(__dmul_rn(o.tau_shift, ts0), z0s, z0c);
                const double mult = l == 0 ? (double)D : l == 1 ? (double)(4 * D) : (double)(32 * kT * D);
                sincos_late(__dmul_rn(o.tau_shift, __dmul_rn(mult, step)), fs, fc);
                const double es = __shfl(fs, 0), ec = __shfl(fc, 0), as = __shfl(fs, 1), ac = __shfl(fc, 1);
                const double bs = __shfl(fs, 2), bc = __shfl(fc, 2);
                auto mul32 = [](float2 a, double c, double sn) {
                    const float cr = (float)c, ci = (float)sn;
                    return make_float2(__fmaf_rn(a.x, cr, -(a.y * ci)), __fmaf_rn(a.x, ci, a.y * cr));
                };
#pragma unroll
                for (int j = 0; j < BW; j++) {
                    double zc = z0c, zs = z0s;
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) {
                        y[j][2 * q4] = mul32(y[j][2 * q4], zc, zs);
                        const double ze_c = __fma_rn(zc, ec, -(zs * es)), ze_s = __fma_rn(zc, es, zs * ec);
                        y[j][2 * q4 + 1] = mul32(y[j][2 * q4 + 1], ze_c, ze_s);
                        const double nc = __fma_rn(zc, ac, -(zs * as)), ns = __fma_rn(zc, as, zs * ac);
                        zc = nc;
                        zs = ns;
                    }
                    const double nc = __fma_rn(z0c, bc, -(z0s * bs)), ns = __fma_rn(z0c, bs, z0s * bc);
                    z0c = nc;
                    z0s = ns;
                }
            }
        }
        stamp(4);
        // Stores: a lane holds eight outputs of each of its tiles, 128 bytes apart from the next lane's --
        // through LDS (16-byte pieces swizzled by the tile index) a wave writes whole 1 KiB rows instead
        // (direct 8-byte stores cost 10 us of the kernel's 46).
        __syncthreads();  // the exchange buffers have been read
        if (fin) {
            float4 *yl = reinterpret_cast<float4 *>(mm_lds) + wave * (BW * 32 * 8);  // [BW blocks][32 tiles][8 pieces]
#pragma unroll
            for (int j = 0; j < BW; j++)
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++)
                    yl[(j * 32 + n) * 8 + ((2 * q4 + h) ^ (n & 7))] =
                        make_float4(y[j][2 * q4].x, y[j][2 * q4].y, y[j][2 * q4 + 1].x, y[j][2 * q4 + 1].y);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 4 * BW; it++) {
                const int idx = it * 64 + l, tile = idx >> 3, piece = idx & 7;  // tile = 32 j + n
                const float4 v = yl[tile * 8 + (piece ^ (tile & 7))];
                const uint32_t rel = (uint32_t)(32 * BW * wave + tile) * kT + 2 * piece;
                const uint32_t mo = m_start + rel;  // (even: v_lo and v_hi are even except the call's last output)
                const bool st = (EXP & 8) == 0 && mo >= v_lo;
                if (st && mo + 1 < v_hi) *reinterpret_cast<float4 *>(out + mo) = v;
                else if (st && mo < v_hi) out[mo] = make_float2(v.x, v.y);
            }
        }
    }
    stamp(5);
    if (!active || tasks_late) {
        if (active) __syncthreads();
        for (int task = wb; task < n_tasks; task += (int)gridDim.x) small_task(task);  // (uniform)
    }
}

// hz_firmm.hip: the decimations the kernel is instantiated for, and the launch (u8 / i8)
bool factor_ok(unsigned D);
int launch_fir(hipStream_t stream, int fmt, unsigned D, const void *in, float2 *out, const float2 *hist, float2 *new_hist,
               const uint8_t *rhist, uint8_t *new_rhist, const float2 *taps, size_t n, const Geom &g, const EwProgram &P,
               const Runs &R, const Fix &F);

}  // namespace mm
}  // namespace hz

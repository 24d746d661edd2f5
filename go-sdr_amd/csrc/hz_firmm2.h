// hz_firmm2.h -- the int8 matrix FIR (hz_firmm.h) re-cut so that its phases overlap: one PERSISTENT
// workgroup per CU whose eight waves pull small passes from a queue.
//
// hz_firmm.h runs 1024 workgroups, each of which loads 34 KB, multiplies for 24 us and finishes 2048
// outputs -- all of them in the same phase at the same time (accumulators for ALL 2^21 outputs of a
// 2^24-sample call are resident at once, so there is exactly one round): the matrix pipe idles through
// the 9 us input burst and the 5 us epilogue, 44 % busy over the kernel.  A round can only be cut into
// shorter ones with FEWER accumulators per wave, and fewer accumulators mean less operand re-use; what
// makes that affordable is the LDS (256 B/clk/CU for ds_read_b128, a quarter of it used by hz_firmm.h)
// once the tap table lives there too -- which needs ONE copy per CU, i.e. one workgroup per CU:
//
//   * a tile = EIGHT consecutive outputs x (re, im) x TWO digit planes = 32 rows of the Toeplitz
//     matrix (hz_firmm.h: 16 outputs x (re, im), the planes split over two waves).  One wave holds all
//     four planes of its outputs (two A fragments), nothing is exchanged between waves, and a tile's
//     window is 56 samples wider than the taps instead of 120 (68 instead of 72 MFMA steps at 1024 taps);
//   * a PASS = 2 column blocks of 32 tiles = 512 outputs (D = 8): 4 accumulators (64 registers), 10 KB of
//     input in a slot of LDS that belongs to the wave.  Per 32-byte step of the window: two A fragments
//     from the table in LDS, two B fragments from the slot, four MFMAs -- 1 KB of LDS reads per MFMA,
//     half of what the LDS delivers, no global load in the loop;
//   * a wave's life: take a pass from the workgroup's queue (an LDS counter), and while it multiplies,
//     the NEXT pass's bytes are in flight into registers (landed in the slot when the loop is done);
//     the finished pass's float64 combination, mixer and stores then run beside the partner wave's
//     loop on the same SIMD.  The eight waves de-phase by themselves (the queue) and by construction
//     (waves 4-7, the second wave of each SIMD, start half a pass late);
//   * the table (19 KB per clock run at 1024 taps) is loaded once per workgroup; a workgroup whose
//     range of passes spans a run boundary reloads it between the runs (a barrier; one or two of the
//     256 workgroups per call);
//   * fix-up tasks (outputs in reference order) and history tasks are wave-sized here and ride on the
//     waves of all workgroups before their first pass.
//
// Geometry, table contents and the arithmetic (four balanced base-256 digit planes of 32-bit
// fixed-point taps, exact int32 sums, one float64 combination, one rounding) are those of hz_firmm.h.
#pragma once
#include <type_traits>

#include "hz_firmm.h"
#include "hz_firmm2_plan.h"

namespace hz {
namespace mm2 {

using mm::v16i;
using mm::v4i;

using mm::find_le;
using mm::task_window;

// The kernel's arguments as ONE struct (the kernarg segment's layout: the explicit arguments in order, each at its
// natural alignment), for the RARE paths -- fix-up and history tasks, edge landings, the elementwise stages other
// than the one Shift the planner marks.  Left to read the arguments the ordinary way, everything those paths derive
// from them (pointers, comparisons, table addresses) is loop-invariant and the compiler hoists it to the kernel's
// head and parks it in spilled scalar registers: 138 of the 621 instructions in front of the first barrier were
// v_writelane, executed by every wave of every launch at the cold instruction cache's ~7 ns per instruction, for
// paths most launches never take.  A rare path starts with HZ_COLD_ARGS instead: the kernarg pointer through an
// empty asm (nothing derived from it can move above that line) and its own names for the arguments.
struct KernArgs {
    const void *in;
    float2 *out;
    const float2 *hist;
    float2 *new_hist;
    const uint8_t *rhist;
    uint8_t *new_rhist;
    const float2 *taps;
    size_t n_in;
    Geom G;
    Plan L;
    EwProgram P;
    Fix F;
    Batch B;
    unsigned long long *stamps;
};
typedef const KernArgs __attribute__((address_space(4))) *KernArgsP;
#define HZ_COLD_ARGS                                                                                                   \
    KernArgsP ca_ = (KernArgsP)__builtin_amdgcn_kernarg_segment_ptr();                                                 \
    asm volatile("" : "+s"(ca_));                                                                                      \
    [[maybe_unused]] const void *const in = ca_->in;                                                                   \
    [[maybe_unused]] float2 *const out = ca_->out;                                                                     \
    [[maybe_unused]] const float2 *const hist = ca_->hist;                                                             \
    [[maybe_unused]] float2 *const new_hist = ca_->new_hist;                                                           \
    [[maybe_unused]] const uint8_t *const rhist = ca_->rhist;                                                          \
    [[maybe_unused]] uint8_t *const new_rhist = ca_->new_rhist;                                                        \
    [[maybe_unused]] const float2 *const taps = ca_->taps;                                                             \
    [[maybe_unused]] const size_t n_in = ca_->n_in;                                                                    \
    [[maybe_unused]] const Geom &G = *(const Geom *)&ca_->G;                                                           \
    [[maybe_unused]] const Plan &L = *(const Plan *)&ca_->L;                                                           \
    [[maybe_unused]] const EwProgram &P = *(const EwProgram *)&ca_->P;                                                 \
    [[maybe_unused]] const Fix &F = *(const Fix *)&ca_->F;                                                             \
    [[maybe_unused]] const Batch &B = *(const Batch *)&ca_->B

// A Shift stage of a program with SEVERAL Shift stages (float64 phases, as hz_firmm.h), over a lane's outputs
// m = mb + 32 kT b + a of a pass: OUT OF LINE.  Inlined, its two float64 Sincos put twenty-two polynomial constants
// into vector registers at the kernel's head -- loop-invariant, so hoisted, for every wave of every launch -- for a
// path that a chain with one Shift never takes.  `y` travels through memory (the caller copies: its own outputs
// stay in registers).
template <int D, int NB>
__device__ __attribute__((noinline)) void other_shift_stage(float2 *y, double tau_shift, double step, double ts0, int l) {
    double cs0, cc0, fs, fc;
    sincos_late(__dmul_rn(tau_shift, ts0), cs0, cc0);
    const double mult = l == 0 ? (double)D : (double)(32 * kT * D);
    sincos_late(__dmul_rn(tau_shift, __dmul_rn(mult, step)), fs, fc);
    const double e_s = __shfl(fs, 0), e_c = __shfl(fc, 0), b_s = __shfl(fs, 1), b_c = __shfl(fc, 1);
    auto mul32 = [](float2 v, double c, double sn) {
        const float cr = (float)c, ci = (float)sn;
        return make_float2(__fmaf_rn(v.x, cr, -(v.y * ci)), __fmaf_rn(v.x, ci, v.y * cr));
    };
#pragma unroll
    for (int b = 0; b < NB; b++) {
        double zc = cc0, zs = cs0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            y[4 * b + q] = mul32(y[4 * b + q], zc, zs);
            if (q < 3) {
                const double nc = __fma_rn(zc, e_c, -(zs * e_s)), ns = __fma_rn(zc, e_s, zs * e_c);
                zc = nc;
                zs = ns;
            }
        }
        if (b + 1 < NB) {
            const double nc = __fma_rn(cc0, b_c, -(cs0 * b_s)), ns = __fma_rn(cc0, b_s, cs0 * b_c);
            cc0 = nc;
            cs0 = ns;
        }
    }
}

// The mixer's complex products as the packed float32 pipe takes them WITHOUT moves: a complex value is a register
// pair (re, im); a broadcast of one half (op_sel / op_sel_hi), the exchange of the halves and a negated half (neg_lo)
// are operand modifiers of v_pk_mul_f32 / v_pk_fma_f32 and cost nothing.  The compiler does not use them for this
// code -- it builds (-s, c) from (c, s) with a v_xor and two v_mov per product and pads every packed instruction with
// an s_nop: ~13 vector instructions per output where 6 do (round 5; every vector instruction of an epilogue costs
// the SIMD partner's matrix loop ~2.5 cycles).  Same operations in the same order as the float expressions they
// replace: bit-identical results.  None of the forms below is the gfx950 hazard's (op_sel:[0,1] with src0 straight,
// hz_firmm.h); tools/fix_pk_opsel.py checks the built library all the same.
typedef float v2f __attribute__((ext_vector_type(2)));
// (cs, sn) = (c0, s0) turned by w = (cos hi, sin hi, cos lo, sin lo):
//   cs = fma(c0, w.x, fma(-s0, w.y, fma(c0, w.z, -(s0 w.w)))),  sn = fma(c0, w.y, fma(s0, w.x, fma(c0, w.w, s0 w.z)))
__device__ __forceinline__ v2f pk_turn(v2f cs0, v2f wxy, v2f wzw) {
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(cs0), "v"(wzw));      // (-s0 w.w, s0 w.z)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(t) : "v"(cs0), "v"(wzw));                           // + (c0 w.z, c0 w.w)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(t) : "v"(cs0), "v"(wxy));  // + (-s0 w.y, s0 w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(t) : "v"(cs0), "v"(wxy));                           // + (c0 w.x, c0 w.y)
    return t;
}
// y (cs, sn):  re = fma(y.x, cs, -(y.y sn)),  im = fma(y.x, sn, y.y cs)
__device__ __forceinline__ v2f pk_cmul(v2f y, v2f f) {
    v2f u, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(u) : "v"(y), "v"(f));           // (-y.y sn, y.y cs)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(y), "v"(f), "v"(u));                       // + (y.x cs, y.x sn)
    return r;
}

// EXP (tools/mfma_fir2.hip; kLibExp in the library): 1 = no input loads, 2 = no matrix loop, 4 = no mixer,
// 8 = no stores, 16 = no stagger, 32 = wave priorities (see the pass loop's end), 64 = stamps, 128 = no explicit vmcnt(0),
// 256 = accumulator checksums per pass and lane, taken right behind the matrix loop and again behind the landing,
// stored (first launch) or compared with the stored ones (tools/mm2_glitch.hip: which register, which lanes,
// which compute unit, stale read or wrong sum), 512 = the mixer's step factors read from LDS in one batch, waited for
// (lgkmcnt(0) and eight wait states) and pinned before their first use, 1024 = the first step factor formed a second
// time from a second, fully waited read and compared with the one the mixer used (records as for 256), 2048 = the
// constant 0 as the first step's C operand instead of cleared accumulator registers, 4096 = the rare paths read the
// kernel arguments as the common path does (HZ_COLD_ARGS below switched off: round 3's code shape, for A/B).
// Round 5, fewer vector instructions per pass (each switchable for A/B; kLibExp = what the library ships):
// 8192 = the mixer in hand-packed form (pk_turn / pk_cmul above), 16384 = the two top digit planes combined in int32
// in front of the float64 step (the host bounds the sums: hz_firmm_plan.h, digit_shift), 32768 = a u8 pass image's
// sign flip by the LDS (ds_xor_b64 behind the landing's writes) instead of 40 v_xor per lane, 65536 = the first
// step's MFMAs with the constant 0 as C through inline assembly (early-clobber destinations: no clears, no overlap of
// destinations and sources), 131072 = the cold multi-Shift path waits for its own scratch reloads and is marked
// unlikely (the compiler otherwise leaves a vmcnt(0) for them at the pass loop's top, behind the prefetch's issue, and
// eight register moves on the hot path where the two meet), 262144 = a u8 pass image's sign flip among the wave's OWN
// MFMAs, near the matrix loop's end (straight-line form).  Measured one by one and together, tools/mfma_fir2.hip AB=1:
// 32768 is 0.8 us SLOWER (LDS atomics), 131072 alone -1.0 us, the rest -0.3 together.
// 524288 = a pass's stores among the MFMAs of the wave's NEXT matrix loop (a global store does not issue beside a
// partner that streams MFMAs, tools/epi_cost.hip: it waits for the partner's loop to end) -- and yet 1 us per buffer
// SLOWER in a call over four buffers, even single: not shipped.  1 << 22 = the mixer's products as SCALAR float32
// instructions: a packed float32 instruction does not issue beside a streaming partner either (the mixer beside a
// partner's loop: 5.6 us packed, 1.2 us scalar -- tools/mfma_fir2.hip BISECT=1, LDS-staged stamps), which pinned the
// epilogue to the partner's loop in every round so far; -0.6 ... -1.4 us per buffer in a call over four.  (1 << 20,
// 1 << 21: the bisection's other switches -- no Sincos, no step factors from LDS; neither matters.)
// NG: the window's groups (ks / GS) when the instantiation is for ONE tap count -- the matrix loop is then
// straight-line code (a loop header drains the operand pipeline: the compiler cannot count outstanding
// loads across a back edge); 0: any window, a loop over the groups.
template <int FMT, int D, int NG = 0, int EXP = 0, int UG = 0>
__global__ __launch_bounds__(kThreads) void fir_mm2_kernel(
    const void *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ hist,
    float2 *__restrict__ new_hist, const uint8_t *__restrict__ rhist, uint8_t *__restrict__ new_rhist,
    const float2 *__restrict__ taps, size_t n_in, Geom G, Plan L, EwProgram P, Fix F, Batch B,
    unsigned long long *stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t mm_lds[];
    constexpr int TB = tile_bytes(D), TS = tile_stride(D), PPT = TB / 16, GS = PPT / 2, NB = blocks_for(D);
    constexpr int kPassOut = pass_out(D);
    // pieces per lane of a pass image: known when the window is (NG groups: pass_tiles - 1 + NG tiles of PPT pieces)
    // the matrix loop as straight-line code (NG > 0, UG = 0) or as a loop of UG groups per trip
    constexpr bool STRAIGHT = NG > 0 && UG == 0;
    constexpr int KU = NG > 0 ? ((pass_tiles(D) - 1 + NG) * PPT + 63) / 64 : kU;
    constexpr bool kWhole = NG > 0 && ((pass_tiles(D) - 1 + NG) * PPT) % 64 == 0;  // every lane has KU pieces
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int wb = blockIdx.x;
    const int n = l & 31, h = l >> 5;
    const size_t tab_lds = table_lds(G.ne), slot_sz = slot_bytes(D, G.ks);
    uint8_t *const tabp = mm_lds;
    unsigned *const ctr = reinterpret_cast<unsigned *>(mm_lds + 2 * tab_lds);
    uint8_t *const slot = mm_lds + 2 * tab_lds + kCtlBytes + (size_t)wave * slot_sz;
    const int64_t n_bytes = 2 * (int64_t)n_in;
    const uint8_t *src = (const uint8_t *)in;
    const int pieces = (int)(image_bytes(D, G.ks) / 16);
    // A call over several buffers (hz_firmm2_plan.h: Batch): `in` / `out` are buffer 0's, buffer j's virtual base is
    // read when a pass of buffer j comes up; everything is indexed by the position in the concatenation.
    const int nbuf = B.nbuf;
    const uint32_t rcp = B.rcp;
    const int64_t nb = (int64_t)B.nb;
    // (pass / ppb by the host's reciprocal: one s_mul_hi_u32 -- a chain of comparisons summed as integers became
    // vector instructions, fifty per pass)
    auto buf_of = [&](uint32_t pass) -> int { return nbuf > 1 ? (int)__umulhi(pass, rcp) : 0; };
    // (as integers, used through GLOBAL pointers: a pointer made from an integer is a generic one to the compiler, and
    // a flat load counts as an LDS operation too -- the matrix loop's first ds_read would wait for the prefetch)
    auto vin_of = [&](int j) -> uint64_t { return j == 0 ? (uint64_t)(uintptr_t)src : B.vin[j]; };
    auto vout_of = [&](int j) -> uint64_t { return j == 0 ? (uint64_t)(uintptr_t)out : B.vout[j]; };
    typedef const v4i __attribute__((address_space(1))) *gload_p;
    // (EXP & 64: s_memrealtime stamps, 10 ns ticks: slot 0 of a wave = its start and end, slots 1 .. 3 its passes)
    // The stamps are kept in LDS (2 KB behind the kernel's own: the harness asks for them) and written out when the wave
    // ends: a global store does not issue while the SIMD partner's MFMAs are back to back (tools/epi_cost.hip), so a
    // stamp stored at once pinned the very phases it was to time to the partner's loop (rounds 3-4 read "the epilogue
    // takes as long as the partner's loop" off such stamps).
    // (1 << 24, with 64: a row of stamps for EVERY pass of the wave, up to fifteen -- a call over four buffers gives a wave
    // eight; 8 KB of staging behind the kernel's own LDS, which the harness asks for -- tools/mfma_fir2.hip PASSES=1)
    constexpr int kStampRows = (EXP & (1 << 24)) != 0 ? 16 : 4;
    int stamp_pass = 0;
    [[maybe_unused]] unsigned long long *const lstamp =
        reinterpret_cast<unsigned long long *>(mm_lds + lds_bytes(D, G.ks, G.ne, G.ntaps)) + (size_t)wave * (8 * kStampRows);
    if constexpr ((EXP & 64) != 0) {
        for (int i = l; i < 8 * kStampRows; i += 64) lstamp[i] = 0;
    }
    auto stamp = [&](int k) {
        if constexpr ((EXP & 64) != 0) {
            if (l == 0 && stamp_pass < kStampRows)
                lstamp[(k == 0 || k == 7 || (k >= 8 && k < 16) ? 0 : stamp_pass) * 8 + (k & 7)] = __builtin_amdgcn_s_memrealtime();
        }
    };
    stamp(0);
    // (with the stamps: the shader clock the workgroup ran at -- s_memtime counts shader cycles, s_memrealtime 100 MHz)
    [[maybe_unused]] unsigned long long clk0 = 0, rt0 = 0;
    if constexpr ((EXP & 64) != 0) clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    if constexpr ((EXP & 32) != 0) __builtin_amdgcn_s_setprio(1);
    const Run run0 = L.run[0];  // (read with the header: most workgroups of most calls are in run 0)
    // Everything the way to the first loads reads from the kernel arguments, wanted HERE: the compiler then issues
    // these scalar loads together and waits once (left alone it reads each next to its use: five load-wait
    // groups, ~0.6 us each through the scalar cache while every wave of the chip asks, before the first byte was
    // requested).
    {
        const uint64_t a0 = (uint64_t)(uintptr_t)in, a1 = (uint64_t)n_in, a2 = (uint64_t)(uintptr_t)run0.tab;
        asm volatile("" ::"s"(a0), "s"(a1), "s"(a2), "s"(G.w0), "s"(G.ks), "s"(G.ne), "s"(G.e0), "s"(G.ntaps), "s"(G.off), "s"(L.n),
                     "s"(L.n_pass), "s"(L.grid), "s"(L.n_task), "s"(L.pass_first[0]), "s"(L.pass_end[0]), "s"(L.pass_first[1]),
                     "s"(L.pass_end[1]), "s"(run0.m_lo), "s"(run0.m_hi), "s"(nbuf), "s"(rcp), "s"(nb));
    }

    // ---- this workgroup's passes [pb0, pb1) of the call -------------------------------------------------
    const uint32_t NP = (uint32_t)L.n_pass;
    // (NP / grid passes each, the first NP % grid workgroups one more; 32-bit arithmetic)
    const uint32_t per = NP / (uint32_t)L.grid, rem = NP - per * (uint32_t)L.grid;
    const uint32_t pb0 = (uint32_t)wb * per + min((uint32_t)wb, rem), pb1 = pb0 + per + ((uint32_t)wb < rem ? 1u : 0u);

    // The bytes of a pass: lane l takes pieces l, l + 64, ...  A pass that lies inside the buffer (all but the
    // call's first and last) is prefetched into registers while the previous pass multiplies (issue, then
    // land); the others are brought in piece by piece when their turn comes (land_edge: bytes before the
    // buffer are the previous call's last samples when the clock run continues, zeros otherwise).
    auto pass_p0 = [&](uint32_t pass) { return 2 * ((int64_t)D * ((int64_t)pass * kPassOut) - G.w0); };
    auto inside = [&](uint32_t pass) {  // (the image lies in ONE buffer)
        const int64_t p0 = pass_p0(pass), lo = (int64_t)buf_of(pass) * nb;
        return p0 >= lo && p0 + 16 * (int64_t)pieces <= lo + nb;
    };
    auto issue = [&](v4i(&x)[KU], uint32_t pass) {
        const gload_p p = (gload_p)(vin_of(buf_of(pass)) + (uint64_t)pass_p0(pass));  // (in pieces of 16 bytes)
        const gload_p pl = p + l;  // (ONE address per lane, the pieces at immediate offsets)
#pragma unroll
        for (int u = 0; u < KU; u++) {
            x[u] = v4i{0, 0, 0, 0};  // (pieces past the image: the last piece again, not landed)
            if constexpr ((EXP & 1) != 0) continue;
            if constexpr (kWhole) x[u] = __builtin_nontemporal_load(pl + 64 * u);
            else x[u] = __builtin_nontemporal_load(p + min(l + u * 64, pieces - 1));
        }
    };
    // The first pass's bytes, requested before anything else is even fetched: the instruction cache is cold at every
    // launch, and the path from the kernel's first instruction to this one decides when the matrix pipes start
    // (measured: 2.4 us of the kernel's first 5.5 went by between two stamps with ~300 instructions in between).
    // Waves 0-3 take passes pb0 .. pb0 + 3 when those lie in run 0 (the group loop below checks that it agrees).
    // (The first run's TABLE asked for here as well, 0.7 us ahead of the group loop's head where it is: medians 37.8-38.4
    // against 37.9-38.9 us, minima 36.7-36.9 against 36.3-36.4, interleaved on one box -- nothing, and not kept.)
    v4i x[KU];
    bool pre_issued = false;
    if (wave < kWaves / 2) {
        const uint32_t p = pb0 + (uint32_t)wave;
        if (pb0 >= (uint32_t)L.pass_first[0] && p < min(pb1, (uint32_t)L.pass_end[0]) && inside(p)) {
            issue(x, p);
            pre_issued = true;
        }
    }
    // ---- the small tasks: waves 0-3 of a workgroup, behind their last pass (the calls are at the kernel's end) ------
    // A FIX-UP TASK = 16 outputs in reference order (direct form, float64).  One wave alone took 18-27 us over one
    // (measured: ~3000 instructions per lane, every load a full trip): workgroup t takes task t, its four old
    // waves stage the window's 1144 samples together (load, convert, the elementwise program: five per thread,
    // all loads in flight at once) into a scratch of their own in LDS while the young waves run the workgroup's
    // last pass, and behind the barrier the same 256 threads -- 16 per output, 64 taps each -- sum and store.  One
    // per workgroup and round: a call with more fix-up tasks than four rounds keeps the transform kernels (the host).
    // A HISTORY TASK = 64 samples of the next call's history, one old wave each.
    using RWT = typename Raw<FMT>::t;
    uint32_t fix_m0 = 0;
    int fix_cnt = 0;
    // (the tasks' scratch in LDS behind the slots: the window's samples, the taps beside them)
    auto task_xs = [&](const Geom &G) { return reinterpret_cast<float2 *>(mm_lds + 2 * table_lds(G.ne) + kCtlBytes + (size_t)kWaves * slot_bytes(D, G.ks)); };
    auto tasks_front_cold = [&](int round, const void *in, const float2 *hist, float2 *new_hist, const uint8_t *rhist, uint8_t *new_rhist,
                                const float2 *taps, size_t n_in, const Geom &G, const Plan &L, const EwProgram &P, const Fix &F, const Batch &B) {
        // (sample pu of the concatenation: its buffer's virtual base)
        auto vbase = [&](int64_t pu) -> const RWT * {
            const void *sb = in;
            for (int i = 1; i < B.nbuf; i++)
                if (2 * pu >= (int64_t)i * (int64_t)B.nb) sb = (const void *)(uintptr_t)B.vin[i];
            return (const RWT *)sb;
        };
        float2 *const xs = task_xs(G);
        float2 *const tl = xs + (G.ntaps + D * (kFixOut - 1));
        const int ftask = (L.grid - 1 - wb) + round * L.grid;  // (from the grid's far end: see the calls)
        if (ftask < L.n_task) {
            const int ct = tid;  // 0 .. 255: the old waves
            const int k = find_le(F.task_first, F.n, ftask);
            fix_m0 = F.m_a[k] + (uint32_t)(ftask - F.task_first[k]) * kFixOut;
            fix_cnt = (int)min((uint32_t)kFixOut, F.m_b[k] - fix_m0);
            if (round == 0)
                for (int q = ct; q < G.ntaps; q += 256) tl[q] = taps[q];
            const int64_t p_lo = (int64_t)D * fix_m0 - (G.ntaps - 1);
            const int n_s = G.ntaps + D * (fix_cnt - 1);
            const NcoWin tw = task_window(P, p_lo, p_lo + n_s - 1);
#pragma unroll 1
            for (int i0 = ct; i0 < n_s; i0 += 5 * 256) {  // (one trip up to 1280 samples: 1160 taps)
                constexpr int W = 5;
                float2 v[W];
#pragma unroll
                for (int u = 0; u < W; u++) {
                    const int64_t pu = p_lo + i0 + u * 256;
                    v[u] = Raw<FMT>::cvt((i0 + u * 256 < n_s && pu >= 0) ? vbase(pu)[pu] : RWT{});
                }
                ew_apply_n<W, 2>(P, v, (uint64_t)(p_lo + i0), tw, (uint64_t)256);
#pragma unroll
                for (int u = 0; u < W; u++) {
                    const int idx = i0 + u * 256;
                    const int64_t pu = p_lo + idx;
                    if (idx < n_s)
                        xs[idx] = pu >= 0 ? v[u] : ((hist && pu + (int64_t)G.off >= 0) ? hist[pu + (int64_t)G.off] : make_float2(0.f, 0.f));
                }
            }
        }
    };
    // A HISTORY TASK = 64 samples of the next call's history (the stages in reference order, and the raw samples): task
    // j goes to workgroup j mod grid, one wave each, and it is the OLD waves that take them, BEHIND their last pass:
    // waves 0-3 are through with their passes ~5 us before waves 4-7, who run the workgroup's last pass alone.  (Round
    // 3 had wave 7 of the first sixteen workgroups do them in front of its first pass -- 5 us of exact Sincos that
    // made those waves, and with them the launch, end 2 us late: tools/mfma_fir2.hip lists the workgroups that end
    // last, and they were workgroups 3 and 11 on every box.)
    auto history_tasks_cold = [&](const void *in0, const float2 *hist, float2 *new_hist, const uint8_t *rhist, uint8_t *new_rhist, size_t n_in,
                                  const Geom &G, const Plan &L, const EwProgram &P, const Batch &B) {
        // (the call's last samples: the last buffer's, which holds at least `off` samples when there are several)
        const void *const in = B.nbuf > 1 ? (const void *)(uintptr_t)B.vin[B.nbuf - 1] : in0;
        const int n_hist_tasks = new_hist ? (int)((G.off + kHistPer - 1) / kHistPer) : 0;
        for (int j = wb + L.grid * wave; j < n_hist_tasks; j += (kWaves / 2) * L.grid) {  // (uniform per wave)
            const unsigned idx = (unsigned)j * kHistPer + l;
            const int64_t h_lo = (int64_t)n_in - (int64_t)G.off + (int64_t)j * kHistPer;
            const NcoWin tw = task_window(P, h_lo, h_lo + kHistPer - 1);
            if (idx < G.off) {
                const int64_t p = (int64_t)n_in - (int64_t)G.off + idx;
                new_hist[idx] = mm::ordered_sample<FMT, 2>(in, P, p, hist, G.off, tw);  // (the tasks' own Sincos flavour: hz_firmm.h)
                reinterpret_cast<RWT *>(new_rhist)[idx] = p >= 0 ? ((const RWT *)in)[p] : reinterpret_cast<const RWT *>(rhist)[p + (int64_t)G.off];
            }
        }
    };
    auto history_tasks = [&]() {
        if (wave >= kWaves / 2 || new_hist == nullptr || wb + L.grid * wave >= (int)((G.off + kHistPer - 1) / kHistPer)) return;
        if constexpr ((EXP & 4096) != 0) {
            history_tasks_cold(in, hist, new_hist, rhist, new_rhist, n_in, G, L, P, B);
        } else {
            HZ_COLD_ARGS;
            history_tasks_cold(in, hist, new_hist, rhist, new_rhist, n_in, G, L, P, B);
        }
    };
    // Round r: workgroup wb takes fix-up task wb + r grid.  Round 0 is called ONCE, in front of the run groups, the
    // further rounds of a call with more tasks than workgroups (short calls across many clock boundaries) BEHIND them
    // -- never inside the groups' loop: what a call site inside a loop derives from loop-invariant values (the exact
    // Sincos' twenty-two polynomial constants, for one) is hoisted to the loop's head, i.e. into every launch's first
    // microseconds.
    auto tasks_front = [&](int round) {
        if (wave >= kWaves / 2) return;
        if ((L.grid - 1 - wb) + round * L.grid >= L.n_task) return;  // (what most workgroups of most calls find)
        if constexpr ((EXP & 4096) != 0) {
            tasks_front_cold(round, in, hist, new_hist, rhist, new_rhist, taps, n_in, G, L, P, F, B);
        } else {
            HZ_COLD_ARGS;
            tasks_front_cold(round, in, hist, new_hist, rhist, new_rhist, taps, n_in, G, L, P, F, B);
        }
    };
    auto tasks_back_cold = [&](float2 *out0, const Geom &G, const Batch &B) {
        float2 *const xs = task_xs(G);
        float2 *const tl = xs + (G.ntaps + D * (kFixOut - 1));
        const int ct = tid;
        const int o = ct >> 4, sl = ct & 15;  // output, tap slice: sixteen lanes per output
        // a lane's 64 (ntaps / 16) taps as two packed float32 fma chains -- the reference's own arithmetic is float32
        // throughout; a chain of 64 terms stays below 1e-7 here -- and the sixteen lanes' partial sums in float64:
        // two vector instructions per tap instead of eight (the workgroup's first passes wait for this)
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f a0{0.f, 0.f}, a1{0.f, 0.f};
        if (o < fix_cnt) {
            const float2 *xo = xs + D * o + (G.ntaps - 1);
#pragma unroll 8
            for (int kk = sl; kk < G.ntaps; kk += 16) {
                const float2 hk = tl[kk], x = xo[-kk];
                a0 = __builtin_elementwise_fma(v2f{x.x, x.x}, v2f{hk.x, hk.y}, a0);   // (xr hr, xr hi)
                a1 = __builtin_elementwise_fma(v2f{-x.y, x.y}, v2f{hk.y, hk.x}, a1);  // (-xi hi, xi hr)
            }
        }
        double ar = (double)a0.x + (double)a1.x, ai = (double)a0.y + (double)a1.y;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            ar += __shfl_xor(ar, d);
            ai += __shfl_xor(ai, d);
        }
        if (o < fix_cnt && sl == 0) {
            float2 *ob = out0;  // (output fix_m0 + o of the concatenation: its buffer's virtual base)
            for (int i = 1; i < B.nbuf; i++)
                if (fix_m0 + o >= (uint32_t)i * B.out_each) ob = (float2 *)(uintptr_t)B.vout[i];
            ob[fix_m0 + o] = make_float2((float)ar, (float)ai);
        }
        fix_cnt = 0;
    };
    auto tasks_back = [&]() {  // behind the barrier that follows tasks_front
        if (wave >= kWaves / 2 || fix_cnt == 0) return;
        if constexpr ((EXP & 4096) != 0) {
            tasks_back_cold(out, G, B);
        } else {
            HZ_COLD_ARGS;
            tasks_back_cold(out, G, B);
        }
    };
    stamp(9);
    auto put = [&](int q, v4i v) {
        if constexpr (FMT == HZSDR_FMT_U8) v ^= (int)0x80808080;  // b - 128 as int8
        *reinterpret_cast<v4i *>(slot + TS * (q / PPT) + 16 * (q % PPT)) = v;
    };
    // The compiler's s_waitcnt pass keeps a load "pending" on every path it cannot prove a wait on (a conditional
    // consumer is enough), right around the pass loop; with stores in flight as well -- loads and stores return out of
    // order -- each later write of such a register becomes s_waitcnt vmcnt(0).  Two of those sat in the pass loop:
    // at its head (the previous pass's stores drained before the queue was read) and in front of the matrix loop
    // (the prefetch just issued waited for, instead of flying under the matrix loop).  An explicit vmcnt(0) where
    // nothing is in flight anyway -- after a landing -- tells the pass so.
    auto vm_clear = [&]() {
        if constexpr ((EXP & 128) == 0) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), the other counters untouched
    };
    // (flipped: the u8 sign flip has been applied to the registers already -- among the MFMAs of the matrix loop the
    // bytes were in flight under, EXP & 262144)
    auto land = [&](v4i(&x)[KU], bool flipped = false) {
        // piece l + 64 u is piece l % PPT of tile l / PPT + (64 / PPT) u: one address and constants
        uint8_t *lp = slot + TS * (l / PPT) + 16 * (l % PPT);
        // (1 << 25: the rare un-flipped landing -- a wave's first pass, a pass behind an inactive one -- as a BRANCH.  The
        // condition is uniform, and the compiler turned `if (!flipped) x ^= c` into 40 v_xor + 40 v_cndmask executed
        // by EVERY landing: 80 of a pass's ~490 vector instructions, each of which costs the SIMD partner's matrix loop
        // ~2.5 cycles -- found in the ISA in round 6.  The empty asm keeps the block from being if-converted again.)
        if constexpr (FMT == HZSDR_FMT_U8 && STRAIGHT && (EXP & 262144) != 0 && (EXP & (1 << 25)) != 0 && (EXP & 32768) == 0) {
            if (!flipped) {
#pragma unroll
                for (int u = 0; u < KU; u++) {
                    x[u] ^= (int)0x80808080;
                    asm volatile("" : "+v"(x[u]));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < KU; u++)
            if (kWhole || l + u * 64 < pieces) {
                if constexpr (FMT == HZSDR_FMT_U8 && (EXP & 32768) == 0) {
                    if constexpr (STRAIGHT && (EXP & 262144) != 0 && (EXP & (1 << 25)) != 0) {
                        // (flipped above, in one uniform branch)
                    } else if constexpr (STRAIGHT && (EXP & 262144) != 0) {
                        if (!flipped) x[u] ^= (int)0x80808080;  // (uniform: a wave's first pass, a pass behind an inactive one)
                    } else {
                        x[u] ^= (int)0x80808080;  // b - 128 as int8
                    }
                }
                *reinterpret_cast<v4i *>(lp + u * (64 / PPT) * TS) = x[u];
            }
        if constexpr (FMT == HZSDR_FMT_U8 && (EXP & 32768) != 0) {
            // the sign flip by the LDS itself: two ds_xor_b64 per piece behind its write (a wave's LDS operations
            // execute in order) instead of four v_xor_b32 -- no vector instruction, and the LDS is half idle
#pragma unroll
            for (int u = 0; u < KU; u++)
                if (kWhole || l + u * 64 < pieces) {
                    unsigned long long *q8 = reinterpret_cast<unsigned long long *>(lp + u * (64 / PPT) * TS);
                    (void)__hip_atomic_fetch_xor(q8, 0x8080808080808080ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    (void)__hip_atomic_fetch_xor(q8 + 1, 0x8080808080808080ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto land_edge_cold = [&](uint32_t pass, const uint8_t *src, const uint8_t *rhist, int64_t n_bytes, const Geom &G, const Plan &L, const Batch &B) {
        const int64_t p0 = 2 * ((int64_t)D * ((int64_t)pass * kPassOut) - G.w0);
        const int pieces = (int)(image_bytes(D, G.ks) / 16);
#pragma unroll 1
        for (int q = l; q < pieces; q += 64) {
            const int64_t p = p0 + (int64_t)q * 16;
            v4i v{0, 0, 0, 0};
            if (p >= 0 && p + 16 <= n_bytes) {
                // (a piece never straddles two buffers: their length is a multiple of 16 bytes)
                uint64_t sb = (uint64_t)(uintptr_t)src;
                for (int i = 1; i < B.nbuf; i++)
                    if (p >= (int64_t)i * (int64_t)B.nb) sb = B.vin[i];
                v = *(const v4i __attribute__((address_space(1))) *)(sb + (uint64_t)p);
            } else if (p < 0 && L.cont && p + 2 * (int64_t)G.off >= 0) {
                v = *reinterpret_cast<const v4i *>(rhist + (p + 2 * (int64_t)G.off));  // (2 off is a multiple of 16)
            } else if (p + 16 > 0 && p < n_bytes) {  // straddles the buffer's end: byte by byte
                uint64_t w0 = 0, w1 = 0;  // (no byte array: it would live in scratch memory)
#pragma unroll 1
                for (int e = 0; e < 16; e++)
                    if (p + e >= 0 && p + e < n_bytes) {
                        const uint64_t bb = src[p + e];
                        if (e < 8) w0 |= bb << (8 * e);
                        else w1 |= bb << (8 * (e - 8));
                    }
                v = v4i{(int)(uint32_t)w0, (int)(uint32_t)(w0 >> 32), (int)(uint32_t)w1, (int)(uint32_t)(w1 >> 32)};
            }
            put(q, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto land_edge = [&](uint32_t pass) {  // (the call's first and last passes)
        if constexpr ((EXP & 4096) != 0) {
            land_edge_cold(pass, src, rhist, n_bytes, G, L, B);
        } else {
            HZ_COLD_ARGS;
            land_edge_cold(pass, (const uint8_t *)in, rhist, 2 * (int64_t)n_in, G, L, B);
        }
    };

    // A fragments: lane n is row (a = n >> 3, c = (n >> 2) & 1, part = (n >> 1) & 1, pl = n & 1) = output
    // i = 4 c + a of the tile -- the accumulator then holds FOUR CONSECUTIVE outputs per lane (row =
    // 8 (q >> 2) + 4 (l >> 5) + (q & 3)): entry (D/8) i - h - 2 s + e0, 64 bytes per entry and fragment row
    const int i_row = 4 * ((n >> 2) & 1) + (n >> 3);
    const int a_base = 64 * ((D / 8) * i_row - h + G.e0) + 16 * (n & 3);
    const int f_stride = G.ne * 64;
    const int groups = NG > 0 ? NG : G.ks / GS;  // (NG > 0: the host launches this instantiation for ks = NG GS only)

    bool first_seg = true, first_stamp = true;
    const double k3 = __hiloint2double((1023 - G.shift) << 20, 0), k2 = k3 * 256.0, k1 = k3 * 65536.0, k0 = k3 * 16777216.0;  // 2^-S 256^d
    // a lane holds outputs 4 h + a (a = 0 .. 3) of tile n of each block: planes 2 f + pl at q = 4 a + 2 part + pl.
    // The planes meet in float64 with power-of-two weights (exact; the constant part of a u8 stream and the scale
    // 2^-S folded in), one rounding to float32.
    auto planes = [&](const v16i(&acc)[2][NB], float2(&y)[NB][4], double dcr, double dci) {
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int aa = 0; aa < 4; aa++) {
                float c2[2];
#pragma unroll
                for (int pt = 0; pt < 2; pt++) {
                    const int q = 4 * aa + 2 * pt;
                    if constexpr ((EXP & (1 << 27)) != 0) {  // (tools/mfma_fir2.hip CLOCKS: what the float64 instructions cost -- NOT exact, timing only)
                        const int hi = (int)(((unsigned)acc[0][b][q] << 8) + (unsigned)acc[0][b][q + 1]);
                        float v = __fmaf_rn((float)acc[1][b][q + 1], (float)k3, (float)(pt ? dci : dcr));
                        v = __fmaf_rn((float)acc[1][b][q], (float)k2, v);
                        c2[pt] = __fmaf_rn((float)hi, (float)k1, v);
                        continue;
                    }
                    double v = __fma_rn((double)acc[1][b][q + 1], k3, pt ? dci : dcr);
                    v = __fma_rn((double)acc[1][b][q], k2, v);
                    if constexpr ((EXP & 16384) != 0) {
                        // planes 0 and 1 meet in int32 (one v_lshl_add_u32 for a conversion and an fma): the host chose
                        // the taps' scale so that 256 |sum_0| + |sum_1| < 2^31 for every input (digit_shift); every
                        // partial sum of the chain is exact in float64 either way, so the result is the same bits
                        const int hi = (int)(((unsigned)acc[0][b][q] << 8) + (unsigned)acc[0][b][q + 1]);
                        v = __fma_rn((double)hi, k1, v);
                    } else {
                        v = __fma_rn((double)acc[0][b][q + 1], k1, v);
                        v = __fma_rn((double)acc[0][b][q], k0, v);
                    }
                    c2[pt] = (float)v;
                }
                y[b][aa] = make_float2(c2[0], c2[1]);
            }
    };
    // the elementwise program over a lane's outputs m = mb + 256 b + a of a pass in the run with phase line
    // (phi_r, dphi) / clock-table entry seg: equally spaced in two directions inside one exactly-linear clock run
    auto program = [&](float2(&y)[NB][4], uint32_t mb, uint64_t phi_r, uint64_t dphi, int seg, const float4 *wtab) {
        // The other stages of a program -- Gain, Multiply, the Shifts of a program with several -- through the cold
        // arguments (HZ_COLD_ARGS above): the stage's kind and parameters are read where they are used.  (Read the
        // ordinary way they were five dependent scalar loads per pass in front of the marked Shift.)
        auto other_stage = [&](int oi, int seg, const EwProgram &P) {
            const EwOp &o = P.op[oi];
            if (o.kind == EW_SCALE) {
#pragma unroll
                for (int b = 0; b < NB; b++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        // (scalar products, kept from being paired: a packed float32 instruction waits for the SIMD
                        // partner's matrix loop -- EXP's 1 << 22)
                        float gx = __fmul_rn(y[b][q].x, o.a), gy = __fmul_rn(y[b][q].y, o.a);
                        asm volatile("" : "+v"(gx), "+v"(gy));
                        y[b][q] = make_float2(gx, gy);
                    }
            } else if (o.kind == EW_ROTATE) {
#pragma unroll
                for (int b = 0; b < NB; b++)
#pragma unroll
                    for (int q = 0; q < 4; q++) y[b][q] = go_cmul(y[b][q], make_float2(o.a, o.b));
            } else {
                // (programs with several Shift stages: out of line, other_shift_stage above)
                const double step = P.segs.step[seg];
                const int64_t dj = (int64_t)((uint64_t)D * mb) - (int64_t)P.segs.first[seg];
                const double ts0 = __fma_rn((double)dj, step, P.segs.t0[seg]);
                float2 t[4 * NB];
#pragma unroll
                for (int b = 0; b < NB; b++)
#pragma unroll
                    for (int q = 0; q < 4; q++) t[4 * b + q] = y[b][q];
                other_shift_stage<D, NB>(t, o.tau_shift, step, ts0, l);
#pragma unroll
                for (int b = 0; b < NB; b++)
#pragma unroll
                    for (int q = 0; q < 4; q++) y[b][q] = t[4 * b + q];
            }
        };
        // (the program's ONE marked Shift stage: the mixer)
        auto shift_stage = [&]() {
                // The stage's phase is a 64-bit accumulator in turns (exact increments, no float64).  ONE Sincos per
                // lane, for its first output; the other seven are that factor turned on by the group's step factors
                // exp(2 pi i (256 b + a) D dphi) -- double-float constants in LDS, two fma pairs and an add per
                // component (0.4 ulp rms against the 0.3 of a Sincos each, a third of the instructions: every vector
                // instruction here runs beside the SIMD partner's matrix loop and costs both waves).
                const uint64_t ph0 = phi_r + (uint64_t)D * mb * dphi;
                float s0, c0;
                // (tools/mfma_fir2.hip BISECT: 1 << 20 = no Sincos, 1 << 21 = the step factors not read from LDS,
                // 1 << 22 = the products in scalar float32 instead of packed)
                if constexpr ((EXP & (1 << 20)) != 0) s0 = __uint_as_float((uint32_t)(ph0 >> 40)), c0 = 1.0f;
                else sincos_turns32((uint32_t)(ph0 >> 32), s0, c0);
                [[maybe_unused]] float4 wpin[4 * NB];
                [[maybe_unused]] float4 chk_w = make_float4(0.f, 0.f, 0.f, 0.f);
                [[maybe_unused]] float chk_cs = 0.f, chk_sn = 0.f;
                if constexpr ((EXP & 512) != 0) {
#pragma unroll
                    for (int i = 1; i < 4 * NB; i++) wpin[i] = wtab[i];
                    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 7" ::: "memory");
#pragma unroll
                    for (int i = 1; i < 4 * NB; i++) asm volatile("" : "+v"(wpin[i].x), "+v"(wpin[i].y), "+v"(wpin[i].z), "+v"(wpin[i].w));
                }
                if constexpr ((EXP & 8192) != 0) {
                    const v2f cs0{c0, s0};
#pragma unroll
                    for (int b = 0; b < NB; b++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            v2f f = cs0;
                            if (b + q > 0) {
                                typedef float v4f_ __attribute__((ext_vector_type(4)));
                                v4f_ w;
                                if constexpr ((EXP & (1 << 21)) != 0) w = v4f_{c0, s0, 1e-8f * (float)(4 * b + q), 1e-9f};
                                else w = *reinterpret_cast<const v4f_ *>(wtab + 4 * b + q);  // (cos hi, sin hi, cos lo, sin lo)
                                if constexpr ((EXP & (1 << 22)) != 0) {
                                    f.x = __fmaf_rn(c0, w.x, __fmaf_rn(-s0, w.y, __fmaf_rn(c0, w.z, -(s0 * w.w))));
                                    f.y = __fmaf_rn(c0, w.y, __fmaf_rn(s0, w.x, __fmaf_rn(c0, w.w, s0 * w.z)));
                                } else
                                f = pk_turn(cs0, w.xy, w.zw);
                            }
                            if constexpr ((EXP & (1 << 22)) != 0) {
                                float rx = __fmaf_rn(y[b][q].x, f.x, -(y[b][q].y * f.y)), ry = __fmaf_rn(y[b][q].x, f.y, y[b][q].y * f.x);
                                asm volatile("" : "+v"(rx), "+v"(ry));  // (kept scalar: no re-vectorisation)
                                y[b][q] = make_float2(rx, ry);
                            } else {
                            const v2f r = pk_cmul(v2f{y[b][q].x, y[b][q].y}, f);
                            y[b][q] = make_float2(r.x, r.y);
                            }
                        }
                } else
#pragma unroll
                for (int b = 0; b < NB; b++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float cs = c0, sn = s0;
                        if (b + q > 0) {
                            float4 w;
                            if constexpr ((EXP & 512) != 0) w = wpin[4 * b + q];
                            else w = wtab[4 * b + q];  // (cos hi, sin hi, cos lo, sin lo)
                            // (the small terms first, the large ones chained on: four instructions per component)
                            cs = __fmaf_rn(c0, w.x, __fmaf_rn(-s0, w.y, __fmaf_rn(c0, w.z, -(s0 * w.w))));
                            sn = __fmaf_rn(c0, w.y, __fmaf_rn(s0, w.x, __fmaf_rn(c0, w.w, s0 * w.z)));
                            if constexpr ((EXP & 1024) != 0) {
                                if (b == 0 && q == 1) chk_w = w, chk_cs = cs, chk_sn = sn;
                            }
                        }
                        y[b][q] = make_float2(__fmaf_rn(y[b][q].x, cs, -(y[b][q].y * sn)), __fmaf_rn(y[b][q].x, sn, y[b][q].y * cs));
                    }
                if constexpr ((EXP & 1024) != 0) {
                    float4 w2 = wtab[1];
                    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 7" : "+v"(w2.x), "+v"(w2.y), "+v"(w2.z), "+v"(w2.w));
                    float c0b = c0, s0b = s0;
                    asm volatile("" : "+v"(c0b), "+v"(s0b));
                    float csl = __fmaf_rn(c0b, w2.z, -(s0b * w2.w)), snl = __fmaf_rn(c0b, w2.w, s0b * w2.z);
                    asm volatile("" : "+v"(csl), "+v"(snl));  // (no packed forms across these)
                    float cs2 = __fmaf_rn(-s0b, w2.y, csl), sn2 = __fmaf_rn(s0b, w2.x, snl);
                    asm volatile("" : "+v"(cs2), "+v"(sn2));
                    cs2 = __fmaf_rn(c0b, w2.x, cs2), sn2 = __fmaf_rn(c0b, w2.y, sn2);
                    if (__float_as_uint(cs2) != __float_as_uint(chk_cs) || __float_as_uint(sn2) != __float_as_uint(chk_sn)) {
                        const unsigned long long slot_r = atomicAdd(&stamps[2], 1ull);
                        if (slot_r < 4000) {
                            unsigned long long *rec = stamps + 8 + 8 * slot_r;
                            auto pk = [](float a, float bq) { return (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(bq) << 32); };
                            rec[0] = (unsigned long long)(mb / kPassOut) | ((unsigned long long)l << 32) | ((unsigned long long)wave << 40);
                            rec[1] = (unsigned long long)(unsigned)wb | ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11)) << 32);
                            rec[2] = pk(c0, s0);
                            rec[3] = pk(chk_cs, chk_sn);
                            rec[4] = pk(cs2, sn2);
                            rec[5] = pk(chk_w.x, chk_w.y);
                            rec[6] = pk(chk_w.z, chk_w.w);
                            rec[7] = pk(w2.z, w2.w);
                        }
                    }
                }
        };
        auto other_stage_at = [&](int oi) {
                if constexpr ((EXP & 4096) != 0) {
                    other_stage(oi, seg, P);
                } else {
                    HZ_COLD_ARGS;
                    other_stage(oi, seg, P);
                }
                // (a multi-Shift stage's outputs come back through scratch memory: waited for HERE -- left pending they
                // become a vmcnt(0) at the pass loop's top, behind the next pass's prefetch)
                if constexpr ((EXP & 131072) != 0) __builtin_amdgcn_s_waitcnt(0x0F70);
        };
        const int n_ops = (EXP & 4) ? 0 : L.n_ops;
        // (1 << 26: the common program -- ONE stage, the marked Shift -- outside the loop over the stages: rolled, the loop
        // carries a lane's sixteen outputs from trip to trip through eight v_mov_b64 that every pass executed)
        if constexpr ((EXP & (1 << 26)) != 0) {
            if (__builtin_expect(n_ops == 1 && L.shift_op == 0, 1)) {
                shift_stage();
                return;
            }
        }
#pragma unroll 1
        for (int oi = 0; oi < n_ops; oi++) {  // uniform
            bool other = L.shift_op != oi;
            if constexpr ((EXP & 131072) != 0) other = __builtin_expect(other, 0);
            if (other) other_stage_at(oi);
            else shift_stage();
        }
    };
    auto store_block = [&](uint64_t outb, float2(&y)[4], uint32_t mo, uint32_t lo, uint32_t hi) {  // (outb: the pass's buffer, virtual base)
        if constexpr ((EXP & 8) != 0) return;
        typedef float v4f __attribute__((ext_vector_type(4)));
        typedef float v2f_ __attribute__((ext_vector_type(2)));
        if (mo >= lo && mo + 4 <= hi) {
            v4f __attribute__((address_space(1))) *o4 = (v4f __attribute__((address_space(1))) *)(outb + 8 * (uint64_t)mo);
            __builtin_nontemporal_store(v4f{y[0].x, y[0].y, y[1].x, y[1].y}, o4);
            __builtin_nontemporal_store(v4f{y[2].x, y[2].y, y[3].x, y[3].y}, o4 + 1);
        } else {
            v2f_ __attribute__((address_space(1))) *o2 = (v2f_ __attribute__((address_space(1))) *)(outb + 8 * (uint64_t)mo);
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (mo + q >= lo && mo + q < hi) o2[q] = v2f_{y[q].x, y[q].y};
        }
    };
    // The runs that matter to this workgroup: those with passes in [pb0, pb1).  Two tables fit the LDS: run A of a
    // group and the run B behind it are ONE queue (A's passes, B's passes); the next group slides on by one run (B
    // becomes A, its table stays) behind a barrier.
    auto has_passes = [&](int rr) { return rr < L.n && max(pb0, (uint32_t)L.pass_first[rr]) < min(pb1, (uint32_t)L.pass_end[rr]); };
    // EXP & 524288 (straight-line form): a pass's STORES wait for the wave's next matrix loop.  Measured
    // (tools/epi_cost.hip): beside a SIMD partner that issues MFMAs back to back a wave's vector instructions, LDS
    // reads and writes and global loads go through as if the partner were idle -- but a global STORE does not issue
    // until the partner's stream of MFMAs pauses (its data leaves the register file by a path the MFMAs' operand reads
    // hold).  An epilogue that ends in stores therefore ends when the partner's loop ends, whatever its length, and the
    // wave reaches its next loop -- queue, prefetch, first operands -- only then: a bubble of ~1 us in the matrix pipe
    // at every change of waves.  Issued among the wave's OWN MFMAs (the partner is in its epilogue by then) the stores
    // cost a few cycles each.  The last pass of a wave stores behind its loop as before.
    // 1 << 23: the same stores, but in ONE place -- at the next matrix loop's TOP, behind the queue, the prefetch's issue
    // and the first operand reads, in front of the first MFMA: while the partner's loop runs the wave gets everything
    // else of its next pass ready and waits at the stores; when the partner's stream of MFMAs ends they issue and the
    // loop starts at once (without this the wave reaches its loop's top only behind the stores: ~0.5 us of idle matrix
    // pipe per pass).  No branch inside the straight-line loop (524288's two cost more than they gave).
    constexpr bool kDeferTop = STRAIGHT && (EXP & (1 << 23)) != 0;
    constexpr bool kDefer = STRAIGHT && ((EXP & 524288) != 0 || kDeferTop);
    [[maybe_unused]] float2 yd[NB][4];
    [[maybe_unused]] uint64_t out_d = 0;
    [[maybe_unused]] uint32_t mb_d = 0, lo_d = 0, hi_d = 0;
    [[maybe_unused]] bool have_d = false;  // uniform
    auto flush_deferred = [&]() {
        if constexpr (kDefer) {
            if (have_d) {
#pragma unroll
                for (int b = 0; b < NB; b++) store_block(out_d, yd[b], mb_d + (uint32_t)(32 * kT) * b, lo_d, hi_d);
                have_d = false;
            }
        }
    };
    int ra = 0;
    while (ra < L.n && !has_passes(ra)) ra++;
    Run ru = run0, rv = run0;
    bool a_done = false;
    int slot_a = 0;  // the half of the table area that holds run A's table
#pragma unroll 1
    while (ra < L.n) {
        const int rb = ra + 1;
        const bool b_pass = has_passes(rb), b_here = b_pass;  // uniform
        uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
        if (!a_done && has_passes(ra)) a0 = max(pb0, (uint32_t)L.pass_first[ra]), a1 = min(pb1, (uint32_t)L.pass_end[ra]);
        if (b_pass) b0 = max(pb0, (uint32_t)L.pass_first[rb]), b1 = min(pb1, (uint32_t)L.pass_end[rb]);
        // items [0, seg_b): pass a0 + q of run A, then pass b0 + q of run B
        const uint32_t n_a = a1 - a0, seg_b = n_a + (b1 - b0);
        auto pass_of = [&](uint32_t q) { return q < n_a ? a0 + q : b0 + (q - n_a); };
        auto prefetchable = [&](uint32_t q) { return q < seg_b && inside(pass_of(q)); };
        if (!first_seg) __syncthreads();  // every wave is done with the table that goes and with the queue
        const bool first_group = first_seg;  // (the workgroup's first group: the small tasks ride on it)
        // Waves 0-3 (one per SIMD) take the queue's first items and put their bytes in flight at once;
        // waves 4-7 start behind the barrier: the first burst is half as large, and the two waves of a SIMD
        // start out of phase.
        constexpr int kEarly = kWaves / 2;
        auto grab = [&]() -> uint32_t {
            unsigned p = 0;
            if (l == 0) p = atomicAdd(ctr, 1u);
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
        };
        uint32_t cur = seg_b;
        bool in0 = false;
        constexpr int kTU = 4;  // table pieces per thread (32 KB at most: the host)
        if (wave < kEarly) {
            cur = min((uint32_t)wave, seg_b);
            in0 = prefetchable(cur);  // uniform
            // (pre_issued: the kernel's first lines have requested pass pb0 + wave -- this one, in the first group)
            if (in0 && !(pre_issued && ra == 0 && !a_done && pass_of(cur) == pb0 + (uint32_t)wave)) issue(x, pass_of(cur));
            pre_issued = false;
        }
        if (first_stamp) stamp(13);
        // (everything the queue reads from the kernel arguments by run, at once)
        // (run A of a later group is the previous group's run B: the lines stay in registers, one at a time is read)
        if (!first_seg) ru = rv;
        else if (ra > 0) ru = L.run[ra];  // (one line)
        if (b_here) rv = L.run[rb];
        const int off_a = slot_a ? (int)tab_lds : 0, off_b = slot_a ? 0 : (int)tab_lds;
        {
            const int tp = (int)(table_bytes(G.ne) + 15) / 16;
            const v4i *tg = (const v4i *)ru.tab, *th = (const v4i *)rv.tab;
            // (one table after the other: both in flight at once, beside the first pass's bytes, spilled registers)
            v4i tq[kTU];
            if (first_seg) {
#pragma unroll
                for (int u = 0; u < kTU; u++) tq[u] = tid + u * kThreads < tp ? tg[tid + u * kThreads] : v4i{0, 0, 0, 0};
                if (first_stamp) stamp(14);
#pragma unroll
                for (int u = 0; u < kTU; u++)
                    if (tid + u * kThreads < tp) *reinterpret_cast<v4i *>(tabp + off_a + 16 * (size_t)(tid + u * kThreads)) = tq[u];
            }
            if (b_here) {
#pragma unroll
                for (int u = 0; u < kTU; u++) tq[u] = tid + u * kThreads < tp ? th[tid + u * kThreads] : v4i{0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < kTU; u++)
                    if (tid + u * kThreads < tp) *reinterpret_cast<v4i *>(tabp + off_b + 16 * (size_t)(tid + u * kThreads)) = tq[u];
            }
        }
        first_seg = false;
        if (first_stamp) stamp(10);
        if (tid == 0) *ctr = kEarly;
        __syncthreads();
        if (first_stamp) stamp(11);
        if constexpr ((EXP & 32) != 0) {
            if (first_group) __builtin_amdgcn_s_setprio(1);
        }
        if (wave >= kEarly) {
            cur = grab();
            in0 = prefetchable(cur);
            if (in0) issue(x, pass_of(cur));
            if constexpr ((EXP & 16) != 0) __builtin_amdgcn_s_sleep(60);  // (experiment: a longer offset)
        }
        if (cur < seg_b) {
            if (in0) land(x);
            else land_edge(pass_of(cur));
        }
        if (first_stamp) stamp(12);
        first_stamp = false;
        vm_clear();
#pragma unroll 1
        while (cur < seg_b) {
            stamp_pass++;
            stamp(16);
            const uint32_t nxt = grab();
            const bool has_next = nxt < seg_b, in_next = prefetchable(nxt);  // uniform
            stamp(23);
            if (in_next) issue(x, pass_of(nxt));
            stamp(1);
            const bool sel = cur >= n_a;  // uniform: the pass belongs to run B
            const uint32_t m_lo = sel ? rv.m_lo : ru.m_lo, m_hi = sel ? rv.m_hi : ru.m_hi;
            const uint64_t dphi = sel ? rv.dphi : ru.dphi, phi_r = sel ? rv.phi : ru.phi;
            const int tab_off = sel ? off_b : off_a;
            const uint32_t m_start = pass_of(cur) * (uint32_t)kPassOut;
            const uint32_t v_lo = max(m_lo, m_start), v_hi = min(m_hi, m_start + (uint32_t)kPassOut);
            const bool active = v_lo < v_hi;  // uniform
            // (EXP & 65536, straight-line form: the first step writes the accumulators -- constant 0 as C -- and an
            // inactive pass never reads them)
            constexpr bool kFirstC0 = STRAIGHT && (EXP & 65536) != 0;
            v16i acc[2][NB];
            if constexpr (!kFirstC0) {
#pragma unroll
                for (int f = 0; f < 2; f++)
#pragma unroll
                    for (int b = 0; b < NB; b++)
#pragma unroll
                        for (int q = 0; q < 16; q++) acc[f][b][q] = 0;
            }
            // (the accumulators as opaque registers: the straight-line form then starts them like every other
            // step -- v_mfma acc, a, b, acc -- instead of with the constant 0 as the C operand.  EXP & 2048 lets the
            // compiler use the constant: 64 clears per pass less, exact sums all the same (tools/mm2_glitch.hip; what
            // round 3 saw "lose terms" with the constant was the mixer's packed instruction, hz_firmm.h) -- and 1.8 us
            // per call SLOWER, measured A/B on one box: the first step's destinations then overlap its sources)
            if constexpr (STRAIGHT && (EXP & 2048) == 0 && !kFirstC0) {
#pragma unroll
                for (int f = 0; f < 2; f++)
#pragma unroll
                    for (int b = 0; b < NB; b++) asm volatile("" : "+v"(acc[f][b]));
            }
            if constexpr (kDefer) {
                if (!(active && (EXP & 2) == 0)) flush_deferred();
            }
            if (active && (EXP & 2) == 0) {
                // step s = GS g + j of the window: the A entries 2 s below the lane's first, B piece 2 j + h of
                // tile n + g -- constants off two per-lane addresses (NG > 0) or off two running ones
                constexpr int KS = NG * GS;
                // (NG > 0: the lane's LAST entry, and the compiler kept from folding the base back to the first
                // one: DS offsets are unsigned, negative ones cost an address add per load)
                int a_off = a_base - (STRAIGHT ? 128 * KS : 0) + tab_off;
                if constexpr (STRAIGHT) asm volatile("" : "+v"(a_off));
                const uint8_t *ap = tabp + a_off;
                const uint8_t *bp = slot + TS * n + 16 * h;
                int s_done = 0, g_done = 0;  // (compile-time values when the groups are unrolled)
                auto load_a = [&](v4i(&a)[2], int s) {
#pragma unroll
                    for (int f = 0; f < 2; f++) {
                        if constexpr (STRAIGHT) a[f] = *reinterpret_cast<const v4i *>(ap + f * (2 * (KS + 4) + (D / 8) * (kT - 1) + 1) * 64 + 128 * (KS - s));
                        else a[f] = *reinterpret_cast<const v4i *>(ap + f * f_stride - 128 * (s - s_done));
                    }
                };
                auto load_b = [&](v4i(&b)[NB], int g, int j) {
#pragma unroll
                    for (int q = 0; q < NB; q++) b[q] = *reinterpret_cast<const v4i *>(bp + TS * (g - g_done) + 32 * j + q * 32 * TS);
                };
                // operands two steps ahead of their MFMAs, in rings of three (straight-line code) or four
                constexpr int RG = STRAIGHT ? 3 : 4;
                v4i a[RG][2], b[RG][NB];
                load_a(a[0], 0);
                load_b(b[0], 0, 0);
                load_a(a[1], 1);
                load_b(b[1], 0, 1);
                if constexpr (kDeferTop) {
                    __builtin_amdgcn_sched_barrier(0);  // (behind the operand reads' issue, in front of the first MFMA)
                    flush_deferred();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // one step; SC: the step as a compile-time value (straight-line form), or -1
                auto step = [&](auto sc, int g, int j, int jx = 0) {
                    constexpr int SC = decltype(sc)::value;
                    const int s = SC >= 0 ? SC : GS * g + j;
                    const int sx = SC >= 0 ? SC : jx;  // (ring positions: the step mod 3, or the step of the trip mod 4)
                    load_a(a[(sx + 2) % RG], s + 2);
                    if (j + 2 < GS) load_b(b[(sx + 2) % RG], g, j + 2);
                    else load_b(b[(sx + 2) % RG], g + 1, j + 2 - GS);
#pragma unroll
                    for (int f = 0; f < 2; f++)
#pragma unroll
                        for (int qq = 0; qq < NB; qq++) {
                            // (1 << 28: the step's MFMAs in "snake" order -- (A0,B0) (A0,B1) (A1,B1) (A1,B0): one operand
                            // changes between neighbours -- an experiment on the operands' toggling, tools/mfma_fir2.hip AB8)
                            const int q = ((EXP & (1 << 28)) != 0 && f == 1) ? NB - 1 - qq : qq;
                            if constexpr (kFirstC0 && SC == 0)
                                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(acc[f][q]) : "v"(a[sx % RG][f]), "v"(b[sx % RG][q]));
                            else
                                acc[f][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[sx % RG][f], b[sx % RG][q], acc[f][q], 0, 0, 0);
                        }
                    // (EXP & 524288: the PREVIOUS pass's stores, one column block each in two early steps)
                    if constexpr (kDefer && !kDeferTop && SC >= 0) {
                        if constexpr (SC >= 3 && (SC - 3) % 4 == 0 && (SC - 3) / 4 < NB) {
                            constexpr int bq = (SC - 3) / 4;
                            if (have_d) store_block(out_d, yd[bq], mb_d + (uint32_t)(32 * kT) * bq, lo_d, hi_d);
                        }
                        if constexpr (SC == 3 + 4 * (NB - 1)) have_d = false;
                    }
                    // (EXP & 262144, u8 sources: the NEXT pass's bytes -- in flight since the loop's first lines, here long
                    // since -- get their sign flip in the shadow of this wave's own MFMAs, where a vector instruction
                    // costs 2-3 cycles; left to the landing they are 40 instructions of the epilogue, which runs
                    // beside the SIMD partner's loop at one instruction per MFMA slot.  Registers without a load in
                    // flight hold stale values that nobody lands.)
                    if constexpr (STRAIGHT && FMT == HZSDR_FMT_U8 && (EXP & 262144) != 0 && SC >= 0) {
                        constexpr int kFlipFirst = KS - 2 - (KU + 1) / 2;  // two registers per step, done two steps before the end
                        if constexpr (SC == kFlipFirst) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                        if constexpr (SC >= kFlipFirst && 2 * (SC - kFlipFirst) < KU) {
#pragma unroll
                            for (int u = 2 * (SC - kFlipFirst); u < KU && u < 2 * (SC - kFlipFirst) + 2; u++) {
                                x[u] ^= (int)0x80808080;
                                asm volatile("" : "+v"(x[u]));  // (here, not at the landing)
                            }
                        }
                    }
                    // the step's loads between its MFMAs, nothing moved across steps
#pragma unroll
                    for (int q = 0; q < 2 + NB; q++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // LDS read
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NB - 2 - NB > 0 ? 2 * NB - 2 - NB : 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                static_assert(GS == 4 || GS == 8, "ring indices repeat per group");
                if constexpr (STRAIGHT) {
                    auto all = [&](auto self, auto sc) {
                        constexpr int S = decltype(sc)::value;
                        if constexpr (S < KS) {
                            step(sc, S / GS, S % GS);
                            self(self, std::integral_constant<int, S + 1>{});
                        }
                    };
                    all(all, std::integral_constant<int, 0>{});
                } else {
                    // UG groups per trip: a back edge costs a drain of the operand pipeline (the compiler cannot count
                    // outstanding loads across it)
                    constexpr int UGv = UG > 0 ? UG : 1;
                    int g = 0;
#pragma unroll 1
                    for (; g + UGv <= groups; g += UGv) {
#pragma unroll
                        for (int gq = 0; gq < UGv; gq++)
#pragma unroll
                            for (int j = 0; j < GS; j++) step(std::integral_constant<int, -1>{}, g + gq, j, GS * gq + j);
                        ap -= 128 * GS * UGv;
                        bp += TS * UGv;
                        s_done += GS * UGv;
                        g_done += UGv;
                    }
                    if constexpr (UGv > 1) {
#pragma unroll 1
                        for (; g < groups; g++) {
#pragma unroll
                            for (int j = 0; j < GS; j++) step(std::integral_constant<int, -1>{}, g, j, j);
                            ap -= 128 * GS;
                            bp += TS;
                            s_done += GS;
                            g_done++;
                        }
                    }
                }
            }
            stamp(2);
            [[maybe_unused]] int cs_e1 = 0, cs_e2 = 0, cs_l1 = 0, cs_l2 = 0;
            [[maybe_unused]] auto csum = [&](int &s1, int &s2) {
#pragma unroll
                for (int f = 0; f < 2; f++)
#pragma unroll
                    for (int b = 0; b < NB; b++) asm volatile("" : "+v"(acc[f][b]));
                s1 = 0, s2 = 0;
#pragma unroll
                for (int f = 0; f < 2; f++)
#pragma unroll
                    for (int b = 0; b < NB; b++)
#pragma unroll
                        for (int q = 0; q < 16; q++) s1 += acc[f][b][q], s2 += (1 + q + 16 * (b + NB * f)) * acc[f][b][q];
                asm volatile("" : "+v"(s1), "+v"(s2));
            };
            if constexpr ((EXP & 256) != 0) csum(cs_e1, cs_e2);
            if (in_next) land(x, active && (EXP & 2) == 0);  // (the matrix loop ran: it flipped the bytes' signs)
            else if (has_next) land_edge(pass_of(nxt));
            vm_clear();
            stamp(3);
            if constexpr ((EXP & 256) != 0) {
                // stamps[0]: 0 = store, 1 = compare; [1]: the stored checksums (int4 per pass and lane); [2]: records
                // written; [8 + 8 r ...]: record r
                csum(cs_l1, cs_l2);
                v4i *ref = reinterpret_cast<v4i *>((uintptr_t)stamps[1]);
                const size_t ci = (size_t)pass_of(cur) * 64 + l;
                if (active) {
                    if (stamps[0] == 0) {
                        ref[ci] = v4i{cs_e1, cs_e2, cs_l1, cs_l2};
                    } else {
                        const v4i r = ref[ci];
                        if (r[0] != cs_e1 || r[1] != cs_e2 || r[2] != cs_l1 || r[3] != cs_l2) {
                            const unsigned long long slot_r = atomicAdd(&stamps[2], 1ull);
                            if (slot_r < 4000) {
                                unsigned long long *rec = stamps + 8 + 8 * slot_r;
                                rec[0] = (unsigned long long)pass_of(cur) | ((unsigned long long)l << 32) | ((unsigned long long)wave << 40);
                                rec[1] = (unsigned long long)(unsigned)wb | ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11)) << 32);
                                rec[2] = (unsigned long long)(unsigned)cs_e1 | ((unsigned long long)(unsigned)cs_e2 << 32);
                                rec[3] = (unsigned long long)(unsigned)cs_l1 | ((unsigned long long)(unsigned)cs_l2 << 32);
                                rec[4] = (unsigned long long)(unsigned)r[0] | ((unsigned long long)(unsigned)r[1] << 32);
                                rec[5] = (unsigned long long)(unsigned)r[2] | ((unsigned long long)(unsigned)r[3] << 32);
                                rec[6] = __builtin_amdgcn_s_memrealtime();
                                rec[7] = (unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(20 | (31 << 11));
                            }
                        }
                    }
                }
            }
            if (active) {
                const double *dc = reinterpret_cast<const double *>(tabp + tab_off + (size_t)G.ne * 128);
                float2 y[NB][4];
                planes(acc, y, dc[0] * k3, dc[1] * k3);
                if constexpr ((EXP & 64) != 0) {  // (the stamp behind the combination, not in the middle of it)
#pragma unroll
                    for (int b = 0; b < NB; b++)
#pragma unroll
                        for (int q = 0; q < 4; q++) asm volatile("" : "+v"(y[b][q].x), "+v"(y[b][q].y));
                }
                stamp(4);
                const uint32_t mb = m_start + (uint32_t)n * kT + 4 * h;
                {
                    // (the run's step factors: behind the constant term at the table's end, hz_firmm2_plan.h)
                    program(y, mb, phi_r, dphi, sel ? rv.seg : ru.seg, reinterpret_cast<const float4 *>(dc + 2));
                    stamp(5);
                    const uint64_t outp = vout_of(buf_of(pass_of(cur)));
                    if constexpr (kDefer) {
                        // (kept for the wave's next matrix loop; a wave without one -- no next pass in this group --
                        // stores now)
#pragma unroll
                        for (int b = 0; b < NB; b++)
#pragma unroll
                            for (int q = 0; q < 4; q++) yd[b][q] = y[b][q];
                        out_d = outp, mb_d = mb, lo_d = v_lo, hi_d = v_hi, have_d = true;
                        if (!has_next) flush_deferred();
                    } else {
#pragma unroll
                        for (int b = 0; b < NB; b++) store_block(outp, y[b], mb + (uint32_t)(32 * kT) * b, v_lo, v_hi);
                    }
                }
            }
            stamp(6);
            // No wave priorities in the library (EXP & 32 turns them on for study: the matrix pipe goes to the wave of
            // higher priority, then to the OLDER one, and ranking a wave on its first pass above one that has finished
            // a pass makes the two waves of a SIMD alternate pass by pass): the kernel is as fast without them
            // (tools/mfma_fir2.hip: 38.6 against 38.9 us).  Round 4, tasks off the head: launch by launch with an event
            // pair each (AB=1) the priorities are worth 0.4-1.8 us of the median -- and in bench.py's back-to-back
            // stream they COST 0.6 us per step (0.0392 against 0.0386, three times over on one box).  Still off.
            if constexpr ((EXP & 32) != 0) __builtin_amdgcn_s_setprio(0);
            cur = nxt;
        }
        if (!b_here) break;
        ra = rb;
        slot_a ^= 1;
        a_done = true;
        if (!has_passes(ra + 1)) break;
    }
    history_tasks();
    // The fix-up tasks, BEHIND the passes: the old waves stage round 0's window while the young ones are still in
    // the workgroup's last pass (~5 us of slack), the sums follow the barrier that ends the passes.  (Rounds 2-4
    // did round 0 in front of the first pass -- "the matrix pipes idle there anyway" -- but the workgroup's first
    // barrier then waited for the window's loads and the young waves started their passes behind the sums: a call
    // with a clock boundary took 3-4 us longer than one without, four calls of 7.5.)
#pragma unroll 1
    // (Task t of a round goes to workgroup grid - 1 - t: the history tasks sit on the first sixteen workgroups' old
    // waves, and the few tasks of a call with one clock boundary should not queue up behind them.)
    for (int round = 0; (L.grid - 1 - wb) + round * L.grid < L.n_task; round++) {  // (uniform; round > 0: more tasks than workgroups)
        if (round > 0) __syncthreads();
        tasks_front(round);
        __syncthreads();
        tasks_back();
    }
    stamp(7);
    if constexpr ((EXP & 64) != 0) {
        for (int i = l; i < 8 * kStampRows; i += 64) stamps[((size_t)wb * kWaves + wave) * (8 * kStampRows) + i] = lstamp[i];
        if (tid == 0) {  // (behind every wave's rows: shader cycles and 10 ns ticks of this workgroup's first wave)
            unsigned long long *ck = stamps + (size_t)L.grid * kWaves * (8 * kStampRows) + 2 * (size_t)wb;
            ck[0] = __builtin_amdgcn_s_memtime() - clk0;
            ck[1] = __builtin_amdgcn_s_memrealtime() - rt0;
        }
    }
}

// hz_firmm2.hip
int launch_fir(hipStream_t stream, int num_cus, int fmt, unsigned D, const void *in, float2 *out, const float2 *hist,
               float2 *new_hist, const uint8_t *rhist, uint8_t *new_rhist, const float2 *taps, size_t n, const Geom &g,
               const Plan &L, const EwProgram &P, const Fix &F, const Batch &B, int loop_form = 0);
// the history tasks alone, as a kernel of their own (n >= off): hz_firmm2.hip
int launch_history(hipStream_t stream, int fmt, const void *in, float2 *new_hist, uint8_t *new_rhist, size_t n, unsigned off, const EwProgram &P);

}  // namespace mm2
}  // namespace hz

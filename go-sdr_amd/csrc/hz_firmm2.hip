// hz_firmm2.hip -- the instantiations of the persistent-pass int8 matrix FIR (hz_firmm2.h) and their launcher.
// make NO_PK_F32=1 (csrc/Makefile), in front of every definition: no packed float32 instruction in this unit's device code -- the gfx950 hazard of
// hz_firmm.h cannot occur and the link-time rewrite (tools/fix_pk_opsel.py) is not needed for it
#if defined(HZSDR_NO_PK_F32) && defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute push(__attribute__((target("no-packed-fp32-ops"))), apply_to = function)
#endif

#include <stdlib.h>

#include <algorithm>

#include "hz_firmm2.h"

namespace hz {
namespace mm2 {

template <class K, class... A>
static int launch(K kernel, dim3 grid, size_t lds, hipStream_t stream, A... args) {
    if (lds > 48 * 1024) HZ_TRY(raise_dynamic_lds((const void *)kernel));
    hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, stream, args...);
    return hipGetLastError() == hipSuccess ? HZSDR_OK : HZSDR_ERR_HIP;
}

// What the library ships of the kernel's switches (hz_firmm2.h, EXP): round 5's instruction cuts.
// (round 6: 1 << 25, the landing's rare sign flip as a uniform branch -- it was 80 vector instructions of every pass --, and
// 1 << 26, the one-Shift program outside the loop over the stages)
constexpr int kLibExp = 8192 | 16384 | 65536 | 131072 | 262144 | (1 << 22) | (1 << 25) | (1 << 26);

template <int FMT>
static int launch_fmt(hipStream_t stream, int num_cus, unsigned D, const void *in, float2 *out, const float2 *hist,
                      float2 *new_hist, const uint8_t *rhist, uint8_t *new_rhist, const float2 *taps, size_t n, const Geom &g,
                      Plan L, const EwProgram &P, const Fix &F, const Batch &B, int rolled) {
    if (!factor_ok(D)) return HZSDR_ERR_INVALID_ARGUMENT;
    const size_t lds = lds_bytes((int)D, g.ks, g.ne, g.ntaps);
    if (D == 16) {  // (256 outputs per pass, one column block; the straight-line loop for the 1024-tap window: 9 groups of 8 steps)
        const int grid16 = std::max(1, L.grid);
        unsigned long long *no_stamps16 = nullptr;
        if (g.ks == 9 * 8 && rolled == 0)
            return launch(fir_mm2_kernel<FMT, 16, 9, kLibExp>, dim3(grid16), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, B, no_stamps16);
        return launch(fir_mm2_kernel<FMT, 16, 0, kLibExp>, dim3(grid16), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, B, no_stamps16);
    }
    // one workgroup per CU; a call with fewer passes than CUs: one pass per workgroup
    const int grid = std::max(1, L.grid);  // (the planner's: min(CUs, passes))
    (void)num_cus;
    unsigned long long *no_stamps = nullptr;
    // (the straight-line matrix loop exists for the 1024-tap window: 17 groups; `rolled`: hzsdr_chain_fir_options'
    // loop form -- 1, 2, 4 groups per trip, anything else the instantiation for any window -- for A/B measurements)
    if (g.ks == 17 * 4 && rolled == 0)
        return launch(fir_mm2_kernel<FMT, 8, 17, kLibExp>, dim3(grid), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, B, no_stamps);
    if (g.ks == 17 * 4 && rolled == 1)
        return launch(fir_mm2_kernel<FMT, 8, 17, kLibExp, 1>, dim3(grid), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, B, no_stamps);
    if (g.ks == 17 * 4 && rolled == 2)
        return launch(fir_mm2_kernel<FMT, 8, 17, kLibExp, 2>, dim3(grid), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, B, no_stamps);
    if (g.ks == 17 * 4 && rolled == 4)
        return launch(fir_mm2_kernel<FMT, 8, 17, kLibExp, 4>, dim3(grid), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, B, no_stamps);
    return launch(fir_mm2_kernel<FMT, 8, 0, kLibExp>, dim3(grid), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, B, no_stamps);
}

int launch_fir(hipStream_t stream, int num_cus, int fmt, unsigned D, const void *in, float2 *out, const float2 *hist,
               float2 *new_hist, const uint8_t *rhist, uint8_t *new_rhist, const float2 *taps, size_t n, const Geom &g,
               const Plan &L, const EwProgram &P, const Fix &F, const Batch &B, int loop_form) {
    if (fmt == HZSDR_FMT_U8)
        return launch_fmt<HZSDR_FMT_U8>(stream, num_cus, D, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, B, loop_form);
    if (fmt == HZSDR_FMT_I8)
        return launch_fmt<HZSDR_FMT_I8>(stream, num_cus, D, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, B, loop_form);
    return HZSDR_ERR_INVALID_ARGUMENT;
}

// The next call's history as a kernel of its own (a pipelined chain: hz_chain_fir.hip): exactly the history tasks of
// fir_mm2_kernel -- the last `off` samples of the call, the stages in reference order, and the raw samples -- for a
// call of at least `off` samples.  One wave per 64 samples.
template <int FMT>
__global__ __launch_bounds__(64) void history_kernel(const void *__restrict__ in, float2 *__restrict__ new_hist, uint8_t *__restrict__ new_rhist,
                                                     size_t n_in, unsigned off, EwProgram P) {
    using RWT = typename Raw<FMT>::t;
    const unsigned idx = blockIdx.x * 64u + threadIdx.x;
    const int64_t h_lo = (int64_t)n_in - (int64_t)off + (int64_t)blockIdx.x * 64;
    const NcoWin tw = mm::task_window(P, h_lo, h_lo + 63);
    if (idx < off) {
        const int64_t p = (int64_t)n_in - (int64_t)off + idx;
        new_hist[idx] = mm::ordered_sample<FMT, 2>(in, P, p, nullptr, off, tw);
        reinterpret_cast<RWT *>(new_rhist)[idx] = ((const RWT *)in)[p];
    }
}

int launch_history(hipStream_t stream, int fmt, const void *in, float2 *new_hist, uint8_t *new_rhist, size_t n, unsigned off, const EwProgram &P) {
    if (n < off || off == 0) return HZSDR_ERR_INVALID_ARGUMENT;
    const dim3 grid((off + 63) / 64), block(64);
    if (fmt == HZSDR_FMT_U8) hipLaunchKernelGGL(history_kernel<HZSDR_FMT_U8>, grid, block, 0, stream, in, new_hist, new_rhist, n, off, P);
    else if (fmt == HZSDR_FMT_I8) hipLaunchKernelGGL(history_kernel<HZSDR_FMT_I8>, grid, block, 0, stream, in, new_hist, new_rhist, n, off, P);
    else return HZSDR_ERR_INVALID_ARGUMENT;
    return hipGetLastError() == hipSuccess ? HZSDR_OK : HZSDR_ERR_HIP;
}

}  // namespace mm2
}  // namespace hz

#if defined(HZSDR_NO_PK_F32) && defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute pop
#endif

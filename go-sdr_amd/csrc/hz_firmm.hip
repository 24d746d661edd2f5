// hz_firmm.hip -- the instantiations of the int8 matrix FIR kernel (hz_firmm.h) and their launcher,
// in a translation unit of their own (they compile in parallel with hz_chain_fir.hip, which plans the
// launch: mm_plan / mm_table_for).
// make NO_PK_F32=1 (csrc/Makefile), in front of every definition: no packed float32 instruction in this unit's device code -- the gfx950 hazard of
// hz_firmm.h cannot occur and the link-time rewrite (tools/fix_pk_opsel.py) is not needed for it
#if defined(HZSDR_NO_PK_F32) && defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute push(__attribute__((target("no-packed-fp32-ops"))), apply_to = function)
#endif

#include <algorithm>

#include "hz_firmm.h"

namespace hz {
namespace mm {

template <class K, class... A>
static int launch(K kernel, dim3 grid, size_t lds, hipStream_t stream, A... args) {
    if (lds > 48 * 1024) HZ_TRY(raise_dynamic_lds((const void *)kernel));
    hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, stream, args...);
    return hipGetLastError() == hipSuccess ? HZSDR_OK : HZSDR_ERR_HIP;
}

bool factor_ok(unsigned D) { return D == 8 || D == 16 || D == 24 || D == 32 || D == 40 || D == 48 || D == 64; }

template <int FMT>
static int launch_fmt(hipStream_t stream, unsigned D, const void *in, float2 *out, const float2 *hist, float2 *new_hist,
                      const uint8_t *rhist, uint8_t *new_rhist, const float2 *taps, size_t n, const Geom &g,
                      const EwProgram &P, const Runs &R, const Fix &F) {
    const size_t lds = std::max(chunk_bytes((int)D, g.ks) + kLookAhead, (size_t)(2 * g.ntaps + D * (kFixOut - 1)) * 8);
    const dim3 grid((unsigned)R.n_wg);  // (the fix-up and history tasks ride on the chunk workgroups)
#define HZ_MM_CASE(DD)                                                                                              \
    case DD:                                                                                                        \
        return launch(fir_mm_kernel<FMT, DD>, grid, lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, P, R, \
                      F, (unsigned long long *)nullptr);
    switch (D) {
        HZ_MM_CASE(8)
        HZ_MM_CASE(16)
        HZ_MM_CASE(24)
        HZ_MM_CASE(32)
        HZ_MM_CASE(40)
        HZ_MM_CASE(48)
        HZ_MM_CASE(64)
    }
#undef HZ_MM_CASE
    return HZSDR_ERR_INVALID_ARGUMENT;
}

int launch_fir(hipStream_t stream, int fmt, unsigned D, const void *in, float2 *out, const float2 *hist, float2 *new_hist,
               const uint8_t *rhist, uint8_t *new_rhist, const float2 *taps, size_t n, const Geom &g, const EwProgram &P,
               const Runs &R, const Fix &F) {
    if (fmt == HZSDR_FMT_U8)
        return launch_fmt<HZSDR_FMT_U8>(stream, D, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, P, R, F);
    if (fmt == HZSDR_FMT_I8)
        return launch_fmt<HZSDR_FMT_I8>(stream, D, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, P, R, F);
    return HZSDR_ERR_INVALID_ARGUMENT;
}

}  // namespace mm
}  // namespace hz

#if defined(HZSDR_NO_PK_F32) && defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute pop
#endif

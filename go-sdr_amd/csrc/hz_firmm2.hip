// hz_firmm2.hip -- the instantiations of the persistent-pass int8 matrix FIR (hz_firmm2.h) and their launcher.
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

template <int FMT>
static int launch_fmt(hipStream_t stream, int num_cus, unsigned D, const void *in, float2 *out, const float2 *hist,
                      float2 *new_hist, const uint8_t *rhist, uint8_t *new_rhist, const float2 *taps, size_t n, const Geom &g,
                      Plan L, const EwProgram &P, const Fix &F, int rolled) {
    if (D != 8) return HZSDR_ERR_INVALID_ARGUMENT;
    const size_t lds = lds_bytes((int)D, g.ks, g.ne, g.ntaps);
    // one workgroup per CU; a call with fewer passes than CUs: one pass per workgroup
    const int grid = std::max(1, L.grid);  // (the planner's: min(CUs, passes))
    (void)num_cus;
    unsigned long long *no_stamps = nullptr;
    // (the straight-line matrix loop exists for the 1024-tap window: 17 groups; `rolled`: hzsdr_chain_fir_options'
    // loop form -- 1, 2, 4 groups per trip, anything else the instantiation for any window -- for A/B measurements)
    if (g.ks == 17 * 4 && rolled == 0)
        return launch(fir_mm2_kernel<FMT, 8, 17>, dim3(grid), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, no_stamps);
    if (g.ks == 17 * 4 && rolled == 1)
        return launch(fir_mm2_kernel<FMT, 8, 17, 0, 1>, dim3(grid), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, no_stamps);
    if (g.ks == 17 * 4 && rolled == 2)
        return launch(fir_mm2_kernel<FMT, 8, 17, 0, 2>, dim3(grid), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, no_stamps);
    if (g.ks == 17 * 4 && rolled == 4)
        return launch(fir_mm2_kernel<FMT, 8, 17, 0, 4>, dim3(grid), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, no_stamps);
    return launch(fir_mm2_kernel<FMT, 8, 0>, dim3(grid), lds, stream, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, no_stamps);
}

int launch_fir(hipStream_t stream, int num_cus, int fmt, unsigned D, const void *in, float2 *out, const float2 *hist,
               float2 *new_hist, const uint8_t *rhist, uint8_t *new_rhist, const float2 *taps, size_t n, const Geom &g,
               const Plan &L, const EwProgram &P, const Fix &F, int loop_form) {
    if (fmt == HZSDR_FMT_U8)
        return launch_fmt<HZSDR_FMT_U8>(stream, num_cus, D, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, loop_form);
    if (fmt == HZSDR_FMT_I8)
        return launch_fmt<HZSDR_FMT_I8>(stream, num_cus, D, in, out, hist, new_hist, rhist, new_rhist, taps, n, g, L, P, F, loop_form);
    return HZSDR_ERR_INVALID_ARGUMENT;
}

}  // namespace mm2
}  // namespace hz

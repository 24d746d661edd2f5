// hz_fft_api.h -- host-side entry points of hz_fft.hip used by hz_conv.hip and hz_chain_fir.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "hz_fftv.h"

struct hzsdr_ctx;

namespace hz {
// lengths fft_device takes: powers of two up to 2^24, anything else up to 2^23 (Bluestein over a 2^24-point plan)
inline bool fft_length_ok(size_t n) { return n >= 1 && (((n & (n - 1)) == 0) ? n <= ((size_t)1 << 24) : n <= ((size_t)1 << 23)); }
// exp(-2 pi i m / n), m < n, float32 from float64; cached per context.
int get_twiddles(hzsdr_ctx *ctx, size_t n, const float2 **out);
// per-pass tables of the packed-math core (hz_fftv.h), n in 256 .. 8192; cached per context
int get_fv_tables(hzsdr_ctx *ctx, size_t n, fv::FvTabs *out);
int get_fv_poly_tables(hzsdr_ctx *ctx, size_t n, unsigned fold, fv::PolyTabs *out);
// `batch` consecutive length-n transforms, device pointers, any length (powers of two directly, others by Bluestein's
// chirp transform over them: scratch slots 12 and 13).
int fft_device(hzsdr_ctx *ctx, const void *in, void *out, size_t n, size_t batch, bool fwd);
// Everything fft_device(.., n, ..) would otherwise build on its first call for a length -- for a length that is not
// a power of two the chirp and its spectrum: ~50 n bytes of host vectors, a float64 transform of M >= 2n - 1 points
// on the host, a blocking upload; seconds at n ~ 2^23 -- built NOW: called where a chain, a convolution closure or a
// plan is created, so that no call on the data path stalls for it.
int fft_prepare(hzsdr_ctx *ctx, size_t n);
// f1 *= f2 (or conj(f2)) with Go complex64 multiply semantics.
void pointwise_mul_device(hzsdr_ctx *ctx, void *f1, const void *f2, size_t n, bool conj);
// the same over nblocks blocks of `period` values each against ONE f2 of `period` values
void pointwise_mul_blocks_device(hzsdr_ctx *ctx, void *f1, const void *f2, size_t period, size_t nblocks);
}  // namespace hz

// hz_fft_api.h -- host-side entry points of hz_fft.hip used by hz_conv.hip and hz_chain_fir.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "hz_fftv.h"

struct hzsdr_ctx;

namespace hz {
// exp(-2 pi i m / n), m < n, float32 from float64; cached per context.
int get_twiddles(hzsdr_ctx *ctx, size_t n, const float2 **out);
// per-pass tables of the packed-math core (hz_fftv.h), n in 256 .. 8192; cached per context
int get_fv_tables(hzsdr_ctx *ctx, size_t n, fv::FvTabs *out);
int get_fv_poly_tables(hzsdr_ctx *ctx, size_t n, unsigned fold, fv::PolyTabs *out);
// `batch` consecutive length-n transforms, device pointers, any power of two.
int fft_device(hzsdr_ctx *ctx, const void *in, void *out, size_t n, size_t batch, bool fwd);
// f1 *= f2 (or conj(f2)) with Go complex64 multiply semantics.
void pointwise_mul_device(hzsdr_ctx *ctx, void *f1, const void *f2, size_t n, bool conj);
}  // namespace hz

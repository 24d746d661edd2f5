/*
 * hzsdr.h -- C ABI of libhzsdr_hip: the MI355X (gfx950) backend for the
 * hz.tools/sdr sample-processing hot path.
 *
 * This is the boundary a cgo package binds (INTEGRATION.md shows the Go side).
 * Plain C: pointers, sizes, ints.  Every entry point returns an hzsdr_status
 * (0 = OK) unless it says otherwise.  Each declaration cites the reference
 * interface (file:line under the go-sdr checkout) it replaces.
 *
 * Conventions
 *   - Lengths and counts are in IQ SAMPLES, never bytes (reader.go:40-43).
 *   - Sample memory layouts are the reference's: u8/i8 = interleaved I,Q bytes
 *     (iq_u8.go:35, iq_i8.go:31); i16 = native-endian int16 pairs (iq_i16.go:50);
 *     c64 = interleaved float32 re,im (iq_c64.go:38).
 *   - A context is bound to one GPU and one hipStream.  Its memory space says
 *     what the sample pointers passed through it are: HZSDR_MEM_HOST (ordinary
 *     process memory, e.g. Go slices: the call stages H2D, runs the kernel,
 *     stages D2H and returns when the result is in `dst`; the library keeps no
 *     pointer after return, as cgo requires) or HZSDR_MEM_DEVICE (HIP device
 *     pointers: the call only enqueues on the context's stream).
 *   - Arguments are validated on the host before the GPU is touched, so the
 *     reference's error known-answer tests hold without running a kernel.
 *   - Entry points may be called from any OS thread (the device is re-selected
 *     inside every call); one context must not be used by two threads at once,
 *     matching the reference's single-consumer Readers (pipe.go:112).
 *   - There is no CPU fallback: without a gfx950 device hzsdr_open fails with
 *     HZSDR_ERR_NO_DEVICE.
 */
#ifndef HZSDR_H
#define HZSDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sdr.SampleFormat values, iq.go:110-126 */
#define HZSDR_FMT_C64 1
#define HZSDR_FMT_U8 2
#define HZSDR_FMT_I16 3
#define HZSDR_FMT_I8 4

/* Status codes.  The first four map 1:1 onto the reference's sentinel errors. */
#define HZSDR_OK 0
#define HZSDR_ERR_FORMAT_MISMATCH 1          /* sdr.ErrSampleFormatMismatch, iq.go:30 */
#define HZSDR_ERR_FORMAT_UNKNOWN 2           /* sdr.ErrSampleFormatUnknown,  iq.go:34 */
#define HZSDR_ERR_DST_TOO_SMALL 3            /* sdr.ErrDstTooSmall,          iq.go:38 */
#define HZSDR_ERR_CONVERSION_NOT_IMPLEMENTED 4 /* sdr.ErrConversionNotImplemented, conv.go:30 */
#define HZSDR_ERR_LENGTH_MISMATCH 5          /* the fmt.Errorf length errors: add.go:34, fft/convolution.go:38,157 */
#define HZSDR_ERR_INVALID_ARGUMENT 6
#define HZSDR_ERR_NO_DEVICE 7                /* no gfx950 GPU / HIP runtime unusable */
#define HZSDR_ERR_HIP 8                      /* a HIP call failed; see hzsdr_last_error */
#define HZSDR_ERR_OUT_OF_MEMORY 9

#define HZSDR_MEM_HOST 0
#define HZSDR_MEM_DEVICE 1

typedef struct hzsdr_ctx hzsdr_ctx;
typedef struct hzsdr_lut hzsdr_lut;
typedef struct hzsdr_rotlut hzsdr_rotlut;
typedef struct hzsdr_nco hzsdr_nco;
typedef struct hzsdr_fft hzsdr_fft;
typedef struct hzsdr_conv hzsdr_conv;
typedef struct hzsdr_chain hzsdr_chain;

/* ---- library / context -------------------------------------------------- */

/* "hip:gfx950": the string a simd.Backends-style report lists
 * (internal/simd/simd.go:29, debug/build.go:69-72).  Returns a static string. */
const char *hzsdr_backend(void);
const char *hzsdr_version(void);
/* Static text for a status code (rtl/error.go:29-36 rvToErr analogue). */
const char *hzsdr_strerror(int status);
/* SampleFormat.Size(), iq.go:93-106: bytes per IQ sample, 0 if unknown. */
int hzsdr_format_size(int format);
/* Number of usable gfx950 devices (the CPUID gate of
 * internal/simd/enabled_amd64.go:35-55 becomes an arch check). */
int hzsdr_device_count(int *count);

int hzsdr_open(int device, int memspace, hzsdr_ctx **out);
int hzsdr_close(hzsdr_ctx *ctx);
/* Detail of the last failure on this context ("" if none). */
const char *hzsdr_last_error(const hzsdr_ctx *ctx);
int hzsdr_memspace(const hzsdr_ctx *ctx);
/* Adopt the caller's hipStream_t (e.g. the stream a host framework already
 * orders its work on).  NULL is HIP's default (null) stream, as everywhere in
 * HIP.  hzsdr_use_own_stream goes back to the context's private stream. */
int hzsdr_set_stream(hzsdr_ctx *ctx, void *hip_stream);
int hzsdr_use_own_stream(hzsdr_ctx *ctx);
void *hzsdr_get_stream(const hzsdr_ctx *ctx);
/* Block until everything enqueued on the context's stream has finished. */
int hzsdr_synchronize(hzsdr_ctx *ctx);
/* How many calls of this library have selected the context's device so far -- every entry point that runs or
 * enqueues GPU work, allocates or synchronises does, once -- for tests and logs that count how many calls a Reader
 * pipeline makes per sample (go/hip/readers.go: nested Readers fuse into one chain and read ahead; a pipeline that
 * makes one call per 32 Ki-sample block is bound by the call's latency, not by the GPU).  No reference counterpart. */
int hzsdr_call_count(const hzsdr_ctx *ctx, unsigned long long *calls);

/* C-owned buffers for Go to wrap with yikes.Samples (yikes/bytes.go:50-71) or
 * to hand out from RingBufferOptions.IQBufferAllocator (stream/ring.go:60-68). */
int hzsdr_malloc_device(hzsdr_ctx *ctx, size_t bytes, void **out);
int hzsdr_free_device(hzsdr_ctx *ctx, void *ptr);
int hzsdr_malloc_pinned(hzsdr_ctx *ctx, size_t bytes, void **out);
int hzsdr_free_pinned(hzsdr_ctx *ctx, void *ptr);
/* Stream-ordered copies between host and device memory. */
int hzsdr_memcpy_h2d(hzsdr_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);
int hzsdr_memcpy_d2h(hzsdr_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);

/* ---- format converters ---------------------------------- */

/* sdr.ConvertBuffer(dst, src), conv.go:55-93; the twelve converters
 * iq_u8.go:75-121, iq_i8.go:71-119, iq_i16.go:116-162, iq_c64.go:77-117 and
 * the native kernels iq_u8_amd64.s:27-90 / iq_u8_amd64.go:26-38.  Equal
 * formats copy min(dst_len, src_len) samples (CopySamples, copy.go:31-52).
 * *n_out = samples written.  Errors: DST_TOO_SMALL, FORMAT_UNKNOWN. */
int hzsdr_convert(hzsdr_ctx *ctx, int dst_format, void *dst, size_t dst_len, int src_format,
                  const void *src, size_t src_len, size_t *n_out);
/* SamplesI16.ShiftLSBToMSBBits(bits), iq_i16.go:103-111; in place. */
int hzsdr_i16_shift_lsb_to_msb(hzsdr_ctx *ctx, void *buf_i16, size_t n, int bits);

/* ---- LookupTable -------------------------------------------- */

/* sdr.NewLookupTable(inputFormat, lookup), iq_lookup_table.go:98-123.  `table`
 * holds exactly 65536 samples of dst_format, indexed by the raw little-endian
 * uint16 of the two source bytes (iq_lookup_table.go:56-64); it is copied. */
int hzsdr_lut_create(hzsdr_ctx *ctx, int src_format, int dst_format, const void *table,
                     size_t table_len, hzsdr_lut **out);
/* LookupTable.Lookup(dst, src), iq_lookup_table.go:129-150 and the gather
 * loops :177-251.  src may alias dst when the formats have equal size. */
int hzsdr_lut_lookup(hzsdr_lut *lut, int dst_format, void *dst, size_t dst_len, int src_format,
                     const void *src, size_t src_len, size_t *n_out);
int hzsdr_lut_free(hzsdr_lut *lut);
/* LookupTableIdentityU8 / I8, iq_lookup_table.go:69-90: fills 65536 samples
 * (2 bytes each) of HOST memory; pure host helper. */
int hzsdr_lut_identity(void *table_host_131072_bytes);

/* ---- c64 vector ops -------------------------------- */

/* simd.ScaleComplex / SamplesC64.Scale / stream.Gain.Scale:
 * internal/simd/mult.go:40, mult_simd_amd64.s:27-55, iq_c64.go:122,
 * stream/gain.go:39-48.  In place. */
int hzsdr_scale(hzsdr_ctx *ctx, void *buf_c64, size_t n, float r);
/* simd.RotateComplex / SamplesC64.Multiply: internal/simd/mult.go:29-33,45,
 * iq_c64.go:128.  In place; Go complex64 multiply semantics. */
int hzsdr_rotate(hzsdr_ctx *ctx, void *buf_c64, size_t n, float re, float im);
/* simd.AddComplex(a, b, c), internal/simd/add.go:33, add_simd_amd64.s:27-71,
 * iq_c64.go:134.  c may alias a or b.  LENGTH_MISMATCH unless na == nb == nc. */
int hzsdr_add(hzsdr_ctx *ctx, const void *a, size_t na, const void *b, size_t nb, void *c,
              size_t nc);
/* The data path of addReader.Read, stream/add.go:121-185: out = 0, then
 * out += bufs[k] for k = 0..count-1 in that order (c64 float adds; i16 / i8
 * wrapping adds, stream/add.go:95-119).  u8 -> FORMAT_UNKNOWN (stream/add.go:55-61). */
int hzsdr_sum(hzsdr_ctx *ctx, int format, void *out, const void *const *bufs, int count, size_t n);

/* ---- stream.Multiply on u8 / i8 ---------------------------- */

/* uint8MultiplyReader / int8MultiplyReader, stream/multiply.go:91-238: a
 * table built by identity -> c64 -> Multiply(m) -> back, rebuilt by
 * SetMultiplier; the u8 variant keeps the reference's private I*255+Q index
 * (stream/multiply.go:106-108) including its aliasing.  format = U8 or I8. */
int hzsdr_rotlut_create(hzsdr_ctx *ctx, int format, float re, float im, hzsdr_rotlut **out);
int hzsdr_rotlut_set_multiplier(hzsdr_rotlut *t, float re, float im);
/* In place over n samples (stream/multiply.go:118-140, :186-205). */
int hzsdr_rotlut_apply(hzsdr_rotlut *t, void *buf, size_t n);
int hzsdr_rotlut_free(hzsdr_rotlut *t);

/* ---- stream.ShiftBuffer NCO -------------------------------- */

/* stream.ShiftBuffer(sampleRate), stream/shifter.go:66-85: the returned
 * closure becomes an object whose float64 time `ts` persists across buffers. */
int hzsdr_nco_create(hzsdr_ctx *ctx, uint64_t sample_rate, hzsdr_nco **out);
/* The closure call fn(freq, buf): buf[j] *= exp(i * (2*pi*freq) * ts_j), in
 * place, advancing ts by n samples. */
int hzsdr_nco_shift(hzsdr_nco *nco, double shift_hz, void *buf_c64, size_t n);
int hzsdr_nco_get_time(const hzsdr_nco *nco, double *ts);
int hzsdr_nco_set_time(hzsdr_nco *nco, double ts);
/* Opt-in, off by default: the rotation factor of stream/shifter.go:81-82 within ONE float32 ulp
 * of the reference's complex64(complex(cos, sin)) instead of bit-identical to it -- the phase as a
 * 32-bit fraction of a turn and float32 polynomials instead of math.Sincos in float64 (a third of
 * the vector instructions: the kernel becomes HBM-bound, 0.59 -> 0.76 of the roofline).  The clock
 * (ts, its wraps at 2 pi, its state across calls) is the reference's either way.  north_star:
 * "within 1 ULP for c64 float ops".  The chain form is hzsdr_chain_shift_ulp1. */
int hzsdr_nco_set_ulp1(hzsdr_nco *nco, int on);
int hzsdr_nco_free(hzsdr_nco *nco);

/* One run of the NCO time sequence that is exactly linear:
 * ts[first + i] = t0 + i * step for 0 <= i < count (float64, exact). */
typedef struct hzsdr_nco_segment {
    uint64_t first;
    uint64_t count;
    double t0;
    double step;
} hzsdr_nco_segment;
/* Pure host helper (no GPU): the piecewise-linear description of the next n
 * values of `ts += 1/rate; if ts > 2pi { ts -= 2pi }` (stream/shifter.go:76-79)
 * starting from ts_start.  Writes at most cap segments; *n_segments is the
 * number needed; *ts_end the state after n samples. */
int hzsdr_nco_segments(uint64_t sample_rate, double ts_start, uint64_t n,
                       hzsdr_nco_segment *segments, size_t cap, size_t *n_segments,
                       double *ts_end);

/* ---- Decimate / Downsample ---------------------------- */

/* stream.DecimateBuffer(to, from, factor, offset), stream/decimate.go:59-101:
 * to[i] = from[factor*i]; u8 / i16 / c64 only; `offset` is accepted and
 * ignored exactly as the reference does. */
int hzsdr_decimate(hzsdr_ctx *ctx, int to_format, void *to, size_t to_len, int from_format,
                   const void *from, size_t from_len, unsigned factor, int64_t offset,
                   size_t *n_out);
/* stream.DownsampleBuffer(to, from, factor, offset), stream/downsample.go:68-127:
 * boxcar mean of `factor` samples converted to c64, accumulated in order. */
int hzsdr_downsample(hzsdr_ctx *ctx, int to_format, void *to, size_t to_len, int from_format,
                     const void *from, size_t from_len, unsigned factor, int64_t offset,
                     size_t *n_out);

/* ---- fft.Planner / fft.Plan and the convolve helpers --- */

#define HZSDR_FFT_BACKWARD 0 /* fft.Backward = false, fft/fft.go:36 */
#define HZSDR_FFT_FORWARD 1  /* fft.Forward  = true,  fft/fft.go:33 */

/* fft.Planner(iq, frequency, direction), fft/fft.go:45-48: binds the two
 * buffers for the life of the plan.  iq_len != freq_len -> DST_TOO_SMALL
 * (testutils/fft.go:127-137).  Forward = exp(-j 2 pi k n / N), natural bin
 * order, backward unnormalised.  ANY length, as the reference's Planner (fft/fft.go:45-48): powers of two up to
 * 2^24 on the power-of-two kernels, every other length up to 2^23 by Bluestein's chirp transform over them (two
 * transforms of the next power of two >= 2 N - 1 and three elementwise passes; the chirp and its spectrum are formed
 * in float64 once per context and length, at plan time). */
int hzsdr_fft_plan(hzsdr_ctx *ctx, void *iq_c64, size_t iq_len, void *freq_c64, size_t freq_len,
                   int direction, hzsdr_fft **out);
/* `batch` independent transforms over consecutive length-n blocks of both buffers. */
int hzsdr_fft_plan_batch(hzsdr_ctx *ctx, void *iq_c64, void *freq_c64, size_t n, size_t batch,
                         int direction, hzsdr_fft **out);
/* Plan.Transform(), fft/fft.go:52-55. */
int hzsdr_fft_transform(hzsdr_fft *plan);
/* Plan.Close(), fft/fft.go:57-58. */
int hzsdr_fft_free(hzsdr_fft *plan);

#define HZSDR_CONV_CONVOLVE 0        /* fft.Convolve,       fft/convolution.go:97-113 */
#define HZSDR_CONV_CROSS_CORRELATE 1 /* fft.CrossCorrelate, fft/convolution.go:119-138 */
/* fft.Convolve / fft.CrossCorrelate: returns the "func() error" closure as an
 * object.  dst may alias iq1 / iq2.  LENGTH_MISMATCH unless all equal. */
int hzsdr_convolve_create(hzsdr_ctx *ctx, void *dst, size_t dst_len, const void *iq1,
                          size_t iq1_len, const void *iq2, size_t iq2_len, int mode,
                          hzsdr_conv **out);
/* fft.ConvolveFreq(planner, dst, src, freq), fft/convolution.go:150-192; `freq`
 * (frequency-domain filter, n bins) is copied at creation. */
int hzsdr_convolve_freq_create(hzsdr_ctx *ctx, void *dst, size_t dst_len, const void *src,
                               size_t src_len, const void *freq, size_t freq_len,
                               hzsdr_conv **out);
/* Calls the closure once: forward transform(s), pointwise multiply with Go
 * complex64 semantics (fft/convolution.go:107-109, :187-189), backward. */
int hzsdr_conv_exec(hzsdr_conv *conv);
/* DIFFERENCE from the reference: fft.ConvolveFreq's closure captures the `freq` slice by
 * reference (fft/convolution.go:183-189), so a Go caller that rewrites the filter bins
 * between calls sees the new filter; here the bins were snapshotted into device memory at
 * creation.  A caller that updates its filter calls this before the next hzsdr_conv_exec
 * (the copy is ordered on the context's stream behind earlier execs).  ConvolveFreq
 * closures only; LENGTH_MISMATCH unless freq_len equals the closure's length. */
int hzsdr_conv_set_filter(hzsdr_conv *conv, const void *freq, size_t freq_len);
int hzsdr_conv_free(hzsdr_conv *conv);
/* The whole-buffer form of stream.ConvolutionReader, stream/convolution.go:36-82:
 * block-circular filtering of consecutive len(filter)-sample blocks; a
 * trailing partial block is not produced.  *n_out = blocks * filter_len.
 * Any filter length a plan takes (stream/convolution.go:57-61 blocks on len(filter), whatever it is): powers of two
 * in 4 ... 8192 run as ONE kernel per launch, every other length as batched transforms through scratch. */
int hzsdr_convolution_blocks(hzsdr_ctx *ctx, void *out, size_t out_len, const void *in,
                             size_t in_len, const void *filter_freq, size_t filter_len,
                             size_t *n_out);

/* ---- Beamform ---------------------------------------------- */

/* stream.BeamformAngles2D / BeamformAngles, stream/beamform.go:57-128.  Pure
 * host float64 math; writes n complex64 values (re, im pairs). */
int hzsdr_beamform_angles_2d(double frequency_hz, double angle_deg, const double center[2],
                             const double *antennas_xy, int n, float *out_c64);
int hzsdr_beamform_angles(double frequency_hz, double angle_deg, const double *distances, int n,
                          float *out_c64);
/* The data path of stream.ReadBeamform, stream/beamform.go:148-171:
 * out = ((0 + w0*x0) + w1*x1) + ...; per channel ConvertReader -> c64
 * (`format` is the channels' format), multiplyReader (skipped when w == 1,
 * stream/multiply.go:59-62), then the ordered sum of stream/add.go:115-119.
 * weights: count complex64 values in HOST memory, passed by value per call
 * (Beamform.SetPhaseAngles, stream/beamform.go:131-139, applies between reads). */
int hzsdr_beamform(hzsdr_ctx *ctx, void *out_c64, int format, const void *const *channels,
                   const float *weights_c64, int count, size_t n);
/* Partial sum for sharding channels over GPUs: as hzsdr_beamform
 * but accumulating onto the current contents of out_c64 when accumulate != 0,
 * so rank r continues the ordered sum that ranks < r started. */
int hzsdr_beamform_partial(hzsdr_ctx *ctx, void *out_c64, int format,
                           const void *const *channels, const float *weights_c64, int count,
                           size_t n, int accumulate);

/* ---- rtl/kerberos coherent sync reductions (a "next" row) -- */

/* The peak search of checkAlignment, rtl/kerberos/internal/align.go:128-149, over
 * a cross-correlation (fft.CrossCorrelate / hzsdr_convolve_create with
 * HZSDR_CONV_CROSS_CORRELATE): index of the largest float64(re*re + im*im)
 * (float32 products), exact zeros skipped, first maximum wins, indices above n/2
 * folded to negative lags.  *lag = -1 when every element is zero. */
int hzsdr_peak_lag(hzsdr_ctx *ctx, const void *corr_c64, size_t n, int64_t *lag);
/* The inner loop of PhaseOffsets for one channel pair, align.go:257-266: the
 * mean over i of Phase(a[i] * conj(b[i])) in float64 (radians).  The caller
 * finishes with cmplx.Rect(1, mean) as the reference does. */
int hzsdr_mean_phase(hzsdr_ctx *ctx, const void *a_c64, const void *b_c64, size_t n,
                     double *mean_phase);

/* FFTShiftAndScale(data, scale), rtl/kerberos/internal/reader.go:47-64: swap the
 * two halves of a spectrum and divide every component by `scale`.  In place. */
int hzsdr_fftshift_scale(hzsdr_ctx *ctx, void *data_c64, size_t n, float scale);
/* One block of GraftReaders, rtl/kerberos/internal/graft.go:63-122 (SURVEY 8f rank
 * 3): forward-transform `count` adjacent bands of n samples each, fftshift + scale
 * by 1/n, concatenate the spectra and run ONE backward transform of count*n points:
 * count*n samples at count times the rate.  n and count*n must be powers of two. */
int hzsdr_graft(hzsdr_ctx *ctx, void *out_c64, size_t out_len, const void *const *channels_c64,
                int count, size_t n);

/* ---- foreign-endian wire / disk formats ----------------- */

/* What sdr.ByteReader / ByteWriter do for a byte order that is not the host's
 * (bytes_io.go:30-64, :150-197): every int16 component (i16) or float32 component
 * (c64) has its bytes reversed; u8 / i8 are untouched.  In place over n samples. */
int hzsdr_byteswap(hzsdr_ctx *ctx, int format, void *buf, size_t n);
/* hzsdr_convert with the swap fused into the converter's load and/or store: one
 * pass instead of three when a foreign-order capture (ByteReader, bytes_io.go:150-197)
 * feeds a ConvertBuffer, or a ConvertBuffer feeds a foreign-order ByteWriter
 * (bytes_io.go:30-64).  *_foreign != 0: that side's int16 / float32 components are
 * in the byte order that is not the host's.  Formats must differ. */
int hzsdr_convert_foreign(hzsdr_ctx *ctx, int dst_format, void *dst, size_t dst_len,
                          int dst_foreign, int src_format, const void *src, size_t src_len,
                          int src_foreign, size_t *n_out);

/* ---- fused operator chains (north_star: one kernel per buffer) ----------- */

/* A chain is the GPU form of nested stream.* Readers over one source
 * (stream/read_transformer.go:92-116 Proc is the per-buffer hook it plugs in
 * at).  Stages are pushed in data-flow order; elementwise stages (convert to
 * c64, shift, gain, rotate) and at most one terminal stage fuse into a single
 * kernel launch per hzsdr_chain_run (the FIR-decimate terminal: one for u8 / i8 sources at
 * factor 8 / 16 -- the int8 matrix form, hzsdr_chain_last_fir_path -- otherwise two, the
 * overlap-save analysis and the small inverse transforms; no full-rate complex64
 * intermediate is ever written). */
int hzsdr_chain_create(hzsdr_ctx *ctx, int src_format, uint64_t sample_rate, hzsdr_chain **out);
/* stream.ShiftReader(r, shift), stream/shifter.go:89-102 (stateful NCO). */
int hzsdr_chain_shift(hzsdr_chain *c, double shift_hz);
/* stream.Gain(r, v), stream/gain.go:30-57. */
int hzsdr_chain_gain(hzsdr_chain *c, float r);
/* stream.Multiply(r, m) on c64, stream/multiply.go:27-89 (skipped when m == 1). */
int hzsdr_chain_rotate(hzsdr_chain *c, float re, float im);
/* Terminal: stream.DecimateReader(r, factor), stream/decimate.go:34-55
 * (32 Ki-sample blocks, phase restarts per block as in the reference). */
int hzsdr_chain_decimate(hzsdr_chain *c, unsigned factor);
/* Terminal: stream.DownsampleReader(r, factor), stream/downsample.go:47-64. */
int hzsdr_chain_downsample(hzsdr_chain *c, unsigned factor);
/* Terminal: stream.ConvolutionReader(r, planner, filter), stream/convolution.go:36-82,
 * optionally followed by DecimateReader(factor) (factor 1 = none).  Any filter length a plan takes; with a
 * DecimateReader behind it the chain consumes whole multiples of lcm(filter_len, 32 Ki) samples. */
int hzsdr_chain_convolution(hzsdr_chain *c, const void *filter_freq, size_t filter_len,
                            unsigned decimate_factor);
/* Terminal: the north-star FIR-decimate (BASELINE.json north_star; not a
 * reference function): y[m] = sum_k taps[k] * x[factor*m - k], history carried across
 * runs; at most 8191 taps (minus the rounding of taps - 1 up to the factor).  taps: n_taps
 * complex64 in HOST memory. */
int hzsdr_chain_fir_decimate(hzsdr_chain *c, const float *taps_c64, size_t n_taps,
                             unsigned factor);
/* Where the elementwise stages run relative to the FIR-decimate terminal.
 * Default (in_order = 0): Shift / Gain / Multiply are multiplications by a complex
 * scalar or by exp(i*tau*ts[n]), and the NCO clock (stream/shifter.go:76-79) is exactly
 * linear inside one binade, so they commute with the filter: blocks whose input lies
 * in one such run are filtered with taps modulated by exp(-i*tau*k*step) and mixed at
 * the DECIMATED rate, with the same Sincos, clock and multiply order; blocks at the
 * stream edges, across a run boundary or across the clock's 2*pi wrap keep reference
 * order.  The two orders agree to float32 rounding (the FIR itself is held to an error
 * bound, not to bits).  in_order = 1: every block mixes each input sample before the
 * filter, exactly as nested stream.ShiftReader -> filter Readers would.
 * A VERIFICATION SWITCH, not a mode to run a stream in: it exists so that the default order can be
 * checked against the reference's own order on the same kernels (tests/test_gpu_latemix.py,
 * tests/test_gpu_fullsize.py hold the two to 3e-7 of each other and each to the oracle).  It keeps a
 * chain on the transform kernels and evaluates the mixer at the INPUT rate -- 2^24 Sincos per 2^24
 * samples where the default evaluates 2^21: 130 us per buffer against 28-36, and no faster form is
 * planned (round 6 closed the item: the outputs it produces are inside the same error bound as the
 * default's, so there is nothing a stream gains from it).  Results with it on are as well defined
 * and as tested as with it off. */
int hzsdr_chain_mix_in_order(hzsdr_chain *c, int in_order);
/* Measurement and test aid, in front of hzsdr_chain_fir_decimate: WHICH implementation the FIR-decimate
 * terminal takes -- all of them compute the same filter (stream/downsample.go:47-64 / a Reader that
 * convolves and decimates) within the bound of DESIGN.md section 2; AUTO is the library's choice by
 * format, factor and tap count and is what every ordinary caller wants.  nfft_min (0: the default):
 * the smallest overlap-save block (a power of two in 256 ... 8192); loop_form (0: the default): the
 * persistent-pass matrix kernel's loop form (1, 2, 4 groups per trip, 99 the any-window instantiation).
 * Replaces the round-2/3 environment switches HZ_FIR_FFT / HZ_MM_V1 / HZ_FIR_NFFT_MIN / HZ_MM_ROLLED,
 * which a library built with -DHZSDR_DIAG still reads, once, as the process-wide default. */
#define HZSDR_FIR_IMPL_AUTO 0
#define HZSDR_FIR_IMPL_TRANSFORMS 1     /* the overlap-save transform kernels */
#define HZSDR_FIR_IMPL_MATRIX_CHUNKS 2  /* the int8 matrix form as chunk workgroups (hz_firmm.h), where eligible */
int hzsdr_chain_fir_options(hzsdr_chain *c, int impl, unsigned nfft_min, int loop_form);
/* OPT-IN, off by default: consecutive calls of a FIR-decimate chain on the int8 matrix path, or of a chain without a
 * terminal stage (convert / Shift / Gain / Multiply maps), may OVERLAP on the GPU.
 * A stream runs its launches one behind the other -- the next call's workgroups wait for the last workgroup of this
 * one and then pay the kernel's head in full: ~5 us of 37 per 2^24-sample buffer.  The calls of one chain depend on
 * each other through the FIR history alone, which is a function of the call's input; a pipelined chain forms it in
 * a small kernel of its own and alternates its calls between two streams it owns.
 * The mode changes NOTHING about hzsdr_chain_run: that call is ordered on the context's stream like every other call
 * of this library (behind what the stream holds, in front of what is enqueued later), and so cannot overlap the call
 * before it -- the stream has just been made to wait for that one.  The overlap is taken by hzsdr_chain_run_after /
 * hzsdr_chain_run_batch_after (below), where the caller states what the call's buffers wait for instead.  Results are
 * bit-identical either way (the same kernels on the same values).  Calls that do not take the matrix path (too many
 * clock boundaries, other source formats and factors) run on the context's stream as before.  (No reference
 * counterpart: stream.* Readers are synchronous, reader.go:39-51; the single-consumer, in-order contract is kept.) */
int hzsdr_chain_pipeline(hzsdr_chain *c, int on);
/* OPT-IN, off by default: a chain WITHOUT a terminal (ShiftReader, or ShiftReader -> Gain: BASELINE config 2)
 * whose buffers allow four samples per lane forms the Shift's rotation factor from the phase in turns with
 * float32 polynomials instead of an operation-for-operation math.Sincos in float64.  The float64 product
 * (tau*shift)*ts and the clock are the reference's (stream/shifter.go:76-82); the factor complex64(cos, sin) is
 * then within ONE float32 ulp per component of the reference's (0.8 ulp from the polynomials, 2^-53 |phase| / 2 pi
 * turns from the reduction: phases up to 1e8 rad), so the outputs are no longer bit-identical to the
 * reference's -- about three times fewer vector instructions: the kernel runs at the HBM rate.  Other
 * programs and terminals are unaffected. */
int hzsdr_chain_shift_ulp1(hzsdr_chain *c, int on);
/* Which kernels the LAST hzsdr_chain_run of a FIR-decimate chain used (for tests, benchmarks and
 * logs; the results are held to the same bound either way):
 *   HZSDR_FIR_PATH_NONE       no run yet / no FIR-decimate terminal
 *   HZSDR_FIR_PATH_TRANSFORM  overlap-save transforms (any source format, any factor)
 *   HZSDR_FIR_PATH_MATRIX     int8 matrix form (csrc/hz_firmm2.h: factor 8 up to ~1100 taps, 16 up to 1040; csrc/hz_firmm.h
 *                             the rest): u8 / i8 sources, factor 8, 16, 24, 32,
 *                             40, 48 or 64, 16..1536 taps (factor 16 / 24: ..2560, larger: ..4096), default mixer order, 16-byte aligned device buffers,
 *                             at least 4096 outputs per call.  Environment HZ_FIR_FFT=1 (read when
 *                             the terminal is created) keeps a chain on the transforms. */
#define HZSDR_FIR_PATH_NONE 0
#define HZSDR_FIR_PATH_TRANSFORM 1
#define HZSDR_FIR_PATH_MATRIX 2
int hzsdr_chain_last_fir_path(const hzsdr_chain *c, int *path);
/* The same question one level down: WHICH matrix kernel (logs, benchmarks, the tests that pin a kernel):
 *   HZSDR_FIR_KERNEL_MATRIX_PASSES  csrc/hz_firmm2.h -- one persistent workgroup per CU, the tap table in LDS,
 *                                   512-output passes from a queue: factor 8, up to ~1150 taps, filters whose two top
 *                                   digit planes add up in int32 for every input (csrc/hz_firmm_plan.h,
 *                                   int32_combine_ok: every practical low-pass; a 1024-tap boxcar takes CHUNKS)
 *   HZSDR_FIR_KERNEL_MATRIX_CHUNKS  csrc/hz_firmm.h -- one round of 2048-output chunk workgroups: the other
 *                                   factors and tap counts of HZSDR_FIR_PATH_MATRIX (and HZ_MM_V1=1)
 * (no reference counterpart: stream.* Readers have one implementation each.) */
#define HZSDR_FIR_KERNEL_NONE 0
#define HZSDR_FIR_KERNEL_TRANSFORM 1
#define HZSDR_FIR_KERNEL_MATRIX_CHUNKS 2
#define HZSDR_FIR_KERNEL_MATRIX_PASSES 3
int hzsdr_chain_last_fir_kernel(const hzsdr_chain *c, int *kernel);
/* Samples the chain would produce for n_in input samples, and how many input
 * samples it consumes (whole blocks only for block-structured terminals). */
int hzsdr_chain_plan(const hzsdr_chain *c, size_t n_in, size_t *n_consumed, size_t *n_out);
/* One buffer through the chain.  out format: c64. */
int hzsdr_chain_run(hzsdr_chain *c, const void *in, size_t n_in, void *out, size_t out_cap,
                    size_t *n_consumed, size_t *n_out);
/* hzsdr_chain_run with the call's START ordered by the caller: the call's buffers -- `in` complete, `out` no longer
 * in use by anyone -- are ready when `ready_event` (a hipEvent_t; NULL: they are ready now) has fired, NOT "when the
 * context's stream gets there".  Everything else is hzsdr_chain_run's contract: the call's reads of `in` and writes
 * of `out` are ordered on the context's stream (work enqueued there afterwards may consume `out` and refill `in`),
 * chain state advances in call order, results are bit-identical.  On a chain in pipelined mode (hzsdr_chain_pipeline)
 * in a DEVICE-space context this is the call that overlaps with the one before it; buffers that overlap those of the
 * two calls before are detected and the call ordered behind them.  On any other chain or context the event is simply
 * waited for where the call runs.  A producer on another stream records the event; a producer on the context's own
 * stream needs nothing of this -- hzsdr_chain_run is already behind it.  (The pinned ring, hzsdr_ring_*, is built on
 * this call: its uploads are the events.)  No reference counterpart. */
int hzsdr_chain_run_after(hzsdr_chain *c, const void *in, size_t n_in, void *out, size_t out_cap,
                          size_t *n_consumed, size_t *n_out, void *ready_event);
/* n_buffers (1..8) consecutive buffers of the stream, n_in samples each, in ONE call: the results, the chain's state
 * and the ordering on the context's stream are those of n_buffers hzsdr_chain_run calls in a row (outs[j]: out_cap
 * samples each; *n_consumed / *n_out: per buffer; a batch's buffers must be consumed whole) -- BIT FOR BIT: what the
 * chain computes for a sample does not depend on how the stream was cut into calls, as stream/shifter.go:68-79's
 * clock does not (round 6: the mixer's phase and the split between the matrix path and the reference-order fix-up
 * outputs belong to the clock run's line in its binade, csrc/hz_firmm2_plan.h run_line; round 5's batches differed from
 * single calls in a last bit).  A FIR-decimate chain on the persistent-pass matrix kernel
 * (hzsdr_chain_last_fir_kernel; u8 / i8 sources at factor 8 or 16) whose buffers hold a whole number of passes (512
 * outputs at factor 8, 256 at 16), at least eight, takes the batch in ONE launch -- the kernel's head, launch and tail
 * are paid once per batch instead of once per buffer: 2^24-sample buffers by four, ~28-30 us per buffer where single
 * calls take ~34-36 -- PROVIDED every buffer by itself would have taken that kernel too (a short buffer that is
 * mostly clock boundaries keeps the transform kernels -- other arithmetic inside the same error bound -- and a call
 * over several then runs them one by one: the single calls are planned on the host first, a few microseconds per
 * buffer).  Every other chain runs the buffers one after the other.  _after: the START as in hzsdr_chain_run_after
 * (all the batch's buffers ready at the event).  No reference counterpart (a Reader hands over one slice per Read). */
int hzsdr_chain_run_batch(hzsdr_chain *c, const void *const *ins, void *const *outs, size_t n_buffers, size_t n_in,
                          size_t out_cap, size_t *n_consumed, size_t *n_out);
int hzsdr_chain_run_batch_after(hzsdr_chain *c, const void *const *ins, void *const *outs, size_t n_buffers, size_t n_in,
                                size_t out_cap, size_t *n_consumed, size_t *n_out, void *ready_event);
/* Forget stream state (NCO time, FIR history). */
int hzsdr_chain_reset(hzsdr_chain *c);
/* The Shift closure's clock `ts` (stream/shifter.go:71,77-80), in [0, 2*pi]: read it to
 * checkpoint a stream, set it to resume one (or to start a test next to the 2*pi wrap).
 * Setting it also prepares the late mixer's spectra for the runs that follow, so that
 * hzsdr_chain_run itself never allocates or waits. */
int hzsdr_chain_set_time(hzsdr_chain *c, double ts);
int hzsdr_chain_time(const hzsdr_chain *c, double *ts);
int hzsdr_chain_free(hzsdr_chain *c);

/* ---- Beamform sharded over GPUs from one process ------------------ */
/* The reference sums its coherent channels in one place (stream/beamform.go:148-171 ->
 * the ordered add of stream/add.go:115-119).  A cgo caller is ONE process with one context
 * per GPU: hzsdr_mgpu_open makes one DEVICE-space context (and stream) per entry of
 * `devices` -- the same GPU may appear more than once, which is how a 1-GPU box exercises
 * the exchange -- and hzsdr_mgpu_beamform runs the weighted sum with channel c resident on
 * shard owner(c) (contiguous ranges: hzsdr_mgpu_shard_channels).
 *   channels[c]  device pointer ON owner(c)'s GPU, `format` samples
 *   weights_c64  n_channels complex64 values in HOST memory
 *   out_c64      device pointer on shard `dst_shard`'s GPU
 * HZSDR_MGPU_ORDERED: bit-identical to hzsdr_beamform on one GPU (shard s owns slice s of
 *   the output, receives slice s of every weighted channel by peer copy and adds the K
 *   pieces in channel order from +0); HZSDR_MGPU_RCCL: per-shard partial sums + one
 *   ncclReduce (float sum) over xGMI: faster, RCCL's summation order (a few ULP), needs one
 *   shard per GPU; librccl is loaded at run time.
 * Nothing waits on the host: later work on the destination shard's stream is ordered
 * behind the result; hzsdr_mgpu_synchronize waits for every shard. */
#define HZSDR_MGPU_ORDERED 0
#define HZSDR_MGPU_RCCL 1
typedef struct hzsdr_mgpu hzsdr_mgpu;
int hzsdr_mgpu_open(const int *devices, int n_devices, hzsdr_mgpu **out);
int hzsdr_mgpu_close(hzsdr_mgpu *m);
int hzsdr_mgpu_shards(const hzsdr_mgpu *m);
/* the DEVICE-space context of a shard (allocate / upload its channels through it) */
int hzsdr_mgpu_ctx(hzsdr_mgpu *m, int shard, hzsdr_ctx **ctx);
/* channels [*lo, *hi) live on `shard` when n_channels are spread over n_shards */
int hzsdr_mgpu_shard_channels(int n_channels, int n_shards, int shard, int *lo, int *hi);
int hzsdr_mgpu_beamform(hzsdr_mgpu *m, void *out_c64, int dst_shard, int format,
                        const void *const *channels, const float *weights_c64, int n_channels,
                        size_t n, int mode);
int hzsdr_mgpu_synchronize(hzsdr_mgpu *m);
const char *hzsdr_mgpu_last_error(const hzsdr_mgpu *m);
/* How the shards reach each other, counted over the ordered pairs of DISTINCT GPUs when the object was opened:
 * *direct = pairs with peer access enabled (hipMemcpyPeerAsync goes over xGMI), *staged = pairs without (the copies
 * stage through the host at PCIe speed; hzsdr_mgpu_last_error names the last such pair).  Both 0 when every shard
 * sits on one GPU.  No reference counterpart (stream/add.go:115-119 sums in one address space). */
int hzsdr_mgpu_peer_pairs(const hzsdr_mgpu *m, int *direct, int *staged);

/* ---- pinned ring in front of a chain -------------------- */

/* The streaming form of a chain for live input: what stream.RingBuffer
 * (stream/ring.go:48-69, :337-392) is to a driver callback.  The ring owns ONE
 * pinned (hipHostMalloc) region of slots * slot_length samples of the chain's
 * source format -- the memory RingBufferOptions.IQBufferAllocator returns
 * (stream/ring.go:60-68), wrapped as sdr.Samples with yikes.Samples
 * (yikes/bytes.go:50-71) -- so driver callbacks (rtl/rx.go:49-68) land IQ where
 * the copy engine reads it.  A submitted slot goes upload -> chain kernel ->
 * download on three HIP streams chained by events: the PCIe transfers of
 * neighbouring slots overlap the kernel.  Slots are used in order; chain state
 * (NCO clock, FIR history) advances in submit order.  slot_length must be a whole
 * number of the chain's blocks (hzsdr_chain_plan consumes all of it).  One
 * producer / one consumer; calls on one ring must not overlap. */
typedef struct hzsdr_ring hzsdr_ring;
int hzsdr_ring_create(hzsdr_chain *c, size_t slot_length, int slots, hzsdr_ring **out);
/* The whole pinned IQ region (slot i starts at i * slot_length samples). */
int hzsdr_ring_iq_buffer(const hzsdr_ring *r, void **base, size_t *n_samples, size_t *slot_length);
/* Write cursor: the next slot and its pinned memory.  DST_TOO_SMALL when every
 * slot is acquired or still in flight (the overrun case: pop first).  Several slots
 * may be acquired before any is submitted (in ring order; stream/ring.go:337-392's
 * writer likewise runs ahead of the reader by whole slots). */
int hzsdr_ring_acquire(hzsdr_ring *r, int *slot, void **iq);
/* The OLDEST acquired slot holds n samples: enqueue its upload, kernel and download.
 * Returns without waiting. */
int hzsdr_ring_submit(hzsdr_ring *r, int slot, size_t n);
/* `count` (1..8) acquired slots, first_slot the oldest, n samples each, as ONE call of the chain: their uploads,
 * hzsdr_chain_run_batch_after's launch over the slots' device copies (one launch of the FIR-decimate terminal's
 * persistent-pass kernel where that form applies -- hzsdr_chain_run_batch's conditions --, slot by slot otherwise),
 * their downloads; hzsdr_ring_pop hands the outputs out slot by slot as before.  The results are those of `count`
 * hzsdr_ring_submit calls, bit for bit.  What a driver callback that fills slots faster than one launch per slot
 * pays for does: submit what has accumulated (stream/ring.go:337-392 has no counterpart -- its reader copies). */
int hzsdr_ring_submit_many(hzsdr_ring *r, int first_slot, int count, size_t n);
/* Give the NEWEST acquired slot back unused (the source had nothing for it: the end of a stream): the next
 * hzsdr_ring_acquire hands out the same slot.  No reference counterpart (stream.RingBuffer's writer simply does not
 * advance: stream/ring.go:337-392). */
int hzsdr_ring_release(hzsdr_ring *r, int slot);
/* Read cursor: wait for the oldest submitted slot; *out (pinned, complex64,
 * *n_out samples) stays valid until that slot is submitted again.
 * INVALID_ARGUMENT when nothing is in flight (the underrun case). */
int hzsdr_ring_pop(hzsdr_ring *r, const void **out, size_t *n_out);
int hzsdr_ring_in_flight(const hzsdr_ring *r);
int hzsdr_ring_free(hzsdr_ring *r);

#ifdef __cplusplus
}
#endif
#endif /* HZSDR_H */

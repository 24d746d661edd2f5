//go:build sdr.hip

package hip

// #include <hzsdr.h>
import "C"

import (
	"unsafe"

	"hz.tools/sdr"
	"hz.tools/sdr/fft"
	"hz.tools/sdr/yikes"
)

type plan struct {
	x *Context
	p *C.hzsdr_fft
}

func (p *plan) Transform() error { return toErr(p.x.c, C.hzsdr_fft_transform(p.p)) }
func (p *plan) Close() error     { return toErr(p.x.c, C.hzsdr_fft_free(p.p)) }

// Planner satisfies fft.Planner (fft/fft.go:45-48): forward = exp(-2 pi i k n / N), natural
// bin order, unnormalised backward; sdr.ErrDstTooSmall on a length mismatch
// (testutils/fft.go:127-137).
//
// A Plan keeps its two buffers for its life, which cgo forbids for Go-heap slices: allocate
// them with PlanBuffers (pinned C memory behind yikes.Samples, yikes/bytes.go:50-71).
func (x *Context) Planner(iq sdr.SamplesC64, frequency []complex64, dir fft.Direction) (fft.Plan, error) {
	d := C.int(C.HZSDR_FFT_BACKWARD)
	if dir == fft.Forward {
		d = C.HZSDR_FFT_FORWARD
	}
	var ip, fp unsafe.Pointer
	if len(iq) > 0 {
		ip = unsafe.Pointer(&iq[0])
	}
	if len(frequency) > 0 {
		fp = unsafe.Pointer(&frequency[0])
	}
	var p *C.hzsdr_fft
	rc := C.hzsdr_fft_plan(x.c, ip, C.size_t(len(iq)), fp, C.size_t(len(frequency)), d, &p)
	if rc != C.HZSDR_OK {
		return nil, toErr(x.c, rc)
	}
	return &plan{x, p}, nil
}

// PlanBatch transforms `batch` consecutive length-n blocks per Transform (spectrograms,
// rtl/kerberos' cross-correlations).
func (x *Context) PlanBatch(iq sdr.SamplesC64, frequency []complex64, n, batch int, dir fft.Direction) (fft.Plan, error) {
	d := C.int(C.HZSDR_FFT_BACKWARD)
	if dir == fft.Forward {
		d = C.HZSDR_FFT_FORWARD
	}
	var p *C.hzsdr_fft
	rc := C.hzsdr_fft_plan_batch(x.c, unsafe.Pointer(&iq[0]), unsafe.Pointer(&frequency[0]), C.size_t(n), C.size_t(batch), d, &p)
	if rc != C.HZSDR_OK {
		return nil, toErr(x.c, rc)
	}
	return &plan{x, p}, nil
}

// PlanBuffers returns n-sample iq and frequency buffers in pinned C memory (safe for a Plan
// to keep, and HOST-space calls on them skip all staging) and a function that frees them.
func (x *Context) PlanBuffers(n int) (iq sdr.SamplesC64, frequency []complex64, free func(), err error) {
	a, err := x.MallocPinned(8 * n)
	if err != nil {
		return nil, nil, nil, err
	}
	b, err := x.MallocPinned(8 * n)
	if err != nil {
		x.FreePinned(a)
		return nil, nil, nil, err
	}
	s, _ := yikes.Samples(uintptr(a), n, sdr.SampleFormatC64)
	f, _ := yikes.Samples(uintptr(b), n, sdr.SampleFormatC64)
	return s.(sdr.SamplesC64), []complex64(f.(sdr.SamplesC64)), func() { x.FreePinned(a); x.FreePinned(b) }, nil
}

type closure struct {
	x *Context
	c *C.hzsdr_conv
}

// Convolve / CrossCorrelate / ConvolveFreq return the reference's "func() error" closure
// (fft/convolution.go:97-113, :119-138, :150-192) plus a release function.
func (x *Context) Convolve(dst, iq1, iq2 sdr.SamplesC64) (func() error, func() error, error) {
	return x.convolve(dst, iq1, iq2, C.HZSDR_CONV_CONVOLVE)
}

func (x *Context) CrossCorrelate(dst, iq1, iq2 sdr.SamplesC64) (func() error, func() error, error) {
	return x.convolve(dst, iq1, iq2, C.HZSDR_CONV_CROSS_CORRELATE)
}

func (x *Context) convolve(dst, iq1, iq2 sdr.SamplesC64, mode C.int) (func() error, func() error, error) {
	var c *C.hzsdr_conv
	rc := C.hzsdr_convolve_create(x.c, base(dst), C.size_t(len(dst)), base(iq1), C.size_t(len(iq1)), base(iq2), C.size_t(len(iq2)), mode, &c)
	if rc != C.HZSDR_OK {
		return nil, nil, toErr(x.c, rc)
	}
	cl := &closure{x, c}
	return cl.exec, cl.free, nil
}

// ConvolveFreq: the filter bins are SNAPSHOTTED at creation (the reference's closure reads
// the slice on every call, fft/convolution.go:183-189): call setFilter after changing them.
func (x *Context) ConvolveFreq(dst, src sdr.SamplesC64, freq []complex64) (exec func() error, setFilter func([]complex64) error, free func() error, err error) {
	var c *C.hzsdr_conv
	rc := C.hzsdr_convolve_freq_create(x.c, base(dst), C.size_t(len(dst)), base(src), C.size_t(len(src)),
		unsafe.Pointer(&freq[0]), C.size_t(len(freq)), &c)
	if rc != C.HZSDR_OK {
		return nil, nil, nil, toErr(x.c, rc)
	}
	cl := &closure{x, c}
	return cl.exec, cl.setFilter, cl.free, nil
}

func (cl *closure) exec() error { return toErr(cl.x.c, C.hzsdr_conv_exec(cl.c)) }
func (cl *closure) free() error { return toErr(cl.x.c, C.hzsdr_conv_free(cl.c)) }
func (cl *closure) setFilter(freq []complex64) error {
	return toErr(cl.x.c, C.hzsdr_conv_set_filter(cl.c, unsafe.Pointer(&freq[0]), C.size_t(len(freq))))
}

// ConvolutionBlocks is the whole-buffer form of stream.ConvolutionReader (stream/convolution.go:36-82).
func (x *Context) ConvolutionBlocks(out, in sdr.SamplesC64, filterFreq []complex64) (int, error) {
	var n C.size_t
	rc := C.hzsdr_convolution_blocks(x.c, base(out), C.size_t(len(out)), base(in), C.size_t(len(in)),
		unsafe.Pointer(&filterFreq[0]), C.size_t(len(filterFreq)), &n)
	return int(n), toErr(x.c, rc)
}

// ---- rtl/kerberos (SURVEY 8f) ---------------------------------------------------------------

// PeakLag: index of the largest |corr[i]|^2, as a signed lag (rtl/kerberos/internal/align.go:128-149).
func (x *Context) PeakLag(corr sdr.SamplesC64) (int64, error) {
	var lag C.int64_t
	rc := C.hzsdr_peak_lag(x.c, base(corr), C.size_t(len(corr)), &lag)
	return int64(lag), toErr(x.c, rc)
}

// MeanPhase: arg(sum a[i] * conj(b[i])) (rtl/kerberos/internal/align.go:257-266).
func (x *Context) MeanPhase(a, b sdr.SamplesC64) (float64, error) {
	var ph C.double
	rc := C.hzsdr_mean_phase(x.c, base(a), base(b), C.size_t(len(a)), &ph)
	return float64(ph), toErr(x.c, rc)
}

// FFTShiftScale: swap the halves of a spectrum and scale it (graft.go:97-114's per-band step).
func (x *Context) FFTShiftScale(data []complex64, scale float32) error {
	return toErr(x.c, C.hzsdr_fftshift_scale(x.c, unsafe.Pointer(&data[0]), C.size_t(len(data)), C.float(scale)))
}

// Graft: K bands of n samples -> one stream of K*n samples (rtl/kerberos/internal/graft.go:97-114).
func (x *Context) Graft(out sdr.SamplesC64, bands []sdr.SamplesC64, n int) error {
	ptrs := (*[64]unsafe.Pointer)(C.malloc(C.size_t(64 * unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(ptrs))
	for i, b := range bands {
		ptrs[i] = base(b)
	}
	return toErr(x.c, C.hzsdr_graft(x.c, base(out), C.size_t(len(out)), (*unsafe.Pointer)(unsafe.Pointer(ptrs)), C.int(len(bands)), C.size_t(n)))
}

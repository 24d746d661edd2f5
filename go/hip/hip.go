//go:build sdr.hip

// Package hip is the MI355X (gfx950) backend for hz.tools/sdr's sample-processing
// hot path: every declaration of include/hzsdr.h bound with cgo, behind the
// reference's own plug points (stream.ReadTransformerConfig.Proc, fft.Planner,
// stream.RingBufferOptions.IQBufferAllocator, yikes.Samples).
//
// UNCOMPILED IN THE BUILD IMAGE: the image that built libhzsdr_hip.so has no Go
// toolchain.  tests/c/test_c_abi.c walks the same call sequences from plain C (gcc,
// the compiler cgo hands these preambles to), and the C++ / Python host layers run the
// reference's known-answer tests through the same entry points.
//
// Opt-in by build tag, like the reference's driver tags (README.md:28-37):
//
//	go build -tags sdr.hip ./...
//
// Conventions: lengths are IQ samples (sdr.Samples.Length()), never bytes.  A
// Context opened with MemHost takes pointers into Go slices: each call copies in,
// runs, copies out and returns; C keeps no Go pointer and never calls back.  Every
// entry point re-selects its device, so goroutines may migrate between OS threads;
// one Context is used by one goroutine at a time, like any sdr.Reader (pipe.go:112).
package hip

// #cgo CFLAGS: -I${SRCDIR}/../../include
// #cgo LDFLAGS: -L${SRCDIR}/../../go-sdr_amd -lhzsdr_hip -Wl,-rpath,${SRCDIR}/../../go-sdr_amd
// #include <stdlib.h>
// #include <hzsdr.h>
import "C"

import (
	"unsafe"

	"hz.tools/sdr"
)

// MemSpace says what the pointers handed to a Context are.
type MemSpace int

const (
	// MemHost: process memory (Go slices, pinned buffers).  Calls are synchronous.
	MemHost MemSpace = C.HZSDR_MEM_HOST
	// MemDevice: HIP device pointers.  Calls only enqueue on the context's stream.
	MemDevice MemSpace = C.HZSDR_MEM_DEVICE
)

// Backend is "hip:gfx950" (for a simd.Backends-style list, internal/simd/simd.go:29).
func Backend() string { return C.GoString(C.hzsdr_backend()) }

// Version of the C library.
func Version() string { return C.GoString(C.hzsdr_version()) }

// FormatSize is sdr.SampleFormat.Size() as the library sees it (iq.go:97-107).
func FormatSize(f sdr.SampleFormat) int { return int(C.hzsdr_format_size(C.int(f))) }

// DeviceCount returns the number of gfx950 GPUs.
func DeviceCount() (int, error) {
	var n C.int
	rc := C.hzsdr_device_count(&n)
	return int(n), toErr(nil, rc)
}

// Context is one GPU, one HIP stream, one memory space.
type Context struct {
	c     *C.hzsdr_ctx
	space MemSpace
}

// Open a context on `device`.
func Open(device int, space MemSpace) (*Context, error) {
	var c *C.hzsdr_ctx
	if rc := C.hzsdr_open(C.int(device), C.int(space), &c); rc != C.HZSDR_OK {
		return nil, toErr(nil, rc)
	}
	return &Context{c: c, space: space}, nil
}

// Close waits for the stream and frees everything the context owns.
func (x *Context) Close() error { return toErr(nil, C.hzsdr_close(x.c)) }

// MemSpace of this context.
func (x *Context) MemSpace() MemSpace { return MemSpace(C.hzsdr_memspace(x.c)) }

// SetStream makes the context enqueue on a caller-owned hipStream_t.
func (x *Context) SetStream(hipStream unsafe.Pointer) error {
	return toErr(x.c, C.hzsdr_set_stream(x.c, hipStream))
}

// UseOwnStream returns to the context's own stream.
func (x *Context) UseOwnStream() error { return toErr(x.c, C.hzsdr_use_own_stream(x.c)) }

// Stream is the hipStream_t the context enqueues on.
func (x *Context) Stream() unsafe.Pointer { return C.hzsdr_get_stream(x.c) }

// Synchronize waits for everything enqueued so far.
// CallCount reports how many calls of the library have run on this context so far (hzsdr_call_count): tests and logs
// count the calls a Reader pipeline makes per sample with it.
func (x *Context) CallCount() (uint64, error) {
	var n C.ulonglong
	rc := C.hzsdr_call_count(x.c, &n)
	return uint64(n), toErr(x.c, rc)
}

func (x *Context) Synchronize() error { return toErr(x.c, C.hzsdr_synchronize(x.c)) }

// MallocDevice / FreeDevice: device memory for MemDevice contexts.
func (x *Context) MallocDevice(bytes int) (unsafe.Pointer, error) {
	var p unsafe.Pointer
	rc := C.hzsdr_malloc_device(x.c, C.size_t(bytes), &p)
	return p, toErr(x.c, rc)
}

func (x *Context) FreeDevice(p unsafe.Pointer) error { return toErr(x.c, C.hzsdr_free_device(x.c, p)) }

// MallocPinned / FreePinned: page-locked, GPU-visible host memory.  HOST-space calls on
// buffers inside such an allocation skip all staging.
func (x *Context) MallocPinned(bytes int) (unsafe.Pointer, error) {
	var p unsafe.Pointer
	rc := C.hzsdr_malloc_pinned(x.c, C.size_t(bytes), &p)
	return p, toErr(x.c, rc)
}

func (x *Context) FreePinned(p unsafe.Pointer) error { return toErr(x.c, C.hzsdr_free_pinned(x.c, p)) }

// MemcpyH2D / MemcpyD2H on the context's stream.
func (x *Context) MemcpyH2D(dstDevice unsafe.Pointer, src sdr.Samples) error {
	return toErr(x.c, C.hzsdr_memcpy_h2d(x.c, dstDevice, base(src), C.size_t(src.Size())))
}

func (x *Context) MemcpyD2H(dst sdr.Samples, srcDevice unsafe.Pointer) error {
	return toErr(x.c, C.hzsdr_memcpy_d2h(x.c, base(dst), srcDevice, C.size_t(dst.Size())))
}

// base is the address of a Samples buffer's first byte (iq_unsafe.go:34-60), nil if empty.
func base(s sdr.Samples) unsafe.Pointer {
	if s == nil || s.Length() == 0 {
		return nil
	}
	b, err := sdr.UnsafeSamplesAsBytes(s)
	if err != nil || len(b) == 0 {
		return nil
	}
	return unsafe.Pointer(&b[0])
}

func cbool(b bool) C.int {
	if b {
		return 1
	}
	return 0
}

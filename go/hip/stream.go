//go:build sdr.hip

package hip

// #include <hzsdr.h>
import "C"

import (
	"fmt"
	"unsafe"

	"hz.tools/rf"
	"hz.tools/sdr"
	"hz.tools/sdr/stream"
)

// ---- SamplesC64.Scale / Multiply / Add (iq_c64.go:122-136, internal/simd) ----------------

func (x *Context) Scale(buf sdr.SamplesC64, r float32) error {
	return toErr(x.c, C.hzsdr_scale(x.c, base(buf), C.size_t(len(buf)), C.float(r)))
}

func (x *Context) Multiply(buf sdr.SamplesC64, m complex64) error {
	return toErr(x.c, C.hzsdr_rotate(x.c, base(buf), C.size_t(len(buf)), C.float(real(m)), C.float(imag(m))))
}

// Add is SamplesC64.Add(b, dst): dst = a + b, lengths must match (internal/simd/add.go:33).
func (x *Context) Add(a, b, dst sdr.SamplesC64) error {
	return toErr(x.c, C.hzsdr_add(x.c, base(a), C.size_t(len(a)), base(b), C.size_t(len(b)), base(dst), C.size_t(len(dst))))
}

// Sum is addReader.Read's data path (stream/add.go:121-185): out = ((+0 + b0) + b1) + ...
func (x *Context) Sum(out sdr.Samples, bufs []sdr.Samples) error {
	ptrs := (*[16]unsafe.Pointer)(C.malloc(C.size_t(16 * unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(ptrs))
	for i, b := range bufs {
		if b.Format() != out.Format() {
			return sdr.ErrSampleFormatMismatch
		}
		ptrs[i] = base(b)
	}
	return toErr(x.c, C.hzsdr_sum(x.c, C.int(out.Format()), base(out), (*unsafe.Pointer)(unsafe.Pointer(ptrs)),
		C.int(len(bufs)), C.size_t(out.Length())))
}

// MultiplyTable is the u8 / i8 form of stream.Multiply (stream/multiply.go:91-238): a
// 65 536-entry table of rotated samples, rebuilt when the multiplier changes.
type MultiplyTable struct {
	x *Context
	t *C.hzsdr_rotlut
}

func (x *Context) NewMultiplyTable(f sdr.SampleFormat, m complex64) (*MultiplyTable, error) {
	var t *C.hzsdr_rotlut
	if rc := C.hzsdr_rotlut_create(x.c, C.int(f), C.float(real(m)), C.float(imag(m)), &t); rc != C.HZSDR_OK {
		return nil, toErr(x.c, rc)
	}
	return &MultiplyTable{x, t}, nil
}

func (t *MultiplyTable) SetMultiplier(m complex64) error {
	return toErr(t.x.c, C.hzsdr_rotlut_set_multiplier(t.t, C.float(real(m)), C.float(imag(m))))
}

func (t *MultiplyTable) Apply(buf sdr.Samples) error {
	return toErr(t.x.c, C.hzsdr_rotlut_apply(t.t, base(buf), C.size_t(buf.Length())))
}

func (t *MultiplyTable) Close() error { return toErr(t.x.c, C.hzsdr_rotlut_free(t.t)) }

// ---- stream.ShiftBuffer (stream/shifter.go:66-85) -------------------------------------------

// Shifter holds the closure's clock `ts` on the C side.
type Shifter struct {
	x *Context
	n *C.hzsdr_nco
}

func (x *Context) NewShifter(sampleRate uint) (*Shifter, error) {
	var n *C.hzsdr_nco
	if rc := C.hzsdr_nco_create(x.c, C.uint64_t(sampleRate), &n); rc != C.HZSDR_OK {
		return nil, toErr(x.c, rc)
	}
	return &Shifter{x, n}, nil
}

// ShiftBuffer has the reference closure's signature: func(rf.Hz, sdr.SamplesC64).
func (s *Shifter) ShiftBuffer(freq rf.Hz, buf sdr.SamplesC64) {
	C.hzsdr_nco_shift(s.n, C.double(float64(freq)), base(buf), C.size_t(len(buf)))
}

func (s *Shifter) Time() float64 {
	var ts C.double
	C.hzsdr_nco_get_time(s.n, &ts)
	return float64(ts)
}

func (s *Shifter) SetTime(ts float64) error { return toErr(s.x.c, C.hzsdr_nco_set_time(s.n, C.double(ts))) }

// SetULP1 opts this Shifter (and the ShiftReader built on it) in to the rotation factor within one float32
// ulp of the reference's instead of bit-identical to it: HBM-bound instead of Sincos-bound. Off by default.
func (s *Shifter) SetULP1(on bool) error { return toErr(s.x.c, C.hzsdr_nco_set_ulp1(s.n, cbool(on))) }
func (s *Shifter) Close() error            { return toErr(s.x.c, C.hzsdr_nco_free(s.n)) }

// ClockSegment is one exactly-linear run of the reference's serial clock recurrence
// (ts += 1/fs, wrap at 2 pi): ts(first + i) = t0 + i*step exactly for i < count.
type ClockSegment struct {
	First, Count uint64
	T0, Step     float64
}

// ClockSegments plans `n` clock values from tsStart; pure host math.
func ClockSegments(sampleRate uint64, tsStart float64, n uint64) (segs []ClockSegment, tsEnd float64, err error) {
	var need C.size_t
	var end C.double
	buf := make([]C.hzsdr_nco_segment, 64)
	for {
		rc := C.hzsdr_nco_segments(C.uint64_t(sampleRate), C.double(tsStart), C.uint64_t(n), &buf[0], C.size_t(len(buf)), &need, &end)
		if rc != C.HZSDR_OK {
			return nil, 0, toErr(nil, rc)
		}
		if int(need) <= len(buf) {
			break
		}
		buf = make([]C.hzsdr_nco_segment, int(need))
	}
	for _, s := range buf[:int(need)] {
		segs = append(segs, ClockSegment{uint64(s.first), uint64(s.count), float64(s.t0), float64(s.step)})
	}
	return segs, float64(end), nil
}

// ---- DecimateBuffer / DownsampleBuffer (stream/decimate.go:59-101, downsample.go:68-127) -----

func (x *Context) DecimateBuffer(to, from sdr.Samples, factor uint, offset int) (int, error) {
	var n C.size_t
	rc := C.hzsdr_decimate(x.c, C.int(to.Format()), base(to), C.size_t(to.Length()), C.int(from.Format()), base(from),
		C.size_t(from.Length()), C.uint(factor), C.int64_t(offset), &n)
	return int(n), toErr(x.c, rc)
}

func (x *Context) DownsampleBuffer(to, from sdr.Samples, factor uint, offset int) (int, error) {
	var n C.size_t
	rc := C.hzsdr_downsample(x.c, C.int(to.Format()), base(to), C.size_t(to.Length()), C.int(from.Format()), base(from),
		C.size_t(from.Length()), C.uint(factor), C.int64_t(offset), &n)
	return int(n), toErr(x.c, rc)
}

// ---- Readers: the reference's constructors with Proc swapped (stream/read_transformer.go:92-116) --

// ConvertReader is stream.ConvertReader (stream/convert.go:30-57) on the GPU.
func (x *Context) ConvertReader(in sdr.Reader, to sdr.SampleFormat) (sdr.Reader, error) {
	return stream.ReadTransformer(in, stream.ReadTransformerConfig{
		InputBufferLength:  32 * 1024, // stream/convert.go:43-44
		OutputBufferLength: 32 * 1024,
		OutputSampleRate:   in.SampleRate(),
		OutputSampleFormat: to,
		Proc:               x.ConvertBuffer,
	})
}

// ConvertWriter is stream.ConvertWriter (stream/convert.go:58-118) on the GPU: a Writer of
// inputFormat in front of `out`; every Write is converted in chunks of 32 Ki samples
// (stream/convert.go:67) into a buffer of out's format and handed on. Same errors as the
// reference: sdr.ErrSampleFormatMismatch for a Write of another format, the downstream
// Writer's error with the count written so far, "Conversion mismatch" if a chunk came back short.
func (x *Context) ConvertWriter(out sdr.Writer, inputFormat sdr.SampleFormat) (sdr.Writer, error) {
	bufSize := 32 * 1024
	buf, err := sdr.MakeSamples(out.SampleFormat(), bufSize)
	if err != nil {
		return nil, err
	}
	return &convWriter{x: x, out: out, inputFormat: inputFormat, buffer: buf}, nil
}

type convWriter struct {
	x           *Context
	out         sdr.Writer
	inputFormat sdr.SampleFormat
	buffer      sdr.Samples
}

func (cw *convWriter) Write(in sdr.Samples) (int, error) {
	if in.Format() != cw.inputFormat {
		return 0, sdr.ErrSampleFormatMismatch
	}
	bufSize := cw.buffer.Length()
	n := 0
	for i := 0; i < in.Length(); i += bufSize {
		ie := i + bufSize
		if ie > in.Length() {
			ie = in.Length()
		}
		leng, err := cw.x.ConvertBuffer(cw.buffer, in.Slice(i, ie))
		if err != nil {
			return n, err
		}
		if ie-i != leng {
			return n, fmt.Errorf("ConvertWriter: Conversion mismatch")
		}
		j, err := cw.out.Write(cw.buffer.Slice(0, leng))
		n += j
		if err != nil {
			return n, err
		}
	}
	return n, nil
}

func (cw *convWriter) SampleFormat() sdr.SampleFormat { return cw.inputFormat }
func (cw *convWriter) SampleRate() uint               { return cw.out.SampleRate() }

// Chain is nested stream.* Readers fused into one launch per buffer.
type Chain struct {
	x        *Context
	c        *C.hzsdr_chain
	inFormat sdr.SampleFormat
	rate     uint
	outRate  uint
}

func (x *Context) NewChain(f sdr.SampleFormat, sampleRate uint) (*Chain, error) {
	var c *C.hzsdr_chain
	if rc := C.hzsdr_chain_create(x.c, C.int(f), C.uint64_t(sampleRate), &c); rc != C.HZSDR_OK {
		return nil, toErr(x.c, rc)
	}
	return &Chain{x: x, c: c, inFormat: f, rate: sampleRate, outRate: sampleRate}, nil
}

func (ch *Chain) Shift(freq rf.Hz) error    { return toErr(ch.x.c, C.hzsdr_chain_shift(ch.c, C.double(float64(freq)))) }
func (ch *Chain) Gain(r float32) error      { return toErr(ch.x.c, C.hzsdr_chain_gain(ch.c, C.float(r))) }
func (ch *Chain) Multiply(m complex64) error { return toErr(ch.x.c, C.hzsdr_chain_rotate(ch.c, C.float(real(m)), C.float(imag(m)))) }

func (ch *Chain) Decimate(factor uint) error {
	ch.outRate = ch.rate / factor
	return toErr(ch.x.c, C.hzsdr_chain_decimate(ch.c, C.uint(factor)))
}

func (ch *Chain) Downsample(factor uint) error {
	ch.outRate = ch.rate / factor
	return toErr(ch.x.c, C.hzsdr_chain_downsample(ch.c, C.uint(factor)))
}

// Convolution is stream.ConvolutionReader's block-circular filter (stream/convolution.go:36-82).
func (ch *Chain) Convolution(filterFreq []complex64, decimate uint) error {
	ch.outRate = ch.rate / decimate
	return toErr(ch.x.c, C.hzsdr_chain_convolution(ch.c, unsafe.Pointer(&filterFreq[0]), C.size_t(len(filterFreq)), C.uint(decimate)))
}

// FIRDecimate: overlap-save FIR with history across buffers, then decimation (north star).
func (ch *Chain) FIRDecimate(taps []complex64, factor uint) error {
	ch.outRate = ch.rate / factor
	return toErr(ch.x.c, C.hzsdr_chain_fir_decimate(ch.c, (*C.float)(unsafe.Pointer(&taps[0])), C.size_t(len(taps)), C.uint(factor)))
}

func (ch *Chain) MixInOrder(inOrder bool) error { return toErr(ch.x.c, C.hzsdr_chain_mix_in_order(ch.c, cbool(inOrder))) }

// FIROptions selects, in front of FIRDecimate, which implementation the terminal takes (0: the library's choice,
// 1: the overlap-save transform kernels, 2: the int8 matrix form as chunk workgroups), the smallest overlap-save
// block and the matrix loop's form: a measurement and test aid (include/hzsdr.h: hzsdr_chain_fir_options).
func (ch *Chain) FIROptions(impl int, nfftMin uint, loopForm int) error {
	return toErr(ch.x.c, C.hzsdr_chain_fir_options(ch.c, C.int(impl), C.uint(nfftMin), C.int(loopForm)))
}

// Pipeline opts a FIR-decimate chain on the int8 matrix path (or a chain of maps) in to overlapping consecutive calls
// (include/hzsdr.h: hzsdr_chain_pipeline).  Run stays an ordinary call on the context's stream; the overlap is taken by
// RunAfter / RunBatchAfter, where the caller states what the call's buffers wait for.  Results are bit-identical.
func (ch *Chain) Pipeline(on bool) error { return toErr(ch.x.c, C.hzsdr_chain_pipeline(ch.c, cbool(on))) }

// ShiftULP1 opts a terminal-less chain (ShiftReader, ShiftReader -> Gain) in to the Shift whose rotation
// factor is within one float32 ulp of the reference's instead of bit-identical to it (include/hzsdr.h).
func (ch *Chain) ShiftULP1(on bool) error { return toErr(ch.x.c, C.hzsdr_chain_shift_ulp1(ch.c, cbool(on))) }

func (ch *Chain) Plan(nIn int) (consumed, out int, err error) {
	var a, b C.size_t
	rc := C.hzsdr_chain_plan(ch.c, C.size_t(nIn), &a, &b)
	return int(a), int(b), toErr(ch.x.c, rc)
}

// Run has the signature of ReadTransformerConfig.Proc.
func (ch *Chain) Run(in, out sdr.Samples) (int, error) {
	var cons, n C.size_t
	rc := C.hzsdr_chain_run(ch.c, base(in), C.size_t(in.Length()), base(out), C.size_t(out.Length()), &cons, &n)
	return int(n), toErr(ch.x.c, rc)
}

// RunAfter is Run whose START the caller orders: the buffers are ready when `ready` (a hipEvent_t; nil: now) has
// fired, not "when the context's stream gets there" (include/hzsdr.h: hzsdr_chain_run_after).  The call's reads and
// writes are ordered on the context's stream as Run's are.
func (ch *Chain) RunAfter(in, out sdr.Samples, ready unsafe.Pointer) (int, error) {
	var cons, n C.size_t
	rc := C.hzsdr_chain_run_after(ch.c, base(in), C.size_t(in.Length()), base(out), C.size_t(out.Length()), &cons, &n, ready)
	return int(n), toErr(ch.x.c, rc)
}

// RunBatch hands len(ins) consecutive buffers of the stream (equal lengths, at most eight) to the chain in one call:
// the results and the chain's state are those of len(ins) Run calls in a row; a FIR-decimate chain on the
// persistent-pass matrix kernel takes them in ONE launch (include/hzsdr.h: hzsdr_chain_run_batch).  Returns the
// samples written per buffer.  The pointer tables live in C memory for the call: cgo may not be handed Go memory
// that itself holds Go pointers.
func (ch *Chain) RunBatch(ins, outs []sdr.Samples) (int, error) { return ch.runBatch(ins, outs, false, nil) }

// RunBatchAfter is RunBatch with RunAfter's ordering (every buffer of the batch ready at the event).
func (ch *Chain) RunBatchAfter(ins, outs []sdr.Samples, ready unsafe.Pointer) (int, error) {
	return ch.runBatch(ins, outs, true, ready)
}

func (ch *Chain) runBatch(ins, outs []sdr.Samples, after bool, ready unsafe.Pointer) (int, error) {
	k := len(ins)
	if k == 0 || k != len(outs) {
		return 0, sdr.ErrDstTooSmall
	}
	psz := C.size_t(unsafe.Sizeof(unsafe.Pointer(nil)))
	pi := (*[8]unsafe.Pointer)(C.malloc(C.size_t(k) * psz))
	po := (*[8]unsafe.Pointer)(C.malloc(C.size_t(k) * psz))
	defer C.free(unsafe.Pointer(pi))
	defer C.free(unsafe.Pointer(po))
	outCap := outs[0].Length()
	for j := 0; j < k && j < 8; j++ {
		if ins[j].Length() != ins[0].Length() {
			return 0, sdr.ErrDstTooSmall
		}
		if outs[j].Length() < outCap {
			outCap = outs[j].Length()
		}
		pi[j], po[j] = base(ins[j]), base(outs[j])
	}
	var cons, n C.size_t
	var rc C.int
	if after {
		rc = C.hzsdr_chain_run_batch_after(ch.c, (*unsafe.Pointer)(unsafe.Pointer(pi)), (*unsafe.Pointer)(unsafe.Pointer(po)), C.size_t(k),
			C.size_t(ins[0].Length()), C.size_t(outCap), &cons, &n, ready)
	} else {
		rc = C.hzsdr_chain_run_batch(ch.c, (*unsafe.Pointer)(unsafe.Pointer(pi)), (*unsafe.Pointer)(unsafe.Pointer(po)), C.size_t(k),
			C.size_t(ins[0].Length()), C.size_t(outCap), &cons, &n)
	}
	return int(n), toErr(ch.x.c, rc)
}

func (ch *Chain) Reset() error             { return toErr(ch.x.c, C.hzsdr_chain_reset(ch.c)) }
func (ch *Chain) SetTime(ts float64) error { return toErr(ch.x.c, C.hzsdr_chain_set_time(ch.c, C.double(ts))) }
func (ch *Chain) Time() (float64, error) {
	var ts C.double
	rc := C.hzsdr_chain_time(ch.c, &ts)
	return float64(ts), toErr(ch.x.c, rc)
}

// LastFIRPath reports which kernels the last Run of a FIR-decimate chain used
// (C.HZSDR_FIR_PATH_TRANSFORM or C.HZSDR_FIR_PATH_MATRIX; C.HZSDR_FIR_PATH_NONE before the first run).
func (ch *Chain) LastFIRPath() (int, error) {
	var p C.int
	rc := C.hzsdr_chain_last_fir_path(ch.c, &p)
	return int(p), toErr(ch.x.c, rc)
}

// LastFIRKernel reports which kernel that was (C.HZSDR_FIR_KERNEL_TRANSFORM, _MATRIX_CHUNKS, _MATRIX_PASSES;
// C.HZSDR_FIR_KERNEL_NONE before the first run): for logs and benchmarks.
func (ch *Chain) LastFIRKernel() (int, error) {
	var k C.int
	rc := C.hzsdr_chain_last_fir_kernel(ch.c, &k)
	return int(k), toErr(ch.x.c, rc)
}
func (ch *Chain) Close() error { return toErr(ch.x.c, C.hzsdr_chain_free(ch.c)) }

// Reader wraps the chain as ONE sdr.Reader behind stream.ReadTransformer: `block` input
// samples per launch (a whole number of the chain's blocks; 1 << 20 keeps the GPU busy).
func (ch *Chain) Reader(in sdr.Reader, block int) (sdr.Reader, error) {
	cons, outN, err := ch.Plan(block)
	if err != nil {
		return nil, err
	}
	return stream.ReadTransformer(in, stream.ReadTransformerConfig{
		InputBufferLength:  cons,
		OutputBufferLength: outN,
		OutputSampleRate:   ch.outRate,
		OutputSampleFormat: sdr.SampleFormatC64,
		Proc:               ch.Run,
	})
}

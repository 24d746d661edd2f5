//go:build sdr.hip

package hip

// #include <hzsdr.h>
import "C"

import (
	"fmt"
	"unsafe"

	"hz.tools/rf"
	"hz.tools/sdr"
	"hz.tools/sdr/fft"
	"hz.tools/sdr/stream"
	"hz.tools/sdr/yikes"
)

// Readers has the reference's stream.* Reader constructors, name for name and argument for
// argument, over one GPU context:
//
//	s := ctx.Readers()
//	r, err := s.ShiftReader(r, 100*rf.KHz)   // stream.ShiftReader(r, 100*rf.KHz)
//	r = s.Gain(r, 0.5)                       // stream.Gain(r, 0.5)
//
// so a caller swaps `stream.` for `s.` and keeps the rest of its pipeline. Each Reader does what
// the reference's does in the same order (same format checks, same sentinel errors, same
// pull-style Read for Shift / Gain / Multiply / Add, the same ReadTransformer scaffold with
// 32 Ki-sample blocks for Decimate / Downsample and filter-length blocks for Convolution);
// only the buffer arithmetic runs on the GPU. A 32 Ki block is launch-bound (INTEGRATION.md,
// "What the drop-in costs"), so by default the constructors FUSE: handed a Reader that one of them made, they extend
// its chain instead of wrapping it, and the resulting Reader reads ahead through a pinned ring -- one launch per
// ReadAhead x 32 Ki samples for the whole nest (fused.go: what is kept of the reference's semantics, what differs).
// ReadersUnfused keeps the reference's own structure, one call per block and stage.
type Readers struct {
	x         *Context
	fuse      bool
	readAhead int // Reader blocks of 32 Ki samples per slot of a fused Reader's ring
}

// Readers returns the constructor set bound to this context: nested Readers fuse and read 32 blocks (2^20 samples) ahead.
func (x *Context) Readers() Readers { return Readers{x: x, fuse: true, readAhead: 32} }

// ReadersUnfused returns the constructors in the reference's own structure: a ReadTransformer per block-structured
// stage, a wrapper per pass-through stage, one GPU call per 32 Ki-sample block and stage.
func (x *Context) ReadersUnfused() Readers { return Readers{x: x} }

// ReadAhead sets how many 32 Ki-sample Reader blocks a fused Reader reads ahead of its consumer (at least 1).
func (s Readers) ReadAhead(blocks int) Readers {
	if blocks < 1 {
		blocks = 1
	}
	s.readAhead = blocks
	return s
}

// ConvertReader is stream.ConvertReader (stream/convert.go:30-57); to c64 it opens (or joins) a fused chain.
func (s Readers) ConvertReader(in sdr.Reader, to sdr.SampleFormat) (sdr.Reader, error) {
	if cr := s.fused(in, func(c *chainReader) bool { return c.extendConvert(to) }); cr != nil {
		return cr, nil
	}
	return s.x.ConvertReader(in, to)
}

// ---- stream.ShiftReader (stream/shifter.go:89-102) ----------------------------------------

type shiftReader struct {
	r     sdr.Reader
	shift rf.Hz
	sh    *Shifter
}

func (sr *shiftReader) SampleFormat() sdr.SampleFormat { return sr.r.SampleFormat() }
func (sr *shiftReader) SampleRate() uint               { return sr.r.SampleRate() }

// Read is shiftReader.Read (stream/shifter.go:44-64): c64 only, the clock carries on from
// buffer to buffer.
func (sr *shiftReader) Read(s sdr.Samples) (int, error) {
	if s.Format() != sdr.SampleFormatC64 {
		return 0, sdr.ErrSampleFormatUnknown
	}
	n, err := sr.r.Read(s)
	if err != nil {
		return n, err
	}
	sr.sh.ShiftBuffer(sr.shift, s.Slice(0, n).(sdr.SamplesC64))
	return n, nil
}

// Close releases the clock held on the C side (the reference's closure is garbage collected).
func (sr *shiftReader) Close() error { return sr.sh.Close() }

// ShiftReader is stream.ShiftReader (stream/shifter.go:89).
func (s Readers) ShiftReader(r sdr.Reader, shift rf.Hz) (sdr.Reader, error) {
	if r.SampleFormat() != sdr.SampleFormatC64 {
		return nil, sdr.ErrSampleFormatUnknown
	}
	if cr := s.fused(r, func(c *chainReader) bool { return c.extendStage(chainStage{kind: 0, shift: shift}) }); cr != nil {
		return cr, nil
	}
	sh, err := s.x.NewShifter(r.SampleRate())
	if err != nil {
		return nil, err
	}
	return &shiftReader{r: r, shift: shift, sh: sh}, nil
}

// ---- stream.Gain (stream/gain.go:30-64) -------------------------------------------------

type gain struct {
	x *Context
	v float32
	r sdr.Reader
}

func (g *gain) SampleFormat() sdr.SampleFormat { return g.r.SampleFormat() }
func (g *gain) SampleRate() uint               { return g.r.SampleRate() }

// Scale is gain.Scale (stream/gain.go:39-48): c64 only.
func (g *gain) Scale(s sdr.Samples) error {
	c, ok := s.(sdr.SamplesC64)
	if !ok {
		return sdr.ErrSampleFormatUnknown
	}
	return g.x.Scale(c, g.v)
}

func (g *gain) Read(s sdr.Samples) (int, error) {
	i, err := g.r.Read(s)
	if err != nil {
		return i, err
	}
	return i, g.Scale(s.Slice(0, i))
}

// Gain is stream.Gain (stream/gain.go:30).
func (s Readers) Gain(r sdr.Reader, v float32) sdr.Reader {
	if r.SampleFormat() == sdr.SampleFormatC64 {
		if cr := s.fused(r, func(c *chainReader) bool { return c.extendStage(chainStage{kind: 1, gain: v}) }); cr != nil {
			return cr
		}
	}
	return &gain{x: s.x, v: v, r: r}
}

// ---- stream.Multiply (stream/multiply.go:27-238) ------------------------------------------

// MultiplyReader is what Multiply returns: an sdr.Reader with the reference's undocumented
// SetMultiplier (stream/multiply.go:34-36; Beamform.SetPhaseAngles calls it).
type MultiplyReader interface {
	sdr.Reader
	SetMultiplier(m complex64)
}

type multiplyReader struct {
	x *Context
	m complex64
	r sdr.Reader
}

func (mr *multiplyReader) SetMultiplier(m complex64)      { mr.m = m }
func (mr *multiplyReader) SampleFormat() sdr.SampleFormat { return mr.r.SampleFormat() }
func (mr *multiplyReader) SampleRate() uint               { return mr.r.SampleRate() }

// Read is multiplyReader.Read (stream/multiply.go:46-70), the m == 1 short cut included.
func (mr *multiplyReader) Read(s sdr.Samples) (int, error) {
	if s.Format() != sdr.SampleFormatC64 {
		return 0, sdr.ErrSampleFormatMismatch
	}
	i, err := mr.r.Read(s)
	if err != nil {
		return i, err
	}
	if mr.m == 1 {
		return i, nil
	}
	return i, mr.x.Multiply(s.Slice(0, i).(sdr.SamplesC64), mr.m)
}

// tableMultiplyReader is uint8MultiplyReader / int8MultiplyReader (stream/multiply.go:91-238):
// the 65 536-entry table of rotated samples lives on the GPU, rebuilt by SetMultiplier.
type tableMultiplyReader struct {
	t *MultiplyTable
	r sdr.Reader
	// SetMultiplier has no error in the reference's MultiplyReader interface (stream/multiply.go:27-35);
	// a failed rebuild of the GPU table would leave the OLD weights in place without a word, so it is
	// kept and returned by every Read until a later SetMultiplier succeeds.
	setErr error
}

func (tr *tableMultiplyReader) SetMultiplier(m complex64)      { tr.setErr = tr.t.SetMultiplier(m) }
func (tr *tableMultiplyReader) SampleFormat() sdr.SampleFormat { return tr.r.SampleFormat() }
func (tr *tableMultiplyReader) SampleRate() uint               { return tr.r.SampleRate() }
func (tr *tableMultiplyReader) Close() error                   { return tr.t.Close() }

func (tr *tableMultiplyReader) Read(s sdr.Samples) (int, error) {
	if s.Format() != tr.r.SampleFormat() {
		return 0, sdr.ErrSampleFormatMismatch
	}
	if tr.setErr != nil {
		return 0, tr.setErr
	}
	i, err := tr.r.Read(s)
	if err != nil {
		return i, err
	}
	return i, tr.t.Apply(s.Slice(0, i))
}

// Multiply is stream.Multiply (stream/multiply.go:74): c64 by arithmetic, u8 / i8 by table.
func (s Readers) Multiply(r sdr.Reader, m complex64) (sdr.Reader, error) {
	switch r.SampleFormat() {
	case sdr.SampleFormatI8, sdr.SampleFormatU8:
		t, err := s.x.NewMultiplyTable(r.SampleFormat(), m)
		if err != nil {
			return nil, err
		}
		return &tableMultiplyReader{t: t, r: r}, nil
	case sdr.SampleFormatC64:
		// (a multiplier of exactly 1 is the reference's short cut, stream/multiply.go:59-62 -- and ReadBeamform's
		// first weight: such a Reader stays a wrapper, it has nothing to launch)
		if m != 1 {
			if cr := s.fused(r, func(c *chainReader) bool { return c.extendStage(chainStage{kind: 2, mult: m}) }); cr != nil {
				return cr, nil
			}
		}
		return &multiplyReader{x: s.x, r: r, m: m}, nil
	default:
		return nil, sdr.ErrSampleFormatUnknown
	}
}

// ---- stream.Add (stream/add.go:41-185) ----------------------------------------------------

type addReader struct {
	x            *Context
	sampleFormat sdr.SampleFormat
	sampleRate   uint
	readers      []sdr.Reader
	err          error
	// The K temporaries of a Read: the reference allocates them anew with sdr.MakeSamples on every call
	// (stream/add.go:133-141) -- pageable memory, so each costs the GPU path a staging copy as well.  Here they are
	// allocated ONCE, pinned (hzsdr_malloc_pinned: the kernel reads them in place), and grow with the largest Read.
	buffers []sdr.Samples
	pinned  []unsafe.Pointer
	bufLen  int
}

// Close releases the pinned temporaries (the reference's are garbage collected).
func (ar *addReader) Close() error {
	for _, p := range ar.pinned {
		_ = ar.x.FreePinned(p)
	}
	ar.pinned, ar.buffers, ar.bufLen = nil, nil, 0
	return nil
}

func (ar *addReader) SampleFormat() sdr.SampleFormat { return ar.sampleFormat }
func (ar *addReader) SampleRate() uint               { return ar.sampleRate }

// Read is addReader.Read (stream/add.go:121-185): ReadFull of every reader in order, then
// out = ((+0 + b0) + b1) + ... in ONE kernel (hzsdr_sum) instead of K + 1 passes; sticky errors.
func (ar *addReader) Read(s sdr.Samples) (int, error) {
	if ar.err != nil {
		return 0, ar.err
	}
	switch s.Format() {
	case sdr.SampleFormatC64, sdr.SampleFormatI16, sdr.SampleFormatI8:
	default:
		return 0, sdr.ErrSampleFormatUnknown
	}
	if ar.buffers == nil || ar.bufLen < s.Length() {
		_ = ar.Close()
		ar.buffers = make([]sdr.Samples, len(ar.readers))
		for i := range ar.readers {
			p, err := ar.x.MallocPinned(s.Format().Size() * s.Length())
			if err != nil {
				ar.err = err
				return 0, err
			}
			ar.pinned = append(ar.pinned, p)
			b, err := yikes.Samples(uintptr(p), s.Length(), s.Format()) // yikes/bytes.go:50-71: C-owned memory as sdr.Samples
			if err != nil {
				ar.err = err
				return 0, err
			}
			ar.buffers[i] = b
		}
		ar.bufLen = s.Length()
	}
	buffers := make([]sdr.Samples, len(ar.readers))
	for i, reader := range ar.readers {
		buffers[i] = ar.buffers[i].Slice(0, s.Length())
		if _, err := sdr.ReadFull(reader, buffers[i]); err != nil {
			ar.err = err
			return 0, err
		}
	}
	if err := ar.x.Sum(s, buffers); err != nil {
		ar.err = err
		return 0, err
	}
	return s.Length(), nil
}

// Add is stream.Add (stream/add.go:41): c64, i16 or i8 readers of one format and rate; at most
// sixteen of them on the GPU (hzsdr_sum).
func (s Readers) Add(readers ...sdr.Reader) (sdr.Reader, error) {
	switch len(readers) {
	case 0:
		return nil, fmt.Errorf("stream.Add: No readers passed")
	case 1:
		return readers[0], nil
	}
	if len(readers) > 16 {
		return nil, fmt.Errorf("hip.Add: at most 16 readers")
	}
	sampleFormat, sampleRate := readers[0].SampleFormat(), readers[0].SampleRate()
	switch sampleFormat {
	case sdr.SampleFormatC64, sdr.SampleFormatI16, sdr.SampleFormatI8:
	default:
		return nil, sdr.ErrSampleFormatUnknown
	}
	for _, reader := range readers {
		if reader.SampleFormat() != sampleFormat {
			return nil, fmt.Errorf("stream.Add: Readers are not all the same format")
		}
		if reader.SampleRate() != sampleRate {
			return nil, fmt.Errorf("stream.Add: Readers are not all the same rate")
		}
	}
	return &addReader{x: s.x, sampleFormat: sampleFormat, sampleRate: sampleRate, readers: readers}, nil
}

// ---- stream.DecimateReader / DownsampleReader (stream/decimate.go:34-57, downsample.go:47-66) ----

// DecimateReader is stream.DecimateReader: 32 Ki-sample blocks, the offset counted (and, as in
// the reference, ignored by DecimateBuffer: the phase restarts with every block).
func (s Readers) DecimateReader(in sdr.Reader, factor uint) (sdr.Reader, error) {
	// (the chain's stream is c64: DecimateReader keeps its input's format, so only a c64 stream joins a chain)
	if in.SampleFormat() == sdr.SampleFormatC64 && factor > 0 {
		if cr := s.fused(in, func(c *chainReader) bool { return c.extendTerminal(chainTerm{kind: 1, factor: factor}, readerBlock) }); cr != nil {
			return cr, nil
		}
	}
	offset := 0
	return stream.ReadTransformer(in, stream.ReadTransformerConfig{
		InputBufferLength:  32 * 1024,
		OutputBufferLength: 32 * 1024,
		OutputSampleRate:   in.SampleRate() / factor,
		OutputSampleFormat: in.SampleFormat(),
		Proc: func(inBuf sdr.Samples, outBuf sdr.Samples) (int, error) {
			n, err := s.x.DecimateBuffer(outBuf, inBuf, factor, offset)
			offset += inBuf.Length()
			return n, err
		},
	})
}

// FirDecimateReader is the north-star terminal as an sdr.Reader: an N-tap FIR at the input rate whose output is kept
// every `factor` samples (BASELINE.json north_star; the reference has no such Reader -- its Downsample is the boxcar,
// stream/downsample.go:47-64 -- so name and signature follow DecimateReader's, stream/decimate.go:34). Always a fused
// Reader: ConvertReader / ShiftReader / Gain / Multiply in front of it join its chain (one kernel per call: for a
// u8 / i8 source at factor 8 or 16 the int8 matrix kernel), `slots` slots in the pinned ring, `group` of them per call
// of the chain (hzsdr_ring_submit_many: one launch each time). slots, group <= 0: nine slots, four per call.
func (s Readers) FirDecimateReader(in sdr.Reader, taps []complex64, factor uint, slots, group int) (sdr.Reader, error) {
	if len(taps) == 0 || factor == 0 {
		return nil, fmt.Errorf("hip.FirDecimateReader: %d taps, factor %d", len(taps), factor)
	}
	if slots <= 0 {
		slots = 9
	}
	if group <= 0 {
		group = 4
	}
	how := func(c *chainReader) bool { return c.extendFir(taps, factor, slots, group) }
	if cr, ok := in.(*chainReader); ok && how(cr) {
		return cr, nil
	}
	cr := s.newChainReader(in)
	if how(cr) {
		return cr, nil
	}
	return nil, fmt.Errorf("hip.FirDecimateReader: the stage does not fit the Reader in front of it")
}

// DownsampleReader is stream.DownsampleReader: the boxcar mean of `factor` samples, c64 out.
func (s Readers) DownsampleReader(in sdr.Reader, factor uint) (sdr.Reader, error) {
	switch in.SampleFormat() {
	case sdr.SampleFormatC64, sdr.SampleFormatU8, sdr.SampleFormatI16:
		if factor > 0 {
			if cr := s.fused(in, func(c *chainReader) bool {
				if !c.c64Here() && c.open() && len(c.stages) == 0 {
					c.converted = true // DownsampleBuffer converts by itself (stream/downsample.go:99-124): so does the chain
				}
				return c.extendTerminal(chainTerm{kind: 2, factor: factor}, readerBlock)
			}); cr != nil {
				return cr, nil
			}
		}
	}
	offset := 0
	return stream.ReadTransformer(in, stream.ReadTransformerConfig{
		InputBufferLength:  32 * 1024,
		OutputBufferLength: 32 * 1024,
		OutputSampleRate:   in.SampleRate() / factor,
		OutputSampleFormat: sdr.SampleFormatC64,
		Proc: func(inBuf sdr.Samples, outBuf sdr.Samples) (int, error) {
			n, err := s.x.DownsampleBuffer(outBuf, inBuf, factor, offset)
			offset += inBuf.Length()
			return n, err
		},
	})
}

// ---- stream.ConvolutionReader (stream/convolution.go:36-82) -------------------------------

// ConvolutionReader is stream.ConvolutionReader: block-circular filtering with blocks of
// len(filter) samples, the filter given in the frequency domain. `planner` keeps the
// reference's signature and is not called: forward transform, bin product and backward
// transform of a block are one kernel (hzsdr_convolution_blocks); pass ctx.Planner or nil.
func (s Readers) ConvolutionReader(r sdr.Reader, planner fft.Planner, filter []complex64) (sdr.Reader, error) {
	_ = planner
	if r.SampleFormat() != sdr.SampleFormatC64 {
		return nil, sdr.ErrSampleFormatUnknown
	}
	fftLength := len(filter)
	// hzsdr_convolution_blocks takes any block length the Planner takes -- powers of two up to 2^24 on the
	// power-of-two kernels (4 ... 8192 bins: forward transform, product and backward transform in ONE kernel), any
	// other length up to 2^23 by Bluestein's chirp transform over them.  The reference fails at construction when its
	// planner refuses a length (fft.ConvolveFreq, fft/convolution.go:150-170); so does this.
	if fftLength < 1 || (fftLength&(fftLength-1) == 0 && fftLength > 1<<24) || (fftLength&(fftLength-1) != 0 && fftLength > 1<<23) {
		return nil, fmt.Errorf("hip.ConvolutionReader: filter length %d: 1 ... 2^24 (a power of two) or 1 ... 2^23 (any other)", fftLength)
	}
	if cr := s.fused(r, func(c *chainReader) bool {
		return c.extendTerminal(chainTerm{kind: 3, filter: append([]complex64(nil), filter...), decimate: 1}, fftLength)
	}); cr != nil {
		return cr, nil
	}
	return stream.ReadTransformer(r, stream.ReadTransformerConfig{
		InputBufferLength:  fftLength,
		OutputBufferLength: fftLength,
		OutputSampleFormat: sdr.SampleFormatC64,
		OutputSampleRate:   r.SampleRate(),
		Proc: func(inI sdr.Samples, outI sdr.Samples) (int, error) {
			in, ok := inI.(sdr.SamplesC64)
			if !ok {
				return 0, sdr.ErrSampleFormatUnknown
			}
			out, ok := outI.(sdr.SamplesC64)
			if !ok {
				return 0, sdr.ErrSampleFormatUnknown
			}
			return s.x.ConvolutionBlocks(out[:in.Length()], in, filter)
		},
	})
}

// ReadersUnfusedHere: the same context's constructors without fusion (ReadBeamform composes ConvertReader -> Multiply ->
// Add per channel and SetPhaseAngles must take effect at the next Read: stream/beamform.go:131-139).
func (s Readers) ReadersUnfusedHere() Readers { return Readers{x: s.x} }

// ---- stream.ReadBeamform (stream/beamform.go:131-171) -------------------------------------

// Beamform is stream.Beamform: the weighted sum of coherent readers as one sdr.Reader.
type Beamform struct {
	sdr.Reader
	readers sdr.Readers
	config  stream.BeamformConfig
}

// SetPhaseAngles is Beamform.SetPhaseAngles (stream/beamform.go:131-139): applied between reads.
func (b *Beamform) SetPhaseAngles(angles []complex64) error {
	if len(angles) != len(b.readers) {
		return fmt.Errorf("Beamform.SetPhaseAngles: angles must match the reader length")
	}
	for i, reader := range b.readers {
		reader.(MultiplyReader).SetMultiplier(angles[i])
	}
	return nil
}

// ReadBeamform is stream.ReadBeamform (stream/beamform.go:148): ConvertReader to c64, Multiply
// by the channel's weight, Add -- the reference's own composition over the GPU Readers, so the
// order of the sum (and its +0 start) is the reference's. (One fused kernel for the whole
// thing: Context.Beamform; several GPUs: MultiGPU.Beamform.)
func (s Readers) ReadBeamform(rs sdr.Readers, cfg stream.BeamformConfig) (*Beamform, error) {
	multReaders := make(sdr.Readers, len(rs))
	for i := range rs {
		reader, err := s.ReadersUnfusedHere().ConvertReader(rs[i], sdr.SampleFormatC64)
		if err != nil {
			return nil, err
		}
		multReaders[i], err = s.ReadersUnfusedHere().Multiply(reader, 1)
		if err != nil {
			return nil, err
		}
	}
	addReader, err := s.Add(multReaders...)
	if err != nil {
		return nil, err
	}
	b := &Beamform{Reader: addReader, readers: multReaders, config: cfg}
	// stream/beamform.go:169 calls SetPhaseAngles and drops its error: a BeamformConfig whose Angles do
	// not match the readers (the zero value, say) still yields a *Beamform with every weight 1.
	_ = b.SetPhaseAngles(cfg.Angles)
	return b, nil
}

//go:build sdr.hip

package hip

// #include <hzsdr.h>
import "C"

import (
	"fmt"
	"io"

	"hz.tools/rf"
	"hz.tools/sdr"
	"hz.tools/sdr/stream"
)

// chainReader is nested stream.* Readers of this package as ONE sdr.Reader. The constructors of Readers do not wrap a
// Reader that is already a chainReader: they extend its chain -- ConvertReader -> ShiftReader -> Gain -> Multiply ->
// DecimateReader / DownsampleReader / ConvolutionReader (-> DecimateReader) collapse into one hzsdr_chain, one launch
// per slot -- and the Reader reads AHEAD: ReadAhead Reader blocks of 32 Ki samples per slot of a pinned ring
// (hzsdr_ring_*: the source reads straight into pinned memory; upload, kernel and download of neighbouring slots
// overlap). The reference's Readers make one call per 32 Ki-sample block (stream/convert.go:43-44,
// stream/decimate.go:41-42); a GPU call of that size is all latency (bench.py small_buffers: no faster than one core).
//
// What the reference's nesting means is kept:
//   - the samples are the nested Readers' bit for bit: the chain's stages are the reference's operations in the
//     reference's order, and the block-structured stages (ConvertReader, DecimateReader, DownsampleReader: 32 Ki
//     blocks; ConvolutionReader: len(filter)) see the same blocks of the same stream whatever the slot size is;
//   - a block-structured stage hands out whole blocks only: a source that ends inside a block loses that partial
//     block, as sdr.ReadFull's ErrUnexpectedEOF does in stream/read_transformer.go:120-135; pass-through stages
//     (ShiftReader, Gain, Multiply over a c64 source) hand out whatever the source delivered;
//   - an error of the source is sticky and surfaces when everything read before it has been handed out
//     (io.ErrUnexpectedEOF becomes io.EOF, as the pipe's CloseWithError makes it).
//
// What differs: the source is read up to ReadAhead blocks ahead of the consumer (the reference reads one), and a
// Multiply's SetMultiplier takes effect from the next slot filled, not the next Read.
type chainReader struct {
	s   Readers
	src sdr.Reader

	srcFormat sdr.SampleFormat
	converted bool // a ConvertReader(.., c64) (or DownsampleReader's own conversion) is part of the chain
	rate      uint
	stages    []chainStage
	term      chainTerm
	block     int // the stream is consumed in whole multiples of this many samples (1: any)

	slots int // slots of the pinned ring (3 unless FirDecimateReader says otherwise)
	group int // slots handed to the chain per call (hzsdr_ring_submit_many): 1, or FirDecimateReader's

	ch       *Chain
	ring     *Ring
	slotLen  int
	inflight int
	pending  sdr.SamplesC64
	queue    []sdr.SamplesC64 // outputs kept across a rebuild (SetMultiplier)
	err      error
}

type chainStage struct {
	kind  int // 0 Shift, 1 Gain, 2 Multiply
	shift rf.Hz
	gain  float32
	mult  complex64
}

type chainTerm struct {
	kind     int // 0 none, 1 Decimate, 2 Downsample, 3 Convolution, 4 the north-star FIR-decimate
	factor   uint
	filter   []complex64 // kind 3: the filter's bins; kind 4: the taps
	decimate uint
}

const readerBlock = 32 * 1024

func gcd(a, b int) int {
	for b != 0 {
		a, b = b, a%b
	}
	return a
}
func lcm(a, b int) int { return a / gcd(a, b) * b }

func (s Readers) newChainReader(src sdr.Reader) *chainReader {
	return &chainReader{s: s, src: src, srcFormat: src.SampleFormat(), rate: src.SampleRate(), block: 1, slots: 3, group: 1}
}

// fused returns r as a chainReader extended by how, or nil when the stage cannot join a chain (the caller then
// builds the reference's own structure).
func (s Readers) fused(r sdr.Reader, how func(*chainReader) bool) *chainReader {
	if !s.fuse {
		return nil
	}
	if cr, ok := r.(*chainReader); ok {
		if how(cr) {
			return cr
		}
		// r's chain is closed (it has its terminal, or has run): a new chain behind it
	}
	cr := s.newChainReader(r)
	if how(cr) {
		return cr
	}
	return nil
}

func (cr *chainReader) open() bool    { return cr.ch == nil && cr.term.kind == 0 }
func (cr *chainReader) c64Here() bool { return cr.srcFormat == sdr.SampleFormatC64 || cr.converted }

func (cr *chainReader) extendConvert(to sdr.SampleFormat) bool {
	if !cr.open() || to != sdr.SampleFormatC64 || len(cr.stages) != 0 || cr.srcFormat == sdr.SampleFormatC64 || cr.converted {
		return false
	}
	cr.converted = true
	cr.block = lcm(cr.block, readerBlock)
	return true
}

func (cr *chainReader) extendStage(st chainStage) bool {
	if !cr.open() || !cr.c64Here() {
		return false
	}
	cr.stages = append(cr.stages, st)
	return true
}

func (cr *chainReader) extendTerminal(t chainTerm, block int) bool {
	if cr.ch != nil {
		return false
	}
	// DecimateReader behind the ConvolutionReader -- only where the filter's blocks tile the DecimateReader's: the nest
	// hands out floor(n / 32 Ki) blocks then, as the fused chain does; any other length would make the chain consume
	// whole multiples of lcm(len, 32 Ki) and drop more of a stream's tail than the nest (such a DecimateReader becomes
	// a second chain behind this one)
	if t.kind == 1 && cr.term.kind == 3 && cr.term.decimate == 1 && readerBlock%len(cr.term.filter) == 0 {
		cr.term.decimate = t.factor
		cr.block = lcm(cr.block, readerBlock)
		cr.rate /= t.factor
		return true
	}
	if !cr.open() || !cr.c64Here() {
		return false
	}
	cr.term = t
	cr.block = lcm(cr.block, block)
	if t.kind == 1 || t.kind == 2 {
		cr.rate /= t.factor
	}
	return true
}

// extendFir makes the chain's terminal the north-star FIR-decimate (hzsdr_chain_fir_decimate; the reference has no such
// Reader): the terminal converts a raw source on its way in, as DownsampleReader does.
func (cr *chainReader) extendFir(taps []complex64, factor uint, slots, group int) bool {
	if !cr.open() || factor == 0 {
		return false
	}
	if !cr.c64Here() {
		cr.converted = true
	}
	cr.term = chainTerm{kind: 4, factor: factor, filter: append([]complex64(nil), taps...)}
	cr.block = lcm(cr.block, int(factor))
	cr.rate /= factor
	if slots < 2 {
		slots = 2
	}
	if group < 1 {
		group = 1
	}
	if group > 8 {
		group = 8
	}
	if group > slots-1 {
		group = slots - 1
	}
	cr.slots, cr.group = slots, group
	return true
}

func (cr *chainReader) SampleFormat() sdr.SampleFormat {
	if cr.c64Here() {
		return sdr.SampleFormatC64
	}
	return cr.srcFormat
}
func (cr *chainReader) SampleRate() uint { return cr.rate }

func (cr *chainReader) build() error {
	ch, err := cr.s.x.NewChain(cr.srcFormat, cr.src.SampleRate())
	if err != nil {
		return err
	}
	for _, st := range cr.stages {
		switch st.kind {
		case 0:
			err = ch.Shift(st.shift)
		case 1:
			err = ch.Gain(st.gain)
		default:
			err = ch.Multiply(st.mult)
		}
		if err != nil {
			ch.Close()
			return err
		}
	}
	switch cr.term.kind {
	case 1:
		err = ch.Decimate(cr.term.factor)
	case 2:
		err = ch.Downsample(cr.term.factor)
	case 3:
		err = ch.Convolution(cr.term.filter, cr.term.decimate)
	case 4:
		if err = ch.FIRDecimate(cr.term.filter, cr.term.factor); err == nil {
			err = ch.Pipeline(true) // consecutive calls overlap: the ring says what each call's buffers wait for
		}
	}
	if err != nil {
		ch.Close()
		return err
	}
	unit := cr.block
	if unit == 1 {
		unit = readerBlock
	}
	per := cr.s.readAhead * readerBlock / unit
	if per < 1 {
		per = 1
	}
	cr.slotLen = per * unit
	var ring *Ring
	alloc := ch.Allocator(&ring)
	if _, err = alloc(cr.srcFormat, stream.RingBufferOptions{Slots: cr.slots, SlotLength: cr.slotLen}); err != nil {
		ch.Close()
		return err
	}
	cr.ch, cr.ring = ch, ring
	return nil
}

// fill reads the source into up to `want` pinned slots and submits them -- the full ones together, ONE call of the
// chain (hzsdr_ring_submit_many), a short last one (the source ended) by itself -- and returns how many it submitted
// (0: nothing more comes). An acquired slot is submitted or released whatever happens (ADVICE r05: a failed Submit left
// the slot acquired and every later Acquire failed); a source that keeps returning (0, nil) ends the stream with
// io.ErrNoProgress instead of spinning.
func (cr *chainReader) fill(want int) int {
	if cr.err != nil {
		return 0
	}
	first, full, done := -1, 0, 0
	// flush submits the full slots gathered so far as one call. `newer`: a slot acquired behind them that is still
	// acquired (-1: none) -- on failure every acquired slot goes back, the newest first, as hzsdr_ring_release asks.
	flush := func(newer int) bool {
		if full == 0 {
			return true
		}
		err := cr.ring.SubmitMany(first, full, cr.slotLen)
		if err != nil {
			cr.err = err
			if newer >= 0 {
				_ = cr.ring.Release(newer)
			}
			for k := full - 1; k >= 0; k-- {
				_ = cr.ring.Release((first + k) % cr.slots)
			}
			done -= full
		} else {
			cr.inflight += full
		}
		full = 0
		return err == nil
	}
	for done < want && cr.err == nil {
		slot, iq, err := cr.ring.Acquire()
		if err != nil {
			cr.err = err
			break
		}
		n, idle := 0, 0
		for n < cr.slotLen {
			i, err := cr.src.Read(iq.Slice(n, cr.slotLen))
			n += i
			if err != nil {
				if err == io.ErrUnexpectedEOF {
					err = io.EOF
				}
				cr.err = err
				break
			}
			if i == 0 {
				if idle++; idle >= 100 {
					cr.err = io.ErrNoProgress
					break
				}
			} else {
				idle = 0
			}
		}
		n = n / cr.block * cr.block // a block-structured stage: whole blocks only
		if n == cr.slotLen {
			if full == 0 {
				first = slot
			}
			full++
			done++
			continue
		}
		// a short slot: everything full in front of it goes first, then it by itself (or back, if it is empty)
		if !flush(slot) {
			return done
		}
		if n == 0 {
			_ = cr.ring.Release(slot)
		} else if err := cr.ring.Submit(slot, n); err != nil {
			cr.err = err
			_ = cr.ring.Release(slot)
		} else {
			cr.inflight++
			done++
		}
		return done
	}
	flush(-1)
	return done
}

func (cr *chainReader) Read(s sdr.Samples) (int, error) {
	out, ok := s.(sdr.SamplesC64)
	if !ok || cr.SampleFormat() != sdr.SampleFormatC64 {
		return 0, sdr.ErrSampleFormatMismatch
	}
	if cr.ch == nil {
		if err := cr.build(); err != nil {
			return 0, err
		}
	}
	if len(cr.pending) == 0 && len(cr.queue) > 0 {
		cr.pending, cr.queue = cr.queue[0], cr.queue[1:]
	}
	if len(cr.pending) == 0 {
		// everything but the slot being handed out is in flight -- refilled `group` slots at a time, one call of the chain
		// per group: a refill waits until that many slots are free, or nothing is in flight
		for {
			free := cr.slots - 1 - cr.inflight
			if free < 1 || (free < cr.group && cr.inflight > 0) {
				break
			}
			if free > cr.group {
				free = cr.group
			}
			if cr.fill(free) == 0 {
				break
			}
		}
		if cr.inflight == 0 {
			if cr.err != nil {
				return 0, cr.err
			}
			return 0, io.EOF
		}
		p, err := cr.ring.Pop()
		cr.inflight--
		if err != nil {
			cr.err = err
			return 0, err
		}
		cr.pending = p
	}
	n := copy(out, cr.pending)
	cr.pending = cr.pending[n:]
	return n, nil
}

// SetMultiplier is MultiplyReader's (stream/multiply.go:34-36) for the chain's last Multiply stage. What has been read
// ahead keeps the old multiplier; the slots filled from now on take the new one: the chain is rebuilt at the clock it
// has reached.
func (cr *chainReader) SetMultiplier(m complex64) {
	idx := -1
	for i, st := range cr.stages {
		if st.kind == 2 {
			idx = i
		}
	}
	if idx < 0 {
		return
	}
	cr.stages[idx].mult = m
	if cr.ch == nil {
		return
	}
	keep := []sdr.SamplesC64{}
	if len(cr.pending) > 0 {
		keep = append(keep, append(sdr.SamplesC64(nil), cr.pending...))
	}
	keep = append(keep, cr.queue...)
	for cr.inflight > 0 {
		p, err := cr.ring.Pop()
		cr.inflight--
		if err != nil {
			cr.err = err
			break
		}
		keep = append(keep, append(sdr.SamplesC64(nil), p...))
	}
	cr.pending, cr.queue = nil, keep
	ts, _ := cr.ch.Time()
	cr.ring.Close()
	cr.ch.Close()
	cr.ch, cr.ring = nil, nil
	if err := cr.build(); err != nil {
		cr.err = err
		return
	}
	if err := cr.ch.SetTime(ts); err != nil {
		cr.err = err
	}
}

// Close releases the chain and the pinned ring (the reference's Readers are garbage collected).
func (cr *chainReader) Close() error {
	if cr.ring != nil {
		cr.ring.Close()
	}
	if cr.ch != nil {
		cr.ch.Close()
	}
	cr.ring, cr.ch = nil, nil
	if c, ok := cr.src.(io.Closer); ok {
		return c.Close()
	}
	return nil
}

var _ MultiplyReader = (*chainReader)(nil)

func (cr *chainReader) String() string {
	return fmt.Sprintf("hip.chainReader{%d stages, terminal %d, slot %d samples}", len(cr.stages), cr.term.kind, cr.slotLen)
}

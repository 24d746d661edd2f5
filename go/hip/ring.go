//go:build sdr.hip

package hip

// #include <hzsdr.h>
import "C"

import (
	"unsafe"

	"hz.tools/sdr"
	"hz.tools/sdr/stream"
	"hz.tools/sdr/yikes"
)

// Ring is the pinned stream.RingBuffer in front of a Chain: ONE hipHostMalloc region of
// Slots * SlotLength samples; a submitted slot goes upload -> chain kernel -> download on
// three HIP streams chained by events.
type Ring struct {
	ch *Chain
	r  *C.hzsdr_ring
	iq sdr.Samples
}

// Allocator plugs into stream.RingBufferOptions.IQBufferAllocator (stream/ring.go:60-68):
// the driver's rx callback (rtl/rx.go:49-68) then writes IQ straight into DMA-able memory.
func (ch *Chain) Allocator(ring **Ring) func(sdr.SampleFormat, stream.RingBufferOptions) (sdr.Samples, error) {
	return func(f sdr.SampleFormat, o stream.RingBufferOptions) (sdr.Samples, error) {
		if f != ch.inFormat {
			return nil, sdr.ErrSampleFormatMismatch
		}
		var r *C.hzsdr_ring
		if rc := C.hzsdr_ring_create(ch.c, C.size_t(o.SlotLength), C.int(o.Slots), &r); rc != C.HZSDR_OK {
			return nil, toErr(ch.x.c, rc)
		}
		var p unsafe.Pointer
		var n, slot C.size_t
		C.hzsdr_ring_iq_buffer(r, &p, &n, &slot)
		iq, err := yikes.Samples(uintptr(p), int(n), f) // yikes/bytes.go:50-71: C-owned, no Go pointer kept by C
		if err != nil {
			C.hzsdr_ring_free(r)
			return nil, err
		}
		*ring = &Ring{ch: ch, r: r, iq: iq}
		return iq, nil
	}
}

// Acquire the next slot to fill (ErrDstTooSmall-style overrun if every slot is in flight).
func (g *Ring) Acquire() (slot int, iq sdr.Samples, err error) {
	var s C.int
	var p unsafe.Pointer
	if rc := C.hzsdr_ring_acquire(g.r, &s, &p); rc != C.HZSDR_OK {
		return -1, nil, toErr(g.ch.x.c, rc)
	}
	_, _, sl := g.geometry()
	return int(s), g.iq.Slice(int(s)*sl, (int(s)+1)*sl), nil
}

// Submit `n` samples of the acquired slot: upload, chain kernel and download are enqueued.
func (g *Ring) Submit(slot, n int) error {
	return toErr(g.ch.x.c, C.hzsdr_ring_submit(g.r, C.int(slot), C.size_t(n)))
}

// SubmitMany hands `count` acquired slots (first the oldest, n samples each) to the chain as ONE call
// (hzsdr_ring_submit_many): one launch of the FIR-decimate terminal's persistent-pass kernel where the slots qualify
// (hzsdr_chain_run_batch's rules), slot by slot otherwise -- the same outputs either way, popped one by one.
func (g *Ring) SubmitMany(first, count, n int) error {
	return toErr(g.ch.x.c, C.hzsdr_ring_submit_many(g.r, C.int(first), C.int(count), C.size_t(n)))
}

// Release gives the acquired slot back unused: the source had nothing for it (hzsdr_ring_release).
func (g *Ring) Release(slot int) error { return toErr(g.ch.x.c, C.hzsdr_ring_release(g.r, C.int(slot))) }

// Pop waits for the oldest slot in flight; the returned samples are pinned memory, valid
// until that slot is submitted again.
func (g *Ring) Pop() (sdr.SamplesC64, error) {
	var p unsafe.Pointer
	var n C.size_t
	if rc := C.hzsdr_ring_pop(g.r, &p, &n); rc != C.HZSDR_OK {
		return nil, toErr(g.ch.x.c, rc)
	}
	s, err := yikes.Samples(uintptr(p), int(n), sdr.SampleFormatC64)
	if err != nil {
		return nil, err
	}
	return s.(sdr.SamplesC64), nil
}

func (g *Ring) InFlight() int { return int(C.hzsdr_ring_in_flight(g.r)) }
func (g *Ring) Close() error  { return toErr(g.ch.x.c, C.hzsdr_ring_free(g.r)) }

func (g *Ring) geometry() (base unsafe.Pointer, n, slot int) {
	var p unsafe.Pointer
	var a, b C.size_t
	C.hzsdr_ring_iq_buffer(g.r, &p, &a, &b)
	return p, int(a), int(b)
}

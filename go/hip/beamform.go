//go:build sdr.hip

package hip

// #include <stdlib.h>
// #include <hzsdr.h>
import "C"

import (
	"unsafe"

	"hz.tools/rf"
	"hz.tools/sdr"
)

// BeamformAngles2D / BeamformAngles are stream.BeamformAngles2D / BeamformAngles
// (stream/beamform.go:57-128); pure host float64 math.
func BeamformAngles2D(frequency rf.Hz, angleDeg float64, center [2]float64, antennas [][2]float64) ([]complex64, error) {
	if len(antennas) == 0 {
		return nil, nil
	}
	out := make([]complex64, len(antennas))
	flat := make([]C.double, 2*len(antennas))
	for i, a := range antennas {
		flat[2*i], flat[2*i+1] = C.double(a[0]), C.double(a[1])
	}
	ctr := [2]C.double{C.double(center[0]), C.double(center[1])}
	rc := C.hzsdr_beamform_angles_2d(C.double(float64(frequency)), C.double(angleDeg), &ctr[0], &flat[0], C.int(len(antennas)),
		(*C.float)(unsafe.Pointer(&out[0])))
	return out, toErr(nil, rc)
}

func BeamformAngles(frequency rf.Hz, angleDeg float64, distances []float64) ([]complex64, error) {
	if len(distances) == 0 {
		return nil, nil
	}
	out := make([]complex64, len(distances))
	rc := C.hzsdr_beamform_angles(C.double(float64(frequency)), C.double(angleDeg), (*C.double)(unsafe.Pointer(&distances[0])),
		C.int(len(distances)), (*C.float)(unsafe.Pointer(&out[0])))
	return out, toErr(nil, rc)
}

func chanPtrs(channels []sdr.Samples) (*unsafe.Pointer, func()) {
	ptrs := (*[16]unsafe.Pointer)(C.malloc(C.size_t(16 * unsafe.Sizeof(uintptr(0)))))
	for i, c := range channels {
		ptrs[i] = base(c)
	}
	return (*unsafe.Pointer)(unsafe.Pointer(ptrs)), func() { C.free(unsafe.Pointer(ptrs)) }
}

// Beamform is the data path of stream.ReadBeamform (stream/beamform.go:148-171):
// out = ((0 + w0*x0) + w1*x1) + ...; the weights travel by value per call
// (Beamform.SetPhaseAngles, stream/beamform.go:131-139, applies between reads).
func (x *Context) Beamform(out sdr.SamplesC64, channels []sdr.Samples, weights []complex64) error {
	p, free := chanPtrs(channels)
	defer free()
	return toErr(x.c, C.hzsdr_beamform(x.c, base(out), C.int(channels[0].Format()), p, (*C.float)(unsafe.Pointer(&weights[0])),
		C.int(len(channels)), C.size_t(len(out))))
}

// BeamformPartial continues (accumulate) or starts the ordered sum over a subset of the channels.
func (x *Context) BeamformPartial(out sdr.SamplesC64, channels []sdr.Samples, weights []complex64, accumulate bool) error {
	p, free := chanPtrs(channels)
	defer free()
	return toErr(x.c, C.hzsdr_beamform_partial(x.c, base(out), C.int(channels[0].Format()), p, (*C.float)(unsafe.Pointer(&weights[0])),
		C.int(len(channels)), C.size_t(len(out)), cbool(accumulate)))
}

// MultiGPU shards Beamform over GPUs from this one process: channel c lives on shard
// Owner(c); ONE exchange combines them (ordered, bit-identical; or RCCL, faster).
type MultiGPU struct {
	m      *C.hzsdr_mgpu
	Shards []*Context // MemDevice contexts, one per entry of `devices`
}

type BeamformMode int

const (
	Ordered BeamformMode = C.HZSDR_MGPU_ORDERED
	RCCL    BeamformMode = C.HZSDR_MGPU_RCCL
)

func OpenMultiGPU(devices []int) (*MultiGPU, error) {
	d := make([]C.int, len(devices))
	for i, v := range devices {
		d[i] = C.int(v)
	}
	var m *C.hzsdr_mgpu
	if rc := C.hzsdr_mgpu_open(&d[0], C.int(len(d)), &m); rc != C.HZSDR_OK {
		return nil, toErr(nil, rc)
	}
	g := &MultiGPU{m: m}
	for s := 0; s < int(C.hzsdr_mgpu_shards(m)); s++ {
		var c *C.hzsdr_ctx
		C.hzsdr_mgpu_ctx(m, C.int(s), &c)
		g.Shards = append(g.Shards, &Context{c: c, space: MemDevice})
	}
	return g, nil
}

// ShardChannels: channels [lo, hi) live on `shard`.
func ShardChannels(nChannels, nShards, shard int) (lo, hi int) {
	var a, b C.int
	C.hzsdr_mgpu_shard_channels(C.int(nChannels), C.int(nShards), C.int(shard), &a, &b)
	return int(a), int(b)
}

// Beamform: channels[c] is a DEVICE pointer on its owner's GPU, out a device pointer on
// shard dst's GPU; `format` is the channels' sample format, n the samples per channel.
func (g *MultiGPU) Beamform(out unsafe.Pointer, dst int, format sdr.SampleFormat, channels []unsafe.Pointer, weights []complex64, n int, mode BeamformMode) error {
	ptrs := (*[16]unsafe.Pointer)(C.malloc(C.size_t(16 * unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(ptrs))
	copy(ptrs[:], channels)
	rc := C.hzsdr_mgpu_beamform(g.m, out, C.int(dst), C.int(format), (*unsafe.Pointer)(unsafe.Pointer(ptrs)),
		(*C.float)(unsafe.Pointer(&weights[0])), C.int(len(channels)), C.size_t(n), C.int(mode))
	if rc != C.HZSDR_OK {
		if e := toErr(nil, rc); e != nil && rc <= C.HZSDR_ERR_CONVERSION_NOT_IMPLEMENTED {
			return e
		}
		return &mgpuError{C.GoString(C.hzsdr_strerror(rc)), C.GoString(C.hzsdr_mgpu_last_error(g.m))}
	}
	return nil
}

// PeerPairs reports how the shards reach each other: ordered pairs of distinct GPUs with peer access (xGMI copies) and
// without (copies staged through the host); hzsdr_mgpu_peer_pairs.
func (g *MultiGPU) PeerPairs() (direct, staged int) {
	var d, s C.int
	C.hzsdr_mgpu_peer_pairs(g.m, &d, &s)
	return int(d), int(s)
}

func (g *MultiGPU) Synchronize() error { return toErr(nil, C.hzsdr_mgpu_synchronize(g.m)) }
func (g *MultiGPU) Close() error       { return toErr(nil, C.hzsdr_mgpu_close(g.m)) }

type mgpuError struct{ what, detail string }

func (e *mgpuError) Error() string { return "hip: " + e.what + ": " + e.detail }

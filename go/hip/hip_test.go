//go:build sdr.hip

package hip

// The reference's own known-answer tests, run through this package: the vectors come from
// tests/golden/reference_kats.json (inputs and expected outputs transcribed from the reference's *_test.go files;
// each group cites its file:line), the same fixture the Python, C and C++ layers of this repository check against
// the same C-ABI.  UNCOMPILED IN THE BUILD IMAGE (no Go toolchain there): on a machine with Go and an MI355X,
//
//	go test -tags sdr.hip ./go/hip/
//
// Every test opens a MemHost context: plain Go slices in and out, as a caller of the reference would hold them.

import (
	"encoding/json"
	"math"
	"os"
	"testing"

	"hz.tools/rf"
	"hz.tools/sdr"
	"hz.tools/sdr/fft"
)

type kats map[string]json.RawMessage

func loadKATs(t *testing.T) kats {
	t.Helper()
	raw, err := os.ReadFile("../../tests/golden/reference_kats.json")
	if err != nil {
		t.Skipf("fixture not found: %v", err)
	}
	var k kats
	if err := json.Unmarshal(raw, &k); err != nil {
		t.Fatal(err)
	}
	return k
}

func group(t *testing.T, k kats, name string, into interface{}) {
	t.Helper()
	if err := json.Unmarshal(k[name], into); err != nil {
		t.Fatalf("%s: %v", name, err)
	}
}

func open(t *testing.T) *Context {
	t.Helper()
	x, err := Open(0, MemHost)
	if err != nil {
		t.Skipf("no device: %v", err)
	}
	t.Cleanup(func() { x.Close() })
	return x
}

func c64(v [2]float64) complex64 { return complex(float32(v[0]), float32(v[1])) }

// the reference's tolerance: both components compared as 1 + x (testutils' InEpsilon on shifted values)
func near(a, b complex64, eps float64) bool {
	ok := func(x, y float32) bool { return math.Abs(float64(1+x)-float64(1+y)) <= eps*math.Abs(float64(1+y))+1e-7 }
	return ok(real(a), real(b)) && ok(imag(a), imag(b))
}

func filled(n int, v complex64) sdr.SamplesC64 {
	s := make(sdr.SamplesC64, n)
	for i := range s {
		s[i] = v
	}
	return s
}

func cw(n int, freq float64, rate uint, phase float64) sdr.SamplesC64 {
	s := make(sdr.SamplesC64, n)
	for i := range s {
		a := 2*math.Pi*freq*float64(i)/float64(rate) + phase
		s[i] = complex(float32(math.Cos(a)), float32(math.Sin(a)))
	}
	return s
}

// iq_u8_test.go / iq_i8_test.go / iq_i16_test.go / iq_c64_test.go: the converters' end points
func TestConvertKATs(t *testing.T) {
	k, x := loadKATs(t), open(t)
	var cases []struct {
		Cite   string       `json:"cite"`
		SrcFmt string       `json:"src_fmt"`
		DstFmt string       `json:"dst_fmt"`
		Src    [][2]float64 `json:"src"`
		Dst    [][2]float64 `json:"dst"`
		Eps    float64      `json:"eps"`
	}
	group(t, k, "convert", &cases)
	mk := func(f string, v [][2]float64) sdr.Samples {
		switch f {
		case "u8":
			s := make(sdr.SamplesU8, len(v))
			for i, p := range v {
				s[i] = [2]uint8{uint8(p[0]), uint8(p[1])}
			}
			return s
		case "i8":
			s := make(sdr.SamplesI8, len(v))
			for i, p := range v {
				s[i] = [2]int8{int8(p[0]), int8(p[1])}
			}
			return s
		case "i16":
			s := make(sdr.SamplesI16, len(v))
			for i, p := range v {
				s[i] = [2]int16{int16(p[0]), int16(p[1])}
			}
			return s
		}
		s := make(sdr.SamplesC64, len(v))
		for i, p := range v {
			s[i] = c64(p)
		}
		return s
	}
	for _, c := range cases {
		src, want := mk(c.SrcFmt, c.Src), mk(c.DstFmt, c.Dst)
		dst := mk(c.DstFmt, make([][2]float64, len(c.Dst)))
		n, err := x.ConvertBuffer(dst, src)
		if err != nil || n != src.Length() {
			t.Fatalf("%s: n %d err %v", c.Cite, n, err)
		}
		switch w := want.(type) {
		case sdr.SamplesC64:
			for i := range w {
				if !near(dst.(sdr.SamplesC64)[i], w[i], math.Max(c.Eps, 1e-6)) {
					t.Errorf("%s: sample %d = %v, want %v", c.Cite, i, dst.(sdr.SamplesC64)[i], w[i])
				}
			}
		case sdr.SamplesU8:
			for i := range w {
				if dst.(sdr.SamplesU8)[i] != w[i] {
					t.Errorf("%s: sample %d = %v, want %v", c.Cite, i, dst.(sdr.SamplesU8)[i], w[i])
				}
			}
		case sdr.SamplesI8:
			for i := range w {
				if dst.(sdr.SamplesI8)[i] != w[i] {
					t.Errorf("%s: sample %d = %v, want %v", c.Cite, i, dst.(sdr.SamplesI8)[i], w[i])
				}
			}
		case sdr.SamplesI16:
			for i := range w {
				if dst.(sdr.SamplesI16)[i] != w[i] {
					t.Errorf("%s: sample %d = %v, want %v", c.Cite, i, dst.(sdr.SamplesI16)[i], w[i])
				}
			}
		}
	}
	// iq_u8_test.go:65-85: a sub-slice converts into its own range and nothing else
	var g struct {
		N, Lo, Hi int
		Fill      [2]float64
		Value     [2]float64
		Eps       float64
	}
	group(t, k, "convert_subslice_guard", &g)
	in := make(sdr.SamplesU8, g.N)
	for i := range in {
		in[i] = [2]uint8{uint8(g.Fill[0]), uint8(g.Fill[1])}
	}
	out := make(sdr.SamplesC64, g.N)
	if _, err := x.ConvertBuffer(out[g.Lo:g.Hi], in[g.Lo:g.Hi]); err != nil {
		t.Fatal(err)
	}
	for i, v := range out {
		if inside := i >= g.Lo && i < g.Hi; inside && !near(v, c64(g.Value), g.Eps) || !inside && v != 0 {
			t.Fatalf("sub-slice guard: out[%d] = %v", i, v)
		}
	}
	// conv.go:55-93: the destination must hold the source
	if _, err := x.ConvertBuffer(make(sdr.SamplesC64, 3), make(sdr.SamplesU8, 4)); err != sdr.ErrDstTooSmall {
		t.Errorf("short destination: %v", err)
	}
}

// iq_c64_test.go:110-145, internal/simd: Scale, Multiply, Add
func TestScaleMultiplyAdd(t *testing.T) {
	k, x := loadKATs(t), open(t)
	var s struct {
		N     int
		Fill  [2]float64
		R     float64
		Value [2]float64
	}
	group(t, k, "scale", &s)
	buf := filled(s.N, c64(s.Fill))
	if err := x.Scale(buf, float32(s.R)); err != nil {
		t.Fatal(err)
	}
	for i, v := range buf {
		if v != c64(s.Value) {
			t.Fatalf("Scale: [%d] = %v", i, v)
		}
	}
	var m struct {
		N           int
		Fill, M     [2]float64
		Value       [2]float64
	}
	group(t, k, "multiply", &m)
	buf = filled(m.N, c64(m.Fill))
	if err := x.Multiply(buf, c64(m.M)); err != nil {
		t.Fatal(err)
	}
	for i, v := range buf {
		if v != c64(m.Value) {
			t.Fatalf("Multiply: [%d] = %v", i, v)
		}
	}
	var a struct {
		N           int
		A, B, Value [2]float64
	}
	group(t, k, "add", &a)
	dst := make(sdr.SamplesC64, a.N)
	if err := x.Add(filled(a.N, c64(a.A)), filled(a.N, c64(a.B)), dst); err != nil {
		t.Fatal(err)
	}
	for i, v := range dst {
		if v != c64(a.Value) {
			t.Fatalf("Add: [%d] = %v", i, v)
		}
	}
}

// stream/shifter_test.go:35-72: +shift then -shift returns the carrier
func TestShiftRoundtrip(t *testing.T) {
	k, x := loadKATs(t), open(t)
	var g struct {
		N           int
		Freq, Shift float64
		Rate        uint
		Eps         float64
	}
	group(t, k, "shift_roundtrip", &g)
	for _, ulp1 := range []bool{false, true} {
		want := cw(g.N, g.Freq, g.Rate, 0)
		buf := append(sdr.SamplesC64(nil), want...)
		up, err := x.NewShifter(g.Rate)
		if err != nil {
			t.Fatal(err)
		}
		down, _ := x.NewShifter(g.Rate)
		up.SetULP1(ulp1)
		down.SetULP1(ulp1)
		for lo := 0; lo < g.N; lo += 7000 { // odd block sizes: the clock carries across calls
			hi := lo + 7000
			if hi > g.N {
				hi = g.N
			}
			up.ShiftBuffer(rf.Hz(g.Shift), buf[lo:hi])
			down.ShiftBuffer(rf.Hz(-g.Shift), buf[lo:hi])
		}
		for i := range buf {
			if !near(buf[i], want[i], g.Eps) {
				t.Fatalf("ulp1 %v: [%d] = %v, want %v", ulp1, i, buf[i], want[i])
			}
		}
		up.Close()
		down.Close()
	}
}

// stream/decimate_test.go, stream/downsample_test.go
func TestDecimateDownsample(t *testing.T) {
	k, x := loadKATs(t), open(t)
	var d struct {
		N, Count int
		Factor   uint
		Value    [2]float64
	}
	group(t, k, "decimate_skippy", &d)
	in := make(sdr.SamplesU8, d.N)
	for i := range in {
		in[i] = [2]uint8{uint8(i % 10), uint8(i % 10)}
	}
	out := make(sdr.SamplesU8, d.N)
	n, err := x.DecimateBuffer(out, in, d.Factor, 0)
	if err != nil || n != d.Count {
		t.Fatalf("DecimateBuffer: n %d err %v", n, err)
	}
	for i := 0; i < n; i++ {
		if out[i] != [2]uint8{uint8(d.Value[0]), uint8(d.Value[1])} {
			t.Fatalf("DecimateBuffer: [%d] = %v", i, out[i])
		}
	}
	if _, err := x.DecimateBuffer(make(sdr.SamplesU8, d.N), make(sdr.SamplesC64, d.N), d.Factor, 0); err != sdr.ErrSampleFormatMismatch {
		t.Errorf("DecimateBuffer across formats: %v", err)
	}
	group(t, k, "downsample_calc", &d)
	cin := make(sdr.SamplesC64, d.N)
	for i := range cin {
		cin[i] = complex(float32(i%4), float32(i%4))
	}
	cout := make(sdr.SamplesC64, d.Count)
	n, err = x.DownsampleBuffer(cout, cin, d.Factor, 0)
	if err != nil || n != d.Count {
		t.Fatalf("DownsampleBuffer: n %d err %v", n, err)
	}
	for i, v := range cout {
		if v != c64(d.Value) {
			t.Fatalf("DownsampleBuffer: [%d] = %v", i, v)
		}
	}
}

// testutils/fft.go:54-138: a tone lands in its bin, a bin survives backward + forward, mismatched lengths are refused
func TestPlannerConformance(t *testing.T) {
	k, x := loadKATs(t), open(t)
	var f struct {
		N     int
		Rate  uint
		Cases [][2]float64
	}
	group(t, k, "fft_forward_bins", &f)
	argmax := func(v []complex64) int {
		best, at := -1.0, 0
		for i, c := range v {
			if p := float64(real(c))*float64(real(c)) + float64(imag(c))*float64(imag(c)); p > best {
				best, at = p, i
			}
		}
		return at
	}
	for _, c := range f.Cases {
		iq, freq := cw(f.N, c[0], f.Rate, 0), make([]complex64, f.N)
		p, err := x.Planner(iq, freq, fft.Forward)
		if err != nil {
			t.Fatal(err)
		}
		if err := p.Transform(); err != nil {
			t.Fatal(err)
		}
		if got := argmax(freq); got != int(c[1]) {
			t.Errorf("tone %g Hz: peak in bin %d, want %d", c[0], got, int(c[1]))
		}
		p.Close()
	}
	var b struct {
		N    int
		Bins []int
	}
	group(t, k, "fft_backward_roundtrip", &b)
	for _, bin := range b.Bins {
		iq, freq := make(sdr.SamplesC64, b.N), make([]complex64, b.N)
		freq[bin] = complex(1, 1)
		back, err := x.Planner(iq, freq, fft.Backward)
		if err != nil {
			t.Fatal(err)
		}
		back.Transform()
		back.Close()
		freq[bin] = 0
		fwd, _ := x.Planner(iq, freq, fft.Forward)
		fwd.Transform()
		fwd.Close()
		if got := argmax(freq); got != bin {
			t.Errorf("bin %d came back in bin %d", bin, got)
		}
	}
	if _, err := x.Planner(make(sdr.SamplesC64, 1024), make([]complex64, 128), fft.Forward); err != sdr.ErrDstTooSmall {
		t.Errorf("mismatched lengths: %v", err)
	}
}

// stream/beamform_test.go:34-155: the phase angles of a linear and a planar array
func TestBeamformAngles(t *testing.T) {
	k := loadKATs(t)
	var lin []struct {
		Freq, Angle float64
		Distances   []float64
		Expect      [][2]float64
		Eps         float64
	}
	group(t, k, "beamform_angles", &lin)
	for _, c := range lin {
		got, err := BeamformAngles(rf.Hz(c.Freq), c.Angle, c.Distances)
		if err != nil {
			t.Fatal(err)
		}
		for i := range got {
			if math.Abs(float64(real(got[i]))-c.Expect[i][0]) > math.Max(c.Eps, 1e-4) || math.Abs(float64(imag(got[i]))-c.Expect[i][1]) > math.Max(c.Eps, 1e-4) {
				t.Errorf("angle %g: weight %d = %v, want %v", c.Angle, i, got[i], c.Expect[i])
			}
		}
	}
}

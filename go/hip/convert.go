//go:build sdr.hip

package hip

// #include <hzsdr.h>
import "C"

import (
	"unsafe"

	"hz.tools/sdr"
)

// ConvertBuffer replaces sdr.ConvertBuffer (conv.go:55-93).
func (x *Context) ConvertBuffer(dst, src sdr.Samples) (int, error) {
	var n C.size_t
	rc := C.hzsdr_convert(x.c, C.int(dst.Format()), base(dst), C.size_t(dst.Length()),
		C.int(src.Format()), base(src), C.size_t(src.Length()), &n)
	return int(n), toErr(x.c, rc)
}

// ConvertBufferForeign fuses the byte swap of bytes_io.go:30-64 / :150-197 into the
// converter's load and / or store.
func (x *Context) ConvertBufferForeign(dst sdr.Samples, dstForeign bool, src sdr.Samples, srcForeign bool) (int, error) {
	var n C.size_t
	rc := C.hzsdr_convert_foreign(x.c, C.int(dst.Format()), base(dst), C.size_t(dst.Length()), cbool(dstForeign),
		C.int(src.Format()), base(src), C.size_t(src.Length()), cbool(srcForeign), &n)
	return int(n), toErr(x.c, rc)
}

// ByteSwap reverses every int16 / float32 component in place (foreign-endian payloads).
func (x *Context) ByteSwap(buf sdr.Samples) error {
	return toErr(x.c, C.hzsdr_byteswap(x.c, C.int(buf.Format()), base(buf), C.size_t(buf.Length())))
}

// ShiftLSBToMSBBits replaces SamplesI16.ShiftLSBToMSBBits (iq_i16.go:51-60).
func (x *Context) ShiftLSBToMSBBits(buf sdr.SamplesI16, bits int) error {
	return toErr(x.c, C.hzsdr_i16_shift_lsb_to_msb(x.c, base(buf), C.size_t(len(buf)), C.int(bits)))
}

// LookupTable replaces sdr.LookupTable (iq_lookup_table.go:36-50): 65 536 entries of the
// destination format, indexed by the raw 16 bits of a u8 / i8 sample.
type LookupTable struct {
	x   *Context
	t   *C.hzsdr_lut
	src sdr.SampleFormat
	dst sdr.SampleFormat
}

// NewLookupTable uploads `table` (65 536 samples of its own format).
func (x *Context) NewLookupTable(src sdr.SampleFormat, table sdr.Samples) (*LookupTable, error) {
	var t *C.hzsdr_lut
	rc := C.hzsdr_lut_create(x.c, C.int(src), C.int(table.Format()), base(table), C.size_t(table.Length()), &t)
	if rc != C.HZSDR_OK {
		return nil, toErr(x.c, rc)
	}
	return &LookupTable{x: x, t: t, src: src, dst: table.Format()}, nil
}

// Lookup is LookupTable.Lookup (iq_lookup_table.go:213-251).
func (l *LookupTable) Lookup(dst, src sdr.Samples) (int, error) {
	var n C.size_t
	rc := C.hzsdr_lut_lookup(l.t, C.int(dst.Format()), base(dst), C.size_t(dst.Length()),
		C.int(src.Format()), base(src), C.size_t(src.Length()), &n)
	return int(n), toErr(l.x.c, rc)
}

// Close frees the device copy of the table.
func (l *LookupTable) Close() error { return toErr(l.x.c, C.hzsdr_lut_free(l.t)) }

// IdentityTable fills a 65 536-entry u8 table with the identity mapping
// (iq_lookup_table.go:56-74); host memory, no context needed.
func IdentityTable(table sdr.SamplesU8) error {
	if len(table) < 65536 {
		return sdr.ErrDstTooSmall
	}
	return toErr(nil, C.hzsdr_lut_identity(unsafe.Pointer(&table[0])))
}

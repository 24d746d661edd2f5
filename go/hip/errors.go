//go:build sdr.hip

package hip

// #include <hzsdr.h>
import "C"

import (
	"fmt"

	"hz.tools/sdr"
)

// ErrNoDevice: no gfx950 GPU, or the HIP runtime is unusable.  There is no CPU fallback.
var ErrNoDevice = fmt.Errorf("hip: no MI355X (gfx950) device")

// toErr maps status codes 1..4 onto the reference's sentinel errors 1:1.
func toErr(ctx *C.hzsdr_ctx, rc C.int) error {
	switch rc {
	case C.HZSDR_OK:
		return nil
	case C.HZSDR_ERR_FORMAT_MISMATCH:
		return sdr.ErrSampleFormatMismatch // iq.go:30
	case C.HZSDR_ERR_FORMAT_UNKNOWN:
		return sdr.ErrSampleFormatUnknown // iq.go:34
	case C.HZSDR_ERR_DST_TOO_SMALL:
		return sdr.ErrDstTooSmall // iq.go:38
	case C.HZSDR_ERR_CONVERSION_NOT_IMPLEMENTED:
		return sdr.ErrConversionNotImplemented // conv.go:30
	case C.HZSDR_ERR_NO_DEVICE:
		return ErrNoDevice
	default:
		msg := ""
		if ctx != nil {
			msg = C.GoString(C.hzsdr_last_error(ctx))
		}
		// HZSDR_ERR_LENGTH_MISMATCH carries the reference's own message text
		// (add.go:34, fft/convolution.go:38,157) in last_error
		return fmt.Errorf("hip: %s: %s", C.GoString(C.hzsdr_strerror(rc)), msg)
	}
}

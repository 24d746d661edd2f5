"""go-sdr_amd: the MI355X (gfx950) backend of the hz.tools/sdr sample-processing
hot path, bound from Python through the C ABI of include/hzsdr.h.

This package is the host side used by the tests and bench here (the image has no
Go toolchain; INTEGRATION.md shows the cgo binding of the same C ABI).  It keeps
the reference's vocabulary: sample formats, sentinel errors, buffer-level
functions (ConvertBuffer, DecimateBuffer, ...) and, in `stream` / `fft`, the
Reader-returning operators.

Import with importlib (the directory name carries a hyphen):

    hz = importlib.import_module("go-sdr_amd")
    ctx = hz.Context(0, hz.MEM_DEVICE)

Buffers are numpy arrays for MEM_HOST contexts and torch CUDA tensors for
MEM_DEVICE contexts, laid out as the reference lays out sdr.Samples:
u8/i8 -> shape (n, 2) bytes, i16 -> (n, 2) int16, c64 -> (n,) complex64.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (CONV_CONVOLVE, CONV_CROSS_CORRELATE, FFT_BACKWARD, FFT_FORWARD, FIR_IMPL_AUTO,
                    FIR_IMPL_MATRIX_CHUNKS, FIR_IMPL_TRANSFORMS, FIR_KERNEL_MATRIX_CHUNKS,
                    FIR_KERNEL_MATRIX_PASSES, FIR_KERNEL_NONE, FIR_KERNEL_TRANSFORM, FIR_PATH_MATRIX,
                    FIR_PATH_NONE, FIR_PATH_TRANSFORM, FMT_C64, FMT_I8, FMT_I16, FMT_U8, MEM_DEVICE,
                    MEM_HOST, lib)

SampleFormatC64, SampleFormatU8, SampleFormatI16, SampleFormatI8 = FMT_C64, FMT_U8, FMT_I16, FMT_I8


# ---- errors: the reference's sentinels (iq.go:27-39, conv.go:30) ---------------

class HzsdrError(Exception):
    status = -1


class ErrSampleFormatMismatch(HzsdrError):
    status = _capi.ERR_FORMAT_MISMATCH


class ErrSampleFormatUnknown(HzsdrError):
    status = _capi.ERR_FORMAT_UNKNOWN


class ErrDstTooSmall(HzsdrError):
    status = _capi.ERR_DST_TOO_SMALL


class ErrConversionNotImplemented(HzsdrError):
    status = _capi.ERR_CONVERSION_NOT_IMPLEMENTED


class ErrLengthMismatch(HzsdrError):
    status = _capi.ERR_LENGTH_MISMATCH


class ErrInvalidArgument(HzsdrError):
    status = _capi.ERR_INVALID_ARGUMENT


class ErrNoDevice(HzsdrError):
    status = _capi.ERR_NO_DEVICE


class ErrHip(HzsdrError):
    status = _capi.ERR_HIP


class ErrOutOfMemory(HzsdrError):
    status = _capi.ERR_OUT_OF_MEMORY


_ERRORS = {e.status: e for e in (ErrSampleFormatMismatch, ErrSampleFormatUnknown, ErrDstTooSmall,
                                 ErrConversionNotImplemented, ErrLengthMismatch,
                                 ErrInvalidArgument, ErrNoDevice, ErrHip, ErrOutOfMemory)}


def _check(rc, ctx_handle=None):
    if rc == _capi.OK:
        return
    msg = lib.hzsdr_strerror(rc).decode()
    if ctx_handle:
        detail = lib.hzsdr_last_error(ctx_handle).decode()
        if detail:
            msg = f"{msg}: {detail}"
    raise _ERRORS.get(rc, HzsdrError)(msg)


def backend():
    return lib.hzsdr_backend().decode()


def version():
    return lib.hzsdr_version().decode()


def format_size(fmt):
    return lib.hzsdr_format_size(fmt)


def device_count():
    n = C.c_int(0)
    lib.hzsdr_device_count(C.byref(n))
    return n.value


# ---- buffers --------------------------------------------------------------------

_NP_FMT = {np.dtype(np.complex64): FMT_C64, np.dtype(np.uint8): FMT_U8,
           np.dtype(np.int16): FMT_I16, np.dtype(np.int8): FMT_I8}


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def fmt_of(x):
    """sdr.Samples.Format()."""
    if _is_torch(x):
        import torch
        return {torch.complex64: FMT_C64, torch.uint8: FMT_U8, torch.int16: FMT_I16,
                torch.int8: FMT_I8}[x.dtype]
    return _NP_FMT[x.dtype]


def length(x):
    """sdr.Samples.Length(): IQ samples, not bytes."""
    return int(x.shape[0])


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        assert x.is_contiguous()
        return x.data_ptr()
    assert x.flags["C_CONTIGUOUS"]
    return x.ctypes.data


def _event(ev):
    """A hipEvent_t for the C ABI: None, a raw handle (int), or a torch.cuda.Event that has been recorded (torch
    creates the underlying event lazily, at its first record)."""
    if ev is None:
        return None
    if isinstance(ev, int):
        return ev
    h = getattr(ev, "cuda_event", None)
    if not h:
        raise ValueError("run_after: the torch.cuda.Event has not been recorded yet")
    return int(h)


def make_samples(fmt, n, device=None):
    """sdr.MakeSamples(format, n) (iq.go:128-141); device=None -> numpy."""
    if fmt not in (FMT_C64, FMT_U8, FMT_I16, FMT_I8):
        raise ErrSampleFormatUnknown("sdr: iq sample format is not understood")
    if device is None:
        dt = {FMT_C64: np.complex64, FMT_U8: np.uint8, FMT_I16: np.int16, FMT_I8: np.int8}[fmt]
        return np.zeros(n, dt) if fmt == FMT_C64 else np.zeros((n, 2), dt)
    import torch
    dt = {FMT_C64: torch.complex64, FMT_U8: torch.uint8, FMT_I16: torch.int16, FMT_I8: torch.int8}[fmt]
    shape = (n,) if fmt == FMT_C64 else (n, 2)
    return torch.zeros(shape, dtype=dt, device=device)


def lut_identity():
    """LookupTableIdentityU8 (iq_lookup_table.go:69-78) as a numpy (65536, 2) uint8."""
    t = np.zeros((65536, 2), np.uint8)
    _check(lib.hzsdr_lut_identity(t.ctypes.data))
    return t


def beamform_angles(frequency_hz, angle_deg, distances):
    """stream.BeamformAngles (stream/beamform.go:111-127) -> complex64 array or None."""
    d = np.ascontiguousarray(distances, np.float64)
    if d.size == 0:
        return None
    out = np.zeros(d.size, np.complex64)
    _check(lib.hzsdr_beamform_angles(frequency_hz, angle_deg, d.ctypes.data_as(C.POINTER(C.c_double)),
                                     d.size, out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def beamform_angles_2d(frequency_hz, angle_deg, center, antennas):
    """stream.BeamformAngles2D (stream/beamform.go:57-107)."""
    a = np.ascontiguousarray(antennas, np.float64).reshape(-1, 2)
    if a.shape[0] == 0:
        return None
    ctr = np.ascontiguousarray(center, np.float64)
    out = np.zeros(a.shape[0], np.complex64)
    _check(lib.hzsdr_beamform_angles_2d(frequency_hz, angle_deg,
                                        ctr.ctypes.data_as(C.POINTER(C.c_double)),
                                        a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0],
                                        out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def nco_segments(sample_rate, ts_start, n, cap=1 << 16):
    """hzsdr_nco_segments: ([(first, count, t0, step)...], ts_end).  Pure host."""
    segs = (_capi.NcoSegment * cap)()
    need, te = C.c_size_t(0), C.c_double(0.0)
    _check(lib.hzsdr_nco_segments(sample_rate, ts_start, n, segs, cap, C.byref(need), C.byref(te)))
    if need.value > cap:
        return nco_segments(sample_rate, ts_start, n, need.value)
    return [(s.first, s.count, s.t0, s.step) for s in segs[:need.value]], te.value


# ---- context ----------------------------------------------------------------------

class Context:
    """hzsdr_ctx: one GPU, one hipStream, one memory space."""

    def __init__(self, device=0, memspace=MEM_HOST, stream=None):
        self._h = C.c_void_p()
        _check(lib.hzsdr_open(device, memspace, C.byref(self._h)))
        self.device = device
        self.memspace = memspace
        self._pinned = []
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if self._h:
            for p in self._pinned:
                lib.hzsdr_free_pinned(self._h, C.c_void_p(p))
            self._pinned = []
            lib.hzsdr_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        _check(rc, self._h)

    def set_stream(self, hip_stream):
        """Adopt a hipStream_t given as an int (e.g. torch.cuda.current_stream().cuda_stream)."""
        self._ck(lib.hzsdr_set_stream(self._h, hip_stream))

    def use_own_stream(self):
        self._ck(lib.hzsdr_use_own_stream(self._h))

    def stream(self):
        return lib.hzsdr_get_stream(self._h)

    def synchronize(self):
        self._ck(lib.hzsdr_synchronize(self._h))

    def call_count(self):
        """hzsdr_call_count: library calls made on this context so far (Reader tests count calls per sample)."""
        n = C.c_ulonglong(0)
        self._ck(lib.hzsdr_call_count(self._h, C.byref(n)))
        return n.value

    # -- sdr.ConvertBuffer (conv.go:55) --
    def convert(self, dst, src):
        n = C.c_size_t(0)
        self._ck(lib.hzsdr_convert(self._h, fmt_of(dst), _ptr(dst), length(dst), fmt_of(src),
                                   _ptr(src), length(src), C.byref(n)))
        return n.value

    def convert_raw(self, dst_fmt, dst, dst_len, src_fmt, src, src_len):
        n = C.c_size_t(0)
        self._ck(lib.hzsdr_convert(self._h, dst_fmt, _ptr(dst), dst_len, src_fmt, _ptr(src),
                                   src_len, C.byref(n)))
        return n.value

    def i16_shift_lsb_to_msb(self, buf, bits):
        self._ck(lib.hzsdr_i16_shift_lsb_to_msb(self._h, _ptr(buf), length(buf), bits))

    # -- stream.RingBufferOptions.IQBufferAllocator (stream/ring.go:60-68): sample buffers in
    # pinned, GPU-visible memory the library owns; HOST-space calls on them skip all staging --
    def pinned_samples(self, fmt, n):
        p = C.c_void_p()
        self._ck(lib.hzsdr_malloc_pinned(self._h, max(1, n * format_size(fmt)), C.byref(p)))
        self._pinned.append(p.value)
        return _view(p.value, fmt, n)

    # -- SamplesC64.Scale / Multiply / Add (iq_c64.go:122-136) --
    def scale(self, buf, r):
        self._ck(lib.hzsdr_scale(self._h, _ptr(buf), length(buf), r))

    def rotate(self, buf, m):
        m = complex(m)
        self._ck(lib.hzsdr_rotate(self._h, _ptr(buf), length(buf), m.real, m.imag))

    def add(self, a, b, c):
        self._ck(lib.hzsdr_add(self._h, _ptr(a), length(a), _ptr(b), length(b), _ptr(c), length(c)))

    # -- addReader.Read data path (stream/add.go:121-185) --
    def sum(self, out, bufs):
        k = len(bufs)
        arr = (C.c_void_p * k)(*[_ptr(b) for b in bufs])
        self._ck(lib.hzsdr_sum(self._h, fmt_of(out), _ptr(out), arr, k, length(out)))

    def lut(self, src_fmt, table):
        return LookupTable(self, src_fmt, table)

    def rotlut(self, fmt, m):
        return RotateTable(self, fmt, m)

    def nco(self, sample_rate):
        return Nco(self, sample_rate)

    # -- stream.DecimateBuffer / DownsampleBuffer --
    def decimate(self, to, frm, factor, offset=0):
        n = C.c_size_t(0)
        self._ck(lib.hzsdr_decimate(self._h, fmt_of(to), _ptr(to), length(to), fmt_of(frm), _ptr(frm),
                                    length(frm), factor, offset, C.byref(n)))
        return n.value

    def downsample(self, to, frm, factor, offset=0):
        n = C.c_size_t(0)
        self._ck(lib.hzsdr_downsample(self._h, fmt_of(to), _ptr(to), length(to), fmt_of(frm),
                                      _ptr(frm), length(frm), factor, offset, C.byref(n)))
        return n.value

    # -- fft.Planner (fft/fft.go:45-48) --
    def fft_plan(self, iq, freq, direction, batch=None):
        return FftPlan(self, iq, freq, direction, batch)

    def convolve_freq(self, dst, src, freq):
        return Convolver(self, "freq", dst, src, freq)

    def convolve(self, dst, iq1, iq2):
        return Convolver(self, "convolve", dst, iq1, iq2)

    def cross_correlate(self, dst, iq1, iq2):
        return Convolver(self, "xcorr", dst, iq1, iq2)

    def convolve_once(self, dst, iq1, iq2):
        """fft.ConvolveOnce (fft/convolution.go:200-211)."""
        conv = self.convolve(dst, iq1, iq2)
        try:
            conv()
        finally:
            conv.close()

    def fft_shift(self, frequency):
        """fft.Shift (fft/result.go:230-236, :84-98): swap the halves, 0 Hz to the centre
        (x / 1.0 is exact, so this is FFTShiftAndScale with scale 1)."""
        self.fftshift_scale(frequency, 1.0)

    def convolution_blocks(self, out, inp, filt):
        n = C.c_size_t(0)
        self._ck(lib.hzsdr_convolution_blocks(self._h, _ptr(out), length(out), _ptr(inp), length(inp),
                                              _ptr(filt), length(filt), C.byref(n)))
        return n.value

    # -- stream.ReadBeamform data path --
    def beamform(self, out, channels, weights, accumulate=None):
        k = len(channels)
        arr = (C.c_void_p * k)(*[_ptr(c) for c in channels])
        w = np.ascontiguousarray(weights, np.complex64)
        assert w.size == k
        wp = w.ctypes.data_as(C.POINTER(C.c_float))
        fmt = fmt_of(channels[0])
        if accumulate is None:
            self._ck(lib.hzsdr_beamform(self._h, _ptr(out), fmt, arr, wp, k, length(out)))
        else:
            self._ck(lib.hzsdr_beamform_partial(self._h, _ptr(out), fmt, arr, wp, k, length(out),
                                                1 if accumulate else 0))

    # -- rtl/kerberos coherent sync reductions (align.go:128-149, 257-266) --
    def peak_lag(self, corr):
        lag = C.c_int64(0)
        self._ck(lib.hzsdr_peak_lag(self._h, _ptr(corr), length(corr), C.byref(lag)))
        return lag.value

    def mean_phase(self, a, b):
        v = C.c_double(0.0)
        self._ck(lib.hzsdr_mean_phase(self._h, _ptr(a), _ptr(b), length(a), C.byref(v)))
        return v.value

    def fftshift_scale(self, data, scale):
        self._ck(lib.hzsdr_fftshift_scale(self._h, _ptr(data), length(data), float(scale)))

    # -- one block of GraftReaders (rtl/kerberos/internal/graft.go:63-122) --
    def graft(self, out, channels):
        k = len(channels)
        arr = (C.c_void_p * k)(*[_ptr(c) for c in channels])
        self._ck(lib.hzsdr_graft(self._h, _ptr(out), length(out), arr, k, length(channels[0])))

    # -- foreign-endian ByteReader / ByteWriter payloads (bytes_io.go) --
    def byteswap(self, buf):
        self._ck(lib.hzsdr_byteswap(self._h, fmt_of(buf), _ptr(buf), length(buf)))

    def convert_foreign(self, dst, src, dst_foreign=False, src_foreign=False):
        """ConvertBuffer with a foreign-order ByteReader / ByteWriter payload on either side."""
        n = C.c_size_t(0)
        self._ck(lib.hzsdr_convert_foreign(self._h, fmt_of(dst), _ptr(dst), length(dst), int(dst_foreign),
                                           fmt_of(src), _ptr(src), length(src), int(src_foreign),
                                           C.byref(n)))
        return n.value

    def chain(self, src_fmt, sample_rate=0):
        return Chain(self, src_fmt, sample_rate)


class LookupTable:
    """sdr.LookupTable (iq_lookup_table.go:36-50)."""

    def __init__(self, ctx, src_fmt, table):
        self.ctx = ctx
        self.src_fmt = src_fmt
        self.dst_fmt = fmt_of(table)
        self._h = C.c_void_p()
        ctx._ck(lib.hzsdr_lut_create(ctx._h, src_fmt, self.dst_fmt, _ptr(table), length(table),
                                     C.byref(self._h)))

    def lookup(self, dst, src):
        n = C.c_size_t(0)
        self.ctx._ck(lib.hzsdr_lut_lookup(self._h, fmt_of(dst), _ptr(dst), length(dst), fmt_of(src),
                                          _ptr(src), length(src), C.byref(n)))
        return n.value

    def source_sample_format(self):
        return self.src_fmt

    def destination_sample_format(self):
        return self.dst_fmt

    def close(self):
        if self._h:
            lib.hzsdr_lut_free(self._h)
            self._h = C.c_void_p()


class RotateTable:
    """uint8MultiplyReader / int8MultiplyReader table (stream/multiply.go:91-238)."""

    def __init__(self, ctx, fmt, m):
        self.ctx = ctx
        self.fmt = fmt
        self._h = C.c_void_p()
        m = complex(m)
        ctx._ck(lib.hzsdr_rotlut_create(ctx._h, fmt, m.real, m.imag, C.byref(self._h)))

    def set_multiplier(self, m):
        m = complex(m)
        self.ctx._ck(lib.hzsdr_rotlut_set_multiplier(self._h, m.real, m.imag))

    def apply(self, buf):
        self.ctx._ck(lib.hzsdr_rotlut_apply(self._h, _ptr(buf), length(buf)))

    def close(self):
        if self._h:
            lib.hzsdr_rotlut_free(self._h)
            self._h = C.c_void_p()


class Nco:
    """The closure stream.ShiftBuffer(sampleRate) returns (stream/shifter.go:66-85)."""

    def __init__(self, ctx, sample_rate):
        self.ctx = ctx
        self._h = C.c_void_p()
        ctx._ck(lib.hzsdr_nco_create(ctx._h, int(sample_rate), C.byref(self._h)))

    def __call__(self, freq_hz, buf):
        self.ctx._ck(lib.hzsdr_nco_shift(self._h, float(freq_hz), _ptr(buf), length(buf)))

    @property
    def ts(self):
        v = C.c_double(0)
        lib.hzsdr_nco_get_time(self._h, C.byref(v))
        return v.value

    @ts.setter
    def ts(self, v):
        lib.hzsdr_nco_set_time(self._h, float(v))

    def set_ulp1(self, on=True):
        """Opt in to the <= 1-ulp rotation factor (include/hzsdr.h: hzsdr_nco_set_ulp1); the default is bit-exact."""
        self.ctx._ck(lib.hzsdr_nco_set_ulp1(self._h, int(on)))
        return self

    def close(self):
        if self._h:
            lib.hzsdr_nco_free(self._h)
            self._h = C.c_void_p()


class FftPlan:
    """fft.Plan (fft/fft.go:52-59)."""

    def __init__(self, ctx, iq, freq, direction, batch=None):
        self.ctx = ctx
        self._keep = (iq, freq)  # a plan aliases both buffers for its whole life
        self._h = C.c_void_p()
        fwd = FFT_FORWARD if direction in (True, FFT_FORWARD) else FFT_BACKWARD
        if batch is None:
            ctx._ck(lib.hzsdr_fft_plan(ctx._h, _ptr(iq), length(iq), _ptr(freq), length(freq), fwd,
                                       C.byref(self._h)))
        else:
            n = length(iq) // batch
            ctx._ck(lib.hzsdr_fft_plan_batch(ctx._h, _ptr(iq), _ptr(freq), n, batch, fwd,
                                             C.byref(self._h)))

    def transform(self):
        self.ctx._ck(lib.hzsdr_fft_transform(self._h))

    def close(self):
        if self._h:
            lib.hzsdr_fft_free(self._h)
            self._h = C.c_void_p()


class Convolver:
    """The func() error closures of fft/convolution.go:97,119,150."""

    def __init__(self, ctx, kind, dst, a, b):
        self.ctx = ctx
        self._keep = (dst, a, b)
        self._h = C.c_void_p()
        if kind == "freq":
            ctx._ck(lib.hzsdr_convolve_freq_create(ctx._h, _ptr(dst), length(dst), _ptr(a), length(a),
                                                   _ptr(b), length(b), C.byref(self._h)))
        else:
            mode = CONV_CONVOLVE if kind == "convolve" else CONV_CROSS_CORRELATE
            ctx._ck(lib.hzsdr_convolve_create(ctx._h, _ptr(dst), length(dst), _ptr(a), length(a),
                                              _ptr(b), length(b), mode, C.byref(self._h)))

    def __call__(self):
        self.ctx._ck(lib.hzsdr_conv_exec(self._h))

    def set_filter(self, freq):
        """ConvolveFreq only: the reference closure reads `freq` on every call
        (fft/convolution.go:183-189); here an updated filter is handed over explicitly."""
        self.ctx._ck(lib.hzsdr_conv_set_filter(self._h, _ptr(freq), length(freq)))

    def close(self):
        if self._h:
            lib.hzsdr_conv_free(self._h)
            self._h = C.c_void_p()


class Chain:
    """hzsdr_chain: nested stream.* Readers fused into one launch per buffer."""

    def __init__(self, ctx, src_fmt, sample_rate=0):
        self.ctx = ctx
        self.src_fmt = src_fmt
        self._h = C.c_void_p()
        ctx._ck(lib.hzsdr_chain_create(ctx._h, src_fmt, int(sample_rate), C.byref(self._h)))

    def shift(self, hz):
        self.ctx._ck(lib.hzsdr_chain_shift(self._h, float(hz)))
        return self

    def gain(self, r):
        self.ctx._ck(lib.hzsdr_chain_gain(self._h, float(r)))
        return self

    def rotate(self, m):
        m = complex(m)
        self.ctx._ck(lib.hzsdr_chain_rotate(self._h, m.real, m.imag))
        return self

    def decimate(self, factor):
        self.ctx._ck(lib.hzsdr_chain_decimate(self._h, factor))
        return self

    def downsample(self, factor):
        self.ctx._ck(lib.hzsdr_chain_downsample(self._h, factor))
        return self

    def convolution(self, filter_freq, decimate=1):
        self.ctx._ck(lib.hzsdr_chain_convolution(self._h, _ptr(filter_freq), length(filter_freq),
                                                 decimate))
        return self

    def fir_decimate(self, taps, factor):
        t = np.ascontiguousarray(taps, np.complex64)
        self.ctx._ck(lib.hzsdr_chain_fir_decimate(self._h, t.ctypes.data_as(C.POINTER(C.c_float)),
                                                  t.size, factor))
        return self

    def mix_in_order(self, in_order=True):
        self.ctx._ck(lib.hzsdr_chain_mix_in_order(self._h, int(in_order)))
        return self

    def fir_options(self, impl=0, nfft_min=0, loop_form=0):
        """In front of fir_decimate: which implementation the terminal takes (FIR_IMPL_AUTO / _TRANSFORMS /
        _MATRIX_CHUNKS), the smallest overlap-save block, the matrix loop's form -- include/hzsdr.h:
        hzsdr_chain_fir_options (measurements and tests; AUTO is the library's choice)."""
        self.ctx._ck(lib.hzsdr_chain_fir_options(self._h, int(impl), int(nfft_min), int(loop_form)))
        return self

    def pipeline(self, on=True):
        """Opt in to overlapping consecutive runs of a FIR-decimate chain on the matrix path or of a map chain
        (include/hzsdr.h: hzsdr_chain_pipeline).  run() stays an ordinary call on the context's stream; the overlap
        is taken by run_after() / run_batch_after(), where the caller says what the buffers wait for.  Results are
        bit-identical."""
        self.ctx._ck(lib.hzsdr_chain_pipeline(self._h, int(on)))
        return self

    def shift_ulp1(self, on=True):
        """Opt in to the <= 1-ulp Shift of terminal-less chains (include/hzsdr.h: hzsdr_chain_shift_ulp1)."""
        self.ctx._ck(lib.hzsdr_chain_shift_ulp1(self._h, int(on)))
        return self

    def plan(self, n_in):
        a, b = C.c_size_t(0), C.c_size_t(0)
        self.ctx._ck(lib.hzsdr_chain_plan(self._h, n_in, C.byref(a), C.byref(b)))
        return a.value, b.value

    def run(self, inp, out, n_in=None):
        a, b = C.c_size_t(0), C.c_size_t(0)
        n_in = length(inp) if n_in is None else n_in
        self.ctx._ck(lib.hzsdr_chain_run(self._h, _ptr(inp), n_in, _ptr(out), length(out),
                                         C.byref(a), C.byref(b)))
        return a.value, b.value

    def run_after(self, inp, out, ready_event=None, n_in=None):
        """run() whose start is ordered by the caller (hzsdr_chain_run_after): the buffers are ready when
        `ready_event` (a torch.cuda.Event that has been recorded, a raw hipEvent_t as an int, or None = now) has fired."""
        a, b = C.c_size_t(0), C.c_size_t(0)
        n_in = length(inp) if n_in is None else n_in
        self.ctx._ck(lib.hzsdr_chain_run_after(self._h, _ptr(inp), n_in, _ptr(out), length(out), C.byref(a), C.byref(b),
                                               _event(ready_event)))
        return a.value, b.value

    def run_batch(self, inps, outs, ready_event=None, after=False, n_in=None):
        """len(inps) consecutive buffers of the stream in one call (hzsdr_chain_run_batch; after=True or an event:
        hzsdr_chain_run_batch_after).  Returns (consumed, produced) per buffer."""
        k = len(inps)
        assert k == len(outs) and k >= 1
        pi = (C.c_void_p * k)(*[_ptr(x) for x in inps])
        po = (C.c_void_p * k)(*[_ptr(x) for x in outs])
        a, b = C.c_size_t(0), C.c_size_t(0)
        n_in = length(inps[0]) if n_in is None else n_in
        cap = min(length(o) for o in outs)
        if after or ready_event is not None:
            self.ctx._ck(lib.hzsdr_chain_run_batch_after(self._h, pi, po, k, n_in, cap, C.byref(a), C.byref(b), _event(ready_event)))
        else:
            self.ctx._ck(lib.hzsdr_chain_run_batch(self._h, pi, po, k, n_in, cap, C.byref(a), C.byref(b)))
        return a.value, b.value

    def reset(self):
        self.ctx._ck(lib.hzsdr_chain_reset(self._h))

    def set_time(self, ts):
        """The Shift closure's clock (stream/shifter.go:71): resume a stream / start a test near 2*pi."""
        self.ctx._ck(lib.hzsdr_chain_set_time(self._h, float(ts)))
        return self

    def time(self):
        t = C.c_double(0.0)
        self.ctx._ck(lib.hzsdr_chain_time(self._h, C.byref(t)))
        return t.value

    def last_fir_path(self):
        """FIR_PATH_NONE / _TRANSFORM / _MATRIX: the kernels the last run of a FIR-decimate chain used."""
        p = C.c_int32(0)
        self.ctx._ck(lib.hzsdr_chain_last_fir_path(self._h, C.byref(p)))
        return p.value

    def last_fir_kernel(self):
        """FIR_KERNEL_NONE / _TRANSFORM / _MATRIX_CHUNKS (csrc/hz_firmm.h) / _MATRIX_PASSES (csrc/hz_firmm2.h)."""
        p = C.c_int32(0)
        self.ctx._ck(lib.hzsdr_chain_last_fir_kernel(self._h, C.byref(p)))
        return p.value

    def ring(self, slot_length, slots=4):
        return Ring(self, slot_length, slots)

    def close(self):
        if self._h:
            lib.hzsdr_chain_free(self._h)
            self._h = C.c_void_p()


class _ShardCtx(Context):
    """A shard's context, owned by its MultiGpu (closing the MultiGpu closes it)."""

    def __init__(self, handle, device):
        self._h = handle
        self.device = device
        self.memspace = MEM_DEVICE
        self._pinned = []

    def close(self):
        self._h = C.c_void_p()


MGPU_ORDERED, MGPU_RCCL = 0, 1


class MultiGpu:
    """hzsdr_mgpu_*: Beamform sharded over GPUs from one process (stream/beamform.go:148-171
    with one channel range per GPU).  devices may repeat a GPU (several shards on one)."""

    def __init__(self, devices):
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        _check(lib.hzsdr_mgpu_open(arr, len(devices), C.byref(self._h)))
        self.devices = list(devices)
        self.shards = []
        for s in range(len(devices)):
            h = C.c_void_p()
            _check(lib.hzsdr_mgpu_ctx(self._h, s, C.byref(h)))
            self.shards.append(_ShardCtx(h, devices[s]))

    @staticmethod
    def shard_channels(k, g, s):
        lo, hi = C.c_int(0), C.c_int(0)
        _check(lib.hzsdr_mgpu_shard_channels(k, g, s, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def beamform(self, out, channels, weights, dst_shard=0, mode=MGPU_ORDERED):
        k = len(channels)
        fmt = fmt_of(channels[0])
        arr = (C.c_void_p * k)(*[_ptr(c) for c in channels])
        w = np.ascontiguousarray(weights, np.complex64)
        rc = lib.hzsdr_mgpu_beamform(self._h, _ptr(out), dst_shard, fmt, arr, w.ctypes.data_as(C.POINTER(C.c_float)),
                                     k, length(out), mode)
        if rc != 0:
            msg = lib.hzsdr_mgpu_last_error(self._h)
            raise _ERRORS.get(rc, HzsdrError)(f"hzsdr: {lib.hzsdr_strerror(rc).decode()}: {msg.decode() if msg else ''}")

    def peer_pairs(self):
        """(direct, staged): ordered pairs of distinct GPUs with / without peer access (hzsdr_mgpu_peer_pairs)."""
        d, st = C.c_int(0), C.c_int(0)
        _check(lib.hzsdr_mgpu_peer_pairs(self._h, C.byref(d), C.byref(st)))
        return d.value, st.value

    def synchronize(self):
        _check(lib.hzsdr_mgpu_synchronize(self._h))

    def close(self):
        if self._h:
            lib.hzsdr_mgpu_close(self._h)
            self._h = C.c_void_p()
            for s in self.shards:
                s.close()


def _view(ptr, fmt, n):
    """yikes.Samples(base, len, fmt) (yikes/bytes.go:50-71): numpy over foreign memory."""
    dt = {FMT_C64: np.complex64, FMT_U8: np.uint8, FMT_I16: np.int16, FMT_I8: np.int8}[fmt]
    nbytes = n * format_size(fmt)
    raw = np.frombuffer((C.c_ubyte * nbytes).from_address(ptr), np.uint8) if nbytes else np.zeros(0, np.uint8)
    a = raw.view(dt)
    return a if fmt == FMT_C64 else a.reshape(n, 2)


class Ring:
    """hzsdr_ring_*: the pinned stream.RingBuffer in front of a chain (stream/ring.go:48-69).
    `iq` is the whole IQBufferAllocator region; acquire() -> (slot, samples view),
    submit(slot, n), pop() -> complex64 view of that slot's output."""

    def __init__(self, chain, slot_length, slots=4):
        self.chain, self.ctx = chain, chain.ctx
        self._h = C.c_void_p()
        self.ctx._ck(lib.hzsdr_ring_create(chain._h, slot_length, slots, C.byref(self._h)))
        base, n, sl = C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
        self.ctx._ck(lib.hzsdr_ring_iq_buffer(self._h, C.byref(base), C.byref(n), C.byref(sl)))
        self.slot_length, self.slots, self.fmt = sl.value, slots, chain.src_fmt
        self.iq = _view(base.value, self.fmt, n.value)

    def acquire(self):
        slot, p = C.c_int(-1), C.c_void_p()
        self.ctx._ck(lib.hzsdr_ring_acquire(self._h, C.byref(slot), C.byref(p)))
        return slot.value, self.iq[slot.value * self.slot_length:(slot.value + 1) * self.slot_length]

    def submit(self, slot, n=None):
        self.ctx._ck(lib.hzsdr_ring_submit(self._h, slot, self.slot_length if n is None else n))

    def submit_many(self, first_slot, count, n=None):
        """`count` acquired slots (first_slot the oldest) in ONE call of the chain (hzsdr_ring_submit_many): one launch
        where the chain has that form (hzsdr_chain_run_batch's), the bits of `count` submits either way."""
        self.ctx._ck(lib.hzsdr_ring_submit_many(self._h, first_slot, count, self.slot_length if n is None else n))

    def release(self, slot):
        """The NEWEST acquired slot, unused (hzsdr_ring_release)."""
        self.ctx._ck(lib.hzsdr_ring_release(self._h, slot))

    def pop(self):
        p, n = C.c_void_p(), C.c_size_t(0)
        self.ctx._ck(lib.hzsdr_ring_pop(self._h, C.byref(p), C.byref(n)))
        return _view(p.value, FMT_C64, n.value)

    @property
    def in_flight(self):
        return lib.hzsdr_ring_in_flight(self._h)

    def close(self):
        if self._h:
            self.iq = None
            lib.hzsdr_ring_free(self._h)
            self._h = C.c_void_p()

"""ctypes front-end of the CPU parity oracle (oracle/hzsdr_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the product package (go-sdr_amd/).

Sample buffers are numpy arrays in the reference's memory layouts
(SURVEY.md section 8): u8/i8 = (n, 2) bytes, i16 = (n, 2) int16, c64 = complex64 (n,).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

FMT_C64, FMT_U8, FMT_I16, FMT_I8 = 1, 2, 3, 4
ERR_FORMAT_MISMATCH, ERR_FORMAT_UNKNOWN, ERR_DST_TOO_SMALL, ERR_NOT_IMPLEMENTED, ERR_LENGTH = (
    -1, -2, -3, -4, -5)

DTYPES = {FMT_C64: np.complex64, FMT_U8: np.uint8, FMT_I16: np.int16, FMT_I8: np.int8}


def build(force=False):
    """Compile liboracle.so with gcc (contraction off)."""
    srcs = [os.path.join(_HERE, f) for f in ("hzsdr_oracle.c", "oracle_parallel.c")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_convert.restype = C.c_long
        _lib.orc_lut_apply.restype = C.c_long
        _lib.orc_decimate.restype = C.c_long
        _lib.orc_downsample.restype = C.c_long
        _lib.orc_convolution_reader.restype = C.c_long
        _lib.orc_go_mpi4.restype = C.POINTER(C.c_uint64)
    return _lib


def make_samples(fmt, n):
    if fmt == FMT_C64:
        return np.zeros(n, np.complex64)
    return np.zeros((n, 2), DTYPES[fmt])


def fmt_of(a):
    if a.dtype == np.complex64:
        return FMT_C64
    return {np.dtype(np.uint8): FMT_U8, np.dtype(np.int16): FMT_I16,
            np.dtype(np.int8): FMT_I8}[a.dtype]


def length(a):
    return a.shape[0]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def convert(dst, src, n=None, dst_fmt=None, src_fmt=None):
    """sdr.ConvertBuffer(dst, src) -> count or negative error (conv.go:55)."""
    n = length(src) if n is None else n
    return lib().orc_convert(C.c_int(dst_fmt or fmt_of(dst)), _p(dst), C.c_long(length(dst)),
                             C.c_int(src_fmt or fmt_of(src)), _p(src), C.c_long(n))


def i16_shift_lsb_to_msb(buf, bits):
    lib().orc_i16_shift_lsb_to_msb(_p(buf), C.c_long(length(buf)), C.c_int(bits))


def scale(buf, r):
    lib().orc_scale(C.c_float(r), _p(buf), C.c_long(length(buf)))


def rotate(buf, m):
    m = np.complex64(m)
    lib().orc_rotate(C.c_float(m.real), C.c_float(m.imag), _p(buf), C.c_long(length(buf)))


def add(a, b, c):
    return lib().orc_add(_p(a), C.c_long(length(a)), _p(b), C.c_long(length(b)), _p(c),
                         C.c_long(length(c)))


def sum_(out, bufs):
    """stream.Add Read data path: out = ((0 + b0) + b1) + ..."""
    k = len(bufs)
    arr = (C.c_void_p * k)(*[b.ctypes.data for b in bufs])
    fn = {FMT_C64: lib().orc_sum_c64, FMT_I16: lib().orc_sum_i16, FMT_I8: lib().orc_sum_i8}[
        fmt_of(out)]
    fn(_p(out), arr, C.c_int(k), C.c_long(length(out)))


def lut_identity():
    t = np.zeros((65536, 2), np.uint8)
    lib().orc_lut_identity(_p(t))
    return t


def lut_apply(dst, tab, src):
    return lib().orc_lut_apply(C.c_int(fmt_of(dst)), _p(dst), C.c_long(length(dst)), _p(tab),
                               _p(src), C.c_long(length(src)))


def rotate_table_u8(m):
    m = np.complex64(m)
    t = np.zeros((65535, 2), np.uint8)
    lib().orc_rotate_table_u8(C.c_float(m.real), C.c_float(m.imag), _p(t))
    return t


def rotate_u8_apply(tab, buf):
    lib().orc_rotate_u8_apply(_p(tab), _p(buf), C.c_long(length(buf)))


def rotate_table_i8(m):
    m = np.complex64(m)
    t = np.zeros((65536, 2), np.int8)
    lib().orc_rotate_table_i8(C.c_float(m.real), C.c_float(m.imag), _p(t))
    return t


def go_sincos(x):
    x = np.ascontiguousarray(x, np.float64)
    s = np.empty_like(x)
    c = np.empty_like(x)
    fn = lib().orc_go_sincos
    sv, cv = C.c_double(), C.c_double()
    flat = x.ravel()
    so, co = s.ravel(), c.ravel()
    for i in range(flat.size):
        fn(C.c_double(flat[i]), C.byref(sv), C.byref(cv))
        so[i], co[i] = sv.value, cv.value
    return s, c


def sincos_narrow_check(xs):
    """-> (accepted, wrong): arguments the restated sincos_narrow decides, and those of them whose float32 pair differs
    from complex64(math.Sincos) (the device function's claim is that there are none)."""
    xs = np.ascontiguousarray(xs, np.float64)
    acc, bad = C.c_long(), C.c_long()
    lib().orc_sincos_narrow_check(_p(xs), C.c_long(xs.size), C.byref(acc), C.byref(bad))
    return acc.value, bad.value


def go_mpi4():
    p = lib().orc_go_mpi4()
    return [int(p[i]) for i in range(20)]


class Shifter:
    """stream.ShiftBuffer(sampleRate) closure (stream/shifter.go:66-85)."""

    def __init__(self, sample_rate, use_libm=False):
        self.sample_rate = int(sample_rate)
        self.ts = C.c_double(0.0)
        self.use_libm = int(use_libm)

    def __call__(self, freq_hz, buf):
        lib().orc_shift(C.byref(self.ts), C.c_ulong(self.sample_rate), C.c_double(freq_hz),
                        _p(buf), C.c_long(length(buf)), C.c_int(self.use_libm))

    def ts_sequence(self, n):
        out = np.empty(n, np.float64)
        lib().orc_shift_ts(C.byref(self.ts), C.c_ulong(self.sample_rate), _p(out), C.c_long(n))
        return out


def decimate(to, frm, factor, offset=0):
    return lib().orc_decimate(C.c_int(fmt_of(to)), _p(to), C.c_long(length(to)),
                              C.c_int(fmt_of(frm)), _p(frm), C.c_long(length(frm)),
                              C.c_uint(factor), C.c_long(offset))


def downsample(to, frm, factor, offset=0):
    return lib().orc_downsample(_p(to), C.c_long(length(to)), C.c_int(fmt_of(to)),
                                C.c_int(fmt_of(frm)), _p(frm), C.c_long(length(frm)),
                                C.c_uint(factor), C.c_long(offset))


def fft(inp, out, forward=True):
    return lib().orc_fft(_p(inp), C.c_long(length(inp)), _p(out), C.c_long(length(out)),
                         C.c_int(1 if forward else 0))


def convolve_freq(dst, src, freq):
    return lib().orc_convolve_freq(_p(dst), _p(src), _p(freq), C.c_long(length(src)))


def convolve(dst, iq1, iq2, conj=False):
    return lib().orc_convolve(_p(dst), _p(iq1), _p(iq2), C.c_long(length(iq1)), C.c_int(conj))


def convolution_reader(out, inp, filt):
    return lib().orc_convolution_reader(_p(out), _p(inp), C.c_long(length(inp)), _p(filt),
                                        C.c_long(length(filt)))


def fir_decimate_f64(out, x, taps, d, hist=None):
    lib().orc_fir_decimate_f64(_p(out), _p(x), C.c_long(length(x)), _p(taps),
                               C.c_long(length(taps)), C.c_uint(d),
                               _p(hist) if hist is not None else None)


def beamform_angles_2d(freq_hz, angle_deg, center, antennas):
    ant = np.ascontiguousarray(antennas, np.float64).reshape(-1, 2)
    if ant.shape[0] == 0:
        return None
    ctr = np.ascontiguousarray(center, np.float64)
    out = np.zeros(ant.shape[0], np.complex64)
    lib().orc_beamform_angles_2d(C.c_double(freq_hz), C.c_double(angle_deg), _p(ctr), _p(ant),
                                 C.c_int(ant.shape[0]), _p(out))
    return out


def beamform_angles(freq_hz, angle_deg, distances):
    d = np.ascontiguousarray(distances, np.float64)
    if d.size == 0:
        return None
    out = np.zeros(d.size, np.complex64)
    lib().orc_beamform_angles(C.c_double(freq_hz), C.c_double(angle_deg), _p(d), C.c_int(d.size),
                              _p(out))
    return out


def beamform(out, chans, weights):
    k = len(chans)
    arr = (C.c_void_p * k)(*[c.ctypes.data for c in chans])
    w = np.ascontiguousarray(weights, np.complex64)
    lib().orc_beamform(_p(out), arr, _p(w), C.c_int(k), C.c_long(length(out)))


def cw(n, freq, sample_rate, phase=0.0):
    buf = np.zeros(n, np.complex64)
    lib().orc_cw(_p(buf), C.c_long(n), C.c_double(freq), C.c_long(int(sample_rate)),
                 C.c_double(phase))
    return buf


# --- cpu_baseline drivers (bench.py only) -----------------------------------

def max_threads():
    return lib().orc_max_threads()


def par_u8_to_c64(src, dst, threads):
    lib().orc_par_u8_to_c64(_p(src), _p(dst), C.c_long(length(src)), C.c_int(threads))


def par_shift_gain(ts, sample_rate, freq_hz, gain, buf, threads):
    t = C.c_double(ts)
    lib().orc_par_shift_gain(C.byref(t), C.c_ulong(int(sample_rate)), C.c_double(freq_hz),
                             C.c_float(gain), _p(buf), C.c_long(length(buf)), C.c_int(threads))
    return t.value


def par_fir_decimate_f64(out, x, taps, d, hist=None, threads=None):
    """fir_decimate_f64 over `threads` cores (default: all), bit-identical to the serial form."""
    taps = np.ascontiguousarray(taps, np.complex64)
    lib().orc_par_fir_decimate_f64(_p(out), _p(x), C.c_long(length(x)), _p(taps), C.c_long(len(taps)),
                                   C.c_uint(d), _p(hist) if hist is not None else None,
                                   C.c_int(threads or max_threads()))
    return out


def par_chain_fir(src_u8, out, sample_rate, freq_hz, taps, d, threads, scratch=None):
    """u8 -> c64 -> Shift -> FIR -> decimate by d from a fresh stream, OpenMP over `threads`."""
    n = length(src_u8)
    buf = scratch if scratch is not None else np.zeros(n, np.complex64)
    taps = np.ascontiguousarray(taps, np.complex64)
    lib().orc_par_chain_fir(_p(src_u8), _p(buf), _p(out), C.c_long(n), C.c_ulong(int(sample_rate)),
                            C.c_double(freq_hz), _p(taps), C.c_long(len(taps)), C.c_uint(d), C.c_int(threads))
    return out


def peak_lag(corr):
    lag = C.c_long(0)
    lib().orc_peak_lag(_p(corr), C.c_long(length(corr)), C.byref(lag))
    return lag.value


def mean_phase(a, b):
    fn = lib().orc_mean_phase
    fn.restype = C.c_double
    return fn(_p(a), _p(b), C.c_long(length(a)))


def fftshift_scale(data, scale):
    lib().orc_fftshift_scale(_p(data), C.c_long(length(data)), C.c_float(scale))


def graft(out, bands):
    k = len(bands)
    arr = (C.c_void_p * k)(*[b.ctypes.data for b in bands])
    return lib().orc_graft(_p(out), arr, C.c_int(k), C.c_long(length(bands[0])))


def byteswap(buf):
    """In place over a c64 (4-byte components) or i16 (2-byte components) buffer."""
    width = 4 if buf.dtype == np.complex64 else buf.dtype.itemsize
    if width == 1:
        return
    lib().orc_byteswap(_p(buf), C.c_long(buf.nbytes // width), C.c_int(width))

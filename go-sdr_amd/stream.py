"""Reader-level mirror of the reference's `sdr` / `stream` / `fft` packages for the
hot path, with every per-buffer `Proc` running on the GPU through the C ABI.

Same names, argument meaning and error behaviour as the reference so that tests
read like the reference's own (`stream/*_test.go`):

    sdr.Reader            -> Reader (read / sample_format / sample_rate)
    sdr.ReadFull          -> read_full
    mock.Sdr / sdr.Pipe   -> BufferReader (an in-memory source; the reference tests
                             push known buffers through a Pipe from a goroutine)
    stream.ReadTransformer, ConvertReader, DecimateReader, DownsampleReader,
    ShiftReader, Gain, Multiply, Add, ConvolutionReader, ReadBeamform, fft planner

The reference runs one goroutine per ReadTransformer stage joined by rendezvous
pipes (stream/read_transformer.go:82); here a stage pulls its input block when
its output is read -- same block structure, no threads -- because the data path
is the GPU call, not the plumbing.  Readers work on HOST contexts (numpy buffers),
exactly what a cgo caller with Go slices gets.
"""
import sys

import numpy as np

from . import (FMT_C64, FMT_I8, FMT_I16, FMT_U8, FFT_BACKWARD, FFT_FORWARD, Context,
               ErrDstTooSmall, ErrSampleFormatMismatch, ErrSampleFormatUnknown, HzsdrError,
               beamform_angles, beamform_angles_2d, fmt_of, format_size, length, make_samples)

READER_BLOCK = 32 * 1024  # stream/convert.go:43-44, decimate.go:41-42, downsample.go:54-55


class ErrShortBuffer(HzsdrError):      # reader.go:28-31
    pass


class ErrUnexpectedEOF(HzsdrError):    # reader.go:33-36
    pass


class EOF(Exception):                  # io.EOF
    pass


class Reader:
    """sdr.Reader (reader.go:39-51)."""

    def read(self, samples):
        raise NotImplementedError

    def sample_format(self):
        raise NotImplementedError

    def sample_rate(self):
        raise NotImplementedError


def read_at_least(r, buf, minimum):
    """sdr.ReadAtLeast (reader.go:94-117)."""
    if length(buf) < minimum:
        raise ErrShortBuffer("sdr: short buffer")
    n = 0
    while n < minimum:
        try:
            nn = r.read(buf[n:])
        except EOF as e:
            # Go returns (n, err); the count rides on the exception as `.n`
            err = ErrUnexpectedEOF("sdr: unexpected EOF") if n > 0 else e
            err.n = n
            raise err
        except HzsdrError as e:
            e.n = n
            raise
        n += nn
    return n


def read_full(r, buf):
    """sdr.ReadFull (reader.go:72-74)."""
    return read_at_least(r, buf, length(buf))


class BufferReader(Reader):
    """An in-memory source of one sample format: what mock.Sdr's Rx / a written
    sdr.Pipe are to the reference's tests.  max_read bounds a single Read so that
    callers exercise their ReadFull loops."""

    def __init__(self, data, sample_rate, max_read=None):
        self.data, self.rate, self.pos, self.max_read = data, int(sample_rate), 0, max_read

    def read(self, samples):
        if fmt_of(samples) != fmt_of(self.data):  # testutils/reader.go:87-97
            raise ErrSampleFormatMismatch("sdr: iq sample formats do not match")
        left = length(self.data) - self.pos
        if left == 0:
            raise EOF()
        n = min(left, length(samples))
        if self.max_read:
            n = min(n, self.max_read)
        samples[:n] = self.data[self.pos:self.pos + n]
        self.pos += n
        return n

    def sample_format(self):
        return fmt_of(self.data)

    def sample_rate(self):
        return self.rate


class ReadTransformer(Reader):
    """stream.ReadTransformer (stream/read_transformer.go:45-137): ReadFull an
    input block, Proc(in, out) -> n, hand out the n output samples.  A read error
    on the input (including a partial last block) ends the stream, as `run` does."""

    def __init__(self, inp, input_len, output_len, out_format, out_rate, proc):
        self.inp, self.proc = inp, proc
        self.in_buf = make_samples(inp.sample_format(), input_len)
        self.out_buf = make_samples(out_format, output_len)
        self.out_format, self.out_rate = out_format, int(out_rate)
        self.avail, self.off, self.err = 0, 0, None

    def read(self, samples):
        if fmt_of(samples) != self.out_format:
            raise ErrSampleFormatMismatch("sdr: iq sample formats do not match")
        if self.avail == 0:
            if self.err is not None:
                raise self.err
            try:
                inn = read_full(self.inp, self.in_buf)
                self.avail = self.proc(self.in_buf[:inn], self.out_buf)
                self.off = 0
            except (EOF, HzsdrError) as e:  # CloseWithError (read_transformer.go:121-135)
                self.err = EOF() if isinstance(e, ErrUnexpectedEOF) else e
                raise self.err
            if self.avail == 0:
                return 0
        n = min(self.avail, length(samples))
        samples[:n] = self.out_buf[self.off:self.off + n]
        self.off += n
        self.avail -= n
        return n

    def sample_format(self):
        return self.out_format

    def sample_rate(self):
        return self.out_rate


class Writer:
    """sdr.Writer (writer.go:31-43)."""

    def write(self, samples):
        raise NotImplementedError

    def sample_format(self):
        raise NotImplementedError

    def sample_rate(self):
        raise NotImplementedError


class BufferWriter(Writer):
    """An in-memory sink: the read end of an sdr.Pipe in the reference's writer tests."""

    def __init__(self, fmt, sample_rate):
        self.fmt, self.rate, self.chunks = fmt, int(sample_rate), []

    def write(self, samples):
        if fmt_of(samples) != self.fmt:
            raise ErrSampleFormatMismatch("sdr: iq sample formats do not match")
        self.chunks.append(np.array(samples, copy=True))
        return length(samples)

    def sample_format(self):
        return self.fmt

    def sample_rate(self):
        return self.rate

    def samples(self):
        return np.concatenate(self.chunks) if self.chunks else make_samples(self.fmt, 0)


class ConvertWriter(Writer):
    """stream.ConvertWriter (stream/convert.go:53-118): accepts `input_format`, converts
    32 Ki samples at a time on the GPU and writes them to `out` in out's format."""

    def __init__(self, ctx, out, input_format):
        self.ctx, self.out, self.input_format = ctx, out, input_format
        self.buffer = make_samples(out.sample_format(), READER_BLOCK)

    def sample_format(self):
        return self.input_format

    def sample_rate(self):
        return self.out.sample_rate()

    def write(self, samples):
        if fmt_of(samples) != self.input_format:
            raise ErrSampleFormatMismatch("sdr: iq sample formats do not match")
        n = 0
        for i in range(0, length(samples), READER_BLOCK):
            ie = min(i + READER_BLOCK, length(samples))
            got = self.ctx.convert(self.buffer, samples[i:ie])
            if got != ie - i:
                raise HzsdrError("ConvertWriter: Conversion mismatch")
            n += self.out.write(self.buffer[:got])
        return n


NATIVE_ENDIAN = sys.byteorder  # "little" on every platform this runs on


class ByteReader(Reader):
    """sdr.ByteReader (bytes_io.go:196-219): IQ samples decoded from an io stream of raw
    bytes.  Native byte order reads straight into the buffer (bytes_io.go:170-181); the
    other order (byteReaderForeign, :125-160) reads the same bytes and reverses every
    int16 / float32 component on the GPU.  Like binary.Read, the foreign i16 / c64 path
    wants the whole buffer (a short read is ErrUnexpectedEOF, nothing at all is EOF)."""

    def __init__(self, ctx, r, byte_order, samples_per_second, sample_format):
        self.ctx, self.r, self.rate, self.fmt = ctx, r, int(samples_per_second), sample_format
        self.foreign = byte_order != NATIVE_ENDIAN

    def sample_format(self):
        return self.fmt

    def sample_rate(self):
        return self.rate

    def read(self, samples):
        if fmt_of(samples) != self.fmt:
            raise ErrSampleFormatMismatch("sdr: iq sample formats do not match")
        raw = samples.view(np.uint8).reshape(-1)
        whole = self.foreign and self.fmt in (FMT_I16, FMT_C64)
        got = 0
        while got < raw.size:
            chunk = self.r.read(raw.size - got)
            if not chunk:
                break
            raw[got:got + len(chunk)] = np.frombuffer(chunk, np.uint8)
            got += len(chunk)
            if not whole:
                break
        size = format_size(self.fmt)
        if got == 0 and raw.size:
            raise EOF()
        if whole and got < raw.size:
            raise ErrUnexpectedEOF("sdr: unexpected EOF")
        n = got // size
        if whole:
            self.ctx.byteswap(samples)
        return n


class ByteWriter:
    """sdr.ByteWriter (bytes_io.go:98-121): the inverse; a foreign order swaps a copy
    on the GPU (byteWriterForeign, :30-64) and writes that."""

    def __init__(self, ctx, w, byte_order, samples_per_second, sample_format):
        self.ctx, self.w, self.rate, self.fmt = ctx, w, int(samples_per_second), sample_format
        self.foreign = byte_order != NATIVE_ENDIAN

    def sample_format(self):
        return self.fmt

    def sample_rate(self):
        return self.rate

    def write(self, samples):
        if fmt_of(samples) != self.fmt:
            raise ErrSampleFormatMismatch("sdr: iq sample formats do not match")
        out = samples
        if self.foreign and self.fmt in (FMT_I16, FMT_C64):
            out = np.ascontiguousarray(samples).copy()
            self.ctx.byteswap(out)
        self.w.write(out.view(np.uint8).reshape(-1).tobytes())
        return length(samples)


class Stream:
    """The operators, bound to one GPU context (the reference's package-level
    functions take no context; a cgo shim would hold a package-level one)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    # stream.ConvertReader, stream/convert.go:37-51
    def convert_reader(self, inp, to):
        return ReadTransformer(inp, READER_BLOCK, READER_BLOCK, to, inp.sample_rate(),
                               lambda i, o: self.ctx.convert(o, i))

    # stream.ConvertWriter, stream/convert.go:53-118
    def convert_writer(self, out, input_format):
        return ConvertWriter(self.ctx, out, input_format)

    # stream.DecimateReader, stream/decimate.go:34-55
    def decimate_reader(self, inp, factor):
        state = {"offset": 0}

        def proc(i, o):
            n = self.ctx.decimate(o, i, factor, state["offset"])
            state["offset"] += length(i)
            return n
        return ReadTransformer(inp, READER_BLOCK, READER_BLOCK, inp.sample_format(),
                               inp.sample_rate() // factor, proc)

    # stream.DownsampleReader, stream/downsample.go:47-64
    def downsample_reader(self, inp, factor):
        state = {"offset": 0}

        def proc(i, o):
            n = self.ctx.downsample(o, i, factor, state["offset"])
            state["offset"] += length(i)
            return n
        return ReadTransformer(inp, READER_BLOCK, READER_BLOCK, FMT_C64,
                               inp.sample_rate() // factor, proc)

    # stream.ShiftReader, stream/shifter.go:89-102
    def shift_reader(self, r, shift_hz):
        if r.sample_format() != FMT_C64:
            raise ErrSampleFormatUnknown("sdr: iq sample format is not understood")
        return _ShiftReader(self.ctx, r, shift_hz)

    # stream.Gain, stream/gain.go:30-57
    def gain(self, r, v):
        return _GainReader(self.ctx, r, v)

    # stream.Multiply, stream/multiply.go:74-89
    def multiply(self, r, m):
        f = r.sample_format()
        if f == FMT_C64:
            return _MultiplyReader(self.ctx, r, m)
        if f in (FMT_U8, FMT_I8):
            return _TableMultiplyReader(self.ctx, r, m)
        raise ErrSampleFormatUnknown("sdr: iq sample format is not understood")

    # stream.Add, stream/add.go:41-82
    def add(self, *readers):
        if len(readers) == 0:
            raise HzsdrError("stream.Add: No readers passed")
        if len(readers) == 1:
            return readers[0]
        f, rate = readers[0].sample_format(), readers[0].sample_rate()
        if f not in (FMT_C64, FMT_I16, FMT_I8):
            raise ErrSampleFormatUnknown("sdr: iq sample format is not understood")
        for r in readers:
            if r.sample_format() != f:
                raise HzsdrError("stream.Add: Readers are not all the same format")
            if r.sample_rate() != rate:
                raise HzsdrError("stream.Add: Readers are not all the same rate")
        return _AddReader(self.ctx, readers, f, rate)

    # stream.ConvolutionReader, stream/convolution.go:36-82
    def convolution_reader(self, r, filter_bins):
        if r.sample_format() != FMT_C64:
            raise ErrSampleFormatUnknown("sdr: iq sample format is not understood")
        flen = len(filter_bins)
        filt = np.ascontiguousarray(filter_bins, np.complex64)
        iq = make_samples(FMT_C64, flen)
        conv = self.ctx.convolve_freq(iq, iq, filt)  # fft.ConvolveFreq(planner, iq, iq, filter)

        def proc(i, o):
            iq[:length(i)] = i
            conv()
            o[:length(i)] = iq[:length(i)]
            return length(i)
        return ReadTransformer(r, flen, flen, FMT_C64, r.sample_rate(), proc)

    # stream.ReadBeamform, stream/beamform.go:148-171
    def read_beamform(self, readers, angles):
        return Beamform(self, readers, angles)

    # fft.Planner, fft/fft.go:45-48
    def planner(self, iq, frequency, direction):
        return self.ctx.fft_plan(iq, frequency, FFT_FORWARD if direction else FFT_BACKWARD)


class _Wrap(Reader):
    def __init__(self, ctx, r):
        self.ctx, self.r = ctx, r

    def sample_format(self):
        return self.r.sample_format()

    def sample_rate(self):
        return self.r.sample_rate()


class _ShiftReader(_Wrap):
    def __init__(self, ctx, r, shift_hz):
        super().__init__(ctx, r)
        self.shift = float(shift_hz)
        self.fn = ctx.nco(r.sample_rate())  # ShiftBuffer(r.SampleRate())

    def read(self, s):
        if fmt_of(s) != FMT_C64:  # stream/shifter.go:45-50
            raise ErrSampleFormatUnknown("sdr: iq sample format is not understood")
        n = self.r.read(s)
        self.fn(self.shift, s[:n])
        return n


class _GainReader(_Wrap):
    def __init__(self, ctx, r, v):
        super().__init__(ctx, r)
        self.v = float(v)

    def read(self, s):
        n = self.r.read(s)
        if fmt_of(s) != FMT_C64:  # gain.Scale: stream/gain.go:39-48
            raise ErrSampleFormatUnknown("sdr: iq sample format is not understood")
        self.ctx.scale(s[:n], self.v)
        return n


class _MultiplyReader(_Wrap):
    def __init__(self, ctx, r, m):
        super().__init__(ctx, r)
        self.m = np.complex64(m)

    def set_multiplier(self, m):  # stream/multiply.go:34-36
        self.m = np.complex64(m)

    def read(self, s):
        if fmt_of(s) != FMT_C64:  # stream/multiply.go:47-52
            raise ErrSampleFormatMismatch("sdr: iq sample formats do not match")
        n = self.r.read(s)
        if self.m == 1:  # :59-62
            return n
        self.ctx.rotate(s[:n], self.m)
        return n


class _TableMultiplyReader(_Wrap):
    """uint8MultiplyReader / int8MultiplyReader (stream/multiply.go:91-238)."""

    def __init__(self, ctx, r, m):
        super().__init__(ctx, r)
        self.tab = ctx.rotlut(r.sample_format(), m)

    def set_multiplier(self, m):
        self.tab.set_multiplier(m)

    def read(self, s):
        if fmt_of(s) != self.r.sample_format():
            raise ErrSampleFormatMismatch("sdr: iq sample formats do not match")
        n = self.r.read(s)
        self.tab.apply(s[:n])
        return n


class _AddReader(Reader):
    """addReader (stream/add.go:84-185): K temporaries, ReadFull each reader in
    order, zero + ordered sum; errors are sticky."""

    def __init__(self, ctx, readers, fmt, rate):
        self.ctx, self.readers, self.fmt, self.rate, self.err = ctx, list(readers), fmt, rate, None

    def sample_format(self):
        return self.fmt

    def sample_rate(self):
        return self.rate

    def read(self, s):
        if self.err is not None:
            raise self.err
        if fmt_of(s) not in (FMT_C64, FMT_I16, FMT_I8):
            raise ErrSampleFormatUnknown("sdr: iq sample format is not understood")
        bufs = []
        try:
            for r in self.readers:
                b = make_samples(fmt_of(s), length(s))
                read_full(r, b)
                bufs.append(b)
        except (EOF, HzsdrError) as e:
            self.err = e
            raise
        self.ctx.sum(s, bufs)
        return length(s)


class Beamform(Reader):
    """stream.Beamform (stream/beamform.go:36-40, 131-171).  The reference builds
    ConvertReader -> Multiply(.,1) -> Add per channel; the GPU form reads one block
    per channel and makes ONE call that converts, rotates and sums in the
    reference's order."""

    def __init__(self, stream, readers, angles):
        self.stream, self.readers = stream, list(readers)
        self.rate = readers[0].sample_rate()
        self.fmt_in = readers[0].sample_format()
        self.angles = None
        self.err = None
        self.set_phase_angles(angles)

    def set_phase_angles(self, angles):
        if len(angles) != len(self.readers):
            raise HzsdrError("Beamform.SetPhaseAngles: angles must match the reader length")
        self.angles = np.ascontiguousarray(angles, np.complex64)

    def sample_format(self):
        return FMT_C64

    def sample_rate(self):
        return self.rate

    def read(self, s):
        if self.err is not None:
            raise self.err
        if fmt_of(s) != FMT_C64:
            raise ErrSampleFormatMismatch("sdr: iq sample formats do not match")
        chans = []
        try:
            for r in self.readers:
                b = make_samples(self.fmt_in, length(s))
                read_full(r, b)
                chans.append(b)
        except (EOF, HzsdrError) as e:
            self.err = e
            raise
        self.stream.ctx.beamform(s, chans, self.angles)
        return length(s)


__all__ = ["Reader", "BufferReader", "ReadTransformer", "Stream", "Beamform", "read_full",
           "read_at_least", "EOF", "ErrShortBuffer", "ErrUnexpectedEOF", "READER_BLOCK",
           "beamform_angles", "beamform_angles_2d"]

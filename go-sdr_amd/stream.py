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

from . import (FMT_C64, FMT_I8, FMT_I16, FMT_U8, FFT_BACKWARD, FFT_FORWARD, Context, Ring,
               ErrDstTooSmall, ErrSampleFormatMismatch, ErrSampleFormatUnknown, HzsdrError,
               beamform_angles, beamform_angles_2d, fmt_of, format_size, length, make_samples)

READER_BLOCK = 32 * 1024  # stream/convert.go:43-44, decimate.go:41-42, downsample.go:54-55


class ErrShortBuffer(HzsdrError):      # reader.go:28-31
    pass


class ErrUnexpectedEOF(HzsdrError):    # reader.go:33-36
    pass


class ErrNoProgress(HzsdrError):       # io.ErrNoProgress: a Reader that keeps returning (0, nil)
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


class ChainReader(Reader):
    """Nested stream.* Readers of this package as ONE Reader: the constructors below do not wrap a Reader that is
    already one of ours -- they extend its chain (ConvertReader -> ShiftReader -> Gain -> Multiply -> DecimateReader /
    DownsampleReader / ConvolutionReader collapse into one hzsdr_chain: one launch per slot) -- and the Reader reads
    AHEAD: `readahead` Reader blocks of 32 Ki samples per slot of a pinned ring (hzsdr_ring_*: the source reads
    straight into pinned memory, upload, kernel and download of neighbouring slots overlap).  The reference's
    Readers make one call per 32 Ki-sample block (stream/convert.go:43-44, decimate.go:41-42); a GPU call of that size
    is all latency (bench.py `small_buffers`: no faster than one CPU core).

    What the reference's nesting means is kept (go/hip/readers.go has the same rules):
      * the samples are those of the nested Readers bit for bit: the chain's stages are the reference's operations in
        the reference's order, the block-structured stages (ConvertReader, DecimateReader, DownsampleReader: 32 Ki
        blocks; ConvolutionReader: len(filter)) see the same blocks of the same stream, whatever the slot size is;
      * a block-structured stage hands out whole blocks only: a source that ends inside a block loses that partial
        block, as ReadFull's ErrUnexpectedEOF does in read_transformer.go:120-135; pass-through stages (ShiftReader,
        Gain, Multiply over a c64 source) hand out whatever the source delivered;
      * an error of the source is sticky and surfaces when everything read before it has been handed out.
    What differs: the source is read up to `readahead` blocks ahead of the consumer (the reference reads one), and a
    Multiply's SetMultiplier takes effect from the next slot, not the next Read."""

    def __init__(self, stream, src, readahead=32, slots=3, group=1):
        self.stream, self.ctx, self.src = stream, stream.ctx, src
        # group: slots handed to the chain per call (hzsdr_ring_submit_many): the Reader fills up to `group` free slots
        # and submits them together -- ONE launch over all of them where the chain has that form (the FIR-decimate
        # terminal's persistent-pass kernel: fir_decimate_reader), slot by slot otherwise; the samples are the same
        self.group = max(1, min(int(group), 8))
        self.src_fmt, self.rate = src.sample_format(), int(src.sample_rate())
        self.stages = []        # ("shift", hz) | ("gain", v) | ("rotate", m)
        self.terminal = None    # ("decimate", f) | ("downsample", f) | ("convolution", bins, decimate) | ("fir", taps, f)
        self.block = 1          # the stream is consumed in whole multiples of this many samples (1: any)
        self.readahead, self.nslots = int(readahead), int(slots)
        self.chain = self.ring = None
        self.pending, self.off, self.err, self.inflight = None, 0, None, 0
        self.converted = False
        self.queue = []  # outputs drained from a ring that a changed multiplier made us rebuild

    # ---- construction: each returns the Reader to use (self, extended) or None if the stage does not fit ----
    def _open(self):
        return self.chain is None and self.terminal is None

    def extend_convert(self, to):
        # ConvertReader(r, c64): the chain converts on its way in; it makes the stream block-structured
        if not self._open() or to != FMT_C64 or self.stages or self.src_fmt == FMT_C64:
            return None
        self.block = max(self.block, READER_BLOCK)
        self.converted = True
        return self

    def _c64_here(self):
        return self.src_fmt == FMT_C64 or self.converted

    def extend_op(self, op):
        if not self._open() or not self._c64_here():
            return None
        self.stages.append(op)
        return self

    def extend_terminal(self, term, block):
        if not self._open() or not self._c64_here():
            return None
        self.terminal, self.block = term, _lcm(self.block, block)
        if term[0] in ("decimate", "downsample"):
            self.rate //= term[1]
        if term[0] == "fir":
            self.rate //= term[2]
        return self

    def extend_decimate_after_convolution(self, factor):
        if self.chain is not None or self.terminal is None or self.terminal[0] != "convolution" or self.terminal[2] != 1:
            return None
        # (only where the filter's blocks tile the DecimateReader's: the nest hands out floor(n / 32 Ki) blocks then, as
        # the fused chain does; any other length would make the chain consume whole multiples of lcm(len, 32 Ki) and
        # drop more of a stream's tail than the nest -- such a DecimateReader becomes a second chain behind this one)
        if READER_BLOCK % len(self.terminal[1]) != 0:
            return None
        self.terminal = ("convolution", self.terminal[1], factor)
        self.block = _lcm(self.block, READER_BLOCK)
        self.rate //= factor
        return self

    def sample_format(self):
        return FMT_C64 if self._c64_here() else self.src_fmt

    def sample_rate(self):
        return self.rate

    def set_multiplier(self, m):
        """stream/multiply.go:34-36 for the chain's (last) Multiply stage.  What has been read ahead keeps the old
        multiplier; the slots filled from now on take the new one: the chain is rebuilt at the clock it has reached."""
        idx = [i for i, st in enumerate(self.stages) if st[0] == "rotate"]
        if not idx:
            raise HzsdrError("ChainReader.set_multiplier: the chain has no Multiply stage")
        self.stages[idx[-1]] = ("rotate", complex(m))
        if self.chain is None:
            return
        while self.inflight:  # what is in flight was multiplied by the old value: keep it, in order
            self.queue.append(np.array(self.ring.pop(), copy=True))
            self.inflight -= 1
        if self.pending is not None and self.off < len(self.pending):
            self.queue.insert(0, np.array(self.pending[self.off:], copy=True))
        self.pending, self.off = None, 0
        ts = self.chain.time()
        self.ring.close()
        self.chain.close()
        self.chain = self.ring = None
        self._build()
        self.chain.set_time(ts)

    # ---- the data path ----
    def _build(self):
        ch = self.ctx.chain(self.src_fmt, self.src.sample_rate())
        for kind, v in self.stages:
            ch = ch.shift(v) if kind == "shift" else ch.gain(v) if kind == "gain" else ch.rotate(v)
        t = self.terminal
        if t is not None and t[0] == "fir":
            ch = ch.fir_decimate(t[1], t[2])
            ch.pipeline(True)  # (consecutive calls overlap: the ring says what each call's buffers wait for)
        elif t is not None:
            ch = ch.decimate(t[1]) if t[0] == "decimate" else ch.downsample(t[1]) if t[0] == "downsample" else ch.convolution(t[1], decimate=t[2])
        self.chain = ch
        unit = self.block if self.block > 1 else READER_BLOCK
        per = max(1, self.readahead * READER_BLOCK // unit) * unit  # a slot: whole blocks, about `readahead` Reader blocks
        cons, _ = ch.plan(per)
        assert cons == per, (cons, per)
        self.slot_len = per
        self.ring = Ring(ch, per, self.nslots)

    def _fill(self, want):
        """Read the source into up to `want` pinned slots and submit them -- the full ones together, ONE call of the chain
        (hzsdr_ring_submit_many), a short last one (the source ended) by itself.  -> slots submitted (0: nothing more comes).
        A slot that was acquired is submitted or released whatever the source raises (an OSError from a file or
        socket source left the slot acquired and every later acquire failed); a source that keeps returning 0 samples
        without raising ends the stream with ErrNoProgress instead of spinning (io.ErrNoProgress)."""
        if self.err is not None:
            return 0
        first, full, done = None, 0, 0
        while done < want and self.err is None:
            slot, iq = self.ring.acquire()
            n, idle = 0, 0
            try:
                while n < self.slot_len:
                    got = self.src.read(iq[n:])
                    n += got
                    idle = idle + 1 if got == 0 else 0
                    if idle >= 100:
                        raise ErrNoProgress("multiple Read calls return no data or error")
            except (EOF, HzsdrError) as e:
                self.err = e
            except BaseException as e:  # (OSError of a file / socket source, KeyboardInterrupt, ...: sticky, the slot goes back)
                self.err = e if isinstance(e, Exception) else HzsdrError("source interrupted: %r" % (e,))
                self.ring.release(slot)
                if full:
                    self.ring.submit_many(first, full, self.slot_len)
                    self.inflight += full
                if not isinstance(e, Exception):
                    raise
                return done
            n = n // self.block * self.block  # a block-structured stage: whole blocks only
            if n == self.slot_len:
                first = slot if first is None else first
                full += 1
                done += 1
                continue
            # a short slot: everything full in front of it goes first, then it by itself (or back, if it is empty)
            if full:
                self.ring.submit_many(first, full, self.slot_len)
                self.inflight += full
                full = 0
            if n == 0:
                self.ring.release(slot)
            else:
                self.ring.submit(slot, n)
                self.inflight += 1
                done += 1
            return done
        if full:
            self.ring.submit_many(first, full, self.slot_len)
            self.inflight += full
        return done

    def read(self, samples):
        if fmt_of(samples) != self.sample_format():
            raise ErrSampleFormatMismatch("sdr: iq sample formats do not match")
        if self.chain is None:
            self._build()
        if (self.pending is None or self.off >= len(self.pending)) and self.queue:
            self.pending, self.off = self.queue.pop(0), 0
        if self.pending is None or self.off >= len(self.pending):
            # keep the ring busy: everything but the slot being consumed is in flight -- refilled `group` slots at a time
            # (one call of the chain per group), so a refill waits until that many slots are free, or nothing is in flight
            while True:
                free = self.nslots - 1 - self.inflight
                if free < 1 or (free < self.group and self.inflight > 0):
                    break
                if self._fill(min(self.group, free)) == 0:
                    break
            if self.inflight == 0:
                raise self.err if self.err is not None else EOF()
            self.pending, self.off = self.ring.pop(), 0
            self.inflight -= 1
            if len(self.pending) == 0:
                return 0
        n = min(len(self.pending) - self.off, length(samples))
        samples[:n] = self.pending[self.off:self.off + n]
        self.off += n
        return n

    def close(self):
        if self.ring is not None:
            self.ring.close()
        if self.chain is not None:
            self.chain.close()
        self.ring = self.chain = None


def _lcm(a, b):
    from math import gcd
    return a // gcd(a, b) * b


class Stream:
    """The operators, bound to one GPU context (the reference's package-level
    functions take no context; a cgo shim would hold a package-level one)."""

    def __init__(self, ctx: Context, fuse=False, readahead=32):
        # fuse: nested Readers collapse into one ChainReader that reads `readahead` Reader blocks ahead through a pinned
        # ring (what go/hip/readers.go does by default); off, every constructor is the reference's own structure -- a
        # ReadTransformer per block-structured stage, a wrapper per pass-through stage, one call per 32 Ki block
        self.ctx, self.fuse, self.readahead = ctx, bool(fuse), int(readahead)

    def _fused(self, r, how):
        """r as a ChainReader extended by `how(chain_reader)`, or None when the stage cannot join a chain."""
        if not self.fuse:
            return None
        cr = r if isinstance(r, ChainReader) else ChainReader(self, r, self.readahead)
        got = how(cr)
        if got is None and cr is r:  # r's chain is closed (it has its terminal, or has run): a new chain behind it
            got = how(ChainReader(self, r, self.readahead))
        return got

    # stream.ConvertReader, stream/convert.go:37-51
    def convert_reader(self, inp, to):
        f = self._fused(inp, lambda c: c.extend_convert(to))
        if f is not None:
            return f
        return ReadTransformer(inp, READER_BLOCK, READER_BLOCK, to, inp.sample_rate(),
                               lambda i, o: self.ctx.convert(o, i))

    # stream.ConvertWriter, stream/convert.go:53-118
    def convert_writer(self, out, input_format):
        return ConvertWriter(self.ctx, out, input_format)

    # stream.DecimateReader, stream/decimate.go:34-55
    def decimate_reader(self, inp, factor):
        if self.fuse and isinstance(inp, ChainReader):
            f = inp.extend_decimate_after_convolution(factor) or inp.extend_terminal(("decimate", factor), READER_BLOCK)
            if f is not None:
                return f
        elif inp.sample_format() == FMT_C64:
            f = self._fused(inp, lambda c: c.extend_terminal(("decimate", factor), READER_BLOCK))
            if f is not None:
                return f
        state = {"offset": 0}

        def proc(i, o):
            n = self.ctx.decimate(o, i, factor, state["offset"])
            state["offset"] += length(i)
            return n
        return ReadTransformer(inp, READER_BLOCK, READER_BLOCK, inp.sample_format(),
                               inp.sample_rate() // factor, proc)

    def fir_decimate_reader(self, inp, taps, factor, slots=None, group=None):
        """The north-star terminal as a Reader: an N-tap FIR at the input rate whose output is kept every `factor`
        samples (BASELINE.json north_star; the reference has no such Reader -- its Downsample is the boxcar,
        stream/downsample.go:47-64 -- so the name and the signature follow DecimateReader's, stream/decimate.go:34).
        Always a fused Reader: ConvertReader / ShiftReader / Gain / Multiply in front of it join its chain (one kernel
        per slot: for a u8 / i8 source at factor 8 the int8 matrix kernel, csrc/hz_firmm2.h), the slots of its pinned
        ring go to the chain `group` at a time (one launch each time: hzsdr_ring_submit_many)."""
        taps = np.ascontiguousarray(taps, np.complex64)
        cr = inp if isinstance(inp, ChainReader) and inp._open() and inp._c64_here() else None
        if cr is None:
            cr = ChainReader(self, inp, self.readahead)
            if not cr._c64_here():
                cr.converted = True  # (the terminal converts on its way in, as DownsampleReader does)
        if slots is not None:
            cr.nslots = int(slots)
        cr.group = max(1, min(int(group), 8)) if group is not None else max(1, min(4, cr.nslots - 1))
        got = cr.extend_terminal(("fir", taps, int(factor)), int(factor))
        if got is None:
            raise HzsdrError("fir_decimate_reader: the stage does not fit the Reader in front of it")
        return got

    # stream.DownsampleReader, stream/downsample.go:47-64
    def downsample_reader(self, inp, factor):
        # (a raw u8 / i16 source: DownsampleReader converts by itself -- the chain does the same on its way in)
        if self.fuse and (isinstance(inp, ChainReader) or inp.sample_format() in (FMT_C64, FMT_U8, FMT_I16)):
            def how(c):
                if not c._c64_here():
                    c.converted = True  # DownsampleBuffer's own conversion (stream/downsample.go:99-124)
                return c.extend_terminal(("downsample", factor), READER_BLOCK)
            f = self._fused(inp, how)
            if f is not None:
                return f
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
        f = self._fused(r, lambda c: c.extend_op(("shift", float(shift_hz))))
        if f is not None:
            return f
        return _ShiftReader(self.ctx, r, shift_hz)

    # stream.Gain, stream/gain.go:30-57
    def gain(self, r, v):
        if r.sample_format() == FMT_C64:
            f = self._fused(r, lambda c: c.extend_op(("gain", float(v))))
            if f is not None:
                return f
        return _GainReader(self.ctx, r, v)

    # stream.Multiply, stream/multiply.go:74-89
    def multiply(self, r, m):
        f = r.sample_format()
        if f == FMT_C64:
            fr = self._fused(r, lambda c: c.extend_op(("rotate", complex(m)))) if complex(m) != 1 else None
            if fr is not None:
                return fr
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
        f = self._fused(r, lambda c: c.extend_terminal(("convolution", filt, 1), flen))
        if f is not None:
            return f
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


__all__ = ["Reader", "BufferReader", "ReadTransformer", "Stream", "Beamform", "ChainReader", "read_full",
           "read_at_least", "EOF", "ErrShortBuffer", "ErrUnexpectedEOF", "READER_BLOCK",
           "beamform_angles", "beamform_angles_2d"]

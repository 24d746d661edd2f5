"""rtl/kerberos coherent-sync DSP on the GPU (a "next" row):
the step right before Beamform.  Mirrors rtl/kerberos/internal/align.go:

    CrossCorrelater.Correlate   two forward FFTs, bins1 * conj(bins2), backward FFT
    checkAlignment              peak of |corr|^2 -> sample lag per channel
    PhaseOffsets                mean phase of a * conj(b) -> unit rotation per channel

Hardware I/O (the four RTL dongles) is out of scope; buffers come from Readers.
"""
import math

import numpy as np

from . import (FMT_C64, ErrDstTooSmall, ErrSampleFormatMismatch, HzsdrError, fmt_of, length,
               make_samples)
from .stream import EOF, ErrUnexpectedEOF, Reader, read_full

SYNC_LENGTH = 1024 * 64  # align.go:248, :278


class CrossCorrelater:
    """align.go:36-106.  Owns its buffers like the reference (bufIn1/2, bufOut)."""

    def __init__(self, ctx, fft_length):
        self.ctx, self.n = ctx, fft_length
        self.in1 = make_samples(FMT_C64, fft_length)
        self.in2 = make_samples(FMT_C64, fft_length)
        self.out = make_samples(FMT_C64, fft_length)
        self._cc = ctx.cross_correlate(self.out, self.in1, self.in2)

    def correlate(self, buf1, buf2):
        if length(buf1) != self.n or length(buf2) != self.n:  # align.go:63-65
            raise ErrDstTooSmall("sdr: destination sample buffer is too small")
        self.in1[:] = buf1
        self.in2[:] = buf2
        self._cc()
        return self.out.copy()

    def close(self):
        self._cc.close()


def read_buffers(readers, bufs):
    for r, b in zip(readers, bufs):
        read_full(r, b)


GRAFT_LENGTH = 1024 * 64  # graft.go:128


class GraftReader(Reader):
    """GraftReaders (rtl/kerberos/internal/graft.go:124-159): adjacent bands from
    len(readers) tuners stitched in frequency space into ONE reader at len(readers)
    times the rate.  Each trip ReadFulls `fft_size` samples per band and hands out
    len(readers) * fft_size output samples (graft.go:85-120); a read error on any
    band ends the stream."""

    def __init__(self, ctx, readers, fft_size=GRAFT_LENGTH):
        self.ctx, self.readers, self.n = ctx, list(readers), fft_size
        self.bufs = [make_samples(FMT_C64, fft_size) for _ in self.readers]
        self.out = make_samples(FMT_C64, fft_size * len(self.readers))
        self.rate = len(self.readers) * self.readers[0].sample_rate()  # graft.go:132
        self.avail, self.off, self.err = 0, 0, None

    def sample_format(self):
        return FMT_C64

    def sample_rate(self):
        return self.rate

    def read(self, samples):
        if fmt_of(samples) != FMT_C64:
            raise ErrSampleFormatMismatch("sdr: iq sample formats do not match")
        if self.avail == 0:
            if self.err is not None:
                raise self.err
            try:
                read_buffers(self.readers, self.bufs)
            except (EOF, HzsdrError) as e:
                self.err = EOF() if isinstance(e, ErrUnexpectedEOF) else e
                raise self.err
            self.ctx.graft(self.out, self.bufs)
            self.avail, self.off = length(self.out), 0
        n = min(self.avail, length(samples))
        samples[:n] = self.out[self.off:self.off + n]
        self.off += n
        self.avail -= n
        return n


def graft_readers(ctx, readers, fft_size=GRAFT_LENGTH):
    return GraftReader(ctx, readers, fft_size)


def check_alignment(ctx, readers, bufs, reference_quirk=False):
    """align.go:112-153.  Lag of each channel against channel 0.  The reference
    correlates bufs[0] with bufs[1] for EVERY i (align.go:125 passes bufs[1], not
    bufs[i]); reference_quirk=True reproduces that, the default correlates with
    bufs[i] as the surrounding code intends."""
    ccr = CrossCorrelater(ctx, length(bufs[0]))
    try:
        read_buffers(readers, bufs)
        ret = [0] * len(readers)
        for i in range(1, len(bufs)):
            cc = ccr.correlate(bufs[0], bufs[1] if reference_quirk else bufs[i])
            ret[i] = ctx.peak_lag(cc)
        return ret
    finally:
        ccr.close()


def phase_offsets(ctx, readers, n=SYNC_LENGTH):
    """align.go:244-271: per-channel rotation that cancels its mean phase against
    channel 0.  phases[0] = 1 before the division is the reference's own quirk."""
    bufs = [make_samples(FMT_C64, n) for _ in readers]
    read_buffers(readers, bufs)
    phases = [0.0] * len(readers)
    for j in range(1, len(bufs)):
        phases[j] = ctx.mean_phase(bufs[0], bufs[j]) * n  # the reference divides below
    phases[0] = 1.0
    ret = np.zeros(len(readers), np.complex64)
    for i, p in enumerate(phases):
        p /= float(n)
        ret[i] = np.complex64(complex(math.cos(p), math.sin(p)))  # cmplx.Rect(1, p)
    return ret


def guess_alignment(readings):
    """align.go:155-165: the lags are trusted only if every measurement of the batch agrees."""
    ret = list(readings[0])
    for r in readings:
        if list(r) != ret:
            return None, False
    return ret, True


def align_step(alignments, readers):
    """align.go:167-238 (alignReaders).  A positive lag: reader 0 is that many samples behind reader n;
    consume the largest one from reader 0, slide the others, then consume each remaining (negative)
    lag from its own reader.  Returns True when every lag is 0 (sample lock)."""
    alignments = list(alignments)
    lo, hi = min([0] + alignments), max([0] + alignments)
    if lo == 0 and hi == 0:
        return True
    if hi > 0:
        read_full(readers[0], make_samples(FMT_C64, hi))
        for i in range(1, len(alignments)):
            alignments[i] -= hi
    for i, a in enumerate(alignments):
        if a != 0:
            read_full(readers[i], make_samples(FMT_C64, -a))
    return False


def align_readers(ctx, readers, n=SYNC_LENGTH, measurements=10, max_rounds=64):
    """align.go:273-305 (AlignReaders): measure the lags `measurements` times on fresh buffers
    (cross-correlation on the GPU, hzsdr_peak_lag), act only on a unanimous batch, repeat until the
    readers are in sample lock.  The reference loops forever on a stream that never agrees;
    `max_rounds` bounds that here (HzsdrError)."""
    bufs = [make_samples(FMT_C64, n) for _ in readers]
    for _ in range(max_rounds):
        batch = [check_alignment(ctx, readers, bufs) for _ in range(measurements)]
        alignment, ok = guess_alignment(batch)
        if not ok:
            continue
        if align_step(alignment, readers):
            return
    raise HzsdrError("kerberos: readers did not reach sample lock")

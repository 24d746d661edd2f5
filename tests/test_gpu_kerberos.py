"""rtl/kerberos coherent sync on the GPU ("next" row): cross-correlation lag and
mean phase against the oracle's restatement of rtl/kerberos/internal/align.go
(the reference has no tests for this package)."""
import importlib
import math

import numpy as np
import pytest

from util import rand_c64, zeros

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hz():
    return importlib.import_module("go-sdr_amd")


@pytest.fixture(scope="module")
def ctx(hz):
    c = hz.Context(0, hz.MEM_HOST)
    yield c
    c.close()


def test_peak_lag_matches_sequential_scan(ctx, orc):
    n = 65536
    x = rand_c64(1, n)
    x[100] = 0  # zeros are skipped
    assert ctx.peak_lag(x) == orc.peak_lag(x)
    y = x.copy()
    y[40000] = 50 + 0j
    y[123] = 50 + 0j  # equal maxima: the first one wins
    assert ctx.peak_lag(y) == orc.peak_lag(y) == 123
    y[123] = 0
    assert ctx.peak_lag(y) == orc.peak_lag(y) == 40000 - n  # above n/2 folds negative
    assert ctx.peak_lag(zeros("c64", 1000)) == orc.peak_lag(zeros("c64", 1000)) == -1
    assert ctx.peak_lag(zeros("c64", 0)) == -1


@pytest.mark.parametrize("delay", [0, 7, -19, 300])
def test_cross_correlation_finds_the_delay(hz, ctx, orc, delay):
    """Two copies of one noise burst, one delayed: the 64 Ki-point correlator
    (two-step FFT path) must peak at the delay, like the oracle's float64 one."""
    K = importlib.import_module("go-sdr_amd.kerberos")
    S = importlib.import_module("go-sdr_amd.stream")
    n = K.SYNC_LENGTH
    base = rand_c64(5, n + 1000)
    a = base[500:500 + n].copy()
    b = base[500 - delay:500 - delay + n].copy()
    b += (0.05 * rand_c64(6, n)).astype(np.complex64)
    readers = [S.BufferReader(a, 2_400_000), S.BufferReader(b, 2_400_000)]
    bufs = [zeros("c64", n), zeros("c64", n)]
    lags = K.check_alignment(ctx, readers, bufs)
    want = zeros("c64", n)
    orc.convolve(want, a, b, conj=True)
    assert lags == [0, orc.peak_lag(want)]
    assert lags[1] == -delay
    with pytest.raises(hz.ErrDstTooSmall):  # align.go:63-65
        K.CrossCorrelater(ctx, 1024).correlate(a[:512], b[:1024])


def test_mean_phase_and_phase_offsets(hz, ctx, orc):
    K = importlib.import_module("go-sdr_amd.kerberos")
    S = importlib.import_module("go-sdr_amd.stream")
    n = K.SYNC_LENGTH
    a = rand_c64(9, n)
    rot = [np.complex64(1), np.complex64(np.exp(0.7j)), np.complex64(np.exp(-2.1j))]
    chans = []
    for k, r in enumerate(rot):
        y = a.copy()
        orc.rotate(y, r)
        y += (0.01 * rand_c64(20 + k, n)).astype(np.complex64)
        chans.append(y)
    for j in (1, 2):
        got, want = ctx.mean_phase(chans[0], chans[j]), orc.mean_phase(chans[0], chans[j])
        # float64 sum in tree order vs sequential, OCML atan2 vs libm: 1e-10 rad
        assert abs(got - want) < 1e-10
        assert abs(got - (-np.angle(rot[j]))) < 1e-2
    offs = K.phase_offsets(ctx, [S.BufferReader(c, 2_400_000) for c in chans])
    for j in (1, 2):
        want = orc.mean_phase(chans[0], chans[j])
        assert abs(offs[j] - np.complex64(complex(math.cos(want), math.sin(want)))) < 1e-6
    assert abs(offs[0] - np.complex64(complex(math.cos(1 / n), math.sin(1 / n)))) < 1e-7  # the reference's phases[0] = 1


def test_align_readers_reaches_sample_lock(hz, ctx):
    """AlignReaders (align.go:273-305): three receivers that started 0, 37 and 211 samples apart;
    batches of unanimous lag measurements, the consume steps of alignReaders, until every lag is 0.
    Afterwards the three readers hand out the same samples."""
    K = importlib.import_module("go-sdr_amd.kerberos")
    S = importlib.import_module("go-sdr_amd.stream")
    n = 1 << 14
    base = rand_c64(21, 40 * n)
    starts = [211, 174, 0]  # reader 0 is 37 behind reader 1 and 211 ahead of reader 2
    readers = [S.BufferReader(base[s:].copy(), 2_400_000) for s in starts]
    # the pieces: unanimity, and one consume step
    assert K.guess_alignment([[0, 3, -2], [0, 3, -2]]) == ([0, 3, -2], True)
    assert K.guess_alignment([[0, 3, -2], [0, 3, -1]]) == (None, False)
    assert K.align_step([0, 0, 0], readers) is True
    K.align_readers(ctx, readers, n=n, measurements=3)
    a, b, c = (zeros("c64", 1000) for _ in range(3))
    for r, buf in zip(readers, (a, b, c)):
        S.read_full(r, buf)
    assert a.tobytes() == b.tobytes() == c.tobytes()
    # a stream that never agrees ends in an error here (the reference retries forever)
    noise = [S.BufferReader(rand_c64(30 + i, 8 * n), 2_400_000) for i in range(2)]
    with pytest.raises(hz.HzsdrError):
        K.align_readers(ctx, noise, n=n, measurements=2, max_rounds=1)

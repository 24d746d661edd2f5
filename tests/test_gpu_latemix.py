"""The FIR-decimate terminal's late mixer (include/hzsdr.h, hzsdr_chain_mix_in_order):
inside one exactly-linear run of the NCO clock the Shift / Gain / Multiply stages
commute with the filter, so the default path filters converted samples with modulated
taps and mixes at the decimated rate.  Both orders are held to the SAME bound against
the oracle (reference-order Shift, float64 direct-form FIR):
|err| <= 6e-7 * sum|h| * max|x| per output and relative L2 <= 3e-7 (tests/util.py,
assert_fir_close), and to 3e-7 relative L2 against each other (tests/util.py, CROSS_REL_L2)."""
import importlib

import numpy as np
import pytest

from util import CROSS_REL_L2, assert_fir_close, bits_equal, rand_c64, rand_u8, zeros

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hz():
    return importlib.import_module("go-sdr_amd")


@pytest.fixture(scope="module")
def ctx(hz):
    c = hz.Context(0, hz.MEM_HOST)
    yield c
    c.close()


def taps_for(ntaps):
    k = np.arange(ntaps) - (ntaps - 1) / 2
    return (np.sinc(k / 16) / 16 * np.hamming(ntaps) * np.exp(0.3j * k)).astype(np.complex64)


def oracle_chain(orc, x, rate, ops, taps, D, ts0=0.0):
    """Reference order: convert, then every elementwise stage per sample, then the FIR."""
    xc = zeros("c64", len(x))
    if x.dtype == np.complex64:
        xc[:] = x
    else:
        orc.convert(xc, x)
    shifters = []
    for kind, arg in ops:
        if kind == "shift":
            sh = orc.Shifter(rate)
            sh.ts.value = ts0
            sh(arg, xc)
            shifters.append(sh)
        elif kind == "gain":
            orc.scale(xc, arg)
        else:
            orc.rotate(xc, arg)
    want = zeros("c64", len(x) // D)
    orc.fir_decimate_f64(want, xc, taps, D)
    return want, xc


def build(hz, ctx, fmt, rate, ops, taps, D, in_order, fir_opts=None):
    ch = ctx.chain(fmt, rate)
    for kind, arg in ops:
        ch = ch.shift(arg) if kind == "shift" else ch.gain(arg) if kind == "gain" else ch.rotate(arg)
    if fir_opts:
        ch.fir_options(**fir_opts)
    ch.fir_decimate(taps, D)
    ch.mix_in_order(in_order)
    return ch


CASES = {
    # the BASELINE chain shape; 2^21 samples at 20 Msps stay inside a few long clock runs
    "north_star": dict(fmt="u8", rate=20_000_000, ops=[("shift", -2.5e6)], ntaps=1024, D=8, n=1 << 21),
    # 250 ksps: the clock wraps at 2*pi s = 1 570 796 samples, inside the second call
    # (one call from ts = 0 across the wrap needs > 32 clock runs, the device-table form,
    # which keeps reference order throughout)
    "clock_wrap": dict(fmt="u8", rate=250_000, ops=[("shift", 31_250.0)], ntaps=1024, D=8, n=1 << 21,
                       cuts=[0, 1_200_000, 1 << 21]),
    # several stages in front of the filter, two of them Shifts on the shared clock
    "program": dict(fmt="c64", rate=2_400_000, ops=[("gain", 0.5), ("shift", 100e3), ("rotate", 0.6 - 0.8j),
                                                     ("shift", -350e3)], ntaps=257, D=4, n=1 << 20),
    # D = 2: the folded inverse is 2048 points on 128 lanes (two waves)
    "d2": dict(fmt="u8", rate=20_000_000, ops=[("shift", -2.5e6)], ntaps=1024, D=2, n=1 << 20),
    # shorter filters: N_fft 2048 / 1024, polyphase with a radix-8 / radix-4 second pass
    "n2048_d8": dict(fmt="u8", rate=20_000_000, ops=[("shift", 3e6)], ntaps=300, D=8, n=1 << 20),
    "n2048_d2": dict(fmt="i16", rate=2_400_000, ops=[("shift", -5e5), ("gain", 0.7)], ntaps=400, D=2, n=1 << 19),
    "n1024_d4": dict(fmt="c64", rate=20_000_000, ops=[("shift", 1.25e6)], ntaps=200, D=4, n=1 << 19),
    "n1024_d2": dict(fmt="u8", rate=20_000_000, ops=[("shift", -4e6)], ntaps=150, D=2, n=1 << 19),
    "d16": dict(fmt="u8", rate=20_000_000, ops=[("shift", 1e6)], ntaps=600, D=16, n=1 << 20),
}


@pytest.mark.parametrize("name", list(CASES))
def test_late_and_in_order_mixers_meet_the_same_bound(hz, ctx, orc, name):
    c = CASES[name]
    n, D, taps = c["n"], c["D"], taps_for(c["ntaps"])
    from util import rand_i16
    x = {"u8": rand_u8, "i16": rand_i16, "c64": rand_c64}[c["fmt"]](77, n)
    fmt = {"u8": hz.FMT_U8, "i16": hz.FMT_I16, "c64": hz.FMT_C64}[c["fmt"]]
    want, xc = oracle_chain(orc, x, c["rate"], c["ops"], taps, D)
    xmax = float(np.abs(xc).max())
    outs = {}
    for in_order in (False, True):
        ch = build(hz, ctx, fmt, c["rate"], c["ops"], taps, D, in_order)
        out = zeros("c64", n // D)
        cuts = c.get("cuts", [0, n])
        for a, b in zip(cuts[:-1], cuts[1:]):
            assert ch.run(x[a:b], out[a // D:b // D]) == (b - a, (b - a) // D)
        assert_fir_close(out, want, taps, xmax, (name, in_order))
        outs[in_order] = out
        ch.close()
    # the default really is a different computation, not the same kernel twice
    assert not bits_equal(outs[False], outs[True])
    # ... and the two agree with each other to float32 rounding
    d = outs[False].astype(np.complex128) - outs[True]
    assert np.linalg.norm(d) <= CROSS_REL_L2 * np.linalg.norm(want.astype(np.complex128)), name


def test_late_mixer_stream_continuity(hz, ctx, orc):
    """History and NCO time carried across runs: three ragged calls against one call
    and against the oracle over the concatenated stream."""
    rate, D, taps = 20_000_000, 8, taps_for(1024)
    n = 3 * (1 << 19)
    cuts = [0, (1 << 19) + 8 * 1234, (1 << 20) - 8 * 77, n]
    x = rand_u8(5, n)
    want, xc = oracle_chain(orc, x, rate, [("shift", -2.5e6)], taps, D)
    ch = build(hz, ctx, hz.FMT_U8, rate, [("shift", -2.5e6)], taps, D, False)
    out = zeros("c64", n // D)
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert ch.run(x[a:b], out[a // D:b // D]) == (b - a, (b - a) // D)
    assert_fir_close(out, want, taps, float(np.abs(xc).max()), "continuity")
    ch.close()


def test_chains_without_a_shift(hz, ctx, orc):
    """No clock involved: Gain / Multiply commute with the filter in every block; a bare
    filter takes the same route (samples straight into the first pass's registers)."""
    taps, n = taps_for(1024), 1 << 18
    x = rand_u8(3, n)
    for ops in ([("gain", 0.25), ("rotate", 0.6 + 0.8j)], []):
        want, xc = oracle_chain(orc, x, 20_000_000, ops, taps, 8)
        outs = []
        for in_order in (False, True):
            ch = build(hz, ctx, hz.FMT_U8, 20_000_000, ops, taps, 8, in_order)
            out = zeros("c64", n // 8)
            ch.run(x, out)
            assert_fir_close(out, want, taps, float(np.abs(xc).max()), (ops, in_order))
            outs.append(out)
            ch.close()
        # two instantiations of the same transform (the compiler contracts their butterflies
        # differently): not the same bits, but equal to float32 rounding
        d = outs[0].astype(np.complex128) - outs[1]
        assert np.linalg.norm(d) <= CROSS_REL_L2 * np.linalg.norm(want.astype(np.complex128))


@pytest.mark.parametrize("nmin,ntaps,D,fmt", [(256, 50, 3, "c64"), (256, 64, 8, "u8"), (512, 100, 2, "i16"), (512, 120, 5, "c64")])
def test_small_overlap_save_blocks(hz, ctx, orc, nmin, ntaps, D, fmt):
    """The library picks N_fft >= 1024 (>= 256 D for the polyphase factors) because the fast forms of
    the kernels need it; the 256- and 512-point instantiations stay correct (hzsdr_chain_fir_options:
    the smallest block, and the transform kernels for the byte source that would otherwise take the
    matrix form)."""
    from util import rand_i16
    fir_opts = dict(impl=hz.FIR_IMPL_TRANSFORMS, nfft_min=nmin)
    n, rate, taps = 1 << 18, 2_400_000, taps_for(ntaps)
    ops = [("shift", 2.5e5), ("gain", 0.5)]
    x = {"u8": rand_u8, "i16": rand_i16, "c64": rand_c64}[fmt](3, n)
    f = {"u8": hz.FMT_U8, "i16": hz.FMT_I16, "c64": hz.FMT_C64}[fmt]
    want, xc = oracle_chain(orc, x, rate, ops, taps, D)
    for in_order in (False, True):
        ch = build(hz, ctx, f, rate, ops, taps, D, in_order, fir_opts)
        out = zeros("c64", n // D)
        cons, outn = ch.run(x, out)
        assert (cons, outn) == (n // D * D, n // D)
        assert ch.last_fir_path() == hz.FIR_PATH_TRANSFORM
        assert_fir_close(out, want[:outn], taps, float(np.abs(xc).max()), (nmin, ntaps, D, fmt, in_order))
        ch.close()

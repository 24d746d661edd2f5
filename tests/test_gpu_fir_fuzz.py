"""Seeded random FIR-decimate chains against the oracle: tap counts on both sides of the
FFT-size thresholds, every decimation class (folded inverse for 2 / 4 / 8 / 16, full
inverse for 1 / 3 / 5 / 10), all source formats, one to three elementwise stages, ragged
multi-call streams, clock starting points near a binade edge and near the 2*pi wrap --
both mixer orders, one bound (tests/util.py assert_fir_close): |err| <= 6e-7 * sum|h| * max|x|
per output and relative L2 <= 3e-7."""
import importlib

import numpy as np
import pytest

from util import assert_fir_close, rand_c64, rand_i16, rand_i8, rand_u8, zeros

pytestmark = pytest.mark.gpu

GEN = {"u8": rand_u8, "i8": rand_i8, "i16": rand_i16, "c64": rand_c64}


@pytest.fixture(scope="module")
def hz():
    return importlib.import_module("go-sdr_amd")


@pytest.fixture(scope="module", params=["host", "device"])
def space(request, hz):
    import torch
    if request.param == "host":
        ctx = hz.Context(0, hz.MEM_HOST)
        put, get = (lambda a: a), (lambda a: a)
    else:
        ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
        put = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731

        def get(t):
            ctx.synchronize()
            return t.cpu().numpy()
    yield ctx, put, get
    ctx.close()


def make_case(seed):
    r = np.random.default_rng(seed)
    fmt = ["u8", "i8", "i16", "c64"][seed % 4]
    D = [8, 1, 4, 3, 16, 2, 10, 5][seed % 8]
    ntaps = int(r.choice([1, 7, 33, 64, 65, 129, 256, 257, 600, 1024, 1500, 2047]))
    rate = int(r.choice([250_000, 2_400_000, 20_000_000]))
    ops = []
    for _ in range(int(r.integers(1, 4))):
        kind = ["shift", "gain", "rotate"][int(r.integers(0, 3))]
        if kind == "shift":
            ops.append(("shift", float(r.uniform(-0.4, 0.4)) * rate))
        elif kind == "gain":
            ops.append(("gain", float(np.float32(r.uniform(0.1, 1.5)))))
        else:
            a = r.uniform(0, 2 * np.pi)
            ops.append(("rotate", complex(np.complex64(np.exp(1j * a)))))
    n_calls = int(r.integers(1, 4))
    lens = [int(r.integers(20, 200)) * 1000 * D // D * D for _ in range(n_calls)]
    lens = [max(D, x // D * D) for x in lens]
    k = np.arange(ntaps) - (ntaps - 1) / 2
    taps = (np.sinc(k / (2.5 * max(D, 2))) / (2.5 * max(D, 2)) * np.hamming(ntaps)
            * np.exp(1j * float(r.uniform(-0.5, 0.5)) * k)).astype(np.complex64)
    # start the clock somewhere interesting: 0, just under a binade edge, just under 2*pi
    # (reached by running that many samples through first, so only at the lower rates)
    ts0 = [0.0, 4.0 - 3e-3, 2 * np.pi - 2e-3, 1.0 - 1e-4][seed % 4]
    if ts0 and rate > 2_400_000:
        rate = 250_000 if seed % 8 < 4 else 2_400_000
    return dict(fmt=fmt, D=D, taps=taps, rate=rate, ops=ops, lens=lens, ts0=ts0)


@pytest.mark.parametrize("seed", range(16))
def test_random_fir_chain(hz, space, orc, seed):
    ctx, put, get = space
    c = make_case(seed)
    fmt, D, taps, rate = c["fmt"], c["D"], c["taps"], c["rate"]
    n = sum(c["lens"])
    x = GEN[fmt](1000 + seed, n)
    # the clock start is reached the way a stream reaches it: by running samples through
    warm = int(round(c["ts0"] * rate)) // D * D
    xc = zeros("c64", warm + n)
    xw = GEN[fmt](2000 + seed, warm) if warm else None
    if warm:
        tmp = zeros("c64", warm)
        orc.convert(tmp, xw) if fmt != "c64" else tmp.__setitem__(slice(None), xw)
        xc[:warm] = tmp
    tmp = zeros("c64", n)
    orc.convert(tmp, x) if fmt != "c64" else tmp.__setitem__(slice(None), x)
    xc[warm:] = tmp
    for kind, arg in c["ops"]:
        if kind == "shift":
            orc.Shifter(rate)(arg, xc)
        elif kind == "gain":
            orc.scale(xc, arg)
        else:
            orc.rotate(xc, arg)
    want = zeros("c64", n // D)
    t = len(taps)
    hist = None
    if t > 1:  # the t-1 samples in front of the measured stretch, zero-padded at the stream start
        hist = zeros("c64", t - 1)
        tail = xc[max(0, warm - (t - 1)):warm]
        if len(tail):
            hist[-len(tail):] = tail
    orc.fir_decimate_f64(want, xc[warm:].copy(), taps, D, hist)
    F = {"u8": hz.FMT_U8, "i8": hz.FMT_I8, "i16": hz.FMT_I16, "c64": hz.FMT_C64}[fmt]
    for in_order in (False, True):
        ch = ctx.chain(F, rate)
        for kind, arg in c["ops"]:
            ch = ch.shift(arg) if kind == "shift" else ch.gain(arg) if kind == "gain" else ch.rotate(arg)
        ch.fir_decimate(taps, D).mix_in_order(in_order)
        if warm:
            scratch = put(zeros("c64", warm // D))
            assert ch.run(put(xw), scratch) == (warm, warm // D)
        out = put(zeros("c64", n // D))
        dx = put(x)
        pos = 0
        for ln in c["lens"]:
            assert ch.run(dx[pos:pos + ln], out[pos // D:(pos + ln) // D]) == (ln, ln // D)
            pos += ln
        got = get(out)
        assert_fir_close(got, want, taps, float(np.abs(xc).max()),
                         (seed, c["fmt"], D, len(taps), rate, c["ops"], c["lens"], in_order))
        ch.close()

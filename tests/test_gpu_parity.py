"""GPU parity: every entry point of the C ABI against the CPU oracle on the same
seeded inputs, in both memory spaces (HOST = numpy through staging, DEVICE =
torch CUDA tensors on torch's current stream).

Bars (BASELINE.json north_star): bit-exact for integer / LUT / single-rounding
float paths; <= 1 ULP for complex64 ops whose only freedom is float64 sincos
(here the GPU runs the same restated math.Sincos as the oracle, so those are
checked bit-exact too); FFT-based ops against an error bound, written in each
test, because the reference has no FFT of its own to be identical to."""
import importlib
import math

import numpy as np
import pytest

from util import (assert_fir_close, bits_equal, filled, in_epsilon, rand_c64, rand_i16, rand_i8, rand_u8, samples,
                  ulp_diff, zeros)

pytestmark = pytest.mark.gpu

FMTS = {"c64": 1, "u8": 2, "i16": 3, "i8": 4}


@pytest.fixture(scope="module")
def hz():
    return importlib.import_module("go-sdr_amd")


class Env:
    """One memory space: moves numpy buffers in and out of it."""

    def __init__(self, hz, kind):
        import torch
        self.hz, self.kind, self.torch = hz, kind, torch
        if kind == "host":
            self.ctx = hz.Context(0, hz.MEM_HOST)
        else:
            self.ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)

    def put(self, a):
        if self.kind == "host":
            return np.ascontiguousarray(a).copy()
        return self.torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def get(self, x):
        if self.kind == "host":
            return x
        self.ctx.synchronize()
        return x.cpu().numpy()

    def zeros(self, fmt, n):
        return self.put(zeros(fmt, n))


@pytest.fixture(scope="module", params=["host", "device"])
def env(request, hz):
    e = Env(hz, request.param)
    yield e
    e.ctx.close()


def c(pair):
    return np.complex64(complex(pair[0], pair[1]))


# ---- converters ----------------------------------------------------------------------

ALL_PAIRS = [(s, d) for s in FMTS for d in FMTS if s != d]


def _exhaustive(fmt):
    allb = np.arange(65536, dtype=np.uint32)
    if fmt == "u8":
        return np.stack([(allb & 255), (allb >> 8)], 1).astype(np.uint8)
    if fmt == "i8":
        return np.stack([(allb & 255), (allb >> 8)], 1).astype(np.uint8).view(np.int8)
    if fmt == "i16":
        v = allb.astype(np.uint16).view(np.int16)
        return np.stack([v, v[::-1]], 1).copy()
    x = rand_c64(11, 65536)
    x[:8] = samples("c64", [[1, -1], [0, 0], [-0.0, 0.5], [1e12, -1e12], [float("nan"), float("inf")],
                             [-3, 3], [0.999999, -0.999999], [2.0, -2.0]])
    return x


@pytest.mark.parametrize("src_fmt,dst_fmt", ALL_PAIRS)
def test_convert_exhaustive_bit_exact(env, orc, src_fmt, dst_fmt):
    """All 65 536 (I,Q) byte pairs / all int16 values / float edge cases."""
    src = _exhaustive(src_fmt)
    want = zeros(dst_fmt, len(src))
    assert orc.convert(want, src) == len(src)
    dst = env.zeros(dst_fmt, len(src))
    assert env.ctx.convert(dst, env.put(src)) == len(src)
    assert bits_equal(env.get(dst), want)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 8, 9, 31, 255, 1000, 4097, 70001])
def test_convert_ragged_lengths(env, orc, n):
    src = rand_u8(n + 1, n)
    want, dst = zeros("c64", n), env.zeros("c64", n)
    orc.convert(want, src)
    assert env.ctx.convert(dst, env.put(src)) == n
    assert bits_equal(env.get(dst), want)
    src = rand_c64(n + 2, n)
    want, dst = zeros("i16", n), env.zeros("i16", n)
    orc.convert(want, src)
    assert env.ctx.convert(dst, env.put(src)) == n
    assert bits_equal(env.get(dst), want)


def test_convert_kats(env, kats):
    for k in kats["convert"]:
        src = samples(k["src_fmt"], k["src"])
        dst = env.zeros(k["dst_fmt"], len(src))
        assert env.ctx.convert(dst, env.put(src)) == len(src)
        got, exp = env.get(dst), samples(k["dst_fmt"], k["dst"])
        if "eps" in k:
            g, w = got.view(np.float32).astype(np.float64), exp.view(np.float32).astype(np.float64)
            if k.get("plus_one"):
                g, w = g + 1, w + 1
            assert in_epsilon(w, g, k["eps"]), k["cite"]
        else:
            assert bits_equal(got, exp), k["cite"]


def test_convert_subslice_guard(env, kats):
    """iq_u8_test.go:65-85: converting [32:69] must leave the guards untouched."""
    k = kats["convert_subslice_guard"]
    lo, hi = k["lo"], k["hi"]
    src, dst = env.put(filled("u8", k["n"], k["fill"])), env.zeros("c64", k["n"])
    assert env.ctx.convert(dst[lo:hi], src[lo:hi]) == hi - lo
    got = env.get(dst)
    assert np.all(got[:lo] == 0) and np.all(got[hi:] == 0)
    assert in_epsilon(1.0, got[lo:hi].real, k["eps"]) and in_epsilon(1.0, got[lo:hi].imag, k["eps"])
    # odd offsets: only sample-aligned pointers
    for lo, hi in ((1, 1000), (3, 4), (5, 5), (7, 1024)):
        dst = env.zeros("c64", k["n"])
        env.ctx.convert(dst[lo:hi], src[lo:hi])
        got = env.get(dst)
        assert np.all(got[:lo] == 0) and np.all(got[hi:] == 0) and np.all(got[lo:hi] == got[lo:lo + 1])


def test_convert_errors(env):
    hz = env.hz
    with pytest.raises(hz.ErrDstTooSmall):  # iq_u8.go:104-106
        env.ctx.convert(env.zeros("c64", 4), env.zeros("u8", 8))
    with pytest.raises(hz.ErrSampleFormatUnknown):
        env.ctx.convert_raw(9, env.zeros("c64", 8), 8, hz.FMT_U8, env.zeros("u8", 8), 8)
    # same format: CopySamples copies min(len) (copy.go:31-52)
    src = env.put(rand_u8(1, 8))
    d = env.zeros("u8", 4)
    assert env.ctx.convert(d, src) == 4 and bits_equal(env.get(d), rand_u8(1, 8)[:4])
    d = env.zeros("u8", 16)
    assert env.ctx.convert(d, src) == 8 and bits_equal(env.get(d)[:8], rand_u8(1, 8))


def test_i16_shift_lsb_to_msb(env, orc):
    a = rand_i16(5, 1001)
    want = a.copy()
    orc.i16_shift_lsb_to_msb(want, 12)
    d = env.put(a)
    env.ctx.i16_shift_lsb_to_msb(d, 12)
    assert bits_equal(env.get(d), want)


# ---- c64 vector ops --------------------------------------------------------------------

def test_scale_rotate_add_kats(env, kats):
    k = kats["scale"]
    b = env.put(filled("c64", k["n"], k["fill"]))
    env.ctx.scale(b, k["r"])
    assert np.all(env.get(b) == c(k["value"]))
    k = kats["multiply"]
    b = env.put(filled("c64", k["n"], k["fill"]))
    env.ctx.rotate(b, c(k["m"]))
    assert np.all(env.get(b) == c(k["value"]))
    k = kats["add"]
    a, b = env.put(filled("c64", k["n"], k["a"])), env.put(filled("c64", k["n"], k["b"]))
    env.ctx.add(a, b, a)
    assert np.all(env.get(a) == c(k["value"])) and np.all(env.get(b) == c(k["b"]))
    for k in kats["simd_add"]:
        a, b = env.put(filled("c64", k["n"], k["a"])), env.put(filled("c64", k["n"], k["b"]))
        out = a if k.get("in_place") else env.zeros("c64", k["n"])
        env.ctx.add(a, b, out)
        assert np.all(env.get(out) == c(k["value"]))
    k = kats["simd_add_subslice_guard"]
    a, b, o = (env.put(filled("c64", k["n"], k["a"])), env.put(filled("c64", k["n"], k["b"])),
               env.zeros("c64", k["n"]))
    env.ctx.add(a[k["lo"]:k["hi"]], b[k["lo"]:k["hi"]], o[k["lo"]:k["hi"]])
    got = env.get(o)
    assert np.all(got[:k["lo"]] == 0) and np.all(got[k["hi"]:] == 0)
    assert np.all(got[k["lo"]:k["hi"]] == c(k["value"]))
    k = kats["simd_scale_subslice_guard"]
    b = env.put(filled("c64", k["n"], k["fill"]))
    env.ctx.scale(b[:k["hi"]], k["r"])
    got = env.get(b)
    assert np.all(got[:k["hi"]] == c(k["value"])) and np.all(got[k["hi"]:] == c(k["fill"]))
    k = kats["simd_rotate"]
    b = env.put(filled("c64", k["n"], k["fill"]))
    env.ctx.rotate(b, c(k["m"]))
    assert np.all(env.get(b) == c(k["value"]))
    with pytest.raises(env.hz.ErrLengthMismatch):
        env.ctx.add(env.zeros("c64", 3), env.zeros("c64", 4), env.zeros("c64", 3))


@pytest.mark.parametrize("n", [1, 2, 5, 1023, 100003])
def test_scale_rotate_add_random_bit_exact(env, orc, n):
    x, y = rand_c64(n, n), rand_c64(n + 7, n)
    for off in (0, 1):  # 16-B aligned and 8-B aligned starts
        want = x.copy()
        orc.scale(want[off:], 0.3)
        d = env.put(x)
        env.ctx.scale(d[off:], 0.3)
        assert bits_equal(env.get(d), want)
        want = x.copy()
        orc.rotate(want[off:], 0.70710678 + 0.25881904j)
        d = env.put(x)
        env.ctx.rotate(d[off:], 0.70710678 + 0.25881904j)
        assert bits_equal(env.get(d), want)
        want = zeros("c64", n)
        orc.add(x[off:], y[off:], want[off:])
        d = env.zeros("c64", n)
        env.ctx.add(env.put(x)[off:], env.put(y)[off:], d[off:])
        assert bits_equal(env.get(d), want)


def test_stream_add_sum(env, orc, kats):
    k = kats["stream_add_c64"]
    bufs = [env.put(filled("c64", k["n"], k["fill"])) for _ in range(k["k"])]
    out = env.put(filled("c64", k["n"], [7, 7]))
    env.ctx.sum(out, bufs)
    assert np.all(env.get(out) == c(k["value"]))
    for name, fmt in (("stream_add_i8", "i8"), ("stream_add_i16", "i16")):
        k = kats[name]
        out = env.zeros(fmt, k["n"])
        env.ctx.sum(out, [env.put(filled(fmt, k["n"], k["fill"])) for _ in range(k["k"])])
        assert np.all(env.get(out) == np.asarray(k["value"]))
    # random, 4-way, ordered float sum; wrapping integer sums; odd length
    for fmt, gen in (("c64", rand_c64), ("i16", rand_i16), ("i8", rand_i8)):
        n = 10007
        src = [gen(30 + i, n) for i in range(4)]
        want = zeros(fmt, n)
        orc.sum_(want, src)
        out = env.zeros(fmt, n)
        env.ctx.sum(out, [env.put(s) for s in src])
        assert bits_equal(env.get(out), want)
    with pytest.raises(env.hz.ErrSampleFormatUnknown):  # stream/add.go:55-61
        env.ctx.sum(env.zeros("u8", 4), [env.zeros("u8", 4)])
    out = env.zeros("c64", 2)  # -0 inputs give +0
    env.ctx.sum(out, [env.put(samples("c64", [[-0.0, -0.0], [-0.0, 1.0]]))])
    g = env.get(out)
    assert not np.signbit(g[0].real) and not np.signbit(g[0].imag)


# ---- lookup tables -----------------------------------------------------------------------

@pytest.mark.parametrize("dst_fmt", ["u8", "i8", "i16", "c64"])
@pytest.mark.parametrize("src_fmt", ["u8", "i8"])
def test_lookup_table(env, orc, src_fmt, dst_fmt):
    ident = orc.lut_identity()
    tab = zeros(dst_fmt, 65536)
    if dst_fmt == src_fmt:
        tab = ident.view(tab.dtype).copy()
    else:
        orc.convert(tab, ident.view(np.int8) if src_fmt == "i8" else ident)
    n = 50001
    src = rand_u8(9, n) if src_fmt == "u8" else rand_i8(9, n)
    want = zeros(dst_fmt, n)
    assert orc.lut_apply(want, tab, src.view(np.uint8)) == n
    lut = env.ctx.lut(FMTS[src_fmt], env.put(tab))
    dst = env.zeros(dst_fmt, n)
    assert lut.lookup(dst, env.put(src)) == n
    assert bits_equal(env.get(dst), want)
    assert lut.source_sample_format() == FMTS[src_fmt] and lut.destination_sample_format() == FMTS[dst_fmt]
    with pytest.raises(env.hz.ErrDstTooSmall):
        lut.lookup(env.zeros(dst_fmt, 10), env.put(src))
    other = "i16" if dst_fmt != "i16" else "c64"
    with pytest.raises(env.hz.ErrSampleFormatMismatch):
        lut.lookup(env.zeros(other, n), env.put(src))
    lut.close()


def test_lookup_table_create_errors(env):
    with pytest.raises(env.hz.HzsdrError):
        env.ctx.lut(env.hz.FMT_U8, env.zeros("u8", 100))  # must be 65536 samples
    with pytest.raises(env.hz.ErrSampleFormatUnknown):
        env.ctx.lut(env.hz.FMT_C64, env.zeros("u8", 65536))


def _counter_u8(n):
    i = np.arange(n, dtype=np.uint32) & 0xFFFF
    return np.stack([i & 0xFF, (i & 0xFF00) >> 8], 1).astype(np.uint8)


def test_rotate_lut_u8(env, orc, kats):
    """stream/multiply_test.go:71-112 + the I*255+Q aliasing quirk."""
    k = kats["rotate_lut_u8"]
    vals = _counter_u8(k["n"])
    for m in (c(k["m"]), np.complex64(0.6 + 0.3j), np.complex64(1.7 - 0.2j)):
        want = vals.copy()
        orc.rotate_u8_apply(orc.rotate_table_u8(m), want)
        t = env.ctx.rotlut(env.hz.FMT_U8, m)
        buf = env.put(vals)
        t.apply(buf)
        assert bits_equal(env.get(buf), want)
        t.close()
    # the KAT proper: equals convert -> multiply -> convert done with the GPU ops
    m = c(k["m"])
    cbuf, ref = env.zeros("c64", k["n"]), env.zeros("u8", k["n"])
    env.ctx.convert(cbuf, env.put(vals))
    env.ctx.rotate(cbuf, m)
    env.ctx.convert(ref, cbuf)
    t = env.ctx.rotlut(env.hz.FMT_U8, 1)
    t.set_multiplier(m)  # SetMultiplier rebuilds the table
    buf = env.put(vals)
    t.apply(buf)
    assert bits_equal(env.get(buf), env.get(ref))
    a = env.put(samples("u8", [[3, 255], [4, 0]]))
    t.set_multiplier(1)
    t.apply(a)
    g = env.get(a)
    assert g[0].tolist() == g[1].tolist() == [4, 0]
    t.close()


def test_rotate_lut_i8(env, orc, kats):
    k = kats["rotate_lut_i8"]
    i = np.arange(k["n"], dtype=np.int64) & 0xFFFF
    vals = np.stack([(i & 0xFF), ((i & 0xFF00) >> 8) - 127], 1).astype(np.int8)
    for m in (c(k["m"]), np.complex64(-0.4 + 0.9j)):
        want = zeros("i8", k["n"])
        orc.lut_apply(want, orc.rotate_table_i8(m), vals.view(np.uint8))
        t = env.ctx.rotlut(env.hz.FMT_I8, m)
        buf = env.put(vals)
        t.apply(buf)
        assert bits_equal(env.get(buf), want)
        t.close()


# ---- NCO / Shift ---------------------------------------------------------------------------

@pytest.mark.parametrize("rate,shift,n", [
    (20_000_000, 2.5e6, 300_000), (1_800_000, 1000.0, 61_440), (200_000_000, -70e6, 200_000),
    (1000, 123.0, 50_000), (48_000, -7000.5, 100_001)])
def test_shift_bit_exact_and_state_carries(env, orc, rate, shift, n):
    """The GPU runs the same restated math.Sincos as the oracle on an exactly
    reproduced clock, so the <= 1 ULP bar is met with 0 ULP."""
    x = rand_c64(2, n)
    want = x.copy()
    ref = orc.Shifter(rate)
    nco = env.ctx.nco(rate)
    d = env.put(x)
    cuts = [0, n // 3 + 1, n // 2, n]  # three buffers, odd boundaries
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        ref(shift, want[lo:hi])
        nco(shift, d[lo:hi])
        assert nco.ts == ref.ts.value
    got = env.get(d)
    assert ulp_diff(got, want).max() <= 1
    assert bits_equal(got, want)
    nco.close()


def test_shift_bit_exact_at_volume(hz, orc):
    """2^25 samples in one buffer: bit-identical to the oracle end to end (the
    clock table, Sincos restatement and complex multiply at volume)."""
    import torch
    n, rate, shift = 1 << 25, 20_000_000, 2.5e6
    x = rand_c64(77, n)
    want = x.copy()
    ref = orc.Shifter(rate)
    ref.ts.value = 3.25
    ref(shift, want)
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    nco = ctx.nco(rate)
    nco.ts = 3.25
    d = torch.from_numpy(x).cuda()
    nco(shift, d)
    ctx.synchronize()
    got = d.cpu().numpy()
    assert nco.ts == ref.ts.value
    assert bits_equal(got, want)
    nco.close()
    ctx.close()


@pytest.mark.parametrize("rate,shift,t0,n", [
    (20_000_000, 2.5e6, 0.0, 1 << 22),            # fs/8: half the factors have a component at the rounding noise of the phase
    (20_000_000, -2.5e6, 2 * math.pi - 0.05, 1 << 22),  # across the 2 pi wrap: a burst of short clock runs
    (20_000_000, 0.0, 1.0, 70_001),                # phase +0 throughout
    (20_000_000, -0.0, 0.0, 70_001),               # phase -0: Sincos(-0) = (-0, 1)
    (20_000_000, 1e-32, 0.0, 70_001),              # phases below 2^-60: float32 denormals, outside the straight path
    (20_000_000, 3e-13, 0.0, 300_001),             # phases around 2^-60: tiles either side of the limit
    (200_000_000, 95e6, 0.85, 300_001),            # phases crossing 2^29: tiles either side of the Payne-Hanek switch
    (1000, 123.0, 0.0, 50_000),                    # a millisecond per sample
    (2_400_000, 1e6 / 3, 0.0, (1 << 20) + 3)])
def test_shift_straight_path_and_queue_bit_exact(hz, orc, rate, shift, t0, n):
    """shift_exact_kernel: the factor from sincos_narrow where its check can tell, go_sincos through the wave's
    queue elsewhere -- in place, at both alignments, bit for bit against the oracle's math.Sincos restatement;
    inputs with signed zeros so that the sign of a zero factor component shows."""
    import torch
    x = rand_c64(31, n + 1)
    x[5::7] = np.complex64(complex(-0.0, 0.0))
    x[6::11] = np.complex64(complex(1.0, -0.0))
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    for off in (0, 1):  # 16-byte aligned, and a Go sub-slice starting one sample in
        want = x[off:off + n].copy()
        ref = orc.Shifter(rate)
        ref.ts.value = t0
        nco = ctx.nco(rate)
        nco.ts = t0
        d = torch.from_numpy(x.copy()).cuda()
        cuts = [0, n // 2 + 1, n]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            ref(shift, want[lo:hi])
            nco(shift, d[off + lo:off + hi])
            assert nco.ts == ref.ts.value
        ctx.synchronize()
        got = d.cpu().numpy()
        assert bits_equal(got[off:off + n], want)
        assert bits_equal(got[:off], x[:off]) and bits_equal(got[off + n:], x[off + n:])
        nco.close()
    ctx.close()


@pytest.mark.parametrize("seed", range(6))
def test_shift_random_rates_shifts_and_clocks_bit_exact(hz, orc, seed):
    """Eight random (sample rate, shift, clock start, length, alignment) per seed against the oracle, bit for bit:
    rates 1 kHz .. 1 GHz, shifts up to the rate and beyond, clocks anywhere in [0, 2 pi) (and, rarely, set far
    outside by hand), buffers cut in two calls at a random sample."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    for case in range(8):
        rate = int(10.0 ** rng.uniform(3, 9))
        shift = float(rng.uniform(-1.2, 1.2) * rate) if rng.integers(0, 4) else float(10.0 ** rng.uniform(-6, 9.5) * rng.choice([-1, 1]))
        t0 = float(rng.uniform(0, 2 * math.pi)) if rng.integers(0, 8) else float(10.0 ** rng.uniform(-12, 6))
        n = int(rng.integers(1, 400_000))
        off = int(rng.integers(0, 2))
        cut = int(rng.integers(0, n + 1))
        x = rand_c64(100 * seed + case, n + off)
        want = x[off:].copy()
        ref = orc.Shifter(rate)
        ref.ts.value = t0
        nco = ctx.nco(rate)
        nco.ts = t0
        d = torch.from_numpy(x.copy()).cuda()
        for lo, hi in ((0, cut), (cut, n)):
            if hi > lo:
                ref(shift, want[lo:hi])
                nco(shift, d[off + lo:off + hi])
        ctx.synchronize()
        assert nco.ts == ref.ts.value, (rate, shift, t0, n, off, cut)
        assert bits_equal(d.cpu().numpy()[off:], want), (rate, shift, t0, n, off, cut)
        nco.close()
    ctx.close()


@pytest.mark.parametrize("rate,shift,t0", [(20_000_000, 2.5e6, 3.25), (20_000_000, -7.3e6, 2 * math.pi - 0.5),
                                            (200_000_000, 95e6, 5.0)])
def test_nco_shift_ulp1_is_within_one_ulp_of_the_factor_at_volume(hz, orc, rate, shift, t0):
    """hzsdr_nco_set_ulp1 (opt-in; what ShiftReader / ShiftBuffer bind to): unit inputs over 2^25 samples, so the
    output IS the rotation factor.  The reference's factor is the true value rounded once (half an ulp), the
    opt-in's is within one ulp of the true value: components of size <= 1 are at most 1.5 x 2^-24 apart.  The
    clock crosses 2*pi in the second case (1.6 s of samples from 2*pi - 0.5); in the third the phase is ~3e9
    rad, math.Sincos' Payne-Hanek range, where the opt-in reduces with a double-double 1 / 2 pi.  The clock itself
    (ts after the call) is the reference's bit for bit, and switching the option off again restores bit equality."""
    import torch
    n = 1 << 25
    x = np.ones(n, np.complex64)
    want = x.copy()
    ref = orc.Shifter(rate)
    ref.ts.value = t0
    ref(shift, want)
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    nco = ctx.nco(rate).set_ulp1()
    nco.ts = t0
    d = torch.from_numpy(x).cuda()
    nco(shift, d)
    ctx.synchronize()
    got = d.cpu().numpy()
    assert nco.ts == ref.ts.value
    dd = got.view(np.float32).astype(np.float64) - want.view(np.float32).astype(np.float64)
    assert np.abs(dd).max() <= 1.5 * 2.0 ** -24, np.abs(dd).max() * 2.0 ** 24
    assert not bits_equal(got, want)  # (it really is the other kernel)
    # a sample-aligned sub-slice (8-byte, not 32-byte aligned; a length that is not a multiple of four)
    m = 100_003
    want_s = np.ones(m, np.complex64)
    ref.ts.value = t0
    ref(shift, want_s)
    nco.ts = t0
    d = torch.from_numpy(x[:m + 1].copy()).cuda()
    nco(shift, d[1:])
    ctx.synchronize()
    got_s = d.cpu().numpy()
    assert got_s[0] == 1.0  # (the sample in front of the slice is not touched)
    dd = got_s[1:].view(np.float32).astype(np.float64) - want_s.view(np.float32).astype(np.float64)
    assert np.abs(dd).max() <= 1.5 * 2.0 ** -24
    nco.set_ulp1(False)
    nco.ts = t0
    d = torch.from_numpy(x).cuda()
    nco(shift, d)
    ctx.synchronize()
    assert bits_equal(d.cpu().numpy(), want)
    nco.close()
    ctx.close()


def test_shift_large_phase_uses_payne_hanek(env, orc):
    """2*pi*shift*ts beyond 2^29 rad takes trigReduce (src/math/trig_reduce.go)."""
    n, rate, shift = 100_000, 200_000_000, 95e6
    x = rand_c64(4, n)
    want = x.copy()
    ref = orc.Shifter(rate)
    ref.ts.value = 5.0  # phase ~ 3e9 rad
    ref(shift, want)
    nco = env.ctx.nco(rate)
    nco.ts = 5.0
    d = env.put(x)
    nco(shift, d)
    assert bits_equal(env.get(d), want)
    # and stays within 1 ULP of an independent libm evaluation
    lib = x.copy()
    ref2 = orc.Shifter(rate, use_libm=True)
    ref2.ts.value = 5.0
    ref2(shift, lib)
    assert ulp_diff(env.get(d), lib).max() <= 1
    nco.close()


def test_shift_roundtrip_kat(env, orc, kats):
    k = kats["shift_roundtrip"]
    cw = orc.cw(k["n"], k["freq"], k["rate"], 0.0)
    d = env.put(cw)
    hi, lo = env.ctx.nco(k["rate"]), env.ctx.nco(k["rate"])
    hi(k["shift"], d)
    lo(-k["shift"], d)
    got = env.get(d)
    assert in_epsilon(1 + cw.real, 1 + got.real, k["eps"]) and in_epsilon(1 + cw.imag, 1 + got.imag, k["eps"])


# ---- decimate / downsample -------------------------------------------------------------------

def test_decimate(env, orc, kats):
    hz = env.hz
    k = kats["decimate_count"]
    for fmt in k["formats"]:
        assert env.ctx.decimate(env.zeros(fmt, k["n"]), env.zeros(fmt, k["n"]), k["factor"]) == k["count"]
    k = kats["decimate_skippy"]
    i = (np.arange(k["n"]) % 10).astype(np.uint8)
    dst = env.put(filled("u8", k["count"], [9, 9]))
    assert env.ctx.decimate(dst, env.put(np.stack([i, i], 1)), k["factor"]) == k["count"]
    assert np.all(env.get(dst) == 0)
    with pytest.raises(hz.ErrSampleFormatMismatch):
        env.ctx.decimate(env.zeros("u8", 32768), env.zeros("c64", 32768), 10)
    with pytest.raises(hz.ErrDstTooSmall):
        env.ctx.decimate(env.zeros("u8", 10), env.zeros("u8", 32768), 10)
    with pytest.raises(hz.ErrSampleFormatUnknown):  # no i8 case: stream/decimate.go:85-97
        env.ctx.decimate(env.zeros("i8", 8), env.zeros("i8", 8), 2)
    for fmt, gen in (("c64", rand_c64), ("i16", rand_i16), ("u8", rand_u8)):
        for factor, n in ((7, 100001), (8, 65536), (1, 1000), (3, 2)):
            x = gen(1, n)
            want = zeros(fmt, n // factor + 3)
            cnt = orc.decimate(want, x, factor)
            d = env.zeros(fmt, n // factor + 3)
            assert env.ctx.decimate(d, env.put(x), factor) == cnt
            assert bits_equal(env.get(d), want)


def test_downsample(env, orc, kats):
    hz = env.hz
    k = kats["downsample_calc"]
    e = (np.arange(k["n"]) % 4).astype(np.float32)
    dst = env.zeros("c64", k["n"])
    assert env.ctx.downsample(dst, env.put((e + 1j * e).astype(np.complex64)), k["factor"]) == k["count"]
    assert np.all(env.get(dst)[:k["count"]] == c(k["value"]))
    for fmt, gen in (("i16", rand_i16), ("u8", rand_u8), ("c64", rand_c64)):
        for factor, n in ((8, 8 * 4096 + 5), (4, 4 * 1000), (3, 3001), (16, 16 * 513), (1, 77), (5, 4)):
            x = gen(4, n)
            want = zeros("c64", n // factor + 1)
            cnt = orc.downsample(want, x, factor)
            d = env.zeros("c64", n // factor + 1)
            assert env.ctx.downsample(d, env.put(x), factor) == cnt
            assert bits_equal(env.get(d), want), (fmt, factor, n)
    with pytest.raises(hz.ErrSampleFormatMismatch):
        env.ctx.downsample(env.zeros("u8", 8), env.zeros("i16", 64), 8)
    with pytest.raises(hz.ErrDstTooSmall):
        env.ctx.downsample(env.zeros("c64", 7), env.zeros("i16", 64), 8)
    with pytest.raises(hz.ErrSampleFormatUnknown):
        env.ctx.downsample(env.zeros("c64", 8), env.zeros("i8", 8), 2)


# ---- FFT / convolution -------------------------------------------------------------------------

def _rel_l2(got, want):
    want = want.astype(np.complex128)
    return float(np.linalg.norm(got.astype(np.complex128) - want) / max(np.linalg.norm(want), 1e-30))


@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536,
                               1 << 17, 1 << 18, 1 << 20, 1 << 22])
def test_fft_against_float64(env, n):
    """Tolerance: relative L2 error <= 3e-7 * log2(N) + 1e-7 against numpy's
    float64 FFT (float32 butterflies; the reference pins no values: SURVEY 2b).
    Sizes: radix-4 core (4..128, 8192), radix-16 core (256..4096), global radix-2
    (1, 2, 2^14), two-step (2^16..2^24: kerberos 64 Ki, graft 256 Ki)."""
    tol = 3e-7 * max(1, math.log2(n)) + 1e-7
    batch = 3 if n <= 8192 else (2 if n <= (1 << 18) else 1)
    x = rand_c64(n, n * batch)
    X = np.fft.fft(x.astype(np.complex128).reshape(batch, n), axis=1).reshape(-1)
    iq, fr = env.put(x), env.zeros("c64", n * batch)
    p = env.ctx.fft_plan(iq, fr, env.hz.FFT_FORWARD, batch=batch)
    p.transform()
    assert _rel_l2(env.get(fr), X) < tol
    p.close()
    Y = np.fft.ifft(x.astype(np.complex128).reshape(batch, n), axis=1).reshape(-1) * n  # unnormalised
    fr, iq = env.put(x), env.zeros("c64", n * batch)
    p = env.ctx.fft_plan(iq, fr, env.hz.FFT_BACKWARD, batch=batch)
    p.transform()
    assert _rel_l2(env.get(iq), Y) < tol
    p.close()


@pytest.mark.parametrize("n,batch", [(1 << 14, 3), (1 << 14, 40), (1 << 15, 5), (1 << 16, 3), (1 << 16, 24), (1 << 17, 9),
                                     (1 << 18, 8), (1 << 19, 2), (1 << 21, 1), (1 << 21, 3), (1 << 23, 1), (1 << 24, 1)])
def test_fft_two_step_tilings(env, n, batch):
    """The two-step lengths over batches whose tile counts are and are not multiples of eight (the XCD-contiguous
    tile order applies to the first only), every column / row kernel width (N1 = 256 .. 4096; rows of 64 .. 4096
    points on 256, 512 and 1024 lanes), forward and backward.  Same bound as test_fft_against_float64."""
    tol = 3e-7 * math.log2(n) + 1e-7
    x = rand_c64(n + batch, n * batch)
    xs = x.astype(np.complex128).reshape(batch, n)
    iq, fr = env.put(x), env.zeros("c64", n * batch)
    p = env.ctx.fft_plan(iq, fr, env.hz.FFT_FORWARD, batch=batch)
    p.transform()
    got = env.get(fr).reshape(batch, n)
    want = np.fft.fft(xs, axis=1)
    for b in range(batch):  # per transform: one transform's tiles in the wrong place must not hide in the batch's norm
        assert _rel_l2(got[b], want[b]) < tol, b
    p.close()
    fr, iq = env.put(x), env.zeros("c64", n * batch)
    p = env.ctx.fft_plan(iq, fr, env.hz.FFT_BACKWARD, batch=batch)
    p.transform()
    got = env.get(iq).reshape(batch, n)
    want = np.fft.ifft(xs, axis=1) * n
    for b in range(batch):
        assert _rel_l2(got[b], want[b]) < tol, b
    p.close()


@pytest.mark.parametrize("n", [3, 5, 12, 1000, 1200, 1536, 4099, 3 << 10, 3 << 15, 100_003, (1 << 20) + 7])
def test_fft_any_length_against_float64(env, n):
    """fft.Planner takes whatever length its buffers have (fft/fft.go:45-48): lengths that are not powers of two run as
    Bluestein's chirp transform over the power-of-two kernels.  Same tolerance as the power-of-two plans -- relative L2
    <= 3e-7 * log2(N) + 1e-7 against numpy's float64 transform -- forward and (unnormalised) backward, batched."""
    tol = 3e-7 * max(1, math.log2(n)) + 1e-7
    batch = 3 if n <= 8192 else (2 if n <= (1 << 17) else 1)
    x = rand_c64(n, n * batch)
    X = np.fft.fft(x.astype(np.complex128).reshape(batch, n), axis=1).reshape(-1)
    iq, fr = env.put(x), env.zeros("c64", n * batch)
    p = env.ctx.fft_plan(iq, fr, env.hz.FFT_FORWARD, batch=batch)
    p.transform()
    p.transform()  # (a plan is used many times: the chirp tables and the scratch are the context's)
    assert _rel_l2(env.get(fr), X) < tol
    p.close()
    Y = np.fft.ifft(x.astype(np.complex128).reshape(batch, n), axis=1).reshape(-1) * n  # unnormalised
    fr, iq = env.put(x), env.zeros("c64", n * batch)
    p = env.ctx.fft_plan(iq, fr, env.hz.FFT_BACKWARD, batch=batch)
    p.transform()
    assert _rel_l2(env.get(iq), Y) < tol
    p.close()


def test_fft_any_length_at_its_upper_limit(env):
    """The largest length that is not a power of two: 2^23 - 3 runs as a chirp transform over the 2^24-point plan
    (include/hzsdr.h: any length up to 2^23, powers of two up to 2^24); 2^23 + 1 is refused when the plan is made, as a
    planner's refusal is in the reference (fft/fft.go:45-48 returns an error).  Tolerance as above."""
    if env.kind != "device":
        pytest.skip("1 GiB of scratch once is enough")
    hz = env.hz
    n = (1 << 23) - 3
    x = rand_c64(77, n)
    X = np.fft.fft(x.astype(np.complex128))
    iq, fr = env.put(x), env.zeros("c64", n)
    p = env.ctx.fft_plan(iq, fr, hz.FFT_FORWARD)
    p.transform()
    assert _rel_l2(env.get(fr), X) < 3e-7 * 23 + 1e-7
    p.close()
    big = env.zeros("c64", (1 << 23) + 1)
    with pytest.raises(hz.ErrInvalidArgument):
        env.ctx.fft_plan(big, env.zeros("c64", (1 << 23) + 1), hz.FFT_FORWARD)


def test_fft_conformance_at_a_length_that_is_not_a_power_of_two(env, orc):
    """testutils/fft.go:54-138 at N = 1000: a CW tone lands in its bin (forward), a single bin comes back as that bin
    (backward then forward), mismatched lengths are refused."""
    hz = env.hz
    n, rate = 1000, 1000.0
    for freq, idx in ((0.0, 0), (10.0, 10), (125.0, 125), (-10.0, 990), (499.0, 499)):
        iq, out = env.put(orc.cw(n, freq, rate, 0.0)), env.zeros("c64", n)
        p = env.ctx.fft_plan(iq, out, hz.FFT_FORWARD)
        p.transform()
        p.close()
        assert int(np.argmax(np.abs(env.get(out).astype(np.complex128)))) == idx, freq
    for b in (0, 1, 333, 999):
        f = zeros("c64", n)
        f[b] = 1 + 1j
        fr, iq = env.put(f), env.zeros("c64", n)
        p = env.ctx.fft_plan(iq, fr, hz.FFT_BACKWARD)
        p.transform()
        p.close()
        fr2 = env.zeros("c64", n)
        p = env.ctx.fft_plan(iq, fr2, hz.FFT_FORWARD)
        p.transform()
        p.close()
        got = env.get(fr2).astype(np.complex128)
        assert int(np.argmax(np.abs(got))) == b
        assert abs(got[b] - n * (1 + 1j)) <= 2e-5 * n  # backward is unnormalised: the round trip scales by N
    with pytest.raises(hz.ErrDstTooSmall):
        env.ctx.fft_plan(env.zeros("c64", 1000), env.zeros("c64", 1001), hz.FFT_FORWARD)


def test_fft_conformance_kats(env, orc, kats):
    """testutils/fft.go:54-138, the suite any Planner must pass."""
    hz = env.hz
    k = kats["fft_forward_bins"]
    for freq, idx in k["cases"]:
        iq, out = env.put(orc.cw(k["n"], freq, k["rate"], 0.0)), env.zeros("c64", k["n"])
        p = env.ctx.fft_plan(iq, out, hz.FFT_FORWARD)
        p.transform()
        p.close()
        assert int(np.argmax(np.abs(env.get(out).astype(np.complex128)))) == idx
    k = kats["fft_backward_roundtrip"]
    for b in k["bins"]:
        f = zeros("c64", k["n"])
        f[b] = 1 + 1j
        fr, iq = env.put(f), env.zeros("c64", k["n"])
        p = env.ctx.fft_plan(iq, fr, hz.FFT_BACKWARD)
        p.transform()
        p.close()
        fr2 = env.zeros("c64", k["n"])
        p = env.ctx.fft_plan(iq, fr2, hz.FFT_FORWARD)
        p.transform()
        p.close()
        assert int(np.argmax(np.abs(env.get(fr2)))) == b
    for a, b, d in kats["fft_mismatch"]["cases"]:
        with pytest.raises(hz.ErrDstTooSmall):
            env.ctx.fft_plan(env.zeros("c64", a), env.zeros("c64", b),
                             hz.FFT_FORWARD if d == "forward" else hz.FFT_BACKWARD)


def _lowpass_bins(n, taps=None):
    t = np.arange(n) - (n - 1) / 2
    h = np.sinc(t / 8) / 8 * np.hamming(n)
    return np.fft.fft(h.astype(np.complex128) / n).astype(np.complex64)


@pytest.mark.parametrize("flen", [4, 64, 1024, 2048, 8192, 1, 2, 3, 1000, 1200, 1536, 4099, 16384])
def test_convolution_blocks_reference_semantics(env, orc, flen):
    """stream.ConvolutionReader: block-circular, no overlap; vs the oracle's
    float64-FFT restatement.  Tolerance: relative L2 <= 2e-6."""
    nblk = 5
    x = rand_c64(3, nblk * flen + flen // 2)
    H = _lowpass_bins(flen)
    want = zeros("c64", len(x))
    assert orc.convolution_reader(want, x, H) == nblk * flen
    out = env.zeros("c64", len(x))
    assert env.ctx.convolution_blocks(out, env.put(x), env.put(H)) == nblk * flen
    got = env.get(out)
    assert _rel_l2(got[:nblk * flen], want[:nblk * flen]) < 2e-6
    assert np.all(got[nblk * flen:] == 0)  # the partial block is never produced


@pytest.mark.parametrize("flen", [2048, 4096, 8192])
def test_convolution_blocks_walk_far_more_blocks_than_the_grid_holds(env, orc, flen):
    """The direct c64 form walks its blocks in a loop (one workgroup, many blocks, the next one's loads in
    flight) and from 2048 bins on several waves share one transform's LDS: every trip has to be fenced from the
    previous trip's last reads.  8192 blocks are 4 ... 16 trips per workgroup on a 256-CU chip; blocks are
    independent (block-circular), so the oracle is run on a sample of them -- the first, the last, and
    every 97th -- and each is compared alone.  Tolerance as above: relative L2 <= 2e-6 per block."""
    if env.kind != "device":
        pytest.skip("one memory space is enough for a kernel-internal ordering")
    nblk = 8192
    rng = np.random.default_rng(flen)
    x = (rng.standard_normal(nblk * flen, dtype=np.float32) + 1j * rng.standard_normal(nblk * flen, dtype=np.float32)).astype(np.complex64)
    H = _lowpass_bins(flen)
    out = env.zeros("c64", len(x))
    assert env.ctx.convolution_blocks(out, env.put(x), env.put(H)) == nblk * flen
    got = env.get(out)
    pick = sorted(set(list(range(0, nblk, 97)) + [1, 2, nblk - 2, nblk - 1]))
    xs = np.concatenate([x[b * flen:(b + 1) * flen] for b in pick])
    want = zeros("c64", len(xs))
    assert orc.convolution_reader(want, xs, H) == len(xs)
    for i, b in enumerate(pick):
        assert _rel_l2(got[b * flen:(b + 1) * flen], want[i * flen:(i + 1) * flen]) < 2e-6, (flen, b)


@pytest.mark.parametrize("flen,nblk", [(1024, 8192 + 5), (512, 20000 + 3), (256, 40000 + 37)])
def test_convolution_blocks_shared_tables_walk_ragged_block_counts(env, orc, flen, nblk):
    """Blocks of 1024 points and fewer run in workgroups of sixteen waves that share the tables in LDS and never
    meet after the first barrier (conv_blocks_shared_kernel): a wave holds one, two or four transforms, walks
    its blocks with the next ones' loads in flight, and the last trip is ragged -- some waves, and some of a wave's
    transforms, have no block left.  Sampled blocks against the oracle (blocks are independent), the partial block
    behind the last whole one untouched.  Tolerance: relative L2 <= 2e-6 per block."""
    if env.kind != "device":
        pytest.skip("one memory space is enough for a kernel-internal ordering")
    rng = np.random.default_rng(flen)
    n = nblk * flen + flen // 2
    x = (rng.standard_normal(n, dtype=np.float32) + 1j * rng.standard_normal(n, dtype=np.float32)).astype(np.complex64)
    H = _lowpass_bins(flen)
    out = env.zeros("c64", n)
    assert env.ctx.convolution_blocks(out, env.put(x), env.put(H)) == nblk * flen
    got = env.get(out)
    assert np.all(got[nblk * flen:] == 0)
    pick = sorted(set(list(range(0, nblk, 211)) + [1, 2, 3, 63, 64, 65] + list(range(nblk - 70, nblk))))
    xs = np.concatenate([x[b * flen:(b + 1) * flen] for b in pick])
    want = zeros("c64", len(xs))
    assert orc.convolution_reader(want, xs, H) == len(xs)
    for i, b in enumerate(pick):
        assert _rel_l2(got[b * flen:(b + 1) * flen], want[i * flen:(i + 1) * flen]) < 2e-6, (flen, b)


@pytest.mark.parametrize("n", [8, 1024, 32768, 1000, 4099])
def test_convolve_closures(env, orc, n):
    hz = env.hz
    a, b = rand_c64(5, n), rand_c64(6, n)
    H = _lowpass_bins(n)
    for kind in ("freq", "convolve", "xcorr"):
        want = zeros("c64", n)
        if kind == "freq":
            orc.convolve_freq(want, a, H)
            da, db, dd = env.put(a), env.put(H), env.zeros("c64", n)
            cv = env.ctx.convolve_freq(dd, da, db)
        else:
            orc.convolve(want, a, b, conj=(kind == "xcorr"))
            da, db, dd = env.put(a), env.put(b), env.zeros("c64", n)
            cv = (env.ctx.convolve if kind == "convolve" else env.ctx.cross_correlate)(dd, da, db)
        cv()
        assert _rel_l2(env.get(dd), want) < 3e-6, kind
        cv.close()
    with pytest.raises(hz.ErrLengthMismatch):  # fft/convolution.go:156-158
        env.ctx.convolve_freq(env.zeros("c64", 8), env.zeros("c64", 8), env.zeros("c64", 4))
    with pytest.raises(hz.ErrLengthMismatch):  # fft/convolution.go:37-39
        env.ctx.convolve(env.zeros("c64", 8), env.zeros("c64", 4), env.zeros("c64", 8))


# ---- beamform ------------------------------------------------------------------------------------

@pytest.mark.parametrize("fmt", ["c64", "u8", "i16", "i8"])
def test_beamform_bit_exact(env, orc, fmt):
    n = 40001
    gen = {"c64": rand_c64, "u8": rand_u8, "i16": rand_i16, "i8": rand_i8}[fmt]
    ch = [gen(20 + i, n) for i in range(4)]
    w = env.hz.beamform_angles(433e6, 30.0, [0.0, 0.1, 0.2, 0.3])
    assert w[0] == 1  # the first channel's multiply is skipped
    chc = []
    for x in ch:  # the per-channel ConvertReader of stream/beamform.go:151
        y = zeros("c64", n)
        orc.convert(y, x)
        chc.append(y)
    want = zeros("c64", n)
    orc.beamform(want, chc, w)
    out = env.zeros("c64", n)
    dch = [env.put(x) for x in ch]
    env.ctx.beamform(out, dch, w)
    assert bits_equal(env.get(out), want)
    # sharded form: ranks continue the ordered sum
    out2 = env.zeros("c64", n)
    env.ctx.beamform(out2, dch[:2], w[:2], accumulate=False)
    env.ctx.beamform(out2, dch[2:], w[2:], accumulate=True)
    assert bits_equal(env.get(out2), want)


def test_weighted_channels_then_ordered_sum_is_beamform(env, orc):
    """The pieces multigpu.ordered_alltoall moves: one weighted channel per launch
    (0 + w_c * x_c) and the K-way ordered Add reproduce the fused Beamform bit for bit."""
    n = 30_001
    ch = [rand_c64(60 + i, n) for i in range(4)]
    w = env.hz.beamform_angles(433e6, 30.0, [0.0, 0.1, 0.2, 0.3])
    want = zeros("c64", n)
    orc.beamform(want, ch, w)
    weighted = []
    for x, wc in zip(ch, w):
        y = env.zeros("c64", n)
        env.ctx.beamform(y, [env.put(x)], [wc])
        weighted.append(y)
    out = env.zeros("c64", n)
    env.ctx.sum(out, weighted)
    assert bits_equal(env.get(out), want)


def test_beamform_and_sum_past_the_cache(hz, orc):
    """Working sets above 192 MiB take the non-temporal form of the two kernels
    (4 x 2^23 c64 channels + output = 320 MiB): same bits as the oracle."""
    import torch
    n = 1 << 23
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    ch = [rand_c64(40 + i, n) for i in range(4)]
    w = hz.beamform_angles(433e6, 30.0, [0.0, 0.1, 0.2, 0.3])
    dch = [torch.from_numpy(x).cuda() for x in ch]
    out = torch.zeros(n, dtype=torch.complex64, device="cuda")
    ctx.beamform(out, dch, w)
    ctx.synchronize()
    want = zeros("c64", n)
    orc.beamform(want, ch, w)
    assert bits_equal(out.cpu().numpy(), want)
    ctx.sum(out, dch)
    ctx.synchronize()
    orc.sum_(want, ch)
    assert bits_equal(out.cpu().numpy(), want)
    ctx.close()


# ---- fused chains ----------------------------------------------------------------------------------

def test_chain_convert_shift_gain_equals_separate_ops(env, orc):
    """BASELINE config 2 shape: u8 -> c64 -> Shift -> Gain in ONE kernel must be
    bit-identical to the three reference ops applied one after another."""
    rate, shift, gain = 20_000_000, 2.5e6, 0.5
    n = 150_001
    x = rand_u8(9, n)
    want = zeros("c64", n)
    orc.convert(want, x)
    ref = orc.Shifter(rate)
    ch = env.ctx.chain(env.hz.FMT_U8, rate).shift(shift).gain(gain)
    out = env.zeros("c64", n)
    dx = env.put(x)
    cuts = [0, 50_002, 100_003, n]  # slices aligned for 4-, 2- and 1-sample vectors
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        ref(shift, want[lo:hi])
        orc.scale(want[lo:hi], gain)
        cons, outn = ch.run(dx[lo:hi], out[lo:hi])
        assert (cons, outn) == (hi - lo, hi - lo)
    assert bits_equal(env.get(out), want)
    ch.close()


@pytest.mark.parametrize("fmt,gain", [("u8", None), ("i8", 0.25), ("i16", None), ("c64", 3.0)])
def test_chain_shift_from_every_source_format_bit_exact(env, orc, fmt, gain):
    """shift_exact_kernel behind each converter (Shift, and Shift then Gain): the converter's float32, the factor and the
    product bit for bit as the separate reference operations give them; a clock that crosses run boundaries."""
    hz = env.hz
    rate, shift, n = 2_400_000, -333_333.25, 262_144 + 2
    x = {"u8": rand_u8, "i8": rand_i8, "i16": rand_i16, "c64": rand_c64}[fmt](21, n)
    want = zeros("c64", n)
    orc.convert(want, x)
    ref = orc.Shifter(rate)
    ref.ts.value = 6.1
    ref(shift, want)
    if gain is not None:
        orc.scale(want, gain)
    ch = env.ctx.chain({"u8": hz.FMT_U8, "i8": hz.FMT_I8, "i16": hz.FMT_I16, "c64": hz.FMT_C64}[fmt], rate).shift(shift)
    if gain is not None:
        ch = ch.gain(gain)
    ch.set_time(6.1)
    out = env.zeros("c64", n)
    assert ch.run(env.put(x), out) == (n, n)
    assert bits_equal(env.get(out), want)
    ch.close()


def test_chain_shift_gain_past_the_cache_bit_exact(hz, orc):
    """A call of 128 MiB takes the non-temporal form of shift_exact_kernel (out of place, 2^23 c64 samples): the same
    bits as the oracle."""
    import torch
    n, rate, shift, gain = 1 << 23, 20_000_000, 2.5e6, 0.5
    x = rand_c64(55, n)
    want = x.copy()
    ref = orc.Shifter(rate)
    ref.ts.value = 1.75
    ref(shift, want)
    orc.scale(want, gain)
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    ch = ctx.chain(hz.FMT_C64, rate).shift(shift).gain(gain)
    ch.set_time(1.75)
    out = torch.zeros(n, dtype=torch.complex64, device="cuda")
    assert ch.run(torch.from_numpy(x).cuda(), out) == (n, n)
    ctx.synchronize()
    assert bits_equal(out.cpu().numpy(), want)
    ch.close()
    ctx.close()


def test_chain_shift_ulp1_is_within_one_ulp_of_the_factor(env, orc):
    """hzsdr_chain_shift_ulp1 (opt-in): unit inputs, so the output IS the rotation factor times the gain 1.
    The reference's factor is complex64(math.Sincos(float64 phase)) = the true value rounded once (half an
    ulp); the opt-in's is within one ulp of the true value: at most 1.5 ulp of 1.0-sized components apart
    (2^-24 each: the bound below), over clock values on both sides of the 2*pi wrap and phases up to 1e8
    rad.  Off (the default), the chain stays bit-identical to the reference."""
    rate, shift = 20_000_000, 2.5e6
    n = 1 << 20
    x = np.zeros(n, np.complex64)
    x[:] = 1.0
    dx = env.put(x)
    for t0 in (0.0, 3.0, 2 * math.pi - 0.02):
        want = x.copy()
        ref = orc.Shifter(rate)
        ref.ts.value = t0
        ref(shift, want)
        for on in (False, True):
            ch = env.ctx.chain(env.hz.FMT_C64, rate).shift(shift).gain(1.0)
            if on:
                ch.shift_ulp1()
            ch.set_time(t0)
            out = env.zeros("c64", n)
            assert ch.run(dx, out) == (n, n)
            got = env.get(out)
            assert ch.time() == ref.ts.value
            ch.close()
            if not on:
                assert bits_equal(got, want)
                continue
            d = got.view(np.float32).astype(np.float64) - want.view(np.float32).astype(np.float64)
            assert np.abs(d).max() <= 1.5 * 2.0 ** -24, (t0, np.abs(d).max() * 2.0 ** 24)
            assert not bits_equal(got, want)  # (it really is the other kernel)


def test_chain_shift_roundtrip_two_ncos(env, orc, kats):
    k = kats["shift_roundtrip"]
    cw = orc.cw(k["n"], k["freq"], k["rate"], 0.0)
    ch = env.ctx.chain(env.hz.FMT_C64, k["rate"]).shift(k["shift"]).shift(-k["shift"])
    out = env.zeros("c64", k["n"])
    ch.run(env.put(cw), out)
    got = env.get(out)
    assert in_epsilon(1 + cw.real, 1 + got.real, k["eps"]) and in_epsilon(1 + cw.imag, 1 + got.imag, k["eps"])
    want = cw.copy()
    a, b = orc.Shifter(k["rate"]), orc.Shifter(k["rate"])
    a(k["shift"], want)
    b(-k["shift"], want)
    assert bits_equal(got, want)
    ch.close()


@pytest.mark.parametrize("factor", [8, 10])
def test_chain_decimate_and_downsample_follow_reader_blocks(env, orc, factor):
    """DecimateReader / DownsampleReader work in 32 Ki blocks and restart their
    phase per block (stream/decimate.go:34-55, downsample.go:47-64)."""
    B = 32768
    n = 3 * B + 1234
    x = rand_i16(4, n)
    xc = zeros("c64", n)
    orc.convert(xc, x)
    orc.rotate(xc, 0.5 - 0.25j)
    per = B // factor
    for term in ("decimate", "downsample"):
        want = zeros("c64", 3 * per)
        for b in range(3):
            fn = orc.decimate if term == "decimate" else orc.downsample
            assert fn(want[b * per:(b + 1) * per], xc[b * B:(b + 1) * B], factor) == per
        ch = env.ctx.chain(env.hz.FMT_I16).rotate(0.5 - 0.25j)
        ch = ch.decimate(factor) if term == "decimate" else ch.downsample(factor)
        assert ch.plan(n) == (3 * B, 3 * per)
        out = env.zeros("c64", 3 * per)
        assert ch.run(env.put(x), out) == (3 * B, 3 * per)
        assert bits_equal(env.get(out), want), term
        with pytest.raises(env.hz.ErrDstTooSmall):
            ch.run(env.put(x), env.zeros("c64", 10))
        ch.close()


def test_chain_reference_convolution_then_decimate(env, orc):
    """u8 -> c64 -> Shift -> ConvolutionReader(1024 bins) -> DecimateReader(8),
    the chain spelt with the reference's own operators.  The first two stages
    are bit-exact, so the only difference from the oracle is FFT round-off:
    relative L2 <= 2e-6."""
    rate, shift, flen, D = 20_000_000, -2.5e6, 1024, 8
    n = 4 * 32768
    x = rand_u8(9, n)
    H = _lowpass_bins(flen)
    xc = zeros("c64", n)
    orc.convert(xc, x)
    orc.Shifter(rate)(shift, xc)
    conv = zeros("c64", n)
    orc.convolution_reader(conv, xc, H)
    want = zeros("c64", n // D)
    for b in range(4):
        orc.decimate(want[b * 4096:(b + 1) * 4096], conv[b * 32768:(b + 1) * 32768], D)
    ch = env.ctx.chain(env.hz.FMT_U8, rate).shift(shift).convolution(env.put(H), decimate=D)
    out = env.zeros("c64", n // D)
    assert ch.run(env.put(x), out) == (n, n // D)
    assert _rel_l2(env.get(out), want) < 2e-6
    ch.close()


@pytest.mark.parametrize("fmt,flen,nblk", [("u8", 1024, 4096 + 7), ("i16", 512, 8192 + 3), ("i8", 256, 16384 + 21)])
def test_chain_convert_then_convolution_from_byte_sources(env, orc, fmt, flen, nblk):
    """ConvertReader -> ConvolutionReader fused, nothing elementwise between them: the shared-table block kernel
    converts on its way into the first pass's registers (csrc/hz_chain_dev.h: conv_blocks_shared_kernel<N, W, FMT>).
    The conversion is bit-exact, so what differs from the oracle is FFT round-off: relative L2 <= 2e-6 per block,
    on a sample of the blocks; the partial block is not consumed."""
    if env.kind != "device":
        pytest.skip("one memory space is enough for a kernel-internal path")
    gen = {"u8": rand_u8, "i8": rand_i8, "i16": rand_i16}[fmt]
    n = nblk * flen + flen // 3
    x = gen(41, n)
    H = _lowpass_bins(flen)
    ch = env.ctx.chain(getattr(env.hz, "FMT_" + fmt.upper()), 2_400_000).convolution(env.put(H))
    out = env.zeros("c64", n)
    assert ch.run(env.put(x), out) == (nblk * flen, nblk * flen)
    got = env.get(out)
    ch.close()
    pick = sorted(set(list(range(0, nblk, 331)) + [1, 2, 3, 64] + list(range(nblk - 40, nblk))))
    xs = np.concatenate([x[b * flen:(b + 1) * flen] for b in pick])
    xc = zeros("c64", len(xs))
    orc.convert(xc, xs)
    want = zeros("c64", len(xs))
    assert orc.convolution_reader(want, xc, H) == len(xs)
    for i, b in enumerate(pick):
        assert _rel_l2(got[b * flen:(b + 1) * flen], want[i * flen:(i + 1) * flen]) < 2e-6, (fmt, flen, b)


@pytest.mark.parametrize("flen,D", [(1536, 8), (96, 4), (1200, 1), (4099, 1)])
def test_chain_convolution_of_any_length_then_decimate(env, orc, flen, D):
    """The same chain with a filter whose length is not a power of two (stream/convolution.go:57-61 blocks on
    len(filter), whatever it is): i16 -> c64 -> Shift -> Gain -> ConvolutionReader(flen bins) [-> DecimateReader(D)].
    With a DecimateReader behind it the chain consumes whole multiples of lcm(flen, 32 Ki) samples; a ragged tail is
    left unconsumed.  Tolerance as above: relative L2 <= 2e-6 (the elementwise stages are bit-exact)."""
    from math import gcd
    rate, shift = 2_400_000, 3.1e5
    blk = flen if D == 1 else flen * 32768 // gcd(flen, 32768)  # (3 x 32 Ki for both decimated cases)
    n = (5 * blk if D == 1 else blk) + blk // 3
    x = rand_i16(19, n)
    H = _lowpass_bins(flen)
    xc = zeros("c64", n)
    orc.convert(xc, x)
    ch = env.ctx.chain(env.hz.FMT_I16, rate).shift(shift).gain(0.5).convolution(env.put(H), decimate=D)
    cons, outn = ch.plan(n)
    assert cons == n // blk * blk and outn == (cons if D == 1 else cons // 32768 * (32768 // D))
    orc.Shifter(rate)(shift, xc[:cons])
    orc.scale(xc[:cons], 0.5)
    conv = zeros("c64", cons)
    assert orc.convolution_reader(conv, xc[:cons], H) == cons
    if D == 1:
        want = conv
    else:
        per = 32768 // D
        want = zeros("c64", outn)
        for b in range(cons // 32768):
            orc.decimate(want[b * per:(b + 1) * per], conv[b * 32768:(b + 1) * 32768], D)
    out = env.zeros("c64", outn)
    assert ch.run(env.put(x), out) == (cons, outn)
    assert _rel_l2(env.get(out), want) < 2e-6
    ch.close()


@pytest.mark.parametrize("ntaps,D", [(1024, 8), (33, 4), (129, 10), (1, 1), (2048, 16)])
def test_chain_fir_decimate_overlap_save(env, orc, ntaps, D):
    """North-star op: y[m] = sum_k h[k] x[D m - k] with history across runs, vs
    the oracle's float64 direct form.  Tolerance (tests/util.py assert_fir_close):
    |err| <= 6e-7 * sum|h| * max|x| per output and relative L2 <= 3e-7."""
    rate, shift = 20_000_000, -2.5e6
    n1, n2 = 40 * D * 100, 13 * D * 100
    x = rand_u8(9, n1 + n2)
    k = np.arange(ntaps) - (ntaps - 1) / 2
    taps = (np.sinc(k / 16) / 16 * np.hamming(ntaps) * np.exp(0.3j * k)).astype(np.complex64)
    xc = zeros("c64", n1 + n2)
    orc.convert(xc, x)
    orc.Shifter(rate)(shift, xc)
    want = zeros("c64", (n1 + n2) // D)
    orc.fir_decimate_f64(want, xc, taps, D)  # zero initial history
    ch = env.ctx.chain(env.hz.FMT_U8, rate).shift(shift).fir_decimate(taps, D)
    out = env.zeros("c64", (n1 + n2) // D)
    dx = env.put(x)
    assert ch.run(dx[:n1], out[:n1 // D]) == (n1, n1 // D)
    assert ch.run(dx[n1:], out[n1 // D:]) == (n2, n2 // D)
    got = env.get(out)
    xmax = float(np.abs(xc).max())
    assert_fir_close(got, want, taps, xmax, (ntaps, D))
    ch.reset()  # forget history and NCO time: first run reproduces
    out2 = env.zeros("c64", n1 // D)
    ch.run(dx[:n1], out2)
    assert_fir_close(env.get(out2), want[:n1 // D], taps, xmax, (ntaps, D, "after reset"))
    ch.close()

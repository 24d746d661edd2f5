"""Pins the CPU oracle against the reference's own known-answer tests
(tests/golden/reference_kats.json) and against independent numpy/mpmath
restatements.  CPU only."""
import math

import numpy as np
import pytest

from util import (FMT, bits_equal, filled, in_epsilon, rand_c64, rand_i16, rand_i8, rand_u8,
                  samples, ulp_diff, zeros)


def c(pair):
    return np.complex64(complex(pair[0], pair[1]))


# ---- converters ------------------------------------------------------------

def test_convert_kats(orc, kats):
    for k in kats["convert"]:
        src = samples(k["src_fmt"], k["src"])
        dst = zeros(k["dst_fmt"], len(src))
        assert orc.convert(dst, src) == len(src), k["cite"]
        exp = samples(k["dst_fmt"], k["dst"])
        if "eps" in k:
            got = dst.view(np.float32).astype(np.float64)
            want = exp.view(np.float32).astype(np.float64)
            if k.get("plus_one"):
                got, want = got + 1, want + 1
            assert in_epsilon(want, got, k["eps"]), k["cite"]
        else:
            assert bits_equal(dst, exp), (k["cite"], dst, exp)


def test_convert_u8_midpoint(orc, kats):
    k = kats["convert_u8_midpoint"]
    src = samples("u8", k["src"])
    dst = zeros("c64", 16)
    orc.convert(dst, src)
    s = dst[0] + dst[1]
    assert in_epsilon(1.0, 1 + s.real, k["eps"]) and in_epsilon(1.0, 1 + s.imag, k["eps"])


def test_convert_subslice_guard(orc, kats):
    k = kats["convert_subslice_guard"]
    src = filled("u8", k["n"], k["fill"])
    dst = zeros("c64", k["n"])
    lo, hi = k["lo"], k["hi"]
    assert orc.convert(dst[lo:hi], src[lo:hi]) == hi - lo
    assert np.all(dst[:lo] == 0) and np.all(dst[hi:] == 0)
    assert in_epsilon(1.0, dst[lo:hi].real, k["eps"]) and in_epsilon(1.0, dst[lo:hi].imag, k["eps"])


def test_convert_matches_numpy_restatement(orc):
    """Second, independent restatement (numpy float32 IEEE) of every converter,
    exhaustive over the byte / int16 domains."""
    allb = np.arange(65536, dtype=np.uint32)
    u8 = np.stack([(allb & 255), (allb >> 8)], 1).astype(np.uint8)
    i8 = u8.view(np.int8)
    i16 = np.stack([allb.astype(np.uint16).view(np.int16)] * 2, 1).copy()
    i16[:, 1] = i16[::-1, 0]
    f = np.float32

    out = zeros("c64", 65536)
    orc.convert(out, u8)
    want = ((u8.astype(f) - f(127.5)) / f(127.5)).astype(f)
    assert bits_equal(out.view(f).reshape(-1, 2), want)

    orc.convert(out, i8)
    assert bits_equal(out.view(f).reshape(-1, 2), (i8.astype(f) / f(128)).astype(f))

    orc.convert(out, i16)
    assert bits_equal(out.view(f).reshape(-1, 2), (i16.astype(f) / f(32767)).astype(f))

    o8 = zeros("i8", 65536)
    orc.convert(o8, u8)
    assert bits_equal(o8, (u8.astype(np.int16) - 128).astype(np.int8))
    o16 = zeros("i16", 65536)
    orc.convert(o16, u8)
    assert bits_equal(o16, ((u8.astype(np.int32) << 8) - 32768).astype(np.int16))
    ou8 = zeros("u8", 65536)
    orc.convert(ou8, i8)
    assert bits_equal(ou8, (i8.astype(np.int16) + 128).astype(np.uint8))
    orc.convert(o16, i8)
    assert bits_equal(o16, (i8.astype(np.int16) << 8).astype(np.int16))
    orc.convert(ou8, i16)
    assert bits_equal(ou8, ((i16.astype(np.int32) + 32768) >> 8).astype(np.uint8))
    orc.convert(o8, i16)
    assert bits_equal(o8, (i16 >> 8).astype(np.int8))

    x = rand_c64(11, 65536)
    xf = x.view(f).reshape(-1, 2)
    orc.convert(ou8, x)
    assert bits_equal(ou8, ((xf * f(127.5)).astype(f) + f(127.5)).astype(f).astype(np.int32).astype(np.uint8))
    orc.convert(o16, x)
    assert bits_equal(o16, (xf * f(32767)).astype(f).astype(np.int32).astype(np.int16))
    orc.convert(o8, x)
    assert bits_equal(o8, (xf * f(127)).astype(f).astype(np.int32).astype(np.int8))


def test_convert_out_of_range_follows_amd64(orc):
    """c64 -> int with |x*scale| beyond int32: CVTTSS2SL gives 0x80000000, whose
    low byte / word is 0; in-range negatives wrap through int32 truncation."""
    x = samples("c64", [[1e12, -1e12], [float("nan"), float("inf")], [-3.0, 3.0]])
    o8, ou8, o16 = zeros("i8", 3), zeros("u8", 3), zeros("i16", 3)
    orc.convert(o8, x), orc.convert(ou8, x), orc.convert(o16, x)
    assert o8[:2].tolist() == [[0, 0], [0, 0]] and ou8[:2].tolist() == [[0, 0], [0, 0]]
    assert o16[:2].tolist() == [[0, 0], [0, 0]]
    # -3*127.5+127.5 = -255 -> int32 -255 -> uint8 1 ; 3*127.5+127.5 = 510 -> 254
    assert ou8[2].tolist() == [1, 254]
    # -3*127 = -381 -> int8(-381) = -125 ; 381 -> 125
    assert o8[2].tolist() == [-125, 125]


def test_convert_errors(orc):
    src = zeros("u8", 8)
    assert orc.convert(zeros("c64", 4), src) == orc.ERR_DST_TOO_SMALL  # iq_u8.go:104-106
    assert orc.convert(zeros("c64", 4), src, dst_fmt=9) == orc.ERR_FORMAT_UNKNOWN
    # same format -> CopySamples copies min(len) (copy.go:31-52)
    assert orc.convert(zeros("u8", 4), src) == 4
    assert orc.convert(zeros("u8", 16), src) == 8


def test_i16_shift_lsb_to_msb(orc):
    a = samples("i16", [[1, 0x0FFF], [-1, 0x0800]])
    orc.i16_shift_lsb_to_msb(a, 12)  # iq_i16.go:103-111
    assert a.tolist() == [[16, -16], [-16, -32768]]


# ---- c64 vector ops ---------------------------------------------------------

def test_scale_multiply_add_kats(orc, kats):
    k = kats["scale"]
    b = filled("c64", k["n"], k["fill"])
    orc.scale(b, k["r"])
    assert np.all(b == c(k["value"]))
    k = kats["multiply"]
    b = filled("c64", k["n"], k["fill"])
    orc.rotate(b, c(k["m"]))
    assert np.all(b == c(k["value"]))
    k = kats["add"]
    a, b = filled("c64", k["n"], k["a"]), filled("c64", k["n"], k["b"])
    assert orc.add(a, b, a) == 0
    assert np.all(a == c(k["value"])) and np.all(b == c(k["b"]))
    for k in kats["simd_add"]:
        a, b = filled("c64", k["n"], k["a"]), filled("c64", k["n"], k["b"])
        out = a if k.get("in_place") else zeros("c64", k["n"])
        assert orc.add(a, b, out) == 0
        assert np.all(out == c(k["value"]))
    k = kats["simd_add_subslice_guard"]
    a, b, o = filled("c64", k["n"], k["a"]), filled("c64", k["n"], k["b"]), zeros("c64", k["n"])
    orc.add(a[k["lo"]:k["hi"]], b[k["lo"]:k["hi"]], o[k["lo"]:k["hi"]])
    assert np.all(o[:k["lo"]] == 0) and np.all(o[k["hi"]:] == 0)
    assert np.all(o[k["lo"]:k["hi"]] == c(k["value"]))
    k = kats["simd_scale_subslice_guard"]
    b = filled("c64", k["n"], k["fill"])
    orc.scale(b[:k["hi"]], k["r"])
    assert np.all(b[:k["hi"]] == c(k["value"])) and np.all(b[k["hi"]:] == c(k["fill"]))
    k = kats["simd_rotate"]
    b = filled("c64", k["n"], k["fill"])
    orc.rotate(b, c(k["m"]))
    assert np.all(b == c(k["value"]))
    assert orc.add(zeros("c64", 3), zeros("c64", 4), zeros("c64", 3)) == orc.ERR_LENGTH


def test_rotate_is_f64_widened(orc):
    """Go computes complex64 products in float64 and narrows once; a float32
    evaluation differs on a measurable fraction of inputs."""
    x = rand_c64(5, 200000)
    m = np.complex64(0.70710678 + 0.25881904j)
    got = x.copy()
    orc.rotate(got, m)
    xd = x.astype(np.complex128)
    md = np.complex128(m)
    wr = (xd.real * md.real - xd.imag * md.imag).astype(np.float32)
    wi = (xd.real * md.imag + xd.imag * md.real).astype(np.float32)
    assert bits_equal(got.real, wr) and bits_equal(got.imag, wi)
    f = np.float32
    nr = (x.real * f(m.real)).astype(f) - (x.imag * f(m.imag)).astype(f)
    assert ulp_diff(got.real, nr.astype(f)).max() >= 1  # the two really do differ


def test_stream_add(orc, kats):
    k = kats["stream_add_c64"]
    bufs = [filled("c64", k["n"], k["fill"]) for _ in range(k["k"])]
    out = filled("c64", k["n"], [7, 7])
    orc.sum_(out, bufs)
    assert np.all(out == c(k["value"]))
    for name, fmt in (("stream_add_i8", "i8"), ("stream_add_i16", "i16")):
        k = kats[name]
        bufs = [filled(fmt, k["n"], k["fill"]) for _ in range(k["k"])]
        out = zeros(fmt, k["n"])
        orc.sum_(out, bufs)
        assert np.all(out == np.asarray(k["value"]))
    # integer adds wrap (stream/add.go:95-113)
    out = zeros("i8", 4)
    orc.sum_(out, [filled("i8", 4, [100, -100])] * 2)
    assert out[0].tolist() == [-56, 56]
    # -0 inputs give +0 because the sum starts from +0 (stream/add.go:165-167)
    out = zeros("c64", 2)
    orc.sum_(out, [samples("c64", [[-0.0, -0.0], [-0.0, 1.0]])])
    assert not np.signbit(out[0].real) and not np.signbit(out[0].imag)


# ---- lookup tables ----------------------------------------------------------

def test_lut_identity_and_apply(orc):
    t = orc.lut_identity()
    src = rand_u8(3, 5000)
    dst = zeros("u8", 5000)
    assert orc.lut_apply(dst, t, src) == 5000
    assert bits_equal(dst, src)
    tc = zeros("c64", 65536)
    orc.convert(tc, t)
    d2, want = zeros("c64", 5000), zeros("c64", 5000)
    orc.lut_apply(d2, tc, src)
    orc.convert(want, src)
    assert bits_equal(d2, want)
    assert orc.lut_apply(zeros("u8", 10), t, src) == orc.ERR_DST_TOO_SMALL


def _counter_u8(n):
    i = np.arange(n, dtype=np.uint32) & 0xFFFF
    return np.stack([i & 0xFF, (i & 0xFF00) >> 8], 1).astype(np.uint8)


def test_rotate_lut_u8_kat(orc, kats):
    k = kats["rotate_lut_u8"]
    m = c(k["m"])
    vals = _counter_u8(k["n"])
    cbuf, ref = zeros("c64", k["n"]), zeros("u8", k["n"])
    orc.convert(cbuf, vals)
    orc.rotate(cbuf, m)
    orc.convert(ref, cbuf)
    tab = orc.rotate_table_u8(m)
    buf = vals.copy()
    orc.rotate_u8_apply(tab, buf)
    assert bits_equal(buf, ref)


def test_rotate_lut_u8_alias_quirk(orc):
    """index = I*255 + Q (stream/multiply.go:106-108): (r, 255) aliases (r+1, 0)
    and the later write wins (fill order :157-163)."""
    tab = orc.rotate_table_u8(np.complex64(1))  # multiply by 1 -> c64 round trip of the key
    a = samples("u8", [[3, 255], [4, 0]])
    orc.rotate_u8_apply(tab, a)
    assert a[0].tolist() == a[1].tolist() == [4, 0]


def test_rotate_lut_i8_kat(orc, kats):
    k = kats["rotate_lut_i8"]
    m = c(k["m"])
    i = np.arange(k["n"], dtype=np.int64) & 0xFFFF
    vals = np.stack([(i & 0xFF), ((i & 0xFF00) >> 8) - 127], 1).astype(np.int8)
    cbuf, ref = zeros("c64", k["n"]), zeros("i8", k["n"])
    orc.convert(cbuf, vals)
    orc.rotate(cbuf, m)
    orc.convert(ref, cbuf)
    tab = orc.rotate_table_i8(m)
    out = zeros("i8", k["n"])
    orc.lut_apply(out, tab, vals.view(np.uint8))
    assert bits_equal(out, ref)


def test_rotate_cw_kat(orc, kats):
    k = kats["rotate_cw"]
    p0 = orc.cw(k["n"], k["freq"], k["rate"], 0.0)
    p90 = orc.cw(k["n"], k["freq"], k["rate"], math.pi / 2)
    orc.rotate(p90, c(k["m"]))
    assert in_epsilon(1 + p0.real, 1 + p90.real, k["eps"])
    assert in_epsilon(1 + p0.imag, 1 + p90.imag, k["eps"])


# ---- NCO / Shift ------------------------------------------------------------

def test_go_sincos_table_and_accuracy(orc):
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 1400
    v = int(mp.floor(4 / mp.pi * mp.mpf(2) ** (64 * 19)))
    assert [(v >> (64 * (19 - i))) & (2 ** 64 - 1) for i in range(20)] == orc.go_mpi4()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-10, 10, 200), rng.uniform(0, 5.3e8, 200),
                         rng.uniform(5.4e8, 1e12, 200)])
    s, co = orc.go_sincos(xs)
    worst = 0.0
    for x, si, ci in zip(xs, s, co):
        X = mp.mpf(float(x))
        worst = max(worst, float(abs(mp.sin(X) - mp.mpf(float(si)))),
                    float(abs(mp.cos(X) - mp.mpf(float(ci)))))
    assert worst < 2.5 * 2.0 ** -53  # absolute; Cephes + 61-bit Payne-Hanek
    # against libm after the float32 narrowing Shift applies: <= 1 ULP
    assert ulp_diff(s.astype(np.float32), np.sin(xs).astype(np.float32)).max() <= 1
    assert ulp_diff(co.astype(np.float32), np.cos(xs).astype(np.float32)).max() <= 1
    s0, c0 = orc.go_sincos(np.array([0.0, -0.0]))
    assert c0.tolist() == [1.0, 1.0] and np.signbit(s0).tolist() == [False, True]


def test_go_sincos_against_the_go_math_packages_own_vectors(orc):
    """math.Sincos is not under the reference tree (Go's standard library: SURVEY.md 8c (ii)), but the library that
    holds it publishes the vectors it is tested with: src/math/all_test.go's vf / sin / cos, held with `veryclose`
    (relative 4e-16).  The oracle's restatement passes Go's own test; mod 2 pi, with Go's tenfold arguments too
    (TestSincos also runs vf[i] * 10 against the same function: here against a 40-digit evaluation)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "go_math_sincos_kats.json")) as f:
        k = json.load(f)
    vf = np.array(k["vf"], np.float64)
    s, c = orc.go_sincos(vf)
    tol = k["tolerance_relative"]
    for i in range(len(vf)):
        ws, wc = float(k["sin"][i]), float(k["cos"][i])
        assert abs(s[i] - ws) <= tol * abs(ws), (i, s[i], ws)
        assert abs(c[i] - wc) <= tol * abs(wc), (i, c[i], wc)
    try:
        import mpmath as mp
    except ImportError:
        return
    mp.mp.dps = 40
    s10, c10 = orc.go_sincos(vf * 10)
    for i in range(len(vf)):
        ws, wc = mp.sin(mp.mpf(float(vf[i] * 10))), mp.cos(mp.mpf(float(vf[i] * 10)))
        assert abs(mp.mpf(float(s10[i])) - ws) <= tol * abs(ws) and abs(mp.mpf(float(c10[i])) - wc) <= tol * abs(wc), i


def test_sincos_narrow_never_accepts_a_pair_that_differs_from_math_sincos(orc):
    """The straight path of shift_exact_kernel (csrc/hz_device.h sincos_narrow, restated in the oracle): wherever its
    check on the float64 bits accepts, the float32 pair IS complex64(math.Sincos(x)) -- over random phases of every
    size the path takes, the phases of the benchmark (multiples of pi/4 up to the phase's own rounding: half the
    components are rounding noise), and the neighbours of multiples of pi/4 where Go's reduction is least accurate."""
    rng = np.random.default_rng(5)
    n = 3_000_000
    sets = [rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-17, 8.7, n)]
    k = rng.integers(0, 680_000_000, n).astype(np.float64)
    near = k * (np.pi / 4)
    for _ in range(3):
        near = np.where(rng.integers(0, 2, n) == 0, np.nextafter(near, np.inf), np.nextafter(near, -np.inf))
    sets.append(np.where(rng.integers(0, 2, n) == 0, near, -near))
    ts = np.arange(n, dtype=np.float64) * (1.0 / 20e6) + 3.0
    sets.append(((np.pi * 2) * 2.5e6) * ts)
    sets.append(np.array([0.0, -0.0, 2.0 ** -60, -(2.0 ** -60), np.nextafter(536870912.0, 0)]))
    for xs in sets:
        xs = xs[(np.abs(xs) < 536870912.0) & ((np.abs(xs) >= 2.0 ** -60) | (xs == 0))]
        accepted, wrong = orc.sincos_narrow_check(xs)
        assert wrong == 0
        assert accepted >= xs.size * (1 - 2e-5) - 2  # the queue stays short: 2^-20 of the factors


def test_shift_roundtrip_kat(orc, kats):
    k = kats["shift_roundtrip"]
    cw = orc.cw(k["n"], k["freq"], k["rate"], 0.0)
    buf = cw.copy()
    hi, lo = orc.Shifter(k["rate"]), orc.Shifter(k["rate"])
    hi(k["shift"], buf)
    lo(-k["shift"], buf)
    assert in_epsilon(1 + cw.real, 1 + buf.real, k["eps"])
    assert in_epsilon(1 + cw.imag, 1 + buf.imag, k["eps"])


def test_shift_matches_python_restatement_and_libm(orc):
    """Pure-Python restatement of stream/shifter.go:73-84 (small case) and the
    libm-sincos variant: identical time sequence, <= 1 ULP apart."""
    n, fs, f = 3000, 20_000_000, 2.5e6
    x = rand_c64(2, n)
    a, b = x.copy(), x.copy()
    sa, sb = orc.Shifter(fs), orc.Shifter(fs, use_libm=True)
    for lo in range(0, n, 1000):  # state persists across buffers
        sa(f, a[lo:lo + 1000])
        sb(f, b[lo:lo + 1000])
    assert sa.ts.value == sb.ts.value
    assert ulp_diff(a, b).max() <= 1
    ts, inc, tau = 0.0, 1.0 / fs, math.pi * 2
    want = np.empty(n, np.complex64)
    for j in range(n):
        ts += inc
        if ts > tau:
            ts -= tau
        ph = tau * f * ts
        rot = np.complex64(complex(math.cos(ph), math.sin(ph)))
        xr, xi = float(x[j].real), float(x[j].imag)
        rr, ri = float(rot.real), float(rot.imag)
        want[j] = complex(np.float32(xr * rr - xi * ri), np.float32(xr * ri + xi * rr))
    assert ts == sa.ts.value
    assert ulp_diff(a, want).max() <= 1


def test_shift_time_wraps_at_two_pi(orc):
    sh = orc.Shifter(1000)  # wraps every ~6283 samples (stream/shifter.go:77-79)
    ts = sh.ts_sequence(20000)
    assert ts.max() <= 2 * math.pi + 1e-3 and (np.diff(ts) < 0).sum() == 3


# ---- decimate / downsample --------------------------------------------------

def test_decimate_kats(orc, kats):
    k = kats["decimate_count"]
    for fmt in k["formats"]:
        assert orc.decimate(zeros(fmt, k["n"]), zeros(fmt, k["n"]), k["factor"]) == k["count"]
    k = kats["decimate_skippy"]
    i = (np.arange(k["n"]) % 10).astype(np.uint8)
    src = np.stack([i, i], 1)
    dst = filled("u8", k["count"], [9, 9])
    assert orc.decimate(dst, src, k["factor"]) == k["count"]
    assert np.all(dst == 0)
    for e in kats["decimate_errors"]:
        rc = orc.decimate(zeros(e["to_fmt"], e["to_len"]), zeros(e["from_fmt"], e["n"]), e["factor"])
        assert rc == {"format_mismatch": orc.ERR_FORMAT_MISMATCH,
                      "dst_too_small": orc.ERR_DST_TOO_SMALL}[e["err"]]
    # i8 is not handled by the reference's type switch (stream/decimate.go:85-97)
    assert orc.decimate(zeros("i8", 8), zeros("i8", 8), 2) == orc.ERR_FORMAT_UNKNOWN
    x = rand_c64(1, 1001)
    d = zeros("c64", 143)
    assert orc.decimate(d, x, 7) == 143 and bits_equal(d, x[::7][:143])


def test_downsample_kat_and_restatement(orc, kats):
    k = kats["downsample_calc"]
    e = (np.arange(k["n"]) % 4).astype(np.float32)
    src = (e + 1j * e).astype(np.complex64)
    dst = zeros("c64", k["n"])
    assert orc.downsample(dst, src, k["factor"]) == k["count"]
    assert np.all(dst[:k["count"]] == c(k["value"]))
    # i16 input, factor 8: sequential float32 accumulation from +0, IEEE divide
    x = rand_i16(4, 8 * 500 + 3)
    dst = zeros("c64", 500)
    assert orc.downsample(dst, x, 8) == 500
    f = np.float32
    w = (x[:4000].astype(f) / f(32767)).astype(f).reshape(500, 8, 2)
    acc = np.zeros((500, 2), f)
    for j in range(8):
        acc = (acc + w[:, j, :]).astype(f)
    assert bits_equal(dst.view(f).reshape(-1, 2), (acc / f(8)).astype(f))
    assert orc.downsample(zeros("u8", 8), x, 8) == orc.ERR_FORMAT_MISMATCH
    assert orc.downsample(zeros("c64", 8), x, 8) == orc.ERR_DST_TOO_SMALL
    assert orc.downsample(zeros("c64", 8), zeros("i8", 8), 2) == orc.ERR_FORMAT_UNKNOWN


# ---- FFT conformance (testutils/fft.go) -------------------------------------

def test_fft_conformance(orc, kats):
    k = kats["fft_forward_bins"]
    for freq, idx in k["cases"]:
        out = zeros("c64", k["n"])
        assert orc.fft(orc.cw(k["n"], freq, k["rate"], 0.0), out, True) == 0
        assert int(np.argmax(np.abs(out.astype(np.complex128)))) == idx
    k = kats["fft_backward_roundtrip"]
    for b in k["bins"]:
        fr, iq = zeros("c64", k["n"]), zeros("c64", k["n"])
        fr[b] = 1 + 1j
        assert orc.fft(fr, iq, False) == 0
        fr[b] = 0
        assert orc.fft(iq, fr, True) == 0
        assert int(np.argmax(np.abs(fr))) == b
        assert abs(fr[b] - k["n"] * (1 + 1j)) < 1e-2  # backward is unnormalised
    for a, b, _ in kats["fft_mismatch"]["cases"]:
        assert orc.fft(zeros("c64", a), zeros("c64", b), True) == orc.ERR_DST_TOO_SMALL


def test_fft_matches_numpy(orc):
    x = rand_c64(3, 4096)
    out = zeros("c64", 4096)
    orc.fft(x, out, True)
    want = np.fft.fft(x.astype(np.complex128))
    assert np.linalg.norm(out - want) / np.linalg.norm(want) < 1e-7
    orc.fft(x, out, False)
    want = np.fft.ifft(x.astype(np.complex128)) * 4096
    assert np.linalg.norm(out - want) / np.linalg.norm(want) < 1e-7


def test_convolution_reader_is_block_circular(orc):
    n, flen = 4096, 1024
    x = rand_c64(3, n + 100)  # trailing partial block is never produced
    h = np.zeros(flen, np.complex64)
    h[:5] = [0.5, 0.25, 0.125, 0.0625, 0.03125]
    H = np.fft.fft(h.astype(np.complex128)).astype(np.complex64)
    out = zeros("c64", n + 100)
    assert orc.convolution_reader(out, x, H) == n
    for b in range(n // flen):
        blk = x[b * flen:(b + 1) * flen].astype(np.complex128)
        want = np.fft.ifft(np.fft.fft(blk) * H.astype(np.complex128)) * flen
        got = out[b * flen:(b + 1) * flen]
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-6


def test_fir_decimate_truth(orc):
    x = rand_c64(9, 4096)
    taps = rand_c64(10, 33)
    out = zeros("c64", 512)
    orc.fir_decimate_f64(out, x, taps, 8)
    full = np.convolve(x.astype(np.complex128), taps.astype(np.complex128))[:4096:8]
    assert np.allclose(out, full.astype(np.complex64), rtol=0, atol=1e-5)


# ---- beamform ---------------------------------------------------------------

def _phase_conj(z):
    return math.atan2(-float(z.imag), float(z.real))


def test_beamform_angle_kats(orc, kats):
    for k in kats["beamform_angles"]:
        rot = orc.beamform_angles(k["freq"], k["angle"], k["distances"])
        _check_angles(k, rot)
    for k in kats["beamform_angles_2d"]:
        rot = orc.beamform_angles_2d(k["freq"], k["angle"], k["center"], k["antennas"])
        _check_angles(k, rot)
    assert orc.beamform_angles(900e6, 0, []) is None  # stream/beamform_test.go:64-79
    assert orc.beamform_angles_2d(900e6, 0, [0, 10], []) is None


def _check_angles(k, rot):
    eps = k["eps"]
    if "expect" in k:
        for r, e in zip(rot, k["expect"]):
            assert in_epsilon(e[0], r.real, eps) and in_epsilon(1 + e[1], 1 + r.imag, eps), k["cite"]
    if "expect_real" in k:
        for r, e in zip(rot, k["expect_real"]):
            assert in_epsilon(e, r.real, eps), k["cite"]
    for key, fn in (("phase_conj", lambda p: p), ("phase_conj_plus_one", lambda p: 1 + p)):
        if key in k:
            for r, e in zip(rot, k[key]):
                if e is not None:
                    assert in_epsilon(e, fn(_phase_conj(r)), eps), (k["cite"], e, _phase_conj(r))
    if "phase_conj_plus_2pi_deg" in k:
        for r, e in zip(rot, k["phase_conj_plus_2pi_deg"]):
            if e is not None:
                assert in_epsilon(e * math.pi / 180, 2 * math.pi + _phase_conj(r), eps), k["cite"]


def test_beamform_data_path(orc):
    """out = ((0 + w0*x0) + w1*x1) + ...; a weight of exactly 1 skips the multiply
    (stream/multiply.go:59-62)."""
    n = 1000
    ch = [rand_c64(20 + i, n) for i in range(4)]
    w = orc.beamform_angles(433e6, 30.0, [0.0, 0.1, 0.2, 0.3])
    assert w[0] == 1
    out = zeros("c64", n)
    orc.beamform(out, ch, w)
    acc = np.zeros(n, np.complex64)
    for x, wk in zip(ch, w):
        y = x.copy()
        if wk != 1:
            orc.rotate(y, wk)
        t = zeros("c64", n)
        orc.add(acc, y, t)
        acc = t
    assert bits_equal(out, acc)
    assert all(bits_equal(a, b) for a, b in zip(ch, [rand_c64(20 + i, n) for i in range(4)]))


def test_graft_and_wire_restatements(orc):
    """The reference holds no tests for rtl/kerberos/internal or for the foreign
    byte order; the restatements are checked against their definitions instead."""
    from util import rand_i16
    x = rand_c64(3, 9)
    y = x.copy()
    orc.fftshift_scale(y, 4.0)
    assert bits_equal(y[:4], (x[4:8] / np.float32(4)).astype(np.complex64))
    assert bits_equal(y[4:8], (x[:4] / np.float32(4)).astype(np.complex64))
    assert bits_equal(y[8:], x[8:])  # reader.go:58: half = len/2, an odd tail is untouched
    for buf, comp in ((rand_i16(1, 33), np.int16), (rand_c64(2, 33), np.float32)):
        want = buf.view(comp).byteswap().view(np.uint8)
        orc.byteswap(buf)
        assert np.array_equal(buf.view(np.uint8), want)
    n = 1024
    x = rand_c64(4, n)
    out = zeros("c64", n)
    assert orc.graft(out, [x]) == 0  # one band: fftshift in frequency = (-1)^k in time
    want = x * np.where(np.arange(n) % 2 == 0, 1, -1).astype(np.float32)
    assert np.linalg.norm(out - want) / np.linalg.norm(want) < 1e-6
    t = np.arange(n)
    bands = [zeros("c64", n), np.exp(2j * np.pi * 5 * t / n).astype(np.complex64)]
    out = zeros("c64", 2 * n)
    assert orc.graft(out, bands) == 0
    assert int(np.argmax(np.abs(np.fft.fft(out.astype(np.complex128))))) == n + 5 + n // 2

"""Reader-level tests, written after the reference's own (`stream/*_test.go`,
`testutils/reader.go`, `mock/mock_test.go`): known buffers go in through an
in-memory source, the GPU-backed operator's output is read with ReadFull and
compared with the reference's expected values (KATs) and with the oracle.
HOST contexts: what a cgo caller with Go slices would exercise."""
import importlib
import math

import numpy as np
import pytest

from util import bits_equal, filled, in_epsilon, rand_c64, rand_i16, rand_u8, zeros

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hz():
    return importlib.import_module("go-sdr_amd")


@pytest.fixture(scope="module")
def S(hz):
    return importlib.import_module("go-sdr_amd.stream")


@pytest.fixture(scope="module")
def st(hz, S):
    ctx = hz.Context(0, hz.MEM_HOST)
    yield S.Stream(ctx)
    ctx.close()


def c(pair):
    return np.complex64(complex(pair[0], pair[1]))


def check_reader_contract(hz, S, reader):
    """testutils.TestReader (testutils/reader.go:66-97): a buffer of the wrong
    format must be refused with ErrSampleFormatMismatch."""
    wrong = hz.FMT_I16 if reader.sample_format() != hz.FMT_I16 else hz.FMT_U8
    with pytest.raises(hz.ErrSampleFormatMismatch):
        reader.read(hz.make_samples(wrong, 16))


def test_mock_u8_source_through_convert_reader(hz, S, st, orc):
    """BASELINE config 1: u8 -> c64 on a 1 Mi-sample synthetic buffer through an
    in-memory source and ConvertReader's 32 Ki blocks (mock/mock.go:149-163,
    stream/convert.go:37-51)."""
    n = 1 << 20
    x = rand_u8(1, n)
    r = st.convert_reader(S.BufferReader(x, 2_400_000, max_read=10_000), hz.FMT_C64)
    assert r.sample_format() == hz.FMT_C64 and r.sample_rate() == 2_400_000
    check_reader_contract(hz, S, r)
    out = zeros("c64", n)
    assert S.read_full(r, out) == n
    want = zeros("c64", n)
    orc.convert(want, x)
    assert bits_equal(out, want)
    with pytest.raises(S.EOF):
        r.read(zeros("c64", 16))


def test_convert_reader_drops_partial_block(hz, S, st):
    """A trailing partial 32 Ki block never comes out (ReadFull fails first)."""
    x = rand_u8(2, 32768 + 100)
    r = st.convert_reader(S.BufferReader(x, 1000), hz.FMT_C64)
    out = zeros("c64", 40000)
    with pytest.raises(S.ErrUnexpectedEOF) as e:
        S.read_full(r, out)
    assert e.value.n == 32768


def test_convert_writer(hz, S, st, orc):
    """stream/convert_test.go:81-108 (TestConvertWriterBufferU8C64): 8000 c64 samples
    written through ConvertWriter arrive as 8000 u8 samples at the underlying writer;
    values against the oracle, and a length that is not a multiple of the 32 Ki chunk."""
    sink = S.BufferWriter(hz.FMT_U8, 1337)
    w = st.convert_writer(sink, hz.FMT_C64)
    assert w.sample_format() == hz.FMT_C64 and w.sample_rate() == 1337
    assert w.write(zeros("c64", 1000 * 8)) == 1000 * 8
    assert hz.length(sink.samples()) == 1000 * 8
    x = (rand_c64(12, 100_000) * np.float32(0.99)).astype(np.complex64)
    sink2 = S.BufferWriter(hz.FMT_I16, 0)
    assert st.convert_writer(sink2, hz.FMT_C64).write(x) == 100_000
    want = zeros("i16", 100_000)
    orc.convert(want, x)
    assert bits_equal(sink2.samples(), want)
    assert [hz.length(c) for c in sink2.chunks] == [32768, 32768, 32768, 100_000 - 3 * 32768]
    with pytest.raises(hz.ErrSampleFormatMismatch):  # stream/convert.go:86-88
        w.write(zeros("u8", 16))


def test_gain_reader(hz, S, st, kats):
    """stream/gain_test.go:52-80."""
    src = filled("c64", 1024, [10, 10])
    g = st.gain(S.BufferReader(src, 10_000), 0.5)
    check_reader_contract(hz, S, g)
    out = zeros("c64", 1024)
    assert S.read_full(g, out) == 1024
    assert np.all(out == np.complex64(5 + 5j))
    with pytest.raises(hz.ErrSampleFormatUnknown):  # stream/gain.go:45-46
        st.gain(S.BufferReader(zeros("i16", 8), 1), 2.0).read(zeros("i16", 8))


def test_add_reader_kats(hz, S, st, kats):
    """stream/add_test.go:34-135."""
    k = kats["stream_add_c64"]
    rs = [S.BufferReader(filled("c64", k["n"], k["fill"]), 10_000) for _ in range(k["k"])]
    mix = st.add(*rs)
    out = zeros("c64", k["n"])
    assert S.read_full(mix, out) == k["n"]
    assert np.all(out == c(k["value"]))
    for name, fmt in (("stream_add_i8", "i8"), ("stream_add_i16", "i16")):
        k = kats[name]
        rs = [S.BufferReader(filled(fmt, k["n"], k["fill"]), 10_000) for _ in range(k["k"])]
        out = zeros(fmt, k["n"])
        assert S.read_full(st.add(*rs), out) == k["n"]
        assert np.all(out == np.asarray(k["value"]))
    with pytest.raises(hz.ErrSampleFormatUnknown):  # u8 unsupported: stream/add.go:55-61
        st.add(S.BufferReader(zeros("u8", 4), 1), S.BufferReader(zeros("u8", 4), 1))
    with pytest.raises(hz.HzsdrError):
        st.add()
    one = S.BufferReader(zeros("c64", 4), 1)
    assert st.add(one) is one  # stream/add.go:46-47


def test_multiply_reader_cw(hz, S, st, orc, kats):
    """stream/multiply_test.go:36-69: CW at phase 90 deg times -i is CW at phase 0."""
    k = kats["rotate_cw"]
    p0 = orc.cw(k["n"], k["freq"], k["rate"], 0.0)
    p90 = orc.cw(k["n"], k["freq"], k["rate"], math.pi / 2)
    r = st.multiply(S.BufferReader(p90, k["rate"]), c(k["m"]))
    check_reader_contract(hz, S, r)
    buf = zeros("c64", k["n"])
    assert S.read_full(r, buf) == k["n"]
    assert in_epsilon(1 + p0.real, 1 + buf.real, k["eps"]) and in_epsilon(1 + p0.imag, 1 + buf.imag, k["eps"])


def test_multiply_reader_u8_i8_tables(hz, S, st, orc, kats):
    """stream/multiply_test.go:71-112, 189-230: the LUT readers equal
    ConvertBuffer -> Multiply -> ConvertBuffer exactly."""
    k = kats["rotate_lut_u8"]
    i = np.arange(k["n"], dtype=np.uint32) & 0xFFFF
    vals = np.stack([i & 0xFF, (i & 0xFF00) >> 8], 1).astype(np.uint8)
    cb, ref = zeros("c64", k["n"]), zeros("u8", k["n"])
    st.ctx.convert(cb, vals)
    st.ctx.rotate(cb, c(k["m"]))
    st.ctx.convert(ref, cb)
    r = st.multiply(S.BufferReader(vals, 1_800_000), c(k["m"]))
    buf = zeros("u8", k["n"])
    assert S.read_full(r, buf) == k["n"]
    assert bits_equal(buf, ref)
    k = kats["rotate_lut_i8"]
    vals8 = np.stack([(i & 0xFF), ((i.astype(np.int64) & 0xFF00) >> 8) - 127], 1).astype(np.int8)
    cb, ref = zeros("c64", k["n"]), zeros("i8", k["n"])
    st.ctx.convert(cb, vals8)
    st.ctx.rotate(cb, c(k["m"]))
    st.ctx.convert(ref, cb)
    r = st.multiply(S.BufferReader(vals8, 1_800_000), c(k["m"]))
    buf = zeros("i8", k["n"])
    assert S.read_full(r, buf) == k["n"]
    assert bits_equal(buf, ref)
    r.set_multiplier(1)  # SetMultiplier rebuilds the table
    with pytest.raises(hz.ErrSampleFormatUnknown):
        st.multiply(S.BufferReader(zeros("i16", 4), 1), 1j)


def test_shifter_roundtrip(hz, S, st, orc, kats):
    """stream/shifter_test.go:35-72."""
    k = kats["shift_roundtrip"]
    cw = orc.cw(k["n"], k["freq"], k["rate"], 0.0)
    hi = st.shift_reader(S.BufferReader(cw, k["rate"], max_read=7000), k["shift"])
    lo = st.shift_reader(hi, -k["shift"])
    buf = zeros("c64", k["n"])
    assert S.read_full(lo, buf) == k["n"]
    assert in_epsilon(1 + cw.real, 1 + buf.real, k["eps"]) and in_epsilon(1 + cw.imag, 1 + buf.imag, k["eps"])
    want = cw.copy()
    a, b = orc.Shifter(k["rate"]), orc.Shifter(k["rate"])
    a(k["shift"], want)
    b(-k["shift"], want)
    assert bits_equal(buf, want)  # state carried across the 7000-sample reads
    with pytest.raises(hz.ErrSampleFormatUnknown):  # stream/shifter.go:90-95
        st.shift_reader(S.BufferReader(zeros("u8", 4), 1), 1.0)


def test_decimate_reader(hz, S, st, kats):
    """stream/decimate_test.go:95-166."""
    r = st.decimate_reader(S.BufferReader(zeros("u8", 16), 10000), 10)
    assert r.sample_rate() == 1000 and r.sample_format() == hz.FMT_U8
    k = kats["decimate_skippy"]
    z = (np.arange(k["n"]) % 10).astype(np.uint8)
    r = st.decimate_reader(S.BufferReader(np.stack([z, z], 1), 10000), k["factor"])
    check_reader_contract(hz, S, r)
    out = filled("u8", k["count"], [9, 9])
    assert S.read_full(r, out) == k["count"]
    assert np.all(out == 0)


def test_downsample_reader(hz, S, st, kats):
    """stream/downsample_test.go:34-93."""
    k = kats["downsample_calc"]
    e = (np.arange(k["n"]) % 4).astype(np.float32)
    r = st.downsample_reader(S.BufferReader((e + 1j * e).astype(np.complex64), 10000), k["factor"])
    assert r.sample_rate() == 2500 and r.sample_format() == hz.FMT_C64
    out = zeros("c64", k["n"])
    with pytest.raises(S.ErrUnexpectedEOF) as ex:  # the reference asserts an error here too
        S.read_full(r, out)
    assert ex.value.n == k["count"]
    assert np.all(out[:k["count"]] == c(k["value"]))


def test_convolution_reader(hz, S, st, orc):
    """stream/convolution.go:36-82 (no reference test exists): vs the oracle's
    float64-FFT restatement, relative L2 <= 2e-6."""
    flen, nblk = 1024, 6
    x = rand_c64(3, flen * nblk)
    t = np.arange(flen) - (flen - 1) / 2
    H = np.fft.fft((np.sinc(t / 8) / 8 * np.hamming(flen)) / flen).astype(np.complex64)
    r = st.convolution_reader(S.BufferReader(x, 1_000_000, max_read=3000), H)
    check_reader_contract(hz, S, r)
    out = zeros("c64", flen * nblk)
    assert S.read_full(r, out) == flen * nblk
    want = zeros("c64", flen * nblk)
    orc.convolution_reader(want, x, H)
    err = np.linalg.norm(out.astype(np.complex128) - want) / np.linalg.norm(want.astype(np.complex128))
    assert err < 2e-6
    with pytest.raises(hz.ErrSampleFormatUnknown):
        st.convolution_reader(S.BufferReader(zeros("u8", 8), 1), H)


def test_read_beamform(hz, S, st, orc):
    """stream/beamform.go:148-171 data path (the reference has no test for it):
    four u8 channels, kerberos-style; bit-exact with convert -> rotate -> ordered add."""
    n = 50_000
    ch = [rand_u8(40 + i, n) for i in range(4)]
    angles = S.beamform_angles(433e6, 30.0, [0.0, 0.1, 0.2, 0.3])
    bf = st.read_beamform([S.BufferReader(x, 2_400_000, max_read=9000) for x in ch], angles)
    assert bf.sample_format() == hz.FMT_C64 and bf.sample_rate() == 2_400_000
    out = zeros("c64", n)
    assert S.read_full(bf, out[:20_000]) == 20_000
    chc = []
    for x in ch:
        y = zeros("c64", n)
        orc.convert(y, x)
        chc.append(y)
    want = zeros("c64", n)
    orc.beamform(want, chc, angles)
    assert bits_equal(out[:20_000], want[:20_000])
    # SetPhaseAngles applies between reads (stream/beamform.go:131-139)
    new = S.beamform_angles(433e6, -10.0, [0.0, 0.1, 0.2, 0.3])
    bf.set_phase_angles(new)
    assert S.read_full(bf, out[20_000:]) == 30_000
    want2 = zeros("c64", n)
    orc.beamform(want2, chc, new)
    assert bits_equal(out[20_000:], want2[20_000:])
    with pytest.raises(hz.HzsdrError):
        bf.set_phase_angles(new[:2])


def test_fft_planner_conformance(hz, S, st, orc, kats):
    """testutils.TestFFT run against the GPU Planner (testutils/fft.go:40-138)."""
    k = kats["fft_forward_bins"]
    for freq, idx in k["cases"]:
        iq, out = orc.cw(k["n"], freq, k["rate"], 0.0), zeros("c64", k["n"])
        plan = st.planner(iq, out, True)
        plan.transform()
        plan.close()
        assert int(np.argmax(np.abs(out.astype(np.complex128)))) == idx
    for a, b, _ in kats["fft_mismatch"]["cases"]:
        with pytest.raises(hz.ErrDstTooSmall):
            st.planner(zeros("c64", a), zeros("c64", b), True)


def test_cxx_host_mirror_runs_reference_kats():
    """The C++ host mirror (go-sdr_amd/cxx/hzsdr.hpp) over the C ABI: built with
    g++ against libhzsdr_hip.so, runs the reference's KATs the way a cgo shim would."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "build", "test_host_mirror")
    src = os.path.join(ROOT, "tests", "cxx", "test_host_mirror.cpp")
    deps = [src, os.path.join(ROOT, "go-sdr_amd", "cxx", "hzsdr.hpp"), os.path.join(ROOT, "include", "hzsdr.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + ROOT, src, "-L" + os.path.join(ROOT, "go-sdr_amd"),
                               "-lhzsdr_hip", "-L/opt/rocm/lib", "-Wl,-rpath," + os.path.join(ROOT, "go-sdr_amd"),
                               "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "go-sdr_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "all host-mirror tests passed" in p.stdout


def test_plain_c_abi_walkthrough():
    """tests/c/test_c_abi.c: the call sequences of the cgo package go/hip, compiled by gcc as
    C99 (the compiler cgo hands its preambles to) and run against libhzsdr_hip.so."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "build", "test_c_abi")
    src = os.path.join(ROOT, "tests", "c", "test_c_abi.c")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"), src,
                           "-L" + os.path.join(ROOT, "go-sdr_amd"), "-lhzsdr_hip", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "go-sdr_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "go-sdr_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "c-abi ok" in p.stdout


# ---- round 5: nested Readers that fuse into one chain and read ahead (go/hip/fused.go, stream.ChainReader) ---------

@pytest.fixture(scope="module")
def fst(hz, S):
    ctx = hz.Context(0, hz.MEM_HOST)
    yield S.Stream(ctx, fuse=True, readahead=32)
    ctx.close()


def _drain(S, r, chunk, hz):
    """Read r to its end in chunks of `chunk` samples -> (all the samples, the error that ended it)."""
    got = []
    buf = hz.make_samples(r.sample_format(), chunk)
    while True:
        try:
            n = r.read(buf)
        except (S.EOF, hz.HzsdrError) as e:
            return (np.concatenate(got) if got else buf[:0].copy()), e
        got.append(buf[:n].copy())


def test_nested_readers_fuse_into_one_chain_and_read_ahead(hz, S, st, fst, orc):
    """Gain(ShiftReader(ConvertReader(src))) over 2^24 u8 samples: with the fusing constructors the three Readers are ONE
    chain behind a pinned ring -- at most 2^24 / 2^20 launches' worth of calls (counted: hzsdr_call_count), the samples
    the reference's nest gives bit for bit (the oracle's convert -> Shift -> Gain over the whole stream), read through a
    consumer that asks for awkward amounts.  The partial last block of the source is dropped as ConvertReader's ReadFull
    drops it (stream/read_transformer.go:120-135), and the stream ends in EOF."""
    n, rate, shift, gain = (1 << 24) + 12_345, 2_400_000, 3.1e5, 0.5
    x = rand_u8(31, n)
    whole = n // S.READER_BLOCK * S.READER_BLOCK
    want = zeros("c64", whole)
    orc.convert(want, x[:whole])
    orc.Shifter(rate)(shift, want)
    orc.scale(want, gain)
    r = fst.gain(fst.shift_reader(fst.convert_reader(S.BufferReader(x, rate, max_read=100_003), hz.FMT_C64), shift), gain)
    assert isinstance(r, S.ChainReader) and r.sample_format() == hz.FMT_C64 and r.sample_rate() == rate
    check_reader_contract(hz, S, r)
    before = fst.ctx.call_count()
    got, err = _drain(S, r, 77_777, hz)
    calls = fst.ctx.call_count() - before
    assert isinstance(err, S.EOF)
    assert len(got) == whole and bits_equal(got, want)
    slots = -(-whole // (32 * S.READER_BLOCK))
    # per slot: acquire, submit, pop (+ the chain's and the ring's construction): far below one call per 32 Ki block
    assert calls <= 3 * slots + 16, (calls, slots)
    assert calls < whole // S.READER_BLOCK // 4
    r.close()


@pytest.mark.parametrize("kind", ["convert_decimate", "c64_shift_gain_ragged", "downsample_i16", "convolution_decimate",
                                  "convolution1024_decimate", "multiply_then_more", "decimate_then_gain"])
def test_fused_readers_equal_the_nested_ones(hz, S, st, fst, kind):
    """Every fusable nest against the SAME nest built with the unfused constructors (the reference's own structure:
    one ReadTransformer / wrapper and one GPU call per stage and block): the same samples bit for bit, the same end of
    stream -- sources that end inside a block, sources that deliver in dribbles, consumers that read odd amounts."""
    rate = 2_400_000
    rng = np.random.default_rng(5)

    def nest(s, src):
        if kind == "convert_decimate":          # DecimateReader(Multiply(ConvertReader(i16)))
            return s.decimate_reader(s.multiply(s.convert_reader(src, hz.FMT_C64), 0.6 + 0.8j), 5)
        if kind == "c64_shift_gain_ragged":     # pass-through stages over c64: no block structure at all
            return s.gain(s.shift_reader(src, -123_456.0), 1.5)
        if kind == "downsample_i16":            # DownsampleReader converts by itself
            return s.downsample_reader(src, 8)
        if kind in ("convolution_decimate", "convolution1024_decimate"):  # DecimateReader(ConvolutionReader(ShiftReader(c64)))
            L = 1000 if kind == "convolution_decimate" else 1024  # (1000 bins: the DecimateReader is a second chain)
            t = np.arange(L) - (L - 1) / 2
            H = np.fft.fft((np.sinc(t / 8) / 8 * np.hamming(L)).astype(np.complex128) / L).astype(np.complex64)
            return s.decimate_reader(s.convolution_reader(s.shift_reader(src, 50_000.0), H), 4)
        if kind == "multiply_then_more":        # Gain(Multiply(ShiftReader(ConvertReader(u8))))
            return s.gain(s.multiply(s.shift_reader(s.convert_reader(src, hz.FMT_C64), 1e5), 1j), 0.25)
        # a stage behind a terminal opens a second chain: Gain(DecimateReader(ConvertReader(u8)))
        return s.gain(s.decimate_reader(s.convert_reader(src, hz.FMT_C64), 3), 2.0)

    n = {"convolution_decimate": 3 * 4_096_000 // 2 + 777, "c64_shift_gain_ragged": 3_000_017}.get(kind, (1 << 21) + 4_321)
    if kind in ("c64_shift_gain_ragged", "convolution_decimate", "convolution1024_decimate"):
        x = rand_c64(8, n)
    elif kind in ("convert_decimate", "downsample_i16"):
        x = rand_i16(8, n)
    else:
        x = rand_u8(8, n)
    outs = []
    for s in (st, fst):
        r = nest(s, S.BufferReader(x, rate, max_read=int(rng.integers(1_000, 90_000))))
        got, err = _drain(S, r, 50_001, hz)
        assert isinstance(err, S.EOF), (kind, err)
        outs.append((got, r.sample_rate(), r.sample_format()))
        if hasattr(r, "close"):
            r.close()
    assert outs[0][1:] == outs[1][1:]
    assert len(outs[0][0]) == len(outs[1][0]) and len(outs[0][0]) > 0, (kind, len(outs[0][0]), len(outs[1][0]))
    if kind.startswith("convolution"):  # (the fused chain's block kernel and the closure's differ in rounding: both FFTs)
        a, b = outs[0][0].astype(np.complex128), outs[1][0].astype(np.complex128)
        assert np.linalg.norm(a - b) <= 2e-6 * np.linalg.norm(a)
    else:
        assert bits_equal(outs[0][0], outs[1][0]), kind


def test_fused_multiply_reader_takes_a_new_multiplier(hz, S, fst, orc):
    """SetMultiplier on a fused Reader (stream/multiply.go:34-36; Beamform.SetPhaseAngles calls it): what has been read
    ahead keeps the old multiplier, everything behind it the new one, the Shift's clock carries on across the rebuilt
    chain -- so the output is the oracle's with a switch of the multiplier at ONE sample, and that sample is a whole
    number of slots into the stream."""
    n, rate = 6 * (1 << 20), 2_400_000
    x = rand_c64(77, n)
    r = fst.multiply(fst.shift_reader(S.BufferReader(x, rate), 7e4), 2j)
    buf = zeros("c64", 1 << 19)
    first = [buf[:r.read(buf)].copy()]
    r.set_multiplier(0.5)
    rest, err = _drain(S, r, 300_000, hz)
    assert isinstance(err, S.EOF)
    got = np.concatenate(first + [rest])
    assert len(got) == n
    sh = zeros("c64", n)
    sh[:] = x
    orc.Shifter(rate)(7e4, sh)
    a, b = sh.copy(), sh.copy()
    orc.rotate(a, np.complex64(2j))
    orc.rotate(b, np.complex64(0.5))
    same_a = np.flatnonzero(got.view(np.uint64) != a.view(np.uint64))
    k = int(same_a[0]) if len(same_a) else n
    assert k % (1 << 20) == 0 and (1 << 20) <= k <= 3 * (1 << 20), k  # (one slot being read, up to two more in flight)
    assert bits_equal(got[:k], a[:k]) and bits_equal(got[k:], b[k:])
    r.close()

"""SURVEY.md 8f rank 3/4 "next" rows on the GPU: foreign-endian payload swap
(bytes_io.go), FFTShiftAndScale (rtl/kerberos/internal/reader.go:47-64) and one
block of GraftReaders (rtl/kerberos/internal/graft.go:63-122) against the oracle.
Byte and division work is bit-exact; graft goes through three FFT stages and is
held to a relative L2 bound written in the test."""
import importlib

import numpy as np
import pytest

from util import bits_equal, rand_c64, rand_i16, rand_i8, rand_u8, zeros

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hz():
    return importlib.import_module("go-sdr_amd")


class Env:
    def __init__(self, hz, kind):
        import torch
        self.kind, self.torch = kind, torch
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


@pytest.fixture(scope="module", params=["host", "device"])
def env(request, hz):
    e = Env(hz, request.param)
    yield e
    e.ctx.close()


@pytest.mark.parametrize("fmt,gen", [("i16", rand_i16), ("c64", rand_c64)])
@pytest.mark.parametrize("n,skip", [(0, 0), (1, 0), (7, 0), (4096, 0), (100_003, 0), (100_003, 1),
                                    (1 << 20, 3)])
def test_byteswap_bit_exact(env, orc, fmt, gen, n, skip):
    """Odd lengths and slices that start off a 16-byte boundary take the scalar tail."""
    x = gen(31, n + skip)
    want = x[skip:].copy()
    orc.byteswap(want)
    if n:  # the oracle against numpy's own byteswap
        comp = want.view(np.float32 if fmt == "c64" else np.int16)
        assert np.array_equal(comp.view(np.uint8), x[skip:].view(comp.dtype).byteswap().view(np.uint8))
    d = env.put(x)
    env.ctx.byteswap(d[skip:])
    got = env.get(d)
    assert bits_equal(got[skip:], want)
    assert bits_equal(got[:skip], x[:skip])
    env.ctx.byteswap(d[skip:])  # an involution
    assert bits_equal(env.get(d), x)


GENS = {"u8": rand_u8, "i8": rand_i8, "i16": rand_i16, "c64": rand_c64}


@pytest.mark.parametrize("src_fmt,dst_fmt", [(s, d) for s in GENS for d in GENS if s != d])
@pytest.mark.parametrize("src_foreign,dst_foreign", [(True, False), (False, True), (True, True)])
def test_convert_foreign_is_swap_convert_swap(env, orc, src_fmt, dst_fmt, src_foreign, dst_foreign):
    """The fused converter equals byteswap -> ConvertBuffer -> byteswap done by the
    oracle one pass at a time; n = 100 003 leaves a scalar tail, skip = 1 an unaligned slice."""
    n, skip = 100_003, 1
    native = GENS[src_fmt](51, n + skip)
    if src_fmt == "c64":
        native *= np.float32(0.99)  # in-range for the float -> int converters
    wire = native.copy()
    if src_foreign:
        orc.byteswap(wire)
    want = zeros(dst_fmt, n)
    orc.convert(want, native[skip:].copy())
    if dst_foreign:
        orc.byteswap(want)
    d_src, d_dst = env.put(wire), env.put(zeros(dst_fmt, n + skip))
    assert env.ctx.convert_foreign(d_dst[skip:], d_src[skip:], dst_foreign, src_foreign) == n
    got = env.get(d_dst)
    assert bits_equal(got[skip:], want)
    assert not got[:skip].any()


def test_convert_foreign_errors(hz, env):
    with pytest.raises(hz.ErrDstTooSmall):
        env.ctx.convert_foreign(env.put(zeros("c64", 10)), env.put(zeros("i16", 11)), False, True)
    with pytest.raises(hz.ErrConversionNotImplemented):
        env.ctx.convert_foreign(env.put(zeros("i16", 10)), env.put(zeros("i16", 10)), False, True)


@pytest.mark.parametrize("order", ["little", "big"])
def test_bytes_io_kat(hz, order):
    """bytes_io_test.go:69-127 (TestBytesIOLE / TestBytesIOBE): ten writes of 1024 c64
    samples with sample 10 = 20+10i through ByteWriter -> bytes -> ByteReader + ReadFull."""
    import io
    S = importlib.import_module("go-sdr_amd.stream")
    ctx = hz.Context(0, hz.MEM_HOST)
    pipe = io.BytesIO()
    w = S.ByteWriter(ctx, pipe, order, 0, hz.FMT_C64)
    for _ in range(10):
        wb = zeros("c64", 1024)
        wb[10] = 20 + 10j
        assert w.write(wb) == 1024
    wire = pipe.getvalue()
    # the wire really is in that order: numpy decodes it with an explicit-endian dtype
    dec = np.frombuffer(wire, np.dtype("<f4" if order == "little" else ">f4")).astype(np.float32)
    assert dec[2 * 10] == 20 and dec[2 * 10 + 1] == 10
    r = S.ByteReader(ctx, io.BytesIO(wire), order, 0, hz.FMT_C64)
    buf = zeros("c64", 1024 * 10)
    assert S.read_full(r, buf) == 1024 * 10
    for i in range(10):
        assert buf[i * 1024 + 10] == np.complex64(20 + 10j)
    with pytest.raises(S.EOF):
        r.read(zeros("c64", 4))
    with pytest.raises(hz.ErrSampleFormatMismatch):  # bytes_io.go:126-128
        r.read(zeros("u8", 4))
    ctx.close()


@pytest.mark.parametrize("fmt,gen,wire_dt", [("i16", rand_i16, ">i2"), ("c64", rand_c64, ">f4"),
                                             ("u8", rand_u8, "u1")])
def test_byte_reader_decodes_big_endian_captures(hz, fmt, gen, wire_dt):
    """A big-endian capture file read through ByteReader equals what numpy decodes."""
    import io
    S = importlib.import_module("go-sdr_amd.stream")
    ctx = hz.Context(0, hz.MEM_HOST)
    x = gen(9, 50_000)
    comp = x.view(np.float32) if fmt == "c64" else x
    wire = comp.astype(np.dtype(wire_dt)).tobytes()
    r = S.ByteReader(ctx, io.BytesIO(wire), "big", 2_400_000, hz.FMT_C64 if fmt == "c64" else getattr(hz, "FMT_" + fmt.upper()))
    got = zeros(fmt, 50_000)
    assert S.read_full(r, got) == 50_000
    assert bits_equal(got, x)
    # a capture that ends mid-buffer: binary.Read semantics for the foreign i16 / c64 path
    r2 = S.ByteReader(ctx, io.BytesIO(wire[:len(wire) // 2 + 1]), "big", 0, r.sample_format())
    if fmt != "u8":
        with pytest.raises(S.ErrUnexpectedEOF):
            r2.read(zeros(fmt, 50_000))
    else:
        assert r2.read(zeros(fmt, 50_000)) == (len(wire) // 2 + 1) // 2
    ctx.close()


@pytest.mark.parametrize("gen", [rand_u8, rand_i8])
def test_byteswap_bytes_have_no_order(env, gen):
    x = gen(5, 1000)
    d = env.put(x)
    env.ctx.byteswap(d)
    assert bits_equal(env.get(d), x)


@pytest.mark.parametrize("n", [0, 1, 2, 9, 1024, 65536, (1 << 20) + 1])
@pytest.mark.parametrize("scale", [1.0, 3.0, 65536.0])
def test_fftshift_scale_bit_exact(env, orc, n, scale):
    x = rand_c64(n + 17, n)
    want = x.copy()
    orc.fftshift_scale(want, scale)
    d = env.put(x)
    env.ctx.fftshift_scale(d, scale)
    assert bits_equal(env.get(d), want)


def test_fft_shift_kat(env):
    """fft/result_test.go:186-204 (TestFFTShift): 2048 bins holding their own signed
    frequency index; Shift puts -1024 first and 0 at 1024, Shift again restores it."""
    f = np.concatenate([np.arange(1024), np.arange(-1024, 0)]).astype(np.complex64)
    d = env.put(f)
    env.ctx.fft_shift(d)
    got = env.get(d).copy()
    assert got[0] == -1024 and got[1024] == 0 and got[2047] == 1023
    env.ctx.fft_shift(d)
    assert bits_equal(env.get(d), f)


def test_convolve_once(env, orc):
    """fft.ConvolveOnce (fft/convolution.go:200-211) = plan, run once, close; dst aliases iq1."""
    n = 4096
    a, b = rand_c64(1, n), rand_c64(2, n)
    want = zeros("c64", n)
    orc.convolve(want, a, b)
    da, db = env.put(a), env.put(b)
    env.ctx.convolve_once(da, da, db)
    got = env.get(da)
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 2e-6


@pytest.mark.parametrize("count,n", [(1, 1024), (2, 512), (4, 4096), (2, 65536), (4, 65536),
                                     (8, 65536)])
def test_graft_matches_oracle(env, orc, count, n):
    bands = [rand_c64(100 + k, n) for k in range(count)]
    want = zeros("c64", count * n)
    assert orc.graft(want, bands) == 0
    out = env.put(zeros("c64", count * n))
    env.ctx.graft(out, [env.put(b) for b in bands])
    got = env.get(out)
    err = np.linalg.norm(got.astype(np.complex128) - want) / np.linalg.norm(want)
    # float32 FFTs of 2^10..2^19 points, twice in series, against a float64 one
    assert err < 2e-6, err


def test_graft_of_one_band_is_a_sign_flip(env):
    """fftshift in frequency is (-1)^k in time: with one band the stitched output is
    the input with every odd sample negated (size-independent property)."""
    n = 1 << 20
    x = rand_c64(77, n)
    out = env.put(zeros("c64", n))
    env.ctx.graft(out, [env.put(x)])
    got = env.get(out)
    want = x * np.where(np.arange(n) % 2 == 0, 1, -1).astype(np.float32)
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 2e-6


def test_graft_places_each_band_in_its_own_slice(env):
    """A tone at +f in band c comes out at ((c + 1/2) n + f n/fs ... ) of the wide
    spectrum: the forward FFT of the output peaks inside slice c."""
    n, count = 4096, 4
    k0 = 37  # bin of the tone inside its band
    t = np.arange(n)
    for c in range(count):
        bands = [zeros("c64", n) for _ in range(count)]
        bands[c] = np.exp(2j * np.pi * k0 * t / n).astype(np.complex64)
        out = env.put(zeros("c64", count * n))
        env.ctx.graft(out, [env.put(b) for b in bands])
        spec = np.abs(np.fft.fft(env.get(out).astype(np.complex128)))
        peak = int(np.argmax(spec))
        assert peak == c * n + (k0 + n // 2) % n


def test_graft_argument_errors(hz, env):
    out = env.put(zeros("c64", 4096))
    with pytest.raises(hz.ErrDstTooSmall):
        env.ctx.graft(out, [env.put(zeros("c64", 4096)) for _ in range(2)])
    with pytest.raises(hz.HzsdrError):  # 3 bands: count * n is not a power of two
        env.ctx.graft(env.put(zeros("c64", 3 * 1024)), [env.put(zeros("c64", 1024)) for _ in range(3)])


def test_graft_reader_stream(hz, orc):
    """GraftReaders end to end over BufferReaders: two trips, then EOF."""
    K = importlib.import_module("go-sdr_amd.kerberos")
    S = importlib.import_module("go-sdr_amd.stream")
    ctx = hz.Context(0, hz.MEM_HOST)
    n, count, trips = 8192, 2, 2
    data = [rand_c64(200 + k, n * trips + 100) for k in range(count)]  # the ragged tail is dropped
    r = K.graft_readers(ctx, [S.BufferReader(d, 2_048_000, max_read=5000) for d in data], fft_size=n)
    assert r.sample_rate() == count * 2_048_000 and r.sample_format() == hz.FMT_C64
    got = zeros("c64", count * n * trips)
    assert S.read_full(r, got) == len(got)
    for t in range(trips):
        want = zeros("c64", count * n)
        orc.graft(want, [d[t * n:(t + 1) * n].copy() for d in data])
        g = got[t * count * n:(t + 1) * count * n]
        assert np.linalg.norm(g - want) / np.linalg.norm(want) < 2e-6
    with pytest.raises(S.EOF):
        r.read(zeros("c64", 16))
    with pytest.raises(hz.ErrSampleFormatMismatch):
        r.read(zeros("u8", 16))
    ctx.close()

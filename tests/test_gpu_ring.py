"""the pinned ring in front of a chain (hzsdr_ring_*), the
GPU-side stream.RingBuffer with an IQBufferAllocator of hipHostMalloc memory
(stream/ring.go:48-69).  The ring overlaps upload / kernel / download of
neighbouring slots; its results must be bit-identical to the same chain run
synchronously over the same stream of samples."""
import importlib

import numpy as np
import pytest

from util import bits_equal, rand_c64, rand_i16, rand_u8

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hz():
    return importlib.import_module("go-sdr_amd")


@pytest.fixture(scope="module")
def ctx(hz):
    c = hz.Context(0, hz.MEM_HOST)
    yield c
    c.close()


def lowpass(ntaps, cutoff):
    k = np.arange(ntaps) - (ntaps - 1) / 2
    return (2 * cutoff * np.sinc(2 * cutoff * k) * np.hamming(ntaps)).astype(np.complex64)


def build(hz, ctx, kind):
    if kind == "shift_gain":
        return ctx.chain(hz.FMT_C64, 20_000_000).shift(2.5e6).gain(0.5), rand_c64, 1
    if kind == "u8_fir8":
        return ctx.chain(hz.FMT_U8, 20_000_000).shift(-2.5e6).fir_decimate(lowpass(1024, 1 / 16), 8), rand_u8, 8
    if kind == "i16_downsample":
        return ctx.chain(hz.FMT_I16, 200_000_000).downsample(8), rand_i16, 8
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["shift_gain", "u8_fir8", "i16_downsample", "u8_fir8_pipelined"])
@pytest.mark.parametrize("slots", [2, 4])
def test_ring_equals_synchronous_chain(hz, ctx, orc, kind, slots):
    slot_len, trips = 1 << 17, 11
    piped = kind.endswith("_pipelined")  # hzsdr_chain_pipeline under the ring: the kernels' streams wait for the uploads
    kind = kind.replace("_pipelined", "")
    chain, gen, D = build(hz, ctx, kind)
    if piped:
        chain.pipeline(True)
    ref_chain, _, _ = build(hz, ctx, kind)
    data = gen(41, slot_len * trips)
    want = np.zeros(slot_len * trips // D, np.complex64)
    for t in range(trips):  # the same stream, one synchronous run per block
        used, made = ref_chain.run(data[t * slot_len:(t + 1) * slot_len],
                                   want[t * slot_len // D:(t + 1) * slot_len // D])
        assert (used, made) == (slot_len, slot_len // D)
    ring = chain.ring(slot_len, slots)
    assert hz.length(ring.iq) == slot_len * slots and hz.fmt_of(ring.iq) == chain.src_fmt
    got = []
    for t in range(trips):
        if ring.in_flight == slots:  # overrun: the consumer must drain first
            with pytest.raises(hz.ErrDstTooSmall):
                ring.acquire()
            got.append(ring.pop().copy())
        slot, iq = ring.acquire()
        assert slot == t % slots
        iq[:] = data[t * slot_len:(t + 1) * slot_len]
        ring.submit(slot)
    while ring.in_flight:
        got.append(ring.pop().copy())
    with pytest.raises(hz.HzsdrError):  # underrun
        ring.pop()
    got = np.concatenate(got)
    assert bits_equal(got, want)
    if kind == "shift_gain":  # and the synchronous chain is the oracle's Shift + Scale
        sh = orc.Shifter(20_000_000)
        y = data.copy()
        sh(2.5e6, y)
        orc.scale(y, 0.5)
        assert bits_equal(got, y)
    ring.close()
    chain.close()
    ref_chain.close()


def test_ring_partial_slot_and_argument_errors(hz, ctx):
    chain = ctx.chain(hz.FMT_U8, 2_400_000).downsample(4)  # 32 Ki-sample reader blocks
    with pytest.raises(hz.HzsdrError):
        chain.ring(1000, 4)  # not a whole number of blocks
    with pytest.raises(hz.HzsdrError):
        chain.ring(32768, 1)
    ring = chain.ring(4 * 32768, 3)
    slot, iq = ring.acquire()
    x = rand_u8(3, 2 * 32768)
    iq[:2 * 32768] = x
    with pytest.raises(hz.HzsdrError):
        ring.submit(slot, 40000)  # ragged
    with pytest.raises(hz.HzsdrError):
        ring.submit((slot + 1) % 3, 32768)  # not the acquired slot
    ring.submit(slot, 2 * 32768)  # a short (but whole-block) slot is fine
    out = ring.pop()
    want = np.zeros(2 * 32768 // 4, np.complex64)
    ref = ctx.chain(hz.FMT_U8, 2_400_000).downsample(4)
    ref.run(x, want)
    assert bits_equal(out, want)
    ring.close()
    ref.close()
    chain.close()

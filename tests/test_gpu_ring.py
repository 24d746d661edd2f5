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


@pytest.mark.parametrize("group", [2, 4, 8])
@pytest.mark.parametrize("kind", ["u8_fir8_pipelined", "u8_fir8", "i16_downsample"])
def test_ring_submit_many_is_one_call_per_group(hz, ctx, orc, kind, group):
    """hzsdr_ring_submit_many: `group` acquired slots in ONE call of the chain -- for the north-star chain one launch of
    the persistent-pass kernel over all of them (hzsdr_chain_run_batch_after under the ring), for a chain without that
    form slot by slot.  The outputs equal the synchronous chain's over the same stream BIT FOR BIT (the mixer's phase
    and the matrix / task split belong to the clock run, not to the call: csrc/hz_firmm2_plan.h), across the 2 pi wrap;
    hzsdr_call_count counts one library call per group; the FIR chain reports the persistent-pass kernel."""
    TAU = 6.283185307179586476925286766559
    slot_len, slots = 1 << 18, 9  # (32 768 outputs per slot: the slot that holds the wrap's ~2 700 fix-up outputs still takes the matrix path by itself)
    piped = kind.endswith("_pipelined")
    kind = kind.replace("_pipelined", "")
    chain, gen, D = build(hz, ctx, kind)
    if piped:
        chain.pipeline(True)
    ref_chain, _, _ = build(hz, ctx, kind)
    trips = 3 * group + 1  # three groups, then a single slot
    data = gen(43, slot_len * trips)
    ts0 = TAU - 0.004  # (20 Msps: the wrap 80 000 samples in -- inside the first group)
    if kind == "u8_fir8":
        chain.set_time(ts0), ref_chain.set_time(ts0)
    want = np.zeros(slot_len * trips // D, np.complex64)
    for t in range(trips):
        assert ref_chain.run(data[t * slot_len:(t + 1) * slot_len], want[t * slot_len // D:(t + 1) * slot_len // D]) == (slot_len, slot_len // D)
    ring = chain.ring(slot_len, slots)
    got, t = [], 0
    while t < trips:
        k = min(group, trips - t)
        first = None
        for j in range(k):
            slot, iq = ring.acquire()
            first = slot if first is None else first
            iq[:] = data[(t + j) * slot_len:(t + j + 1) * slot_len]
        before = ctx.call_count()
        ring.submit_many(first, k)
        # (acquire / pop / in_flight are not counted: one counted call per group)
        assert ctx.call_count() == before + 1
        if kind == "u8_fir8" and k > 1:
            assert chain.last_fir_kernel() == hz.FIR_KERNEL_MATRIX_PASSES
        t += k
        assert ring.in_flight == k
        for _ in range(k):
            got.append(ring.pop().copy())
    got = np.concatenate(got)
    assert bits_equal(got, want), int((got.view(np.int64) != want.view(np.int64)).sum())
    assert chain.time() == ref_chain.time()
    # errors: a group that is not the oldest acquired slots, too many slots
    s0, _ = ring.acquire()
    s1, _ = ring.acquire()
    with pytest.raises(hz.HzsdrError):
        ring.submit_many(s1, 1)
    with pytest.raises(hz.HzsdrError):
        ring.submit_many(s0, 3)
    with pytest.raises(hz.HzsdrError):
        ring.release(s0)  # (only the newest acquired slot goes back)
    ring.release(s1)
    ring.release(s0)
    ring.close()
    chain.close()
    ref_chain.close()


@pytest.mark.parametrize("piped", [False, True])
def test_ring_submit_many_of_slots_that_do_not_qualify_goes_slot_by_slot(hz, ctx, piped):
    """Slots of 2^16 samples across the 2 pi wrap: the slot that holds the wrap is mostly fix-up outputs and keeps the
    transform kernels when it runs by itself, so a call over several slots must not take it into one launch of the matrix
    kernel either (hzsdr_chain_run_batch promises the bits of single calls): hzsdr_ring_submit_many falls back to one
    launch per slot -- overlapped where the chain is pipelined -- and the outputs equal the synchronous chain's bit for
    bit."""
    TAU = 6.283185307179586476925286766559
    slot_len, slots, group, trips = 1 << 16, 9, 4, 12
    chain, gen, D = build(hz, ctx, "u8_fir8")
    if piped:
        chain.pipeline(True)
    ref_chain, _, _ = build(hz, ctx, "u8_fir8")
    data = gen(47, slot_len * trips)
    ts0 = TAU - 0.004  # (the wrap 80 000 samples in: slot 1)
    chain.set_time(ts0), ref_chain.set_time(ts0)
    want = np.zeros(slot_len * trips // D, np.complex64)
    kinds = []
    for t in range(trips):
        assert ref_chain.run(data[t * slot_len:(t + 1) * slot_len], want[t * slot_len // D:(t + 1) * slot_len // D]) == (slot_len, slot_len // D)
        kinds.append(ref_chain.last_fir_path())
    assert hz.FIR_PATH_TRANSFORM in kinds and hz.FIR_PATH_MATRIX in kinds  # (the stream meets both implementations)
    ring = chain.ring(slot_len, slots)
    got = []
    for t in range(0, trips, group):
        first = None
        for j in range(group):
            slot, iq = ring.acquire()
            first = slot if first is None else first
            iq[:] = data[(t + j) * slot_len:(t + j + 1) * slot_len]
        ring.submit_many(first, group)
        for _ in range(group):
            got.append(ring.pop().copy())
    got = np.concatenate(got)
    assert bits_equal(got, want), int((got.view(np.int64) != want.view(np.int64)).sum())
    assert chain.time() == ref_chain.time()
    ring.close()
    chain.close()
    ref_chain.close()


def test_fir_decimate_reader_is_the_north_star_chain_as_one_reader(hz, orc):
    """Stream.fir_decimate_reader(ShiftReader(ConvertReader(u8 source))): the north-star chain behind sdr.Reader --
    ONE chain (the int8 matrix kernel), a pinned ring that reads ahead, four slots per launch.  Against the oracle
    (reference-order convert + Shift, float64 direct form) within the FIR bound, bit for bit against the same chain
    run synchronously, and counted: one library call per group of slots."""
    from importlib import import_module
    from util import assert_fir_close, zeros
    st = import_module("go-sdr_amd.stream")
    TAU = 6.283185307179586476925286766559
    # (22 full slots, a short one of three Reader blocks, and 40 samples more: the ConvertReader in the chain hands out
    # whole 32 Ki blocks only -- a source that ends inside a block loses that block, read_transformer.go:120-135)
    fs, D, n_src = 20_000_000, 8, 22 * (1 << 18) + 3 * 32768 + 40
    n = n_src - 40
    taps = lowpass(1024, 1 / 16)
    x = rand_u8(57, n_src)
    hctx = hz.Context(0, hz.MEM_HOST)
    S = st.Stream(hctx, fuse=True, readahead=8)  # slots of 8 x 32 Ki = 2^18 samples
    src = st.BufferReader(x, fs)
    rd = S.fir_decimate_reader(S.shift_reader(S.convert_reader(src, hz.FMT_C64), -fs / 8), taps, D, slots=9, group=4)
    assert isinstance(rd, st.ChainReader) and rd.sample_rate() == fs // D and rd.sample_format() == hz.FMT_C64
    out = zeros("c64", n // D + 16)
    before = hctx.call_count()
    got = 0
    while True:
        try:
            k = rd.read(out[got:])
        except st.EOF:
            break
        got += k
    calls = hctx.call_count() - before
    assert got == n // D
    assert rd.chain.last_fir_kernel() == hz.FIR_KERNEL_MATRIX_PASSES
    # counted calls: 23 pops (one per slot), the submits -- 22 full slots in groups of four and a short one: at most
    # 6 + 1 + 1 -- and the construction of chain and ring (a dozen); one submit per slot would be 23 more
    assert calls <= 23 + 8 + 12, calls
    rd.close()
    xc = zeros("c64", n)
    orc.convert(xc, x[:n])
    orc.Shifter(fs)(-fs / 8, xc)
    want = zeros("c64", n // D)
    orc.par_fir_decimate_f64(want, xc, taps, D)
    assert_fir_close(out[:got], want, taps, float(np.abs(xc).max()), "fir_decimate_reader")
    # the same stream through the synchronous chain, cut at the slots' boundaries: the same bits
    ch = hctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D)
    ref = zeros("c64", n // D)
    sl = 1 << 18
    for a in range(0, n // D * D, sl):
        b = min(a + sl, n // D * D)
        assert ch.run(x[a:b], ref[a // D:b // D]) == (b - a, (b - a) // D)
    assert bits_equal(out[:got], ref)
    ch.close()
    hctx.close()

"""The int8 matrix form of the FIR-decimate terminal (csrc/hz_firmm.h): u8 / i8 sources,
decimation 8 or 16.  Every case is held to the oracle (reference-order Shift, float64 direct
form) with the bounds of tests/util.py, and the cases assert WHICH kernels ran
(hzsdr_chain_last_fir_path): the matrix form where it applies, the transform kernels where
it must not (other factors, misaligned buffers, short calls, HZ_FIR_FFT=1, in-order mixer).

What the matrix form adds to the late mixer's tests (test_gpu_latemix.py):
  * the filter sums are exact integer arithmetic on the quantised taps: a relative L2 error an
    order of magnitude under the transform path's is asserted;
  * chunks of 2048 outputs on the call's grid, owned by one clock run each; outputs whose window
    crosses a run boundary, the stream start or a run without a table come from the fix-up
    tasks -- calls that start on a boundary, cross several, wrap at 2*pi;
  * the raw history: a window that reaches back into the previous call when the clock run
    continues, the float history otherwise and after a call on the transform kernels."""
import importlib
import os

import numpy as np
import pytest

from util import assert_fir_close, bits_equal, fir_errors, rand_i8, rand_u8, zeros

pytestmark = pytest.mark.gpu

TAU = 6.283185307179586476925286766559


@pytest.fixture(scope="module")
def hz():
    return importlib.import_module("go-sdr_amd")


@pytest.fixture(scope="module")
def ctx(hz):
    c = hz.Context(0, hz.MEM_HOST)
    yield c
    c.close()


def taps_for(ntaps, cutoff=1 / 32, rot=0.3):
    k = np.arange(ntaps) - (ntaps - 1) / 2
    return (2 * cutoff * np.sinc(2 * cutoff * k) * np.hamming(ntaps) * np.exp(1j * rot * k)).astype(np.complex64)


def oracle(orc, x, rate, ops, taps, D, ts0=0.0, cuts=None, ts_at=None):
    """Reference order over the whole stream; `ts_at[i]` resets the clock before call i."""
    n = len(x)
    xc = zeros("c64", n)
    orc.convert(xc, x)
    cuts = cuts or [0, n]
    for kind, arg in ops:
        if kind == "shift":
            sh = orc.Shifter(rate)
            sh.ts.value = ts0
            for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                if ts_at and ts_at.get(i) is not None:
                    sh.ts.value = ts_at[i]
                sh(arg, xc[a:b])
        elif kind == "gain":
            orc.scale(xc, arg)
        else:
            orc.rotate(xc, arg)
    want = zeros("c64", n // D)
    orc.par_fir_decimate_f64(want, xc, taps, D)
    return want, float(np.abs(xc).max())


def build(hz, ctx, fmt, rate, ops, taps, D, impl=0):
    ch = ctx.chain(fmt, rate)
    for kind, arg in ops:
        ch = ch.shift(arg) if kind == "shift" else ch.gain(arg) if kind == "gain" else ch.rotate(arg)
    if impl:
        ch.fir_options(impl)
    return ch.fir_decimate(taps, D)


CASES = {
    "u8_d8_1024": dict(fmt="u8", rate=20_000_000, ops=[("shift", -2.5e6)], ntaps=1024, D=8, n=1 << 20),
    "u8_d16_1023": dict(fmt="u8", rate=20_000_000, ops=[("shift", 1.1e6)], ntaps=1023, D=16, n=1 << 20),
    "u8_d8_64": dict(fmt="u8", rate=2_400_000, ops=[("shift", 3e5)], ntaps=64, D=8, n=1 << 19),
    "u8_d8_129_program": dict(fmt="u8", rate=2_400_000, ops=[("gain", 0.5), ("shift", 1e5), ("rotate", 0.6 - 0.8j),
                                                              ("shift", -3.5e5)], ntaps=129, D=8, n=1 << 19),
    "u8_d16_1000_noshift": dict(fmt="u8", rate=1_000_000, ops=[("gain", 1.5), ("rotate", -0.28 + 0.96j)], ntaps=1000,
                                D=16, n=1 << 20),
    "i8_d8_1024": dict(fmt="i8", rate=20_000_000, ops=[("shift", 4e6)], ntaps=1024, D=8, n=1 << 20),
    "i8_d16_300": dict(fmt="i8", rate=8_000_000, ops=[("shift", -1e6), ("gain", 0.25)], ntaps=300, D=16, n=1 << 19),
    "u8_d8_1536": dict(fmt="u8", rate=20_000_000, ops=[("shift", 2e6)], ntaps=1536, D=8, n=1 << 19),
    "u8_d16_2047": dict(fmt="u8", rate=20_000_000, ops=[("shift", -3e6)], ntaps=2047, D=16, n=1 << 20),
    "u8_d32_1024": dict(fmt="u8", rate=20_000_000, ops=[("shift", -2.5e6)], ntaps=1024, D=32, n=1 << 20),
    "i8_d32_200_program": dict(fmt="i8", rate=2_400_000, ops=[("gain", 0.7), ("shift", 3e5), ("rotate", 0.8 + 0.6j)], ntaps=200,
                               D=32, n=1 << 20),
    "u8_d64_1000": dict(fmt="u8", rate=20_000_000, ops=[("shift", 1.5e6)], ntaps=1000, D=64, n=1 << 21),
    "i8_d24_500": dict(fmt="i8", rate=2_400_000, ops=[("shift", -4e5)], ntaps=500, D=24, n=3 << 18),
    "u8_d40_1024": dict(fmt="u8", rate=20_000_000, ops=[("shift", 2.5e6)], ntaps=1024, D=40, n=5 << 18),
    "u8_d48_300": dict(fmt="u8", rate=20_000_000, ops=[("shift", 1e6), ("gain", 0.5)], ntaps=300, D=48, n=3 << 19),
    "u8_d8_17": dict(fmt="u8", rate=20_000_000, ops=[("shift", 2e6)], ntaps=17, D=8, n=1 << 18),
}


@pytest.mark.parametrize("name", list(CASES))
def test_matrix_form_against_the_oracle(hz, ctx, orc, name):
    c = CASES[name]
    n, D, taps = c["n"], c["D"], taps_for(c["ntaps"])
    x = rand_u8(31, n) if c["fmt"] == "u8" else rand_i8(31, n)
    fmt = hz.FMT_U8 if c["fmt"] == "u8" else hz.FMT_I8
    # three ragged calls: the second starts inside the clock run the first ended in (raw history),
    # all cuts on the decimation grid 
    g = 64 * D  # cuts on the decimation grid of every factor
    cuts = [0, (n // 4) // g * g + 2 * g, (n - n // 4) // g * g, n]
    # (the clock starts at 1 s: from 0 it runs through twenty short binades first, and a call that is
    # mostly clock boundaries stays on the transform kernels -- test_clock_boundaries_and_wrap)
    want, xmax = oracle(orc, x, c["rate"], c["ops"], taps, D, ts0=1.0)
    ch = build(hz, ctx, fmt, c["rate"], c["ops"], taps, D)
    ch.set_time(1.0)
    out = zeros("c64", n // D)
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert ch.run(x[a:b], out[a // D:b // D]) == (b - a, (b - a) // D)
        assert ch.last_fir_path() == hz.FIR_PATH_MATRIX, (name, a)
        # factor 8 up to ~1150 taps: the persistent-pass kernel (csrc/hz_firmm2.h); the rest: chunk workgroups
        # (round 6: factor 16 up to 1040 taps as well -- its fix-up window, taps + 15 x 16 samples, must fit 1280)
        passes = (D == 8 and c["ntaps"] <= 1100) or (D == 16 and c["ntaps"] <= 1040)
        want_kernel = hz.FIR_KERNEL_MATRIX_PASSES if passes else hz.FIR_KERNEL_MATRIX_CHUNKS
        assert ch.last_fir_kernel() == want_kernel, (name, a, ch.last_fir_kernel())
    assert_fir_close(out, want, taps, xmax, name)
    # exact integer filter sums: what is left is one float32 rounding of the filter output and one per
    # elementwise stage (the transform path: 1.7e-7 with a single stage)
    err, bound, rel = fir_errors(out, want, taps, xmax)
    assert rel <= 4e-8 * (1 + len(c["ops"])), (name, rel)
    ch.close()


def test_matrix_and_transform_paths_share_the_history(hz, ctx, orc):
    """Calls alternate between the matrix form and the transform kernels (short calls, a misaligned
    buffer): the float history and the clock carry over in both directions."""
    rate, D, taps = 20_000_000, 8, taps_for(1024)
    n = 1 << 20
    x = rand_u8(5, n + 8)
    cuts = [0, 1 << 18, (1 << 18) + 8 * 1000, 1 << 19, (1 << 19) + (1 << 18) + 8, n]
    ops = [("shift", -2.5e6)]
    want, xmax = oracle(orc, x[:n], rate, ops, taps, D, ts0=1.0)
    ch = build(hz, ctx, hz.FMT_U8, rate, ops, taps, D)
    ch.set_time(1.0)
    out = zeros("c64", n // D)
    paths = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert ch.run(x[a:b], out[a // D:b // D]) == (b - a, (b - a) // D)
        paths.append(ch.last_fir_path())
    # 8000 samples = 1000 outputs: too short for the matrix form
    assert paths == [hz.FIR_PATH_MATRIX, hz.FIR_PATH_TRANSFORM, hz.FIR_PATH_MATRIX, hz.FIR_PATH_MATRIX, hz.FIR_PATH_MATRIX]
    assert_fir_close(out, want, taps, xmax, "alternating paths")
    ch.close()


def test_clock_boundaries_and_wrap(hz, ctx, orc):
    """250 ksps: the clock wraps at 2*pi s = 1 570 796 samples.  Call 1 starts on ts = 0 (a dozen short
    clock runs first: fix-up tasks), call 2 resumes just before the wrap, call 3 continues after it."""
    rate, D, taps = 250_000, 8, taps_for(1024)
    n = 3 * (1 << 19)
    x = rand_u8(6, n)
    cuts = [0, 1 << 19, 1 << 20, n]
    ts2 = TAU - 1.0  # one second before the wrap: inside call 2 (2.1 s per call)
    ops = [("shift", 31_250.0)]
    want, xmax = oracle(orc, x, rate, ops, taps, D, cuts=cuts, ts_at={1: ts2})
    ch = build(hz, ctx, hz.FMT_U8, rate, ops, taps, D)
    out = zeros("c64", n // D)
    paths = []
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if i == 1:
            ch.set_time(ts2)
        assert ch.run(x[a:b], out[a // D:b // D]) == (b - a, (b - a) // D)
        paths.append(ch.last_fir_path())
    assert paths == [hz.FIR_PATH_MATRIX] * 3, paths
    # set_time breaks the stream for the oracle too: compare per call, skipping the outputs whose
    # window crosses the cut into call 2 (the oracle filtered ONE concatenated stream)
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        lo = a // D + (128 if i == 1 else 0)
        assert_fir_close(out[lo:b // D], want[lo:b // D], taps, xmax, ("call", i))
    ch.close()


def test_where_the_matrix_form_must_not_run(hz, ctx, orc):
    rate, taps = 20_000_000, taps_for(1024)
    n = 1 << 19
    x = rand_u8(8, n)
    ops = [("shift", -2.5e6)]

    def run(D, in_order=False, fmt=None, data=None, impl=0):
        ch = build(hz, ctx, fmt or hz.FMT_U8, rate, ops, taps, D, impl)
        ch.mix_in_order(in_order)
        ch.set_time(1.0)
        out = zeros("c64", n // D)
        assert ch.last_fir_path() == hz.FIR_PATH_NONE
        ch.run(data if data is not None else x, out)
        p = ch.last_fir_path()
        ch.close()
        return p, out

    want8, xmax = oracle(orc, x, rate, ops, taps, 8, ts0=1.0)
    p, out = run(8)
    assert p == hz.FIR_PATH_MATRIX
    assert_fir_close(out, want8, taps, xmax, "matrix")
    p, out_io = run(8, in_order=True)  # reference order: the transform kernels
    assert p == hz.FIR_PATH_TRANSFORM
    assert_fir_close(out_io, want8, taps, xmax, "in order")
    p, _ = run(4)  # another factor
    assert p == hz.FIR_PATH_TRANSFORM
    ch = build(hz, ctx, hz.FMT_U8, rate, ops, taps_for(2048), 8)  # past the tap count where the direct form pays
    ch.set_time(1.0)
    ch.run(x, zeros("c64", n // 8))
    assert ch.last_fir_path() == hz.FIR_PATH_TRANSFORM
    ch.close()
    p, out_fft = run(8, impl=hz.FIR_IMPL_TRANSFORMS)  # hzsdr_chain_fir_options, in front of the terminal
    assert p == hz.FIR_PATH_TRANSFORM
    assert_fir_close(out_fft, want8, taps, xmax, "FIR_IMPL_TRANSFORMS")
    # ... and the chunk form of the matrix path where the persistent passes are the default
    ch = build(hz, ctx, hz.FMT_U8, rate, ops, taps, 8, hz.FIR_IMPL_MATRIX_CHUNKS)
    ch.set_time(1.0)
    out_chunks = zeros("c64", n // 8)
    ch.run(x, out_chunks)
    assert ch.last_fir_kernel() == hz.FIR_KERNEL_MATRIX_CHUNKS
    ch.close()
    assert_fir_close(out_chunks, want8, taps, xmax, "FIR_IMPL_MATRIX_CHUNKS")
    # the two implementations agree far inside their common bound
    d = out.astype(np.complex128) - out_fft
    assert np.linalg.norm(d) <= 3e-7 * np.linalg.norm(want8.astype(np.complex128))


def test_flat_filter_leaves_the_int32_plane_sum(hz, ctx, orc):
    """The persistent-pass kernel adds its two top digit planes in int32.  That sum holds for every input only while
    sqrt(2) * sum|h| stays below the bound of hz_firmm_plan.h (int32_combine_ok); a 1024-tap boxcar -- every tap at the
    maximum -- with the input that lines all signs up (bytes 0 and 255) would pass 2^31, so such a chain takes the
    chunk form of the matrix path, whose planes meet in float64, and the result stays inside the bound."""
    rate, D, n = 20_000_000, 8, 1 << 19
    flat = np.full(1024, (1 + 1j) / 1024, np.complex64)
    ch = ctx.chain(hz.FMT_U8, rate).fir_decimate(flat, D)
    for x in (np.zeros((n, 2), np.uint8), np.full((n, 2), 255, np.uint8), rand_u8(77, n)):
        out = zeros("c64", n // D)
        ch.reset()
        ch.run(x, out)
        assert ch.last_fir_path() == hz.FIR_PATH_MATRIX and ch.last_fir_kernel() == hz.FIR_KERNEL_MATRIX_CHUNKS
        want, xmax = oracle(orc, x, rate, [], flat, D)
        assert_fir_close(out, want, flat, xmax, "flat filter")
    ch.close()
    # an ordinary low-pass of the same length keeps the persistent passes
    ch = ctx.chain(hz.FMT_U8, rate).fir_decimate(taps_for(1024), D)
    ch.run(rand_u8(78, n), zeros("c64", n // D))
    assert ch.last_fir_kernel() == hz.FIR_KERNEL_MATRIX_PASSES
    ch.close()


def test_misaligned_device_buffers_take_the_transforms(hz, orc):
    import torch
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    rate, D, taps = 20_000_000, 8, taps_for(1024)
    n = 1 << 19
    x = rand_u8(9, n + 8)
    dx = torch.from_numpy(x).cuda()
    want, xmax = oracle(orc, x[1:n + 1], rate, [("shift", 1e6)], taps, D, ts0=1.0)
    ch = build(hz, ctx, hz.FMT_U8, rate, [("shift", 1e6)], taps, D)
    ch.set_time(1.0)
    out = torch.zeros(n // D, dtype=torch.complex64, device="cuda")
    ch.run(dx[1:n + 1], out)  # one sample into the allocation: 2-byte aligned only
    ctx.synchronize()
    assert ch.last_fir_path() == hz.FIR_PATH_TRANSFORM
    assert_fir_close(out.cpu().numpy(), want, taps, xmax, "misaligned")
    ch.close()
    ctx.close()


def fuzz_case(seed):
    r = np.random.default_rng(7000 + seed)
    fmt = ["u8", "i8"][seed % 2]
    D = [8, 16, 32, 64, 24, 40, 48][(seed // 2) % 7]
    ntaps = int(r.choice([16, 17, 63, 64, 65, 128, 255, 500, 777, 1024] + ([1025, 1400, 1536] if D == 8 else [1000, 1800, 2560] if D <= 24 else [1500, 3000, 4096])))
    rate = int(r.choice([250_000, 2_400_000, 20_000_000]))
    ops = []
    for _ in range(int(r.integers(0, 4))):
        kind = ["shift", "gain", "rotate"][int(r.integers(0, 3))]
        if kind == "shift":
            ops.append(("shift", float(r.uniform(-0.45, 0.45)) * rate))
        elif kind == "gain":
            ops.append(("gain", float(np.float32(r.uniform(0.1, 1.5)))))
        else:
            ops.append(("rotate", complex(np.complex64(np.exp(1j * r.uniform(0, TAU))))))
    # ragged calls on the decimation grid, some too short for the matrix form
    lens = [int(r.integers(3000, 60000 if D < 32 else 60000 * 16 // D)) * D for _ in range(int(r.integers(2, 5)))]
    if seed % 3 == 0:
        lens.insert(1, int(r.integers(10, 3000)) * D)
    # clock starts: inside a long binade, just under a binade edge, just under the 2*pi wrap
    ts0 = [1.3, 2.0 - 40000 / rate, TAU - 30000 / rate, 0.5 - 9000 / rate][seed % 4]
    return dict(fmt=fmt, D=D, taps=taps_for(ntaps, 0.4 / D, float(r.uniform(-0.5, 0.5))), rate=rate, ops=ops,
                lens=lens, ts0=float(ts0))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("HZ_FUZZ_SEEDS", "32"))))
def test_random_matrix_form_streams(hz, ctx, orc, seed):
    """Seeded random chains over byte sources at factor 8 / 16: tap counts around the tile and table
    edges, zero to three elementwise stages, ragged calls (some on the transform kernels, so the two
    histories alternate), clocks that cross a binade edge or the 2*pi wrap inside the stream."""
    c = fuzz_case(seed)
    D, taps, rate = c["D"], c["taps"], c["rate"]
    n = sum(c["lens"])
    x = (rand_u8 if c["fmt"] == "u8" else rand_i8)(8000 + seed, n)
    cuts = np.concatenate([[0], np.cumsum(c["lens"])]).tolist()
    want, xmax = oracle(orc, x, rate, c["ops"], taps, D, ts0=c["ts0"])
    ch = build(hz, ctx, hz.FMT_U8 if c["fmt"] == "u8" else hz.FMT_I8, rate, c["ops"], taps, D)
    ch.set_time(c["ts0"])
    out = zeros("c64", n // D)
    paths = set()
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert ch.run(x[a:b], out[a // D:b // D]) == (b - a, (b - a) // D)
        paths.add(ch.last_fir_path())
    if max(c["lens"]) // D >= 16384:  # (a short call that is mostly clock boundaries may stay on the transforms)
        assert hz.FIR_PATH_MATRIX in paths, (seed, c["lens"])
    assert_fir_close(out, want, taps, xmax, (seed, c["fmt"], D, len(taps), rate, c["ops"], c["lens"], c["ts0"]))
    ch.close()


def test_reset_and_interleaved_chains(hz, ctx, orc):
    """hzsdr_chain_reset drops both histories and the clock; two chains on one context keep their own
    state when their calls interleave."""
    rate, D = 2_400_000, 8
    n = 1 << 18
    xa, xb = rand_u8(41, 2 * n), rand_u8(42, 2 * n)
    ta, tb = taps_for(300), taps_for(777, 1 / 40, -0.2)
    opa, opb = [("shift", 2e5)], [("shift", -7e5), ("gain", 0.5)]
    wa, xma = oracle(orc, xa, rate, opa, ta, D, ts0=1.0)
    wb, xmb = oracle(orc, xb, rate, opb, tb, D, ts0=2.5)
    ca = build(hz, ctx, hz.FMT_U8, rate, opa, ta, D).set_time(1.0)
    cb = build(hz, ctx, hz.FMT_U8, rate, opb, tb, D).set_time(2.5)
    oa, ob = zeros("c64", 2 * n // D), zeros("c64", 2 * n // D)
    for k in range(2):  # a, b, a, b
        sl, so = slice(k * n, (k + 1) * n), slice(k * n // D, (k + 1) * n // D)
        assert ca.run(xa[sl], oa[so]) == (n, n // D)
        assert cb.run(xb[sl], ob[so]) == (n, n // D)
        assert ca.last_fir_path() == cb.last_fir_path() == hz.FIR_PATH_MATRIX
    assert_fir_close(oa, wa, ta, xma, "chain a")
    assert_fir_close(ob, wb, tb, xmb, "chain b")
    # reset: the same first half again, from a clean history and clock 0 -> set back to 1 s
    ca.reset()
    ca.set_time(1.0)
    again = zeros("c64", n // D)
    assert ca.run(xa[:n], again) == (n, n // D)
    assert again.tobytes() == oa[:n // D].tobytes()
    ca.close()
    cb.close()


def test_pipelined_chain_is_bit_identical_to_the_plain_one(hz):
    """hzsdr_chain_pipeline + hzsdr_chain_run_after: consecutive calls overlap (two streams of the chain's own, the next
    call's history from a kernel of its own), the same kernels on the same values -- every output bit as the unpipelined chain gives
    it, over forty calls across binades of the clock and the 2 pi wrap, with a change of the clock, a call too
    short for the matrix path (drains the pipeline) and a switch off and on again in between."""
    import torch
    n, fs, D = 1 << 20, 20_000_000, 8
    taps = taps_for(1024, 1 / 16, 0.0)
    ctxs = [hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.Stream().cuda_stream) for _ in range(2)]
    xs = [torch.from_numpy(rand_u8(300 + i, n)).cuda() for i in range(5)]
    torch.cuda.synchronize()
    outs = []
    for piped, ctx in zip((False, True), ctxs):
        ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D)
        if piped:
            ch.pipeline(True)
        ch.set_time(TAU - 0.6)
        # (the outputs exist, zeroed, before the first call: torch fills them on ITS stream, which nothing here waits for)
        sizes = [2048 if i == 11 else n for i in range(40)]
        ys = [torch.zeros(m // D, dtype=torch.complex64, device="cuda") for m in sizes]
        torch.cuda.synchronize()
        for i in range(40):
            if i == 17:
                ch.set_time(0.9)
            if i == 25 and piped:
                ch.pipeline(False)
            if i == 29 and piped:
                ch.pipeline(True)
            m = sizes[i]  # (call 11 is short: the transform kernels, behind everything in flight)
            run = ch.run_after if piped else ch.run  # (the overlap is run_after's: the buffers are ready now)
            assert run(xs[i % 5][:m], ys[i]) == (m, m // D)
        ctx.synchronize()
        outs.append([torch.view_as_real(y).view(torch.int32).cpu().numpy() for y in ys])
        assert ch.last_fir_path() == hz.FIR_PATH_MATRIX
        ch.close()
    for i, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b), "call %d differs" % i
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("sets", [2, 3, 5])
def test_pipelined_chain_over_a_small_rotation_of_buffers(hz, sets):
    """A caller that rotates two, three or five (input, output) sets under hzsdr_chain_run_after, each call waiting for
    the consumer of its output set's previous contents only: a call writes the output a call two, three or five places back wrote.  Two back is the chain's own stream,
    three back its OTHER stream (the call's stream waits for that kernel: csrc/hz_chain_fir.hip, pipeline_begin), one
    back and the history kernels are found by buffer span.  Each output is copied away on the context's stream right
    behind its call (ordered there, like any consumer) and compared with the plain chain's, bit for bit."""
    import torch
    n, fs, D = 1 << 20, 20_000_000, 8
    taps = taps_for(1024, 1 / 16, 0.0)
    res = []
    for piped in (False, True):
        s = torch.cuda.Stream()
        ctx = hz.Context(0, hz.MEM_DEVICE, stream=s.cuda_stream)
        ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D)
        if piped:
            ch.pipeline(True)
        ch.set_time(TAU - 0.3)
        xs = [torch.from_numpy(rand_u8(500 + i, n)).cuda() for i in range(sets)]
        ys = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(sets)]
        keep = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(24)]
        copied = [torch.cuda.Event() for _ in range(sets)]
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            for i in range(24):
                # (hzsdr_chain_run_after's contract: the call's buffers -- the OUTPUT too -- are free when the event has
                # fired.  The consumer of the set's previous output is the copy below, on the context's stream, which an
                # overlapped call is not ordered behind: the call is told.  Round 5's test passed "nothing to wait for"
                # here and raced that copy -- it lost once on a fast box in round 6, the tail of keep[0] held call 3's outputs.)
                if piped:
                    assert ch.run_after(xs[i % sets], ys[i % sets], copied[i % sets] if i >= sets else None) == (n, n // D)
                else:
                    assert ch.run(xs[i % sets], ys[i % sets]) == (n, n // D)
                keep[i].copy_(ys[i % sets])  # (on the context's stream: behind the call)
                copied[i % sets].record(s)
        ctx.synchronize()
        res.append([torch.view_as_real(k).view(torch.int32).cpu().numpy() for k in keep])
        ch.close()
        ctx.close()
    for i, (a, b) in enumerate(zip(*res)):
        assert np.array_equal(a, b), "call %d differs" % i


@pytest.mark.parametrize("fmt,D,shift", [("i16", 8, True), ("i16", 16, False), ("c64", 4, True), ("u8", 2, True), ("i16", 8, "mixed")])
def test_run_after_on_transform_chains_bit_identical(hz, fmt, D, shift):
    """The transform kernels' two-kernel form (analysis + synthesis: every source format and factor the matrix path does
    not take) under hzsdr_chain_pipeline + hzsdr_chain_run_after.  These calls run on the context's stream whatever the
    mode (an overlapped form was built in round 6, passed this test and was slower: csrc/hz_chain_fir.hip, fir_run);
    the entry point's contract holds all the same.  A rotation of three (input, output) sets, ragged and short calls
    among whole ones, a 2*pi wrap of the clock inside, and ('mixed') ordinary hzsdr_chain_run calls between the others:
    every output and the clock equal the plain chain's bit for bit."""
    import torch
    from util import rand_c64, rand_i16
    fs, sets, calls = 20_000_000, 3, 18
    taps = taps_for(1024 if D >= 8 else 256, 1 / (2 * D), 0.0)
    gen = {"i16": rand_i16, "u8": rand_u8, "c64": rand_c64}[fmt]
    F = {"i16": hz.FMT_I16, "u8": hz.FMT_U8, "c64": hz.FMT_C64}[fmt]
    n = 1 << 19
    sizes = [n, n, n - 16 * D * 37, n, 4096 * D, n, n, n - D, n] * 2
    res, clocks = [], []
    for piped in (False, True):
        s = torch.cuda.Stream()
        ctx = hz.Context(0, hz.MEM_DEVICE, stream=s.cuda_stream)
        ch = ctx.chain(F, fs)
        if shift:
            ch = ch.shift(-fs / 8)
        ch = ch.gain(0.75).fir_decimate(taps, D)
        if piped:
            ch.pipeline(True)
        ch.set_time(TAU - 0.05)
        xs = [torch.from_numpy(gen(700 + i, n)).cuda() for i in range(sets)]
        ys = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(sets)]
        keep = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(calls)]
        copied = [torch.cuda.Event() for _ in range(sets)]
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            for i in range(calls):
                m = sizes[i]
                x, y = xs[i % sets][:m], ys[i % sets][: m // D]
                if piped and not (shift == "mixed" and i % 5 == 3):
                    assert ch.run_after(x, y, copied[i % sets] if i >= sets else None) == (m, m // D)
                else:
                    assert ch.run(x, y) == (m, m // D)
                assert ch.last_fir_path() == hz.FIR_PATH_TRANSFORM
                keep[i][: m // D].copy_(y)  # (on the context's stream: behind the call)
                copied[i % sets].record(s)
        ctx.synchronize()
        clocks.append(ch.time())
        res.append([torch.view_as_real(k).view(torch.int32).cpu().numpy() for k in keep])
        ch.close()
        ctx.close()
    assert clocks[0] == clocks[1]
    for i, (a, b) in enumerate(zip(*res)):
        assert np.array_equal(a, b), "call %d differs" % i


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_pipelined_chain_random_calls_bit_identical(hz, seed):
    """Random call lengths (whole tiles and ragged, some below the matrix path's minimum), i8 and u8 sources, random
    changes of the clock: the pipelined chain's outputs equal the plain chain's bit for bit, and so does its clock."""
    import torch
    rng = np.random.default_rng(seed)
    fs, D = 20_000_000, 8
    fmt, gen = (hz.FMT_I8, rand_i8) if seed == 2 else (hz.FMT_U8, rand_u8)
    taps = taps_for(int(rng.choice([256, 640, 1024])), 1 / 16, 0.1)
    sizes = [int(rng.choice([1 << 18, 1 << 19, 3 * (1 << 17) + 8 * int(rng.integers(0, 4000)), 2048, 40_000])) for _ in range(30)]
    sizes = [m - m % D for m in sizes]
    resets = {int(i): float(rng.uniform(0, TAU)) for i in rng.integers(0, 30, 4)}
    src = torch.from_numpy(gen(500 + seed, 1 << 20)).cuda()
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    outs, clocks = [], []
    for piped in (False, True):
        ch = ctx.chain(fmt, fs).shift(3.1e6).gain(0.7).fir_decimate(taps, D)
        if piped:
            ch.pipeline(True)
        ch.set_time(TAU - 0.05)
        ys = [torch.zeros(m // D, dtype=torch.complex64, device="cuda") for m in sizes]
        torch.cuda.synchronize()
        off = 0
        for i, m in enumerate(sizes):
            if i in resets:
                ch.set_time(resets[i])
            if off + m > (1 << 20):
                off = 0
            run = ch.run_after if piped else ch.run
            assert run(src[off:off + m], ys[i]) == (m, m // D)
            off += m - m % 16  # (16-byte aligned starts: the matrix path's condition)
        ctx.synchronize()
        outs.append([torch.view_as_real(y).view(torch.int32).cpu().numpy() for y in ys])
        clocks.append(ch.time())
        ch.close()
    assert clocks[0] == clocks[1]
    for i, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b), "call %d (%d samples) differs" % (i, sizes[i])
    ctx.close()


@pytest.mark.parametrize("kind", ["c64_shift_gain", "u8_shift", "c64_shift_ulp1"])
def test_pipelined_map_chain_is_bit_identical(hz, kind):
    """hzsdr_chain_pipeline on a chain without a terminal: consecutive calls alternate between two streams (nothing on
    the device carries over; the clock is the host's) -- rotating buffers in, distinct buffers out, every bit as the
    plain chain's over thirty calls across the 2 pi wrap, with a ragged call and a clock change in between."""
    import torch
    from util import rand_c64
    n, fs = 1 << 19, 20_000_000
    fmt, gen = (hz.FMT_U8, rand_u8) if kind == "u8_shift" else (hz.FMT_C64, rand_c64)
    xs = [torch.from_numpy(gen(700 + i, n)).cuda() for i in range(4)]
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    outs, clocks = [], []
    for piped in (False, True):
        ch = ctx.chain(fmt, fs).shift(2.5e6)
        if kind == "c64_shift_gain":
            ch = ch.gain(0.5)
        if kind == "c64_shift_ulp1":
            ch = ch.shift_ulp1()
        if piped:
            ch.pipeline(True)
        ch.set_time(TAU - 0.3)
        sizes = [n - 3 if i == 7 else n for i in range(30)]
        ys = [torch.zeros(m, dtype=torch.complex64, device="cuda") for m in sizes]
        torch.cuda.synchronize()
        for i, m in enumerate(sizes):
            if i == 19:
                ch.set_time(1.9)
            run = ch.run_after if piped else ch.run
            assert run(xs[i % 4][:m], ys[i]) == (m, m)
        ctx.synchronize()
        outs.append([torch.view_as_real(y).view(torch.int32).cpu().numpy() for y in ys])
        clocks.append(ch.time())
        ch.close()
    assert clocks[0] == clocks[1]
    for i, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b), "call %d differs" % i
    ctx.close()


# ---- round 5: the ordering contract of the overlapped calls, and calls over several buffers -----------------------

def _north_chain(hz, ctx, taps, fs=20_000_000, D=8, fmt=None):
    return ctx.chain(hz.FMT_U8 if fmt is None else fmt, fs).shift(-fs / 8).fir_decimate(taps, D)


def test_pipelined_chain_keeps_the_context_streams_order(hz):
    """What round 4's mode got wrong and this round's contract fixes, on ONE input and ONE output buffer used by
    every call: the input is filled by an asynchronous copy enqueued on the context's stream right in front of each
    call, the output is copied away on that stream right behind it, nothing synchronises for 40 calls.
      * run() on a pipelined chain is an ordinary call: behind the copy, in front of the consumer;
      * run_after(event) with the producer on ANOTHER stream: the call waits for the event, and because it uses the
        buffers of the call before it the library orders it behind that call by itself;
      * run_after with three rotating buffer pairs and the consumer on the context's stream: the overlapped form.
    All three equal the plain chain bit for bit, and the plain chain is held to the oracle."""
    import torch
    import oracle as orc
    n, fs, D, calls = 1 << 19, 20_000_000, 8, 40
    taps = taps_for(1024, 1 / 16, 0.0)
    host = [torch.from_numpy(rand_u8(900 + i, n)).pin_memory() for i in range(calls)]
    s_ctx, s_copy = torch.cuda.Stream(), torch.cuda.Stream()
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=s_ctx.cuda_stream)
    results = {}
    for mode in ("plain", "run_on_stream", "after_same_buffers", "after_rotating"):
        ch = _north_chain(hz, ctx, taps)
        if mode != "plain":
            ch.pipeline(True)
        ch.set_time(TAU - 0.05)
        nbuf = 3 if mode == "after_rotating" else 1
        xin = [torch.zeros((n, 2), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
        yout = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
        stash = torch.zeros((calls, n // D), dtype=torch.complex64, device="cuda")
        evs = [torch.cuda.Event() for _ in range(nbuf)]
        consumed = [torch.cuda.Event() for _ in range(nbuf)]
        torch.cuda.synchronize()
        for i in range(calls):
            b = i % nbuf
            if mode in ("plain", "run_on_stream"):
                with torch.cuda.stream(s_ctx):
                    xin[b].copy_(host[i], non_blocking=True)
                assert ch.run(xin[b], yout[b]) == (n, n // D)
            else:
                with torch.cuda.stream(s_copy):
                    if i >= nbuf:
                        s_copy.wait_event(consumed[b])  # (the pair's previous use: read by its call, consumed behind it)
                    xin[b].copy_(host[i], non_blocking=True)
                    evs[b].record(s_copy)
                assert ch.run_after(xin[b], yout[b], evs[b]) == (n, n // D)
            with torch.cuda.stream(s_ctx):  # the consumer, on the context's stream: behind the call like behind any other
                stash[i].copy_(yout[b], non_blocking=True)
                consumed[b].record(s_ctx)
        ctx.synchronize()
        torch.cuda.synchronize()
        results[mode] = torch.view_as_real(stash).view(torch.int32).cpu().numpy()
        assert ch.last_fir_path() == hz.FIR_PATH_MATRIX
        ch.close()
    for mode in ("run_on_stream", "after_same_buffers", "after_rotating"):
        for i in range(calls):
            assert np.array_equal(results["plain"][i], results[mode][i]), (mode, "call %d differs" % i)
    x = np.concatenate([h.numpy() for h in host])
    want, xmax = oracle(orc, x, fs, [("shift", -fs / 8)], taps, D, ts0=TAU - 0.05)
    got = results["plain"].view(np.float32).view(np.complex64).reshape(-1)
    assert_fir_close(got, want, taps, xmax, "40 calls behind their copies")
    ctx.close()


@pytest.mark.parametrize("D", [8, 16])
@pytest.mark.parametrize("piped", [False, True])
def test_run_batch_one_launch_equals_the_stream(hz, piped, D):
    """hzsdr_chain_run_batch: k buffers of the stream in ONE launch of the persistent-pass kernel (separate
    allocations, the first pass of every buffer reaching back into the one before) -- against the oracle over the
    whole stream, and BIT FOR BIT the same stream through single calls (round 6: the mixer's phase and the choice
    between matrix path and fix-up task belong to the clock run's line, csrc/hz_firmm2_plan.h run_line, not to the
    call), batches of 4, 3, 1, 8 in a row across the clock's 2 pi wrap; a batch of buffers too short for whole passes
    runs one by one and equals single calls bit for bit as well."""
    import torch
    import oracle as orc
    # (D = 16: the persistent passes' other factor, 256 outputs per pass -- round 6.  Its buffers are twice as long: the
    # buffer that holds the 2 pi wrap must take the matrix path BY ITSELF for a call over several to take it -- a call
    # over several buffers promises the bits of single calls -- and the wrap's ~2 600 fix-up outputs are more than an
    # eighth of 2^18 / 16 outputs: the planner would keep that single call on the transform kernels)
    n, fs = (1 << 18) * (D // 8), 20_000_000
    taps = taps_for(1024, 1 / 16, 0.0)
    batches = [4, 3, 1, 8]
    total = sum(batches)
    x = rand_u8(77, n * total)
    xs = [torch.from_numpy(x[j * n:(j + 1) * n]).cuda() for j in range(total)]
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.Stream().cuda_stream)
    ts0 = TAU - 0.05  # (wraps 10^6 samples in)
    single = _north_chain(hz, ctx, taps, fs, D)
    single.set_time(ts0)
    ys = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(total)]
    torch.cuda.synchronize()
    for j in range(total):
        assert single.run(xs[j], ys[j]) == (n, n // D)
    ctx.synchronize()
    ref = np.concatenate([y.cpu().numpy() for y in ys])
    ch = _north_chain(hz, ctx, taps, fs, D)
    if piped:
        ch.pipeline(True)
    ch.set_time(ts0)
    zs = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(total)]
    torch.cuda.synchronize()
    j = 0
    for k in batches:
        assert ch.run_batch(xs[j:j + k], zs[j:j + k], after=piped) == (n, n // D)
        assert ch.last_fir_kernel() == hz.FIR_KERNEL_MATRIX_PASSES
        j += k
    ctx.synchronize()
    assert ch.time() == single.time()
    got = np.concatenate([z.cpu().numpy() for z in zs])
    want, xmax = oracle(orc, x, fs, [("shift", -fs / 8)], taps, D, ts0=ts0)
    assert_fir_close(got, want, taps, xmax, "batched stream")
    assert_fir_close(ref, want, taps, xmax, "single calls")
    assert bits_equal(got, ref), "a call over several buffers differs from single calls in %d outputs" % int((got.view(np.int64) != ref.view(np.int64)).sum())
    # the stream goes on in single calls from the batched chain's state: the history a batch leaves is the last buffer's
    tail_in = torch.from_numpy(rand_u8(78, n)).cuda()
    ya, yb = (torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(2))
    torch.cuda.synchronize()
    assert single.run(tail_in, ya) == (n, n // D) and ch.run(tail_in, yb) == (n, n // D)
    ctx.synchronize()
    assert torch.equal(torch.view_as_real(ya).view(torch.int32), torch.view_as_real(yb).view(torch.int32))
    # buffers that do not hold whole passes (n / D not a multiple of 512): one by one, the same bits as single calls
    m = n - D * 24
    ch.set_time(1.0), single.set_time(1.0)
    za = [torch.zeros(m // D, dtype=torch.complex64, device="cuda") for _ in range(3)]
    zb = [torch.zeros(m // D, dtype=torch.complex64, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    assert ch.run_batch([t[:m] for t in xs[:3]], za, after=piped) == (m, m // D)
    for q in range(3):
        assert single.run(xs[q][:m], zb[q]) == (m, m // D)
    ctx.synchronize()
    for q in range(3):
        assert torch.equal(torch.view_as_real(za[q]).view(torch.int32), torch.view_as_real(zb[q]).view(torch.int32)), q
    ch.close(), single.close(), ctx.close()


@pytest.mark.parametrize("nbuf,n", [(8, 4096), (2, 8192), (8, 8192), (3, 3 * 4096), (8, 1 << 15)])
def test_run_batch_of_very_short_buffers(hz, nbuf, n):
    """Buffers of ONE 512-output pass each (4096 samples at D = 8): the reciprocal the kernel finds a pass's buffer
    with does not exist for one pass per buffer (2^32 / 1 + 1 overflows: round 5 sent every pass to buffer 0 and wrote
    past its end -- ADVICE r05), so such a batch goes one by one; from two passes per buffer on it is one launch.
    Either way: the outputs equal single calls bit for bit, the guard zones behind every output stay untouched."""
    import torch
    fs, D = 20_000_000, 8
    taps = taps_for(1024, 1 / 16, 0.0)
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    xs = [torch.from_numpy(rand_u8(300 + j, n)).cuda() for j in range(nbuf)]
    guard = 4096
    outs = {}
    for form in ("batch", "single"):
        ch = _north_chain(hz, ctx, taps)
        ch.set_time(1.0)
        # (a first call of 2^16 samples: the batch then continues a clock run and a history)
        warm = torch.from_numpy(rand_u8(299, 1 << 16)).cuda()
        ch.run(warm, torch.zeros((1 << 16) // D, dtype=torch.complex64, device="cuda"))
        ys = [torch.full((n // D + guard,), 7.0 + 7.0j, dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
        if form == "batch":
            assert ch.run_batch(xs, [y[:n // D] for y in ys]) == (n, n // D)
        else:
            for j in range(nbuf):
                assert ch.run(xs[j], ys[j][:n // D]) == (n, n // D)
        ctx.synchronize()
        outs[form] = ([y.cpu().numpy() for y in ys], ch.time())
        ch.close()
    assert outs["batch"][1] == outs["single"][1]
    for j in range(nbuf):
        a, b = outs["batch"][0][j], outs["single"][0][j]
        assert np.all(a[n // D:] == np.complex64(7 + 7j)), ("guard zone of output %d written" % j)
        assert bits_equal(a, b), j
    ctx.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("HZ_FUZZ_SEEDS", "32"))))
def test_batches_equal_single_calls_on_random_streams(hz, seed):
    """hzsdr_chain_run_batch against the same buffers through hzsdr_chain_run, bit for bit, on random streams: sample
    rate (incl. a power-of-two one, whose clock steps never change), factor 8 / 16, 16 ... 1040 taps, 2 ... 8 buffers of
    16 ... 48 passes, one to three elementwise stages, the clock started a random distance in front of a binade edge or
    of the 2 pi wrap -- so that clock boundaries fall a few samples to a few thousand in front of, behind and onto
    buffer boundaries (round 6 found the one case the directed tests had missed that way: a boundary less than a window
    in front of a buffer's start, where the fix-up outputs read the history a single call keeps and a batch recomputes)
    -- plain and overlapped, two calls in a row.  Whatever path each form takes, the bits must agree."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    fs = int(rng.choice([2_400_000, 8_000_000, 20_000_000, 1_048_576, 200_000_000]))
    D = int(rng.choice([8, 16]))
    ntaps = int(rng.integers(16, 1041))
    taps = taps_for(ntaps, 1 / (2 * D), float(rng.uniform(-0.3, 0.3)))
    pass_out = 512 if D == 8 else 256
    ppb = int(rng.integers(4096 // pass_out, 49))
    n = ppb * pass_out * D
    nbuf = int(rng.integers(2, 9))
    piped = bool(rng.integers(0, 2))
    total = 2 * nbuf
    # the clock: a boundary (a binade edge or the wrap) somewhere in the stream, offset by anything from 0 to a few windows
    edge = TAU if rng.integers(0, 2) else float(2.0 ** -int(rng.integers(0, 8)))
    into = int(rng.integers(0, total * n))                       # the boundary falls `into` samples into the stream ...
    if rng.integers(0, 2):                                       # ... or right around a buffer boundary
        into = int(rng.integers(1, total)) * n + int(rng.integers(-3 * ntaps, 3 * ntaps))
    ts0 = edge - into / fs
    while ts0 < 0.0:
        ts0 += TAU
    ops = [("shift", float(rng.uniform(-0.4, 0.4)) * fs)]
    if rng.integers(0, 3) == 0:
        ops.append(("gain", 0.5))
    if rng.integers(0, 4) == 0:
        ops.insert(0, ("rotate", 0.6 - 0.8j))
    x = rand_u8(5000 + seed, n * total)
    xs = [torch.from_numpy(x[j * n:(j + 1) * n]).cuda() for j in range(total)]
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.Stream().cuda_stream)
    outs, clocks = [], []
    for form in ("single", "batch"):
        ch = build(hz, ctx, hz.FMT_U8, fs, ops, taps, D)
        if piped and form == "batch":
            ch.pipeline(True)
        ch.set_time(ts0)
        ys = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(total)]
        torch.cuda.synchronize()
        if form == "single":
            for j in range(total):
                assert ch.run(xs[j], ys[j]) == (n, n // D)
        else:
            for j in range(0, total, nbuf):
                assert ch.run_batch(xs[j:j + nbuf], ys[j:j + nbuf], after=piped) == (n, n // D)
        ctx.synchronize()
        torch.cuda.synchronize()
        outs.append(np.concatenate([y.cpu().numpy() for y in ys]))
        clocks.append(ch.time())
        ch.close()
    ctx.close()
    assert clocks[0] == clocks[1]
    bad = np.nonzero(outs[0].view(np.int64) != outs[1].view(np.int64))[0]
    assert len(bad) == 0, (len(bad), bad[:8], dict(fs=fs, D=D, ntaps=ntaps, n=n, nbuf=nbuf, piped=piped, ts0=ts0, ops=ops))


def test_run_batch_other_chains_and_errors(hz):
    """A chain without a one-launch form takes a batch buffer by buffer: the same bits as single calls (Shift + Gain map,
    a c64 FIR on the transform kernels, HOST space).  Argument errors: 0 or 9 buffers, a short output, ragged buffers."""
    import torch
    from util import rand_c64
    n, fs = 1 << 16, 20_000_000
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    xs = [torch.from_numpy(rand_c64(40 + j, n)).cuda() for j in range(3)]
    for kind in ("map", "fir_c64"):
        chains = []
        for _ in range(2):
            ch = ctx.chain(hz.FMT_C64, fs).shift(1.7e6).gain(0.25)
            if kind == "fir_c64":
                ch = ch.fir_decimate(taps_for(256, 1 / 16, 0.0), 4)
            chains.append(ch)
        no = n if kind == "map" else n // 4
        ya = [torch.zeros(no, dtype=torch.complex64, device="cuda") for _ in range(3)]
        yb = [torch.zeros(no, dtype=torch.complex64, device="cuda") for _ in range(3)]
        torch.cuda.synchronize()
        assert chains[0].run_batch(xs, ya) == (n, no)
        for j in range(3):
            assert chains[1].run(xs[j], yb[j]) == (n, no)
        ctx.synchronize()
        for j in range(3):
            assert torch.equal(torch.view_as_real(ya[j]).view(torch.int32), torch.view_as_real(yb[j]).view(torch.int32)), (kind, j)
        assert chains[0].time() == chains[1].time()
        with pytest.raises(hz.ErrDstTooSmall):
            chains[0].run_batch(xs, [y[:no - 1] for y in ya])
        for ch in chains:
            ch.close()
    ch = ctx.chain(hz.FMT_C64, fs).gain(2.0)
    with pytest.raises(Exception):
        ch.run_batch([xs[0]] * 9, [torch.zeros(n, dtype=torch.complex64, device="cuda") for _ in range(9)])
    ch.close()
    ctx.close()
    hctx = hz.Context(0, hz.MEM_HOST)
    x = [rand_u8(60 + j, 1 << 15) for j in range(2)]
    a, b = hctx.chain(hz.FMT_U8, fs).shift(1e6), hctx.chain(hz.FMT_U8, fs).shift(1e6)
    ya, yb = [zeros("c64", 1 << 15) for _ in range(2)], [zeros("c64", 1 << 15) for _ in range(2)]
    assert a.run_batch(x, ya) == (1 << 15, 1 << 15)
    for j in range(2):
        b.run(x[j], yb[j])
    assert all(p.tobytes() == q.tobytes() for p, q in zip(ya, yb))
    a.close(), b.close(), hctx.close()

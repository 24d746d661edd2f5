"""The shader clock the chip holds UNDER each of the library's kernels (tools/clock_witness.hip beside the work).

For every workload: the witness samples (shader cycles, 100 MHz ticks) every 20 us on a side stream for 64 ms; the
workload runs back to back on the main stream from about 1 ms in, for about 40 ms (HZ_WORK_US; HZ_TRACE=1 prints the
clock by tenths of the work: under the matrix kernel it is down within the first tenth of a 0.5 ms burst, under the
vector kernels it keeps sinking for milliseconds), between two marks that write the same 100 MHz counter.  Printed: the clock before the work, the median / lowest / highest clock of the samples taken in the SECOND
HALF of the work, and the time of one call by events over the whole of it.
Usage: python tools/clock_watch.py [workload ...]   (default: all)
"""
import ctypes, importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hz = importlib.import_module("go-sdr_amd")
from util import rand_c64

wit = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "libclock_witness.so"))
wit.clock_witness_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint]
wit.clock_witness_mark.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
ctx = hz.Context(0, hz.MEM_DEVICE, stream=main.cuda_stream)
n, fs = 1 << 24, 20_000_000
quiet = os.environ.get("HZ_QUIET_INPUT") == "1"  # constant input instead of random: the same instructions, quiet operands


def c64(seed):
    return torch.from_numpy(rand_c64(seed, n)).cuda() if not quiet else torch.full((n,), 0.25 + 0.5j, dtype=torch.complex64, device="cuda")


def u8(seed):
    if quiet:
        return torch.full((n, 2), 131, dtype=torch.uint8, device="cuda")
    return torch.from_numpy(np.random.default_rng(seed).integers(0, 256, (n, 2), dtype=np.uint8)).cuda()


def lowpass(ntaps, cut):
    k = np.arange(ntaps) - (ntaps - 1) / 2
    return (cut * np.sinc(cut * k) * np.hamming(ntaps)).astype(np.complex64)


def workloads():
    xs, outs = [c64(3 + i) for i in range(2)], [torch.zeros(n, dtype=torch.complex64, device="cuda") for _ in range(2)]
    us = [u8(11 + i) for i in range(4)]
    w = {}
    H = torch.from_numpy(np.fft.fft(np.asarray(lowpass(1024, 2 / 16), np.complex128) / 1024).astype(np.complex64)).cuda()
    w["convolution_blocks_1024"] = lambda i: ctx.convolution_blocks(outs[i & 1], xs[i & 1], H)
    ch_fir = ctx.chain(hz.FMT_C64, fs).fir_decimate(lowpass(1024, 2 / 16), 1)
    w["fir_1024_overlap_save_c64"] = lambda i: ch_fir.run(xs[i & 1], outs[i & 1])
    ch_sg = ctx.chain(hz.FMT_C64, fs).shift(2.5e6).gain(0.5)
    w["shift_gain_c64"] = lambda i: ch_sg.run(xs[i & 1], outs[i & 1])
    # the north-star chain as bench.py runs it: 1024 real taps, / 8; single calls, and four buffers per overlapped call
    taps = (2 / 16 * np.sinc(2 / 16 * (np.arange(1024) - 511.5)) * np.hamming(1024)).astype(np.float32)
    ch_ns = ctx.chain(hz.FMT_U8, fs).shift(2.5e6).fir_decimate(taps, 8)
    o8 = [torch.zeros(n // 8, dtype=torch.complex64, device="cuda") for _ in range(8)]
    w["north_star_chain_u8"] = lambda i: ch_ns.run(us[i & 3], o8[i & 3])
    ch_nb = ctx.chain(hz.FMT_U8, fs).shift(2.5e6).fir_decimate(taps, 8).pipeline(True)
    w["north_star_batch4_overlapped"] = lambda i: ch_nb.run_batch([us[j] for j in range(4)], [o8[4 * (i & 1) + j] for j in range(4)], after=True)
    pf = ctx.fft_plan(xs[0], outs[0], hz.FFT_FORWARD, batch=n >> 16)
    w["fft_64ki"] = lambda i: pf.transform()
    p4 = ctx.fft_plan(xs[0], outs[0], hz.FFT_FORWARD, batch=n >> 12)
    w["fft_4096"] = lambda i: p4.transform()
    w["copy"] = lambda i: outs[i & 1].copy_(xs[i & 1])
    return w


def watch(name, fn, samples=int(os.environ.get("HZ_SAMPLES", "3200")), period=int(os.environ.get("HZ_PERIOD_TICKS", "2000"))):
    buf = torch.zeros(2 * samples, dtype=torch.int64, device="cuda")
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(0); b.record(); torch.cuda.synchronize()
    one = a.elapsed_time(b) * 1e3
    reps = max(4, int(float(os.environ.get("HZ_WORK_US", "40000")) / one))
    rc = wit.clock_witness_launch(side.cuda_stream, buf.data_ptr(), samples, period)
    assert rc == 0, rc
    time.sleep(0.001)
    marks = torch.zeros(2, dtype=torch.int64, device="cuda")
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record(main)
    wit.clock_witness_mark(main.cuda_stream, marks.data_ptr())
    for i in range(reps):
        fn(i)
    wit.clock_witness_mark(main.cuda_stream, marks.data_ptr() + 8)
    s1.record(main)
    torch.cuda.synchronize()
    per_call = s0.elapsed_time(s1) * 1e3 / reps
    d = buf.cpu().numpy().reshape(-1, 2).astype(np.float64)
    m0, m1 = (float(v) for v in marks.cpu().numpy())
    mhz = np.diff(d[:, 0]) / np.diff(d[:, 1]) * 100.0  # sample k: the clock between ticks d[k, 1] and d[k + 1, 1]
    lo, hi = d[:-1, 1], d[1:, 1]
    inside = mhz[(lo >= 0.5 * (m0 + m1)) & (hi <= m1 - 0.02 * (m1 - m0))]
    before = mhz[hi <= m0]
    total = (m1 - m0) / 100.0
    assert len(inside) > 3 and len(before) > 3, (len(inside), len(before), "the witness ended before the work did")
    if os.environ.get("HZ_TRACE") == "1":  # the clock by tenths of the work's span (a short burst: HZ_WORK_US=700 HZ_PERIOD_TICKS=100)
        mid = 0.5 * (lo + hi)
        tenths = [mhz[(mid >= m0 + k * (m1 - m0) / 10) & (mid < m0 + (k + 1) * (m1 - m0) / 10)] for k in range(10)]
        print("    clock by tenths of the work: " + " ".join("%4.0f" % float(np.median(t)) if len(t) else "   -" for t in tenths) + " MHz")
    print("%-28s %8.1f us per call | clock before %5.0f MHz | under the work: median %5.0f, lowest %5.0f, highest %5.0f MHz (%d samples over %.0f us)"
          % (name, per_call, float(np.median(before)), float(np.median(inside)), float(inside.min()), float(inside.max()), len(inside), total))


if __name__ == "__main__":
    W = workloads()
    names = sys.argv[1:] or list(W)
    print("input: %s" % ("constant (quiet operands)" if quiet else "random"))
    for nm in names:
        watch(nm, W[nm])

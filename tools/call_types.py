"""The north-star chain call by kind: inside one clock run, across a binade boundary of the clock, across the 2 pi
wrap -- device time per call (events around the call; the clock is set in front of each)."""
import importlib, sys, os, math, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
hz = importlib.import_module("go-sdr_amd")
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
n, fs, D = 1 << 24, 20_000_000, 8
taps = B.lowpass_taps(1024, 1 / 16)
xs = [torch.from_numpy(B.synth_u8(9 + i, n)).cuda() for i in range(12)]
y = torch.zeros(n // D, dtype=torch.complex64, device="cuda")
ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D)
for i in range(400):
    ch.run(xs[i % 12], y)  # clocks up, tables of every binade built
def kind(ts0, reps=40):
    ms = []
    for i in range(reps):
        ch.set_time(1.1)
        ch.run(xs[(2 * i) % 12], y)  # a call in front: the history continues, the raw history is valid
        ch.set_time(ts0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ch.run(xs[(2 * i + 1) % 12], y); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ms)), float(np.min(ms))
for name, ts0 in (("inside one run (ts 1.1 -> 1.94)", 1.1), ("across the binade at 2.0", 1.6), ("across the binade at 4.0", 3.6),
                  ("across the binade at 1.0", 0.6), ("across the 2 pi wrap, mid-call", 2 * math.pi - 0.42), ("the call after the wrap (ts 0.1 ->)", 0.1)):
    m, lo = kind(ts0)
    print("%-40s median %.1f us  min %.1f us" % (name, m, lo))

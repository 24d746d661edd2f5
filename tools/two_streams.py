"""Two independent north-star chains on two HIP streams of one GPU, launched alternately, against one chain on one
stream: what the serialisation of a stream's consecutive launches costs (the next launch's workgroups wait for this
one's last; its head is paid in full)."""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
hz = importlib.import_module("go-sdr_amd")
n, fs, D = 1 << 24, 20_000_000, 8
taps = B.lowpass_taps(1024, 1 / 16)
xs = [torch.from_numpy(B.synth_u8(9 + i, n)).cuda() for i in range(12)]
streams = [torch.cuda.Stream() for _ in range(2)]
ctxs = [hz.Context(0, hz.MEM_DEVICE, stream=s.cuda_stream) for s in streams]
chains = [c.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D) for c in ctxs]
ys = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(2)]
torch.cuda.synchronize()
def run(k, which):
    for i in range(k):
        j = which[i % len(which)]
        chains[j].run(xs[i % 12], ys[j])
for which, name in (([0], "one stream"), ([0, 1], "two streams, alternating"), ([0], "one stream (again)"), ([0, 1], "two streams (again)")):
    run(1200, which)  # clocks up, tables built
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(900, which)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-28s %.2f us per call (host clock over 900 calls)" % (name, dt / 900 * 1e6))

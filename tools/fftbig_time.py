"""Times the two-step FFT (N = 2^14 .. 2^18, 2^24 points per call) forward and backward, one buffer pair."""
import importlib, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hz = importlib.import_module("go-sdr_amd")
from util import rand_c64
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
n = 1 << 24
x = torch.from_numpy(rand_c64(3, n)).cuda()
out = torch.zeros(n, dtype=torch.complex64, device="cuda")
def timed(f, k=60, w=60):
    for i in range(w): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
for lg in (14, 15, 16, 18, 20):
    pf = ctx.fft_plan(x, out, hz.FFT_FORWARD, batch=n >> lg)
    pb = ctx.fft_plan(x, out, hz.FFT_BACKWARD, batch=n >> lg)
    print("N = 2^%d: forward %.1f us, backward %.1f us per 2^24 points" % (lg, timed(pf.transform), timed(pb.transform)))

"""Times Scale, Rotate, Add and u8 -> c64 over 2^24 samples: one buffer (pair) and a rotation of six."""
import importlib, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hz = importlib.import_module("go-sdr_amd")
from util import rand_c64
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
n = 1 << 24
bufs = [torch.from_numpy(rand_c64(3 + i, n)).cuda() for i in range(6)]
outs = [torch.zeros(n, dtype=torch.complex64, device="cuda") for i in range(6)]
def timed(f, k=100, w=200):
    for i in range(w): f(i)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for i, (a, b) in enumerate(ev):
        a.record(); f(i); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
print("scale:  one buffer %.1f us, rotation %.1f us" % (timed(lambda i: ctx.scale(bufs[0], 0.5)), timed(lambda i: ctx.scale(bufs[i % 6], 0.5))))
print("rotate: one buffer %.1f us, rotation %.1f us" % (timed(lambda i: ctx.rotate(bufs[0], 0.6 + 0.8j)), timed(lambda i: ctx.rotate(bufs[i % 6], 0.6 + 0.8j))))
print("add:    one set %.1f us, rotation %.1f us" % (timed(lambda i: ctx.add(bufs[0], bufs[1], outs[0])), timed(lambda i: ctx.add(bufs[i % 6], bufs[(i + 1) % 6], outs[i % 6]))))

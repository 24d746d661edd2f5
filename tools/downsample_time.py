"""Downsample / 8 from i16 (BASELINE config 4) over 2^24 samples, a rotation of six buffer pairs (480 MiB of input):
per-call HIP events (SURVEY 8d's method), back to back between one event pair; / 16 and u8 / 8 beside it.  HZSDR_LIB
selects the library (A/B builds: csrc/Makefile EXTRA=-DHZ_DOWNSAMPLE_M=...)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hz = importlib.import_module("go-sdr_amd")
from util import rand_i16, rand_u8
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
n, K = 1 << 24, 6
def per_call(f, k=150, w=60):
    for i in range(w): f(i)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for i, (a, b) in enumerate(ev):
        a.record(); f(i); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
def back_to_back(f, k=150, w=60):
    for i in range(w): f(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(k): f(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3
print("library:", os.environ.get("HZSDR_LIB", "default"))
xi = [torch.from_numpy(rand_i16(4 + i, n)).cuda() for i in range(K)]
xu = [torch.from_numpy(rand_u8(14 + i, n)).cuda() for i in range(K)]
for rep in range(2):
    for name, xs, f, bps in (("i16 / 8", xi, 8, 4), ("i16 / 16", xi, 16, 4), ("u8 / 8", xu, 8, 2)):
        outs = [torch.zeros(n // f, dtype=torch.complex64, device="cuda") for _ in range(K)]
        fn = lambda i: ctx.downsample(outs[i % K], xs[i % K], f)
        pc, bb = per_call(fn), back_to_back(fn)
        by = n * (bps + 8.0 / f)
        print("downsample %-9s per call %5.1f us = %.3f of 8 TB/s   back to back %5.1f us = %.3f" % (name, pc, by / pc / 8e6, bb, by / bb / 8e6), flush=True)

"""Config 2 from HBM: Shift and Shift + Gain (bit-exact and <= 1-ulp forms) over 2^24 c64 samples, a rotation of six
buffer pairs (1.5 GiB: nothing of a call is left in the 256 MB memory-side cache for the next), per-call HIP events
(SURVEY 8d's method) and back to back between one event pair; Scale beside them.  HZSDR_LIB selects the library
(A/B builds: csrc/Makefile EXTRA=...)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hz = importlib.import_module("go-sdr_amd")
from util import rand_c64
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
n, fs = 1 << 24, 20_000_000
K = 6
bufs = [torch.from_numpy(rand_c64(3 + i, n)).cuda() for i in range(K)]
outs = [torch.zeros(n, dtype=torch.complex64, device="cuda") for i in range(K)]

def per_call(f, k=120, w=60):
    for i in range(w): f(i)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for i, (a, b) in enumerate(ev):
        a.record(); f(i); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3

def back_to_back(f, k=120, w=60):
    for i in range(w): f(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(k): f(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3

rows = []
def row(name, f):
    pc, bb = per_call(f), back_to_back(f)
    rows.append((name, pc, bb))
    print("%-34s per call %6.1f us = %.3f of 8 TB/s   back to back %6.1f us = %.3f" % (name, pc, 16 * n / pc / 8e6, bb, 16 * n / bb / 8e6), flush=True)

print("library:", os.environ.get("HZSDR_LIB", "default"))
for rep in range(2):
    row("scale in place", lambda i: ctx.scale(bufs[i % K], 0.999))
    for ulp1 in (False, True):
        ch = ctx.chain(hz.FMT_C64, fs).shift(2.5e6).gain(0.5)
        if ulp1: ch.shift_ulp1()
        row("shift+gain out of place" + (" (ulp1)" if ulp1 else ""), lambda i: ch.run(bufs[i % K], outs[i % K]))
        ch.close()
        ch = ctx.chain(hz.FMT_C64, fs).shift(2.5e6)
        if ulp1: ch.shift_ulp1()
        row("shift in place (chain)" + (" (ulp1)" if ulp1 else ""), lambda i: ch.run(bufs[i % K], bufs[i % K]))
        ch.close()
        nco = ctx.nco(fs)
        if ulp1: nco.set_ulp1(True)
        row("hzsdr_nco_shift in place" + (" (ulp1)" if ulp1 else ""), lambda i: nco(2.5e6, bufs[i % K]))
        nco.close()

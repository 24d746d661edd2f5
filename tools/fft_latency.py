"""One transform at a time (batch 1): the planner lengths of rtl/kerberos (64 Ki) and its graft (256 Ki), and the
cross-correlation closure kerberos runs per antenna pair (two forward transforms, a product, one backward)."""
import importlib, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hz = importlib.import_module("go-sdr_amd")
from util import rand_c64
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
def timed(f, k=200, w=50):
    for i in range(w): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(k): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / k
for lg in (12, 14, 16, 18, 20, 22, 24):
    n = 1 << lg
    x = torch.from_numpy(rand_c64(lg, n)).cuda()
    out = torch.zeros(n, dtype=torch.complex64, device="cuda")
    pf = ctx.fft_plan(x, out, hz.FFT_FORWARD)
    print("N = 2^%d, one transform per call: %.1f us back to back (%.2f GB/s of 16 B per point)" % (lg, timed(pf.transform), 16 * n / timed(pf.transform) / 1e3))

import importlib, sys, os, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
hz = importlib.import_module("go-sdr_amd")
from util import rand_c64
n, fs = 1 << 24, 20_000_000
bufs = [torch.from_numpy(rand_c64(3 + i, n)).cuda() for i in range(4)]
outs = [torch.zeros(n, dtype=torch.complex64, device="cuda") for i in range(4)]
streams = [torch.cuda.Stream() for _ in range(2)]
ctxs = [hz.Context(0, hz.MEM_DEVICE, stream=s.cuda_stream) for s in streams]
chains = [c.chain(hz.FMT_C64, fs).shift(2.5e6).gain(0.5) for c in ctxs]
torch.cuda.synchronize()
for which, name in (([0], "one stream"), ([0, 1], "two streams"), ([0], "one stream"), ([0, 1], "two streams")):
    for i in range(400): chains[which[i % len(which)]].run(bufs[i % 4], outs[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(600): chains[which[i % len(which)]].run(bufs[i % 4], outs[i % 4])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("shift+gain, rotation of 4 pairs: %-12s %.2f us per call" % (name, dt / 600 * 1e6))

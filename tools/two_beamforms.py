"""4-channel Beamform on one GPU: one stream against two contexts / streams launched alternately."""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hz = importlib.import_module("go-sdr_amd")
from util import rand_c64
n = 1 << 24
chans = [torch.from_numpy(rand_c64(5 + i, n)).cuda() for i in range(4)]
outs = [torch.zeros(n, dtype=torch.complex64, device="cuda") for _ in range(2)]
w = hz.beamform_angles(433e6, 30.0, [0.0, 0.1, 0.2, 0.3])
streams = [torch.cuda.Stream() for _ in range(2)]
ctxs = [hz.Context(0, hz.MEM_DEVICE, stream=s.cuda_stream) for s in streams]
torch.cuda.synchronize()
for which, name in (([0], "one stream"), ([0, 1], "two streams"), ([0], "one stream"), ([0, 1], "two streams")):
    for i in range(200): ctxs[which[i % len(which)]].beamform(outs[i % 2], chans, w)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(300): ctxs[which[i % len(which)]].beamform(outs[i % 2], chans, w)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("beamform 4 x 2^24: %-12s %.2f us per call" % (name, dt / 300 * 1e6))

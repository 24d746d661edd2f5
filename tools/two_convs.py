"""BASELINE config 3's two forms, one stream against two contexts / streams launched alternately (rotating buffers)."""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hz = importlib.import_module("go-sdr_amd")
from util import rand_c64
n, fs, ntaps = 1 << 24, 20_000_000, 1024
k = np.arange(ntaps) - (ntaps - 1) / 2
taps = (2 / 16 * np.sinc(2 / 16 * k) * np.hamming(ntaps)).astype(np.complex64)
bufs = [torch.from_numpy(rand_c64(3 + i, n)).cuda() for i in range(4)]
outs = [torch.zeros(n, dtype=torch.complex64, device="cuda") for i in range(4)]
H = torch.from_numpy(np.fft.fft(np.asarray(taps, np.complex128) / ntaps).astype(np.complex64)).cuda()
streams = [torch.cuda.Stream() for _ in range(2)]
ctxs = [hz.Context(0, hz.MEM_DEVICE, stream=s.cuda_stream) for s in streams]
chains = [c.chain(hz.FMT_C64, fs).fir_decimate(taps, 1) for c in ctxs]
torch.cuda.synchronize()
def run(f, which, name):
    for i in range(200): f(which[i % len(which)], i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(300): f(which[i % len(which)], i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-40s %.2f us per call" % (name, dt / 300 * 1e6))
for which, nm in (([0], "one stream"), ([0, 1], "two streams")):
    run(lambda j, i: ctxs[j].convolution_blocks(outs[i % 4], bufs[i % 4], H), which, "convolution blocks 1024, " + nm)
for which, nm in (([0], "one stream"), ([0, 1], "two independent chains on two streams")):
    run(lambda j, i: chains[j].run(bufs[i % 4], outs[i % 4]), which, "FIR overlap-save c64, " + nm)

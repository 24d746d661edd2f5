import importlib, sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
hz = importlib.import_module("go-sdr_amd")
from util import rand_c64
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
n, fs = 1 << 24, 20_000_000
bufs = [torch.from_numpy(rand_c64(3 + i, n)).cuda() for i in range(6)]
outs = [torch.zeros(n, dtype=torch.complex64, device="cuda") for i in range(6)]
def timed(f, k=120, w=30):
    for i in range(w): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(k): f(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
for name, mk in (("shift+gain", lambda: ctx.chain(hz.FMT_C64, fs).shift(2.5e6).gain(0.5)), ("shift", lambda: ctx.chain(hz.FMT_C64, fs).shift(2.5e6))):
    ch = mk(); print(name, "chain out of place, rotating %.1f us" % timed(lambda i: ch.run(bufs[i % 6], outs[i % 6]))); ch.close()
    ch = mk(); print(name, "chain in place, rotating     %.1f us" % timed(lambda i: ch.run(bufs[i % 6], bufs[i % 6]))); ch.close()
    ch = mk(); print(name, "chain out of place, one pair  %.1f us" % timed(lambda i: ch.run(bufs[0], outs[0]))); ch.close()
    ch = mk(); print(name, "chain in place, one buffer    %.1f us" % timed(lambda i: ch.run(bufs[0], bufs[0]))); ch.close()
nco = ctx.nco(fs)
print("nco in place, rotating %.1f us" % timed(lambda i: nco(2.5e6, bufs[i % 6])))
print("nco in place, one buffer %.1f us" % timed(lambda i: nco(2.5e6, bufs[0])))
nco.set_ulp1()
print("nco ulp1 in place, rotating %.1f us" % timed(lambda i: nco(2.5e6, bufs[i % 6])))
print("nco ulp1 in place, one buffer %.1f us" % timed(lambda i: nco(2.5e6, bufs[0])))

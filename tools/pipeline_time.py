"""The north-star chain, one call behind the other on one context: plain against hzsdr_chain_pipeline."""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
hz = importlib.import_module("go-sdr_amd")
n, fs, D = 1 << 24, 20_000_000, 8
taps = B.lowpass_taps(1024, 1 / 16)
xs = [torch.from_numpy(B.synth_u8(9 + i, n)).cuda() for i in range(12)]
ys = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(4)]
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for piped in (False, True, False, True):
    ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D)
    if piped:
        ch.pipeline(True)
    for i in range(1200):
        ch.run(xs[i % 12], ys[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(900):
        ch.run(xs[i % 12], ys[i % 4])
    th = time.perf_counter() - t0  # the host is through with its 900 calls
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-10s %.2f us per call (host clock over 900 calls); the host's own share %.2f us per call" % ("pipelined" if piped else "plain", dt / 900 * 1e6, th / 900 * 1e6))
    ch.close()

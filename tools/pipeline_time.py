"""The north-star chain, one call behind the other on one context: plain calls, overlapped calls (hzsdr_chain_pipeline +
hzsdr_chain_run_after), four buffers per call (hzsdr_chain_run_batch), and both."""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
hz = importlib.import_module("go-sdr_amd")
n, fs, D = 1 << 24, 20_000_000, 8
taps = B.lowpass_taps(1024, 1 / 16)
xs = [torch.from_numpy(B.synth_u8(9 + i, n)).cuda() for i in range(12)]
ys = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(8)]
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for piped, batch in ((False, 1), (True, 1), (False, 4), (True, 4), (False, 1), (True, 1), (False, 4), (True, 4)):
    ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D)
    if piped:
        ch.pipeline(True)

    def go(calls):
        for c in range(calls):
            i = c * batch
            if batch == 1:
                (ch.run_after if piped else ch.run)(xs[i % 12], ys[i % 8])
            else:
                ch.run_batch([xs[(i + j) % 12] for j in range(batch)], [ys[(i + j) % 8] for j in range(batch)], after=piped)

    go(1200 // batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(900 // batch)
    th = time.perf_counter() - t0  # the host is through with its calls
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-10s %d buffer(s) per call: %.2f us per BUFFER (host clock over 900 buffers); the host's own share %.2f us per buffer"
          % ("overlapped" if piped else "plain", batch, dt / 900 * 1e6, th / 900 * 1e6))
    ch.close()

"""Times the two convolution forms of BASELINE config 3 over one buffer pair and over a rotation of four."""
import importlib, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hz = importlib.import_module("go-sdr_amd")
from util import rand_c64
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
n, fs, ntaps = 1 << 24, 20_000_000, 1024
k = np.arange(ntaps) - (ntaps - 1) / 2
taps = (2 / 16 * np.sinc(2 / 16 * k) * np.hamming(ntaps)).astype(np.complex64)
bufs = [torch.from_numpy(rand_c64(3 + i, n)).cuda() for i in range(4)]
outs = [torch.zeros(n, dtype=torch.complex64, device="cuda") for i in range(4)]
H = torch.from_numpy(np.fft.fft(np.asarray(taps, np.complex128) / ntaps).astype(np.complex64)).cuda()
def timed(f, k=100, w=200):
    for i in range(w): f(i)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for i, (a, b) in enumerate(ev):
        a.record(); f(i); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
print("convolution_blocks 1024: one pair %.1f us, rotation %.1f us" % (timed(lambda i: ctx.convolution_blocks(outs[0], bufs[0], H)), timed(lambda i: ctx.convolution_blocks(outs[i % 4], bufs[i % 4], H))))
ch = ctx.chain(hz.FMT_C64, fs).fir_decimate(taps, 1)
print("fir overlap-save c64:    one pair %.1f us, rotation %.1f us" % (timed(lambda i: ch.run(bufs[0], outs[0])), timed(lambda i: ch.run(bufs[i % 4], outs[i % 4]))))
# the same FIR on 8192-point overlap-save blocks (7169 outputs each: 1.14x redundant transform work instead of 1.33x, one pass more)
ch8 = ctx.chain(hz.FMT_C64, fs).fir_options(hz.FIR_IMPL_TRANSFORMS, nfft_min=8192).fir_decimate(taps, 1)
print("fir overlap-save c64, N 8192: one pair %.1f us, rotation %.1f us" % (timed(lambda i: ch8.run(bufs[0], outs[0])), timed(lambda i: ch8.run(bufs[i % 4], outs[i % 4]))))

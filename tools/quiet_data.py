"""Does a kernel's time depend on its DATA?  Every streaming / transform kernel over 2^24 samples, a rotation of four
buffer pairs, on random samples and on constant ones -- the same instructions and bytes; a difference is the clock the
chip holds under toggling operands (round 6: the int8 matrix kernel loses a quarter)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hz = importlib.import_module("go-sdr_amd")
import bench as B
from util import rand_c64, rand_i16, rand_u8
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
n, fs, K = 1 << 24, 20_000_000, 4

def b2b(f, k=80, w=40):
    for i in range(w): f(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(k): f(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3

taps = B.lowpass_taps(1024, 1 / 16)
H = torch.from_numpy(np.fft.fft(np.asarray(taps, np.complex128) / 1024).astype(np.complex64)).cuda()
rows = []
for kind in ("random", "constant"):
    if kind == "random":
        c = [torch.from_numpy(rand_c64(3 + i, n)).cuda() for i in range(K)]
        u = [torch.from_numpy(rand_u8(13 + i, n)).cuda() for i in range(K)]
        s16 = [torch.from_numpy(rand_i16(23 + i, n)).cuda() for i in range(K)]
    else:
        c = [torch.full((n,), 0.25 - 0.5j, dtype=torch.complex64, device="cuda") for i in range(K)]
        u = [torch.full((n, 2), 0x80, dtype=torch.uint8, device="cuda") for i in range(K)]
        s16 = [torch.full((n, 2), 1000, dtype=torch.int16, device="cuda") for i in range(K)]
    o = [torch.zeros(n, dtype=torch.complex64, device="cuda") for i in range(K)]
    o8 = [torch.zeros(n // 8, dtype=torch.complex64, device="cuda") for i in range(K)]
    r = {}
    r["scale in place"] = b2b(lambda i: ctx.scale(c[i % K], 0.999))
    ch = ctx.chain(hz.FMT_C64, fs).shift(2.5e6).gain(0.5)
    r["shift+gain (exact)"] = b2b(lambda i: ch.run(c[i % K], o[i % K])); ch.close()
    ch = ctx.chain(hz.FMT_C64, fs).shift(2.5e6).gain(0.5).shift_ulp1()
    r["shift+gain (ulp1)"] = b2b(lambda i: ch.run(c[i % K], o[i % K])); ch.close()
    r["u8 -> c64"] = b2b(lambda i: ctx.convert(o[i % K], u[i % K]))
    r["downsample i16 / 8"] = b2b(lambda i: ctx.downsample(o8[i % K], s16[i % K], 8))
    r["block convolution 1024"] = b2b(lambda i: ctx.convolution_blocks(o[i % K], c[i % K], H), 40, 20)
    ch = ctx.chain(hz.FMT_C64, fs).fir_decimate(taps, 1)
    r["overlap-save FIR 1024 taps c64"] = b2b(lambda i: ch.run(c[i % K], o[i % K]), 40, 20); ch.close()
    ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_options(hz.FIR_IMPL_TRANSFORMS).fir_decimate(taps, 8)
    r["north-star chain, transform kernels"] = b2b(lambda i: ch.run(u[i % K], o8[i % K]), 40, 20); ch.close()
    ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, 8)
    r["north-star chain, matrix kernel"] = b2b(lambda i: ch.run(u[i % K], o8[i % K]), 40, 20); ch.close()
    chans = c
    wts = hz.beamform_angles(433e6, 30.0, [0.0, 0.1, 0.2, 0.3])
    r["beamform 4 channels"] = b2b(lambda i: ctx.beamform(o[i % K], chans, wts), 40, 20)
    pin = ctx.fft_plan(c[0].view(-1, 4096), o[0].view(-1, 4096), hz.FFT_FORWARD) if False else None
    rows.append(r)
    del c, u, s16, o, o8
print("%-38s %10s %10s %7s" % ("kernel (2^24 samples, back to back)", "random us", "const us", "ratio"))
for k in rows[0]:
    print("%-38s %10.1f %10.1f %7.3f" % (k, rows[0][k], rows[1][k], rows[0][k] / rows[1][k]))

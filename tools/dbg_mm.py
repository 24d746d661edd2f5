import importlib, sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from util import rand_u8, zeros
hz = importlib.import_module("go-sdr_amd")
import oracle.oracle as orc
ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
n, fs, D = 1 << 24, 20_000_000, 8
k = np.arange(1024) - 1023 / 2
taps = (2 / 16 * np.sinc(2 / 16 * k) * np.hamming(1024)).astype(np.complex64)
x = rand_u8(9, n)
xc = zeros("c64", n); orc.convert(xc, x); sh = orc.Shifter(fs); sh(-fs / 8, xc)
want = zeros("c64", n // D); orc.par_fir_decimate_f64(want, xc, taps, D)
dx = torch.from_numpy(x).cuda()
ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D)
out = torch.zeros(n // D, dtype=torch.complex64, device="cuda")
os.environ["HZ_DEBUG_MM"] = "1"
ch.run(dx, out); ctx.synchronize()
got = out.cpu().numpy()
err = np.abs(got.astype(np.complex128) - want)
bad = np.nonzero(~(err < 1e-5))[0]
print("bad outputs:", len(bad))
if len(bad):
    # group into ranges
    br = np.split(bad, np.nonzero(np.diff(bad) > 1)[0] + 1)
    for b in br[:20]: print("  [%d, %d) len %d  chunk %d off %d  first err %g" % (b[0], b[-1] + 1, len(b), b[0] // 2048, b[0] % 2048, err[b[0]]))
    b = br[0]
    print("got", got[b[0]-2:b[0]+4]); print("want", want[b[0]-2:b[0]+4])

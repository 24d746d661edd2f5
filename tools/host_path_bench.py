#!/usr/bin/env python3
"""PCIe-inclusive rates: the same ops through a HZSDR_MEM_HOST context (numpy
buffers = what a cgo caller with Go slices hands over: pageable memory, staged
H2D -> kernel -> D2H inside every call).  Never the bench `value`; DESIGN.md
section 5 quotes these next to the HBM-resident numbers."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def best(fn, reps=5):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t)


def main():
    hz = importlib.import_module("go-sdr_amd")
    ctx = hz.Context(0, hz.MEM_HOST)
    n = 1 << 24
    xu8 = B.synth_u8(9, n)
    xc = B.synth_c64(2, n)
    out = np.zeros(n, np.complex64)
    o8 = np.zeros(n // 8, np.complex64)
    taps = B.lowpass_taps(1024, 1 / 16)
    rows = []
    t = best(lambda: ctx.convert(out, xu8))
    rows.append(("u8->c64 ConvertBuffer", n, t, 10))
    nco = ctx.nco(20_000_000)
    t = best(lambda: nco(2.5e6, xc))
    rows.append(("Shift (in place)", n, t, 16))
    ch = ctx.chain(hz.FMT_U8, 20_000_000).shift(-2.5e6).fir_decimate(taps, 8)
    t = best(lambda: ch.run(xu8, o8))
    rows.append(("north-star chain u8 in, c64/8 out", n, t, 3))
    for name, n_, t, bps in rows:
        print(f"{name:36s} {t * 1e3:8.2f} ms  {n_ / t / 1e6:9.1f} Msamples/s  {bps * n_ / t / 1e9:6.1f} GB/s over PCIe")
    ctx.close()


if __name__ == "__main__":
    main()

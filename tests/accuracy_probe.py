#!/usr/bin/env python3
"""Relative L2 / max-abs error of the FIR-decimate chain against the oracle, both mixer
orders, and the distance between the two orders (tests/util.py bounds: 6e-7, 3e-7)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # (tests/ -> the repository)
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as orc  # noqa: E402
from util import fir_errors, rand_c64, rand_u8  # noqa: E402


def main():
    import torch
    hz = importlib.import_module("go-sdr_amd")
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    fs = 20_000_000
    for name, fmt, ntaps, D, n, shift in (("north star", "u8", 1024, 8, 1 << 22, -fs / 8),
                                          ("cfg3 D=1", "c64", 1024, 1, 1 << 20, 2.5e6),
                                          ("2047 taps D=4", "u8", 2047, 4, 1 << 21, 1e6),
                                          ("300 taps D=8", "u8", 300, 8, 1 << 20, 3e6)):
        k = np.arange(ntaps) - (ntaps - 1) / 2
        taps = (np.sinc(k / 16) / 16 * np.hamming(ntaps)).astype(np.complex64)
        x = rand_u8(9, n) if fmt == "u8" else rand_c64(9, n)
        xc = np.zeros(n, np.complex64)
        if fmt == "u8":
            orc.convert(xc, x)
        else:
            xc[:] = x
        orc.Shifter(fs)(shift, xc)
        want = np.zeros(n // D, np.complex64)
        orc.par_fir_decimate_f64(want, xc, taps, D)
        outs = []
        for in_order in (False, True):
            ch = ctx.chain(hz.FMT_U8 if fmt == "u8" else hz.FMT_C64, fs).shift(shift).fir_decimate(taps, D)
            ch.mix_in_order(in_order)
            y = torch.zeros(n // D, dtype=torch.complex64, device="cuda")
            ch.run(torch.from_numpy(x).cuda(), y)
            ctx.synchronize()
            got = y.cpu().numpy()
            err, bound, rel = fir_errors(got, want, taps, float(np.abs(xc).max()))
            print(f"{name:16s} in_order={in_order!s:5s} max_abs {err:.3e} (bound {bound:.3e})  rel_l2 {rel:.3e}")
            outs.append(got.astype(np.complex128))
            ch.close()
        print(f"{name:16s} late vs in-order rel_l2 {np.linalg.norm(outs[0] - outs[1]) / np.linalg.norm(want.astype(np.complex128)):.3e}")
    ctx.close()


if __name__ == "__main__":
    main()

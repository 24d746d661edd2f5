#!/usr/bin/env python3
"""Kernel durations of the north-star chain on a buffer whose clock crosses the 2*pi wrap
(run under rocprofv3 --kernel-trace).  HZ_NO_SLOW_FIRST=1 keeps the analysis blocks in stream order."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def main():
    import torch
    hz = importlib.import_module("go-sdr_amd")
    s = torch.cuda.Stream()
    torch.cuda.set_stream(s)
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=s.cuda_stream)
    n, fs, D = 1 << 24, 20_000_000, 8
    taps = B.lowpass_taps(1024, 1 / 16)
    x = torch.from_numpy(B.synth_u8(9, n)).cuda()
    y = torch.zeros(n // D, dtype=torch.complex64, device="cuda")
    ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D)
    for frac in (0.1, 0.5, 0.9, 0.5, 0.5):
        ch.set_time(6.283185307179586 - frac * n / fs)  # the wrap falls at `frac` of the buffer
        ch.run(x, y)
    ch.set_time(1.0)
    for _ in range(3):
        ch.run(x, y)
    torch.cuda.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()

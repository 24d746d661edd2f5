#!/usr/bin/env python3
"""A few blocks around the clock's 2*pi wrap: the slowest block's latency (rocprofv3 --kernel-trace)."""
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
    fs, D = 20_000_000, 8
    taps = B.lowpass_taps(1024, 1 / 16)
    for nblk, wrap_at in ((16, None), (16, 8), (64, 32), (64, None), (8, 4), (8, 1), (8, 7)):
        n = 3072 * nblk
        x = torch.from_numpy(B.synth_u8(9, n)).cuda()
        y = torch.zeros(n // D, dtype=torch.complex64, device="cuda")
        ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D)
        for _ in range(3):
            ch.set_time(1.0 if wrap_at is None else 6.283185307179586 - wrap_at * 3072 / fs)
            ch.run(x, y)
        ch.close()
    torch.cuda.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Latency of ONE overlap-save block: the north-star chain over buffers of a few blocks
(every block on its own CU), late mixer and reference order (rocprofv3 --kernel-trace)."""
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
    for nblk in (16, 256, 1024):
        n = 3072 * nblk
        x = torch.from_numpy(B.synth_u8(9, n)).cuda()
        y = torch.zeros(n // D, dtype=torch.complex64, device="cuda")
        for in_order in (False, True):
            ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D).mix_in_order(in_order)
            ch.set_time(1.0)
            for _ in range(4):
                ch.run(x, y)
            ch.close()
    torch.cuda.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()

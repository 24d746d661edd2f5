#!/usr/bin/env python3
"""Per-call device time of the north-star chain on 12 rotating 2^24-sample buffers (the bench's
timed loop), for the int8 matrix form and -- PROBE_IMPL=1: the transform kernels, 2: the chunk form -- ; the second
half of the calls is printed one by one (a clock wrap shows as a slower call)."""
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
    n, fs = 1 << 24, 20_000_000
    D = int(os.environ.get("PROBE_D", "8"))
    ntaps = int(os.environ.get("PROBE_TAPS", "1024"))
    taps = B.lowpass_taps(ntaps, 0.5 / D)
    xs = [torch.from_numpy(B.synth_u8(9 + i, n)).cuda() for i in range(12)]
    y = torch.zeros(n // D, dtype=torch.complex64, device="cuda")
    ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_options(int(os.environ.get("PROBE_IMPL", "0"))).fir_decimate(taps, D)
    for i in range(6):
        ch.run(xs[i % 12], y)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record(s)
    for i in range(40):
        ch.run(xs[i % 12], y)
        ev[i + 1].record(s)
    torch.cuda.synchronize()
    per = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(40)]
    kern = {hz.FIR_KERNEL_TRANSFORM: "transform kernels", hz.FIR_KERNEL_MATRIX_CHUNKS: "int8 matrix form, chunk workgroups (hz_firmm.h)",
            hz.FIR_KERNEL_MATRIX_PASSES: "int8 matrix form, persistent passes (hz_firmm2.h)"}.get(ch.last_fir_kernel(), "?")
    print("path:", kern,
          "D", D, "taps", ntaps)
    print("mean %.1f us  median %.1f us  min %.1f us  max %.1f us" %
          (sum(per) / len(per), sorted(per)[len(per) // 2], min(per), max(per)))
    print("calls:", " ".join("%.0f" % p for p in per))
    ctx.close()


if __name__ == "__main__":
    main()

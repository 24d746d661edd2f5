#!/usr/bin/env python3
"""Runs a chosen hot-path kernel a few times on 2^24-sample buffers, and nothing
else, so a rocprofv3 pass (kernel trace or PMC counters) sees a clean stream.
The streaming kernels (convert, scale, rotate, shift, shift_gain, downsample, conv, fir_c64_d1, beamform) walk a
ROTATION of `ROT` buffer sets (default 4: 0.5-1 GiB, more than the 256 MB memory-side cache holds), so their rows
are from-HBM figures like `extra.*.hbm` of bench.py -- the forms the library picks for calls of this size
(non-temporal accesses) are built for that case; ROT=1 is the one-pair form of rounds 1-4.

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/prof_kernels.py chain conv
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d out -- python3 tools/prof_kernels.py chain
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402  (synthetic generators)


def main():
    import torch
    which = sys.argv[1:] or ["chain"]
    reps = int(os.environ.get("REPS", "5"))
    rot = max(1, int(os.environ.get("ROT", "4")))
    turn = [0]

    def nxt():
        turn[0] += 1
        return turn[0] % rot
    log2n = int(os.environ.get("LOG2N", "24"))
    hz = importlib.import_module("go-sdr_amd")
    s = torch.cuda.Stream()
    torch.cuda.set_stream(s)
    ctx = hz.Context(0, hz.MEM_DEVICE, stream=s.cuda_stream)
    n, fs, D = 1 << log2n, 20_000_000, 8
    taps = B.lowpass_taps(1024, 1 / 16)
    xu8 = torch.from_numpy(B.synth_u8(9, n)).cuda()
    xc = torch.from_numpy(B.synth_c64(2, n)).cuda()
    out = torch.zeros(n, dtype=torch.complex64, device="cuda")
    need_rot = any(w in ("convert", "scale", "rotate", "shift", "shift_gain", "downsample", "conv", "fir_c64_d1") for w in which)
    xcs = [xc] + [xc.clone() for _ in range(rot - 1)] if need_rot else [xc]
    outs = [out] + [torch.zeros_like(out) for _ in range(rot - 1)] if need_rot else [out]
    xu8s = [xu8] + [xu8.clone() for _ in range(rot - 1)] if "convert" in which else [xu8]
    for w in which:
        if w == "chain":
            ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D)
            fn = lambda: ch.run(xu8, out[:n // D])
        elif w == "chain_batch4":  # the benchmarked form: four consecutive buffers per call, ONE launch (hzsdr_chain_run_batch)
            ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_decimate(taps, D)
            xb = [xu8] + [torch.from_numpy(B.synth_u8(10 + i, n)).cuda() for i in range(3)]
            ob = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(4)]
            fn = lambda: ch.run_batch(xb, ob)
        elif w == "chain_fft":  # the same chain on the overlap-save transform kernels
            ch = ctx.chain(hz.FMT_U8, fs).shift(-fs / 8).fir_options(hz.FIR_IMPL_TRANSFORMS).fir_decimate(taps, D)
            fn = lambda: ch.run(xu8, out[:n // D])
        elif w == "chain_noshift":
            ch = ctx.chain(hz.FMT_U8, fs).fir_decimate(taps, D)
            fn = lambda: ch.run(xu8, out[:n // D])
        elif w == "chain_c64":
            ch = ctx.chain(hz.FMT_C64, fs).fir_decimate(taps, D)
            fn = lambda: ch.run(xc, out[:n // D])
        elif w == "fir_c64_d1":  # BASELINE config 3 in its north-star form: 1024 taps, no decimation
            ch = ctx.chain(hz.FMT_C64, fs).fir_decimate(taps, 1)
            def fn():
                i = nxt() % len(xcs)
                ch.run(xcs[i], outs[i])
        elif w == "conv":
            H = torch.from_numpy(np.fft.fft(np.asarray(taps, np.complex128) / 1024).astype(np.complex64)).cuda()

            def fn():
                i = nxt() % len(xcs)
                ctx.convolution_blocks(outs[i], xcs[i], H)
        elif w == "fft1024":
            p = ctx.fft_plan(xc, out, hz.FFT_FORWARD, batch=n // 1024)
            fn = p.transform
        elif w == "fft4096":
            p = ctx.fft_plan(xc, out, hz.FFT_FORWARD, batch=n // 4096)
            fn = p.transform
        elif w.startswith("fftbig"):  # fftbig14, fftbig16, ...: one transform size, 2^24 points in all
            lg = int(w[6:])
            p = ctx.fft_plan(xc, out, hz.FFT_FORWARD, batch=n >> lg)
            # Context.fft_plan(batch=) takes the block length from len // batch
            fn = p.transform
        elif w == "shift":
            nco = ctx.nco(fs)
            fn = lambda: nco(2.5e6, xcs[nxt() % len(xcs)])
        elif w == "shift_gain":
            ch = ctx.chain(hz.FMT_C64, fs).shift(2.5e6).gain(0.5)

            def fn():
                i = nxt() % len(xcs)
                ch.run(xcs[i], outs[i])
        elif w == "convert":
            def fn():
                i = nxt()
                ctx.convert(outs[i % len(outs)], xu8s[i % len(xu8s)])
        elif w == "scale":
            fn = lambda: ctx.scale(xcs[nxt() % len(xcs)], 0.999)
        elif w == "rotate":
            fn = lambda: ctx.rotate(xcs[nxt() % len(xcs)], 0.6 + 0.8j)
        elif w == "beamform":
            chans = [torch.from_numpy(B.synth_c64(5 + i, n)).cuda() for i in range(4)]
            wts = hz.beamform_angles(433e6, 30.0, [0.0, 0.1, 0.2, 0.3])
            fn = lambda: ctx.beamform(out, chans, wts)
        elif w == "downsample":
            xis = [torch.from_numpy(B.synth_i16(4, n)).cuda()]
            xis += [xis[0].clone() for _ in range(rot - 1)]

            def fn():
                i = nxt()
                ctx.downsample(outs[i % len(outs)][:n // 8], xis[i % len(xis)], 8)
        else:
            raise SystemExit(f"unknown kernel {w}")
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()

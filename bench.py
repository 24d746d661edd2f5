#!/usr/bin/env python3
"""bench.py -- Msamples/s of the IQ hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the north-star chain (u8 -> c64 -> Shift(-fs/8) -> 1024-tap
FIR -> decimate-by-8: one hzsdr_chain_run, no full-rate c64 intermediate in HBM)
over one 2^24-sample synthetic buffer already resident in HBM.  With N > 1 every
rank runs its own independent stream ("single-stream chains stay on one GPU": replicas, weak scaling, no data-path
collective) and the line also carries the 4-channel Beamform measurement sharded
over the ranks with its RCCL exchange (the other half of the metric).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against HBM
peak with its algorithmic bytes (3 B per input sample) and the
HIP-event duration measured inside the timed region; `cpu_baseline` times the
oracle (a scalar port of the reference algorithms) on a bounded sample.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3    # vector f32
I8_PEAK_TOPS = 5000.0       # MI355X_MICROARCH.md: dense int8 MFMA, 2x the ~2.5 PF of bf16


def splitmix64(seed, n):
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def synth_u8(seed, n):
    return (splitmix64(seed, 2 * n) >> np.uint64(56)).astype(np.uint8).reshape(n, 2)


def synth_i16(seed, n):
    # Gaussian sigma 4096, clipped (SURVEY 8d cfg 4) from two uniform words (Box-Muller)
    u = (splitmix64(seed, 4 * n) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    g = np.sqrt(-2.0 * np.log(u[0::2] + 1e-300)) * np.cos(2 * np.pi * u[1::2])
    return np.clip(np.rint(g * 4096), -32768, 32767).astype(np.int16).reshape(n, 2)


def synth_c64(seed, n):
    u = (splitmix64(seed, 2 * n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (u * 2.0 - 1.0).astype(np.float32).view(np.complex64).reshape(n)


def lowpass_taps(ntaps, cutoff_frac):
    """windowed-sinc low-pass, cutoff = cutoff_frac * fs, Hamming (SURVEY 8d cfg 3)."""
    k = np.arange(ntaps) - (ntaps - 1) / 2
    h = 2 * cutoff_frac * np.sinc(2 * cutoff_frac * k) * np.hamming(ntaps)
    return h.astype(np.complex64)


def timed(torch, fn, steps, warmup):
    """-> (wall seconds for `steps` calls, list of per-call HIP-event ms)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall, [a.elapsed_time(b) for a, b in evs]


def timed_rot(torch, fn_i, steps, warmup):
    """`timed` over a rotation: call i runs fn_i(i) -- the caller indexes its buffer sets with it, so that no call
    finds its data in the 256 MB memory-side cache its predecessor left there."""
    it = iter(range(1 << 30))
    return timed(torch, lambda: fn_i(next(it)), steps, warmup)


def stream_rot(torch, fn_i, steps, warmup):
    """Device time per call of `steps` back-to-back calls of a rotation: ONE event pair around all of them -- what
    rocprofv3 lists per launch; an event pair per call (`timed`) adds the gap between two launches to a short kernel."""
    for i in range(warmup):
        fn_i(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(steps):
        fn_i(warmup + i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def self_launch(args):
    """Parent of a multi-rank run: spawn torch.distributed.run as a child, relay stdout
    (the ONE JSON line rank 0 prints) and stderr, return the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 60 periods of the clock (a 2*pi wrap every 7.5 steps).  The warm-up is 6 ms of the timed kernel: with 5 steps
    # the chip was still ramping its clocks through the timed loop (41.2 .. 45.1 us per step run to run on one box;
    # 40.0 .. 40.3 with these)
    ap.add_argument("--steps", type=int, default=450)
    ap.add_argument("--warmup", type=int, default=150)
    ap.add_argument("--ramp-ms", type=float, default=40.0,
                    help="untimed: the same step repeated for this long BEFORE the W warm-up steps, so that the chip's clocks "
                         "have settled whatever W is (with W = 5 the timed loop still ran through the ramp: 41.2 ... 45.1 us per "
                         "step from run to run on one box, 40.0 ... 40.3 behind 150 steps); reported as clock_ramp_ms")
    ap.add_argument("--log2n", type=int, default=24, help="samples per buffer = 2^log2n")
    ap.add_argument("--no-extra", action="store_true", help="skip the per-config side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle TIMING (the parity check still runs)")
    ap.add_argument("--no-oracle", action="store_true", help="skip everything that needs oracle/ (profiling runs)")
    ap.add_argument("--no-full-parity", action="store_true", help="skip the full-size check of the timed form (2 x batch x 2^log2n samples through the oracle)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="the headline chain WITHOUT hzsdr_chain_pipeline: one launch behind the other, as rounds 1-3 measured")
    ap.add_argument("--batch", type=int, default=4,
                    help="buffers handed to the chain per call (hzsdr_chain_run_batch: ONE launch over `batch` buffers of "
                         "the stream, the kernel's head, launch and tail paid once per call); 1: one call per buffer as "
                         "rounds 1-4 measured.  A step stays one 2^log2n-sample buffer")
    ap.add_argument("--buffers", type=int, default=12,
                    help="distinct 2^log2n-sample input buffers the steps rotate through (12 x 32 MiB of u8 "
                         "is more than the 256 MiB Infinity Cache: every step reads its input from HBM)")
    args = ap.parse_args()

    # `python bench.py --gpus N` run plainly (no launcher): this process -- which has not
    # touched the GPU, torch.cuda or HIP yet -- starts one rank per GPU as a CHILD
    # `python -m torch.distributed.run` job, relays its output and exits with its code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    # ONE JSON line on stdout, whatever the libraries print: file descriptor 1 is pointed at stderr for the run
    # (Gloo's connection banners, RCCL's version lines come from C++, past sys.stdout) and the line goes to the saved one
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    # Debug aid for 1-GPU boxes: HZ_BENCH_SAME_DEVICE=1 HZ_BENCH_BACKEND=gloo runs every
    # rank on cuda:0 over gloo so the multi-rank code path can be exercised without N GPUs.
    if os.environ.get("HZ_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("HZ_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (collectives give up after five minutes instead of the default ten: a rank that died must not hold the
        # job's one JSON line back for longer than the driver waits)
        import datetime
        tmo = datetime.timedelta(seconds=300)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=tmo,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    hz = importlib.import_module("go-sdr_amd")
    # one non-default HIP stream carries torch's work, the library's kernels and
    # the timing events alike
    work_stream = torch.cuda.Stream()
    torch.cuda.set_stream(work_stream)
    ctx = hz.Context(local_rank, hz.MEM_DEVICE, stream=work_stream.cuda_stream)

    n = 1 << args.log2n
    fs, D, ntaps = 20_000_000, 8, 1024
    shift = -fs / 8
    taps = lowpass_taps(ntaps, 1.0 / 16)

    # ---- the headline step: fused north-star chain ---------------------------------
    nbuf = max(1, args.buffers)
    xs = [torch.from_numpy(synth_u8(9 + rank + 101 * i, n)).cuda() for i in range(nbuf)]
    if os.environ.get("HZ_BENCH_CONSTANT_INPUT") == "1":  # (tools/power_watch.sh only: NOT the benchmark -- the line says so)
        xs = [torch.full((n, 2), 0x80, dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
    B = max(1, min(8, args.batch, nbuf))
    # (outputs: two calls' worth in rotation -- an overlapped call must not write what the call before it writes)
    ys = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(max(4, 2 * B))]
    x, y = xs[0], ys[0]
    chain = ctx.chain(hz.FMT_U8, fs).shift(shift).fir_decimate(taps, D)
    # hzsdr_chain_pipeline + hzsdr_chain_run(_batch)_after: consecutive calls of the chain overlap -- the next launch's
    # workgroups start on the compute units as this one's finish instead of behind its last one; bit-identical
    # results.  The call states what its buffers wait for: nothing -- the inputs are resident and complete long before
    # the timed region, the outputs rotate
    piped = not args.no_pipeline
    if piped:
        chain.pipeline(True)
    it = [0]

    def submit(ch, k, after):
        # one stream: the NCO clock and the FIR history carry on from buffer to buffer; k buffers per call
        i = it[0]
        it[0] = i + k
        if k == 1:
            (ch.run_after if after else ch.run)(xs[i % nbuf], ys[i % len(ys)])
        else:
            ch.run_batch([xs[(i + j) % nbuf] for j in range(k)], [ys[(i + j) % len(ys)] for j in range(k)], after=after)

    def run_steps(ch, steps, batch, after):
        full, rem = divmod(steps, batch)
        for _ in range(full):
            submit(ch, batch, after)
        if rem:
            submit(ch, rem, after)

    ramp_steps = 0
    if args.ramp_ms > 0:  # untimed clock ramp (see --ramp-ms), in bursts so the host does not run far ahead
        t_r = time.perf_counter()
        while time.perf_counter() - t_r < args.ramp_ms * 1e-3:
            run_steps(chain, 48, B, piped)
            torch.cuda.synchronize()
            ramp_steps += 48
    run_steps(chain, args.warmup, B, piped)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # one HIP-event pair around the whole timed loop, on the stream the kernels run on:
    # per-step pairs put two marker packets between consecutive launches and cost ~15 %
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run_steps(chain, args.steps, B, piped)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # device time per step, launch gaps included
    # the same chain one launch behind the other, in the same run: what ONE launch takes (and what rocprofv3 lists
    # per launch, `--no-pipeline`), beside the time per step of overlapping launches
    kernel_ms_plain = kernel_ms_batch_plain = None
    if (piped or B > 1) and world == 1:
        plain = ctx.chain(hz.FMT_U8, fs).shift(shift).fir_decimate(taps, D)

        def time_plain(batch):
            run_steps(plain, 144, batch, False)
            torch.cuda.synchronize()
            pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            pe0.record()
            run_steps(plain, 288, batch, False)
            pe1.record()
            torch.cuda.synchronize()
            return pe0.elapsed_time(pe1) / 288

        kernel_ms_plain = time_plain(1)           # one hzsdr_chain_run per buffer, one launch behind the other
        if B > 1:
            kernel_ms_batch_plain = time_plain(B)  # hzsdr_chain_run_batch, ordinary stream order
        plain.close()
    value = world * n * args.steps / elapsed / 1e6  # Msamples/s, whole job
    alg_bytes = (2 + 8 / D) * n                     # 2 B read + 8/D B written per input sample
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    nfft, valid = 4096, 3072
    # per 3072-sample hop (polyphase form): D forward 512-point FFTs up to their last pass,
    # the fused last pass x filter x sum (24 complex multiplies + 30 adds per lane, 256
    # lanes), one 512-point inverse
    m = nfft // D
    flops = (D * 5 * m * (np.log2(m) - 1) + 256 * (24 * 6 + 30 * 2) + 5 * m * np.log2(m)) / valid * n
    matrix = chain.last_fir_path() == hz.FIR_PATH_MATRIX
    kern = chain.last_fir_kernel() if hasattr(chain, "last_fir_kernel") else None
    # ---- the roofline object follows SURVEY.md 8(d): the HBM view leads ------------------------------
    #   achieved = algorithmic bytes per launch (3 B per input sample: 2 read + 8/D written) / the dominant
    #   kernel's average launch duration (one HIP-event pair around the timed steps, on the kernel's stream)
    #   fp32_vector_frac: 8(d)'s >= 170 flop per input sample against the 157.3 TFLOP/s vector peak
    #   mfma_algorithmic_frac: the direct form's irreducible work -- taps x outputs x 4 real products per complex
    #   tap x 2 ops -- against the dense int8 peak
    #   mfma_issue: the int8 operations the kernel EXECUTES (x 4 digit planes of the 32-bit fixed-point taps,
    #   and the tile windows' zero padding): an implementation figure, how busy the matrix pipe is
    t_s = kernel_ms * 1e-3
    hbm = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None}
    fp32_vec = 170.0 * n / t_s / 1e12 / FP32_PEAK_TFLOPS
    if matrix:
        passes_form = kern == getattr(hz, "FIR_KERNEL_MATRIX_PASSES", -1)
        tile, steps = (8, 68) if passes_form else (16, 72)       # outputs per tile, 32-byte window steps (hz_firmm2.h / hz_firmm.h)
        alg_ops = 4 * 2 * ntaps * (n // D)                        # the direct form, no digit planes
        # MFMAs as issued x 65 536 (32 x 32 x 32 x 2).  A tile's 32 matrix rows are its outputs x (re, im) x the planes ONE
        # A fragment holds: the passes form keeps TWO planes per fragment (8 outputs x 2 x 2), so its four planes
        # are 2 fragments per tile and step; the chunk form one plane per fragment (16 outputs x 2), 4 per tile
        frags = 2 if passes_form else 4
        exe_ops = frags * (n // D // tile) * steps * 32 * 32 * 2    # (a 32-tile column block x a step x a fragment = one MFMA = 65 536)
        if passes_form and ntaps == 1024:
            # cross-check against the kernel's code: 272 v_mfma_i32_32x32x32_i8 per 512-output pass (4 per step x 68)
            assert exe_ops == 272 * (n // D // 512) * 65536, (exe_ops, 272 * (n // D // 512) * 65536)
        roof = dict(hbm)
        roof.update({
            "kernel": ("hz::mm2::fir_mm2_kernel<u8, D = 8, 17 groups>" if passes_form else "hz::mm::fir_mm_kernel<u8, D = 8>"),
            "kernel_ms": round(kernel_ms, 4),
            "kernel_ms_is": ("device time per STEP (one 2^%d-sample buffer): one HIP-event pair around the timed loop / steps" % args.log2n
                             + ("; %d buffers per launch (hzsdr_chain_run_batch): the kernel's head, launch and tail are paid once per "
                                "call -- a launch lasts kernel_ms x %d" % (B, B) if B > 1 else "")
                             + ("; the chain is PIPELINED (hzsdr_chain_pipeline, calls through hzsdr_chain_run%s_after with nothing to "
                                "wait for): consecutive launches overlap on two streams of the chain's own (the context's stream, on "
                                "which the events sit, waits for every launch), so this is time per step of the overlapped sequence"
                                % ("_batch" if B > 1 else "") if piped else "")
                             + " -- ONE single-buffer launch by itself takes kernel_ms_unpipelined, which is what rocprofv3 lists per "
                               "launch of the `--no-pipeline --batch 1` run"),
            "kernel_ms_unpipelined": (round(kernel_ms_plain, 4) if kernel_ms_plain else None),
            "kernel_ms_batch_unpipelined": (round(kernel_ms_batch_plain, 4) if kernel_ms_batch_plain else None),
            "buffers_per_launch": B,
            "algorithmic_bytes_per_launch": int(alg_bytes) * B,
            "fp32_vector_frac": round(fp32_vec, 4),
            "mfma_algorithmic_frac": round(alg_ops / t_s / 1e12 / I8_PEAK_TOPS, 4),
            "mfma_issue": {
                "executed_int8_ops_per_launch": int(exe_ops) * B,
                "ops_are": "int8 multiply-adds x 2 as issued: v_mfma_i32_32x32x32_i8 instructions x 65 536 -- the 4 "
                           "base-256 digit planes of the 32-bit fixed-point taps (two planes share an A fragment "
                           "in the passes form), tiles x window steps, the tile windows' zero padding included -- "
                           "an implementation figure, not algorithmic work",
                "achieved": round(exe_ops / t_s / 1e12, 1), "peak": I8_PEAK_TOPS, "unit": "Top/s",
                "frac": round(exe_ops / t_s / 1e12 / I8_PEAK_TOPS, 4),
                # (round 6, tools/mfma_power.hip -> profiles/r06_mfma_power.txt: what the chip SUSTAINS of this instruction on
                # every SIMD -- the nominal peak assumes 2.4 GHz; under random int8 operands the chip holds 1.7 GHz)
                "sustained_peak_random_operands": 3400.0, "frac_of_sustained": round(exe_ops / t_s / 1e12 / 3400.0, 4),
                "sustained_peak_is": "bare back-to-back v_mfma_i32_32x32x32_i8 on all 1024 SIMDs, random operands: 19.5 ns per "
                                     "MFMA per SIMD at the 1.7 GHz the chip then holds (4.4-4.9 Pop/s on zero operands at 2.2-2.4 GHz); "
                                     "measured by tools/mfma_power.hip, not in this run",
            },
            "note": "3 B/sample puts the chain far above the HBM ridge (the 50 MB of a buffer are 8 us at 6.3 TB/s): "
                    "what binds is the issue of int8 MFMAs at the clock the chip's power management holds under TOGGLING "
                    "matrix operands -- measured inside the kernel (tools/mfma_fir2.hip PASSES=1, profiles/r06_pass_breakdown*.txt): "
                    "1.25-1.41 GHz over random bytes, 2.10 GHz over constant input, same instruction stream, 28-31 against 22 us per "
                    "buffer (`extra.power` repeats the comparison in this run).  In CYCLES the matrix pipe is busy 79 % of a "
                    "four-buffer launch (139 k of 175 k per SIMD, profiles/r05_sq_counters.txt); the rest is the launch's first "
                    "~4.5 us, the workgroups' uneven ends, and the share of a matrix loop that runs alone on its SIMD at 93 % "
                    "(per-pass stamps: profiles/r06_pass_breakdown.txt)",
        })
    else:
        roof = dict(hbm,
                    kernel="hz::fir_decimate_kernel16<4096, u8, fold 8, late> + hz::fir_synth_kernel16<4096, 8, late>",
                    kernel_ms=round(kernel_ms, 4),
                    kernel_ms_is="device time per chain_run (both kernels and the gap between them), one "
                                 "HIP-event pair around the timed loop / steps",
                    algorithmic_bytes_per_launch=int(alg_bytes),
                    note="3 B/sample puts this chain above the ridge: FFT vector work and its LDS/barrier "
                         "latency bind, not HBM",
                    fp32_vector_frac=round(flops / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4))
    result = {
        "metric": "Msamples/s: u8->c64->Shift->FIR-decimate chain @1 GPU; 4-ch Beamform @1/2/4 GPU",
        "value": round(value, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "clock_ramp_ms": args.ramp_ms,
        "clock_ramp_steps": ramp_steps,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8 x int32 fixed-point taps (int8 MFMA digit planes, exact) -> c64(f32), NCO phase 64-bit fixed point"
                 if matrix else "u8->c64(f32), NCO phase f64",
        "data": ("DIAGNOSTIC RUN, NOT THE BENCHMARK: constant input (HZ_BENCH_CONSTANT_INPUT=1, tools/power_watch.sh); "
                 if os.environ.get("HZ_BENCH_CONSTANT_INPUT") == "1" else "")
                + f"synthetic (splitmix64 u8 IQ, seeds 9+rank+101i; {nbuf} distinct buffers in rotation, "
                f"{nbuf * 2 * n >> 20} MiB resident in HBM); windowed-sinc taps",
        "config": {
            "workload": ("north-star chain: u8->c64->Shift(-fs/8)->1024-tap FIR->decimate-by-8, one "
                         "chain_run per buffer = ONE kernel: the filter over the raw bytes as an int8 "
                         "matrix product with the clock run's modulated taps, the mixer at the decimated "
                         "rate, outputs across clock boundaries in reference order; input resident in HBM"
                         + ("; %d consecutive buffers per call (hzsdr_chain_run_batch: one launch)" % B if B > 1 else "")
                         + ("; PIPELINED (hzsdr_chain_pipeline + hzsdr_chain_run%s_after): consecutive calls overlap on two streams "
                            "of the chain's own, the next call's FIR history formed from the call's input by a 16-wave kernel "
                            "beside the matrix kernel -- both joined to the context's stream; checked in this form under `parity`"
                            % ("_batch" if B > 1 else "") if piped else ""))
                        if matrix else
                        ("north-star chain: u8->c64->Shift(-fs/8)->1024-tap FIR->decimate-by-8, one "
                         "chain_run per buffer = analysis kernel (convert, 4096-point overlap-save "
                         "FFT, filter, fold by 8) + synthesis kernel (512-point inverse, mixer at the "
                         "decimated rate), input resident in HBM"),
            "samples_per_buffer": n, "sample_rate": fs, "taps": ntaps, "decimation": D,
            "parallelism": "1 stream per GPU (replicas)" if world > 1 else "1 GPU",
            "pipelined": piped,
            "buffers_per_call": B,
        },
        "roofline": roof,
    }
    chain.close()
    # HBM traffic per launch of the dominant kernel comes from separate rocprofv3 --pmc
    # passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_summary.py), not from this process.
    tname = next((t for t in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", t))), "r05_traffic.json")
    tpath = os.path.join(ROOT, "profiles", tname)
    if n == (1 << 24) and os.path.exists(tpath):
        try:
            tk = json.load(open(tpath))["kernels"]
            if matrix:
                keys = [k for k in tk if ("fir_mm2_kernel<2, 8, 17" in k if passes_form else "fir_mm_kernel<2, 8" in k)]
                want = 1
            else:
                keys = [k for k in tk if "fir_decimate_kernel16<4096, 2, 8, true" in k
                        or "fir_synth_kernel16<4096, 8, true>" in k]
                want = 2
            if len(keys) == want:
                # (r05 on: the PMC passes profile a launch over FOUR buffers -- tools/prof_kernels.py chain_batch4 --,
                # earlier rounds' one over one; scaled to this run's buffers per launch)
                per_launch_of = 4 if (tname >= "r05" and matrix) else 1
                result["roofline"]["traffic"] = round(sum(tk[k]["hbm_bytes"] for k in keys) * (B if matrix else 1) / per_launch_of)
                result["roofline"]["traffic_source"] = ("profiles/" + tname + " (rocprofv3 --pmc FETCH_SIZE x2 + "
                                                        "WRITE_SIZE, the kernel(s) of a chain_run)")
        except (StopIteration, KeyError, ValueError):
            pass

    # ---- side measurements: the other BASELINE configs (rank 0, N = 1 semantics) ----
    # (the side measurements are one-GPU figures: in a multi-rank job they would only keep the other ranks waiting at
    # the Beamform section's first collective)
    if rank == 0 and world == 1 and not args.no_extra:
        extra = {}
        k, w = 20, 3  # median of 20 runs after 3 warm-ups, per-call HIP events

        def rate(nsamp, ms, bytes_per_sample):
            return {"Msamples_per_s": round(nsamp / (ms * 1e-3) / 1e6, 1), "kernel_ms": round(ms, 4),
                    "GBps": round(bytes_per_sample * nsamp / (ms * 1e-3) / 1e9, 1),
                    "hbm_frac": round(bytes_per_sample * nsamp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

        # data-independence: the same chain over an ADC-like input (Gaussian about 127.5,
        # sigma 20) instead of uniform bytes
        u = (splitmix64(10, 4 * n) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        g = np.sqrt(-2.0 * np.log(u[0::2] + 1e-300)) * np.cos(2 * np.pi * u[1::2])
        xa = torch.from_numpy(np.clip(np.rint(127.5 + 20.0 * g), 0, 255).astype(np.uint8).reshape(n, 2)).cuda()
        del u, g
        ch = ctx.chain(hz.FMT_U8, fs).shift(shift).fir_decimate(taps, D)
        _, ms = timed(torch, lambda: ch.run(xa, y), k, w)
        extra["chain_adc_like_u8"] = rate(n, float(np.median(ms)), 2 + 8 / D)
        ch.close()
        del xa
        # What the chip's POWER management does to this kernel (round 6, DESIGN.md section 4 "Round 6"): the benchmarked
        # form -- B buffers per call, overlapped -- over CONSTANT input (every byte 0x80: zero behind the u8 sign flip).
        # Same instruction stream, same bytes moved, same MFMA count; the matrix operands do not toggle, the chip holds
        # ~2.1 GHz instead of the ~1.4 it holds over random bytes (tools/mfma_fir2.hip PASSES=1 / ZERO=1 measure the
        # clock inside the kernel), and a buffer takes a quarter less time.
        if matrix and B > 1:
            def headline_form(inputs, steps=288, warm=144):
                chp = ctx.chain(hz.FMT_U8, fs).shift(shift).fir_decimate(taps, D)
                if piped:
                    chp.pipeline(True)
                pos = [0]

                def go(cnt):
                    for _ in range(cnt // B):
                        i = pos[0]
                        pos[0] = i + B
                        chp.run_batch([inputs[(i + j) % len(inputs)] for j in range(B)], [ys[(i + j) % len(ys)] for j in range(B)], after=piped)
                go(warm)
                torch.cuda.synchronize()
                pa, pb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                pa.record()
                go(steps)
                pb.record()
                torch.cuda.synchronize()
                chp.close()
                return pa.elapsed_time(pb) / (steps // B * B)
            xq = [torch.full((n, 2), 0x80, dtype=torch.uint8, device="cuda") for _ in range(min(nbuf, 12))]
            ms_quiet = headline_form(xq)
            del xq
            ms_rand = headline_form(xs)
            extra["power"] = {
                "headline_form_ms_per_step_random_bytes": round(ms_rand, 4), "headline_form_ms_per_step_constant_input": round(ms_quiet, 4),
                "ratio": round(ms_rand / ms_quiet, 3),
                "what": "the benchmarked call form, same run, over the bench's random bytes and over constant input (0x80): identical instruction "
                        "stream and traffic -- the difference is the clock the chip holds under toggling int8 matrix operands "
                        "(profiles/r06_pass_breakdown*.txt: 1.41 GHz against 2.10 GHz measured inside the kernel; profiles/r06_mfma_power.txt: "
                        "bare v_mfma_i32_32x32x32_i8 streams on every SIMD sustain 3.4 Pop/s on random operands, 4.4-4.9 on zeros)"}
        # the same chain on the overlap-save transform kernels (round 2's first implementation)
        ch = ctx.chain(hz.FMT_U8, fs).shift(shift).fir_options(hz.FIR_IMPL_TRANSFORMS).fir_decimate(taps, D)
        _, ms = timed(torch, lambda: ch.run(x, y), k, w)
        extra["chain_transform_kernels"] = rate(n, float(np.median(ms)), 2 + 8 / D)
        ch.close()
        # the same chain on the first int8 matrix kernel (round 2: one round of 2048-output chunk workgroups)
        ch = ctx.chain(hz.FMT_U8, fs).shift(shift).fir_options(hz.FIR_IMPL_MATRIX_CHUNKS).fir_decimate(taps, D)
        ch.set_time(1.0)
        _, ms = timed(torch, lambda: ch.run(x, y), k, w)
        extra["chain_matrix_chunks_kernel"] = dict(rate(n, float(np.median(ms)), 2 + 8 / D), kernel=ch.last_fir_kernel())
        ch.close()
        # ONE launch over 2^26 samples (128 MiB of u8): the kernel's first and last microseconds once per 64 passes
        # of a workgroup instead of once per 16
        if args.log2n == 24:
            n26 = 1 << 26
            x26 = torch.from_numpy(synth_u8(77, n26)).cuda()
            y26 = torch.zeros(n26 // D, dtype=torch.complex64, device="cuda")
            ch = ctx.chain(hz.FMT_U8, fs).shift(shift).fir_decimate(taps, D)
            ch.set_time(1.0)
            _, ms = timed(torch, lambda: ch.run(x26, y26), 10, 2)
            extra["chain_2p26"] = dict(rate(n26, float(np.median(ms)), 2 + 8 / D), kernel=ch.last_fir_kernel(),
                                       ms_per_2p24_samples=round(float(np.median(ms)) / 4, 4))
            ch.close()
            del x26, y26
        # the two implementations of the terminal over tap counts and factors (ms per 2^24 samples)
        paths = {}
        for dd, nt in ((8, 64), (8, 256), (8, 1024), (16, 256), (16, 1024), (16, 2047), (32, 1024), (64, 1024)):
            tt = lowpass_taps(nt, 0.5 / dd)
            yy = y[:n // dd]
            row = {}
            for name, impl in (("matrix", hz.FIR_IMPL_AUTO), ("transform", hz.FIR_IMPL_TRANSFORMS)):
                ch = ctx.chain(hz.FMT_U8, fs).shift(shift).fir_options(impl).fir_decimate(tt, dd)
                ch.set_time(1.0)
                _, ms = timed(torch, lambda: ch.run(x, yy), 12, 2)
                row[name + "_ms"] = round(float(np.median(ms)), 4)
                row[name + "_path"] = ch.last_fir_path()
                ch.close()
            paths[f"D{dd}_taps{nt}"] = row
        extra["fir_decimate_paths_u8"] = paths
        # the same chain with the mixer forced in front of the filter on every block
        ch = ctx.chain(hz.FMT_U8, fs).shift(shift).fir_decimate(taps, D).mix_in_order(True)
        _, ms = timed(torch, lambda: ch.run(x, y), k, w)
        extra["chain_mix_in_order"] = rate(n, float(np.median(ms)), 2 + 8 / D)
        ch.close()
        c = torch.from_numpy(synth_c64(2, n)).cuda()
        out = torch.zeros(n, dtype=torch.complex64, device="cuda")
        # The streaming rows are measured TWICE: over ONE buffer pair, as rounds 1-3 did -- 256 MiB, which the
        # 256 MB memory-side cache largely holds from one call to the next (faster than HBM can be) -- and, under
        # "hbm", over a rotation of kRot buffer pairs (1 GiB), where every call's bytes come from HBM and go to it.
        kRot = 4
        cs = [c] + [torch.from_numpy(synth_c64(20 + i, n)).cuda() for i in range(kRot - 1)]
        outs = [out] + [torch.zeros(n, dtype=torch.complex64, device="cuda") for _ in range(kRot - 1)]

        def back_to_back(fn_i, bytes_per_sample):
            # the same rotation, 3 k calls one behind the other between ONE event pair (no event between the launches)
            ms = stream_rot(torch, fn_i, 3 * k, w)
            return {"stream_ms": round(ms, 4), "stream_GBps": round(bytes_per_sample * n / (ms * 1e-3) / 1e9, 1),
                    "stream_hbm_frac": round(bytes_per_sample * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

        def both(name, fn_i, bytes_per_sample):
            _, ms = timed(torch, lambda: fn_i(0), k, w)
            extra[name] = rate(n, float(np.median(ms)), bytes_per_sample)
            _, ms = timed_rot(torch, fn_i, k, w)
            extra[name]["hbm"] = dict(rate(n, float(np.median(ms)), bytes_per_sample), buffer_pairs=kRot)
            extra[name]["hbm"].update(back_to_back(fn_i, bytes_per_sample))

        # the same-run device copy (8 B read + 8 B written per sample): the practical HBM
        # ceiling the HBM-bound rows below are also quoted against
        # (the library's own CopySamples -- hzsdr_convert with equal formats: non-temporal, two vectors per lane --
        # since round 5; torch's copy_, the yardstick of rounds 1-4, beside it)
        both("device_copy_c64", lambda i: ctx.convert(outs[i % kRot], cs[i % kRot]), 16)
        both("torch_copy_c64", lambda i: outs[i % kRot].copy_(cs[i % kRot]), 16)
        # Scale and Rotate in place (internal/simd/mult.go:25-45): 16 B/sample
        both("scale_c64", lambda i: ctx.scale(cs[i % kRot], 0.999), 16)
        both("rotate_c64", lambda i: ctx.rotate(cs[i % kRot], complex(0.6, 0.8)), 16)
        copy_gbps = extra["device_copy_c64"]["GBps"]
        copy_hbm_gbps = extra["device_copy_c64"]["hbm"]["GBps"]
        # cfg 1 kernel: u8 -> c64 (10 B/sample)
        both("convert_u8_c64", lambda i: ctx.convert(outs[i % kRot], xs[i % len(xs)]), 10)
        # cfg 2: Shift + Gain fused (16 B/sample), bit for bit the reference's
        ch = ctx.chain(hz.FMT_C64, fs).shift(2.5e6).gain(0.5)
        both("shift_gain_c64", lambda i: ch.run(cs[i % kRot], outs[i % kRot]), 16)
        ch.close()
        # ... and with hzsdr_chain_pipeline through hzsdr_chain_run_after (consecutive calls overlap on two streams;
        # the buffers rotate, nothing to wait for)
        ch = ctx.chain(hz.FMT_C64, fs).shift(2.5e6).gain(0.5).pipeline(True)
        _, ms = timed_rot(torch, lambda i: ch.run_after(cs[i % kRot], outs[i % kRot]), k, w)
        extra["shift_gain_c64"]["hbm_pipelined"] = dict(rate(n, float(np.median(ms)), 16), buffer_pairs=kRot,
                                                        note="per-call events around OVERLAPPING calls: see wall_ms")
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        for i in range(k):
            ch.run_after(cs[i % kRot], outs[i % kRot])
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t_w) / k * 1e3
        extra["shift_gain_c64"]["hbm_pipelined"].update(wall_ms=round(wall_ms, 4), wall_GBps=round(16 * n / (wall_ms * 1e-3) / 1e9, 1),
                                                        wall_hbm_frac=round(16 * n / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        ch.close()
        # ... and with the opt-in <= 1-ulp rotation factor (hzsdr_chain_shift_ulp1: not bit-identical, not `value`)
        ch = ctx.chain(hz.FMT_C64, fs).shift(2.5e6).gain(0.5).shift_ulp1()
        both("shift_gain_c64_ulp1", lambda i: ch.run(cs[i % kRot], outs[i % kRot]), 16)
        ch.close()
        # the Reader form of Shift (hzsdr_nco_shift: what ShiftReader / ShiftBuffer bind to), in place, bit-exact
        # and with hzsdr_nco_set_ulp1
        nco = ctx.nco(fs)
        both("shift_c64", lambda i: nco(2.5e6, cs[i % kRot]), 16)
        nco.set_ulp1()
        both("shift_c64_ulp1", lambda i: nco(2.5e6, cs[i % kRot]), 16)
        nco.close()
        # cfg 3: reference ConvolutionReader semantics, 1024 bins (16 B/sample)
        H = torch.from_numpy(np.fft.fft(np.asarray(taps, np.complex128) / ntaps).astype(np.complex64)).cuda()
        both("convolution_1024_circular", lambda i: ctx.convolution_blocks(outs[i % kRot], cs[i % kRot], H), 16)
        # cfg 3, north-star form: 1024-tap FIR by overlap-save (N_fft 4096), no decimation
        chf = ctx.chain(hz.FMT_C64, fs).fir_decimate(taps, 1)
        both("fir_1024_overlap_save_c64", lambda i: chf.run(cs[i % kRot], outs[i % kRot]), 16)
        chf.close()
        # fft.Planner lengths of the reference's own callers, 2^24 points per call (16 B / point: in once, out once;
        # the two-step lengths move their scratch as well -- 32 B / point -- and are quoted by the ALGORITHMIC 16):
        # 4096 (ConvolutionReader blocks), 64 Ki (rtl/kerberos/internal/align.go), 256 Ki (internal/graft.go)
        for lg in (12, 16, 18):
            plan = ctx.fft_plan(cs[0], outs[0], hz.FFT_FORWARD, batch=n >> lg)
            plans = [plan] + [ctx.fft_plan(cs[i], outs[i], hz.FFT_FORWARD, batch=n >> lg) for i in range(1, kRot)]
            both("fft_forward_2p%d" % lg, lambda i: plans[i % kRot].transform(), 16)
            del plan, plans
        del cs[1:], outs[1:]
        # cfg 4: Downsample by 8 from i16 (5 B/input sample)
        xi = torch.from_numpy(synth_i16(4, n)).cuda()
        o8 = torch.zeros(n // 8, dtype=torch.complex64, device="cuda")
        xis = [xi] + [torch.roll(xi, 1000 * (i + 1), 0) for i in range(5)]  # 6 x 80 MiB: past the cache
        o8s = [o8] + [torch.zeros(n // 8, dtype=torch.complex64, device="cuda") for _ in range(5)]
        _, ms = timed(torch, lambda: ctx.downsample(o8, xi, 8), k, w)
        extra["downsample8_i16"] = rate(n, float(np.median(ms)), 5)
        _, ms = timed_rot(torch, lambda i: ctx.downsample(o8s[i % 6], xis[i % 6], 8), k, w)
        extra["downsample8_i16"]["hbm"] = dict(rate(n, float(np.median(ms)), 5), buffer_pairs=6)
        extra["downsample8_i16"]["hbm"].update(back_to_back(lambda i: ctx.downsample(o8s[i % 6], xis[i % 6], 8), 5))
        del xis[1:], o8s[1:]
        # cfg 4, north-star form: a designed FIR-decimate by 8 from i16 (polyphase: 256 and 1024 taps; 4 + 8/8 B per
        # input sample), on the overlap-save transform kernels (i16 has no matrix form)
        for nt in (256, 1024):
            ch = ctx.chain(hz.FMT_I16, 200_000_000).fir_decimate(lowpass_taps(nt, 1.0 / 16), 8)
            _, ms = timed(torch, lambda: ch.run(xi, o8), 12, 2)
            extra[f"fir_decimate8_i16_{nt}taps"] = dict(rate(n, float(np.median(ms)), 5), path=ch.last_fir_path())
            ch.close()
        del xi, o8
        # cfg 5 on one GPU: 4-channel c64 beamform (40 B/output sample)
        chans = [torch.from_numpy(synth_c64(5 + i, n)).cuda() for i in range(4)]
        wts = hz.beamform_angles(433e6, 30.0, [0.0, 0.1, 0.2, 0.3])
        _, ms = timed(torch, lambda: ctx.beamform(out, chans, wts), k, w)
        extra["beamform4_c64_1gpu"] = rate(n, float(np.median(ms)), 40)
        del chans, c, out
        for name, row in extra.items():
            if name != "device_copy_c64" and "GBps" in row:
                row["frac_of_device_copy"] = round(row["GBps"] / copy_gbps, 4)
                if "hbm" in row:
                    row["hbm"]["frac_of_device_copy"] = round(row["hbm"]["GBps"] / copy_hbm_gbps, 4)
                    if "stream_GBps" in row["hbm"]:
                        row["hbm"]["stream_frac_of_device_copy"] = round(row["hbm"]["stream_GBps"] / extra["device_copy_c64"]["hbm"]["stream_GBps"], 4)
        extra["small_buffers"] = small_buffers_gpu(hz, ctx, torch, local_rank)
        result["extra"] = extra

    # ---- Beamform, 4 channels, on sub-groups of 1 / 2 / 4 ranks (the other half of the metric) ----
    from importlib import import_module
    mg = import_module("go-sdr_amd.multigpu")
    xs, ys = xs[:1], ys[:1]  # the rotation's other buffers are no longer needed
    torch.cuda.empty_cache()
    try:
        result["beamform"] = mg.bench_beamform(hz, ctx, torch, dist, rank, world, n,
                                               steps=max(5, args.steps // 2), warmup=2, synth=synth_c64)
    except Exception as e:  # noqa: BLE001  (the multi-GPU exchange has never met real hardware: the headline survives it)
        result["beamform"] = {"error": f"{type(e).__name__}: {e}"}
    # evidence that the collective library really spanned N ranks: an all-reduce of (rank + 1)
    rccl = {"backend": backend if world > 1 else None, "world_size": world}
    if world > 1:
        t = torch.tensor([rank + 1], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rccl["allreduce_sum_of_rank_plus_1"] = int(t.item())
        rccl["expected"] = world * (world + 1) // 2
        if backend == "nccl":
            try:
                rccl["version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                pass
        devs = [None] * world
        dist.all_gather_object(devs, f"{torch.cuda.get_device_name(local_rank)}#{local_rank}")
        rccl["devices"] = devs
    result["rccl"] = rccl

    # ---- the oracle section (the cpu_baseline leg): the checker first, then its timing -------
    parity_ok = True
    if rank == 0 and not args.no_oracle:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc
        ns = min(n, 1 << 22)
        # (a) the benchmarked configuration against the oracle, on a stream whose clock starts
        # 0.1 s short of 2*pi: the check crosses the wrap (reference-order blocks on both sides
        # of it) as well as late-mixer runs; both bounds are tests/util.py's
        ts0 = 6.283185307179586 - 0.1
        xs0 = synth_u8(9, ns)
        buf = np.zeros(ns, np.complex64)
        orc.convert(buf, xs0)
        sh = orc.Shifter(fs)
        sh.ts.value = ts0
        sh(shift, buf)
        want = np.zeros(ns // D, np.complex64)
        orc.par_fir_decimate_f64(want, buf, taps, D)
        chk = ctx.chain(hz.FMT_U8, fs).shift(shift).fir_decimate(taps, D)
        if piped:
            chk.pipeline(True)
        chk.set_time(ts0)
        yg = torch.zeros(ns // D, dtype=torch.complex64, device="cuda")
        xg = torch.from_numpy(xs0).cuda()
        torch.cuda.synchronize()
        # (the benchmarked form: calls of B buffers -- eight pieces of the check's samples -- through the same entry
        # points; with the pipeline on, the calls behind the first read a history formed by the history kernel and overlap)
        q = ns // 8
        pieces_in = [xg[j * q:(j + 1) * q] for j in range(8)]
        pieces_out = [yg[j * q // D:(j + 1) * q // D] for j in range(8)]
        j = 0
        while j < 8:
            kk = min(B, 8 - j)
            if kk == 1:
                (chk.run_after if piped else chk.run)(pieces_in[j], pieces_out[j])
            else:
                chk.run_batch(pieces_in[j:j + kk], pieces_out[j:j + kk], after=piped)
            j += kk
        torch.cuda.synchronize()
        got = yg.cpu().numpy().astype(np.complex128)
        err = float(np.abs(got - want).max())
        bound = 6e-7 * float(np.abs(taps).sum()) * float(np.abs(buf).max())
        rel = float(np.linalg.norm(got - want) / np.linalg.norm(want.astype(np.complex128)))
        parity_ok = bool(err <= bound and rel <= 3e-7 and chk.time() == sh.ts.value)
        result["parity"] = {"checked_outputs": ns // D, "max_abs_err": err, "bound": bound,
                            "rel_l2_err": rel, "rel_l2_bound": 3e-7, "clock_start": ts0,
                            "clock_after_equal": bool(chk.time() == sh.ts.value), "ok": parity_ok,
                            "what": "GPU chain (default mixer order) vs the oracle: reference-order convert + "
                                    "Shift, float64 direct-form FIR, clock started 0.1 s before the 2*pi wrap; "
                                    "bounds: max abs <= 6e-7 * sum|h| * max|x| and relative L2 <= 3e-7"}
        chk.close()
        del yg, got, want
        # (a') the TIMED FORM itself at its full size: ONE call over B separately allocated buffers of n samples (the
        # headline's hzsdr_chain_run_batch[_after]: 16 384 passes per launch at B = 4, a pass's buffer by a 32-bit
        # reciprocal, virtual base pointers), the clock's 2*pi wrap inside the call, then a second call that continues
        # the stream -- EVERY output of both calls against the oracle (the reference-order stages and the float64
        # direct form on every host core), per buffer.  Round 5 checked this form at 8 x 2^19 samples only.
        if B > 1 and n * B >= (1 << 22) and not args.no_full_parity:
            threads = orc.max_threads()
            nb = 2 * B
            ts1 = 6.283185307179586 - 0.6 * B * n / fs  # (the wrap 60 % into the first call)
            while ts1 < 0.0:
                ts1 += 6.283185307179586
            xf = synth_u8(19, nb * n)
            bf = np.zeros(nb * n, np.complex64)
            orc.par_u8_to_c64(xf, bf, threads)
            ts_want = orc.par_shift_gain(ts1, fs, shift, 1.0, bf, threads)  # (gain 1.0: the Shift alone)
            # (every output where the host has the cores for it -- 2 s on the 256-core boxes of this pool; on a small host
            # the first, middle and last 2^16 outputs of every buffer: each slice filtered from 1024 samples in front of it)
            check_all = threads >= 48 and not os.environ.get("HZ_BENCH_SLICE_PARITY")
            wantf = np.zeros(nb * n // D, np.complex64)
            slices = []
            if check_all:
                orc.par_fir_decimate_f64(wantf, bf, taps, D)
            else:
                no, cnt, lead = n // D, min(1 << 16, n // D), (ntaps + D - 1) // D
                for j in range(nb):
                    for m0 in sorted({j * no, j * no + (no - cnt) // 2, j * no + no - cnt}):
                        a = max(0, m0 - lead)
                        tmp = np.zeros(m0 + cnt - a, np.complex64)
                        orc.par_fir_decimate_f64(tmp, bf[a * D:(m0 + cnt) * D], taps, D)
                        if a > 0 or m0 == 0:  # (outputs whose window reaches in front of the slice's samples are dropped; the stream's first are whole)
                            wantf[m0:m0 + cnt] = tmp[m0 - a:]
                            slices.append((m0, m0 + cnt))
            xmaxf = float(np.abs(bf[:1 << 20]).max())
            del bf
            xg = [torch.from_numpy(xf[j * n:(j + 1) * n]).cuda() for j in range(nb)]
            yg = [torch.zeros(n // D, dtype=torch.complex64, device="cuda") for _ in range(nb)]
            del xf
            chk = ctx.chain(hz.FMT_U8, fs).shift(shift).fir_decimate(taps, D)
            if piped:
                chk.pipeline(True)
            chk.set_time(ts1)
            torch.cuda.synchronize()
            kernels = []
            for j in range(0, nb, B):
                chk.run_batch(xg[j:j + B], yg[j:j + B], after=piped)
                kernels.append(chk.last_fir_kernel())
            torch.cuda.synchronize()
            boundf = 6e-7 * float(np.abs(taps).sum()) * xmaxf
            per_buf = []
            checked = 0
            for j in range(nb):
                g = yg[j].cpu().numpy().astype(np.complex128)
                w = wantf[j * (n // D):(j + 1) * (n // D)]
                if not check_all:  # (the slices of this buffer)
                    keep = np.zeros(n // D, bool)
                    for a, b in slices:
                        lo, hi = max(a, j * (n // D)) - j * (n // D), min(b, (j + 1) * (n // D)) - j * (n // D)
                        if lo < hi:
                            keep[lo:hi] = True
                    g, w = g[keep], w[keep]
                checked += len(w)
                per_buf.append((float(np.abs(g - w).max()), float(np.linalg.norm(g - w) / np.linalg.norm(w.astype(np.complex128)))))
            ok_full = bool(all(e <= boundf and r <= 3e-7 for e, r in per_buf) and chk.time() == ts_want
                           and all(kq == hz.FIR_KERNEL_MATRIX_PASSES for kq in kernels))
            result["parity"]["timed_form"] = {
                "calls": nb // B, "buffers_per_call": B, "samples_per_buffer": n, "overlapped": bool(piped),
                "checked_outputs": checked, "checked": "every output" if check_all else "the first, middle and last 2^16 outputs of every buffer",
                "max_abs_err": max(e for e, _ in per_buf), "bound": boundf,
                "rel_l2_err_worst_buffer": max(r for _, r in per_buf), "rel_l2_bound": 3e-7, "clock_start": ts1,
                "clock_after_equal": bool(chk.time() == ts_want), "one_launch_per_call": bool(all(kq == hz.FIR_KERNEL_MATRIX_PASSES for kq in kernels)),
                "ok": ok_full,
                "what": "the benchmarked entry point at the benchmarked size: every output of two consecutive calls over "
                        "separately allocated buffers (the 2*pi wrap inside the first) against the oracle, per buffer"}
            result["parity"]["checked_outputs"] += checked
            result["parity"]["ok"] = parity_ok = bool(parity_ok and ok_full)
            chk.close()
            del xg, yg, wantf
        # (b) CPU baseline: the oracle (scalar port), one thread, bounded sample (N = 1 only)
        if world == 1 and not args.no_cpu_baseline:
            outc = np.zeros(ns // D, np.complex64)
            reps, t0 = 0, time.perf_counter()
            while True:
                orc.convert(buf, xs0)
                orc.Shifter(fs)(shift, buf)
                orc.fir_decimate_f64(outc, buf, taps, D)
                reps += 1
                dt = time.perf_counter() - t0
                if dt >= 10.0 or reps >= 64:
                    break
            result["cpu_baseline"] = {
                "value": round(reps * ns / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": f"{reps} x the first 2^{int(np.log2(ns))} samples of the same chain (convert, Shift "
                          f"with math.Sincos restated, 1024-tap direct-form FIR at the decimated rate, float64 "
                          f"accumulate; gcc -O2, one thread), {dt:.1f} s of CPU work",
            }
            # "reference algorithm, parallelised, not the reference": the same
            # oracle functions under OpenMP on every host core -- the whole chain, and the two
            # stages the side measurements quote
            threads = orc.max_threads()
            allc = {"cores": threads, "kind": "port, OpenMP (chunks equal the serial oracle bit for bit)"}
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 3.0:
                orc.par_chain_fir(xs0, outc, fs, shift, taps, D, threads, scratch=buf)
                reps += 1
            allc["value"] = round(reps * ns / (time.perf_counter() - t0) / 1e6, 1)
            allc["unit"] = "Msamples/s"
            allc["sample"] = f"{reps} x the same 2^{int(np.log2(ns))}-sample chain"
            bufp = np.zeros(ns, np.complex64)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 1.0:
                orc.par_u8_to_c64(xs0, bufp, threads)
                reps += 1
            allc["convert_u8_c64_Msamples_per_s"] = round(reps * ns / (time.perf_counter() - t0) / 1e6, 1)
            reps, t0, ts = 0, time.perf_counter(), 0.0
            while time.perf_counter() - t0 < 1.0:
                ts = orc.par_shift_gain(ts, fs, 2.5e6, 0.5, bufp, threads)
                reps += 1
            allc["shift_gain_c64_Msamples_per_s"] = round(reps * ns / (time.perf_counter() - t0) / 1e6, 1)
            result["cpu_baseline"]["all_cores"] = allc
            if "small_buffers" in result.get("extra", {}):
                small_buffers_cpu(orc, result["extra"]["small_buffers"])

    if world > 1:  # every rank learns the verdict: one exit code for the job
        t = torch.tensor([1 if parity_ok else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        parity_ok = bool(int(t.item()))
    if rank == 0:
        if not parity_ok:  # a fast kernel whose results differ from the reference's is not a result
            result["value"] = None
            result["invalid"] = "parity check failed: see `parity`"
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    os.close(real_stdout)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return 0 if parity_ok else 3


# The reference's own benchmark sizes: us per call, one call at a time.
#   iq_u8_test.go:170-182 (u8 -> c64, 16 Ki), internal/simd/mult_test.go:80-102 (Scale / Rotate,
#   16 Ki), stream/add_test.go:137-191 (Add, 8 Ki x K), stream/multiply_test.go:114-187
#   (Multiply u8 / i8 / c64, 64 Ki), plus ConvertBuffer at the 32 Ki Reader block
#   (stream/convert.go:43-44).
SMALL_CASES = [("convert_u8_c64", 16 * 1024), ("convert_u8_c64", 32 * 1024), ("scale_c64", 16 * 1024),
               ("rotate_c64", 16 * 1024), ("add_x2", 8 * 1024), ("add_x4", 8 * 1024), ("add_x16", 8 * 1024),
               ("multiply_u8", 64 * 1024), ("multiply_i8", 64 * 1024), ("multiply_c64", 64 * 1024)]


def _small_op(hz, ctx, name, n, mk_in, mk_out):
    """-> a zero-argument callable running `name` once on buffers made by mk_in / mk_out."""
    m = 0.6 + 0.8j
    if name == "convert_u8_c64":
        src, dst = mk_in(hz.FMT_U8, n, 1), mk_out(hz.FMT_C64, n)
        return lambda: ctx.convert(dst, src)
    if name == "scale_c64":
        buf = mk_in(hz.FMT_C64, n, 2)
        return lambda: ctx.scale(buf, 0.999)
    if name in ("rotate_c64", "multiply_c64"):
        buf = mk_in(hz.FMT_C64, n, 3)
        return lambda: ctx.rotate(buf, m)
    if name.startswith("add_x"):
        k = int(name[5:])
        bufs, out = [mk_in(hz.FMT_C64, n, 10 + i) for i in range(k)], mk_out(hz.FMT_C64, n)
        return lambda: ctx.sum(out, bufs)
    fmt = hz.FMT_U8 if name == "multiply_u8" else hz.FMT_I8
    tab, buf = ctx.rotlut(fmt, m), mk_in(fmt, n, 4)
    return lambda: tab.apply(buf)


def small_buffers_gpu(hz, dctx, torch, device):
    rows = {}
    hctx = hz.Context(device, hz.MEM_HOST)

    def np_in(fmt, n, seed):
        return synth_c64(seed, n).copy() if fmt == hz.FMT_C64 else synth_u8(seed, n).view(
            np.uint8 if fmt == hz.FMT_U8 else np.int8)

    def np_out(fmt, n):
        return np.zeros(n, np.complex64)

    def pin_in(fmt, n, seed):
        a = hctx.pinned_samples(fmt, n)
        a[...] = np_in(fmt, n, seed)
        return a

    def dev_in(fmt, n, seed):
        return torch.from_numpy(np_in(fmt, n, seed)).cuda()

    for name, n in SMALL_CASES:
        row = {"samples": n}
        for label, ctx, mk_in, mk_out in (("host_us", hctx, np_in, np_out),
                                          ("host_pinned_us", hctx, pin_in, lambda f, k: hctx.pinned_samples(f, k))):
            fn = _small_op(hz, ctx, name, n, mk_in, mk_out)
            for _ in range(20):
                fn()
            t = []
            for _ in range(200):
                t0 = time.perf_counter()
                fn()
                t.append(time.perf_counter() - t0)
            row[label] = round(float(np.median(t)) * 1e6, 2)
        fn = _small_op(hz, dctx, name, n, dev_in, lambda f, k: torch.zeros(k, dtype=torch.complex64, device="cuda"))
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        torch.cuda.synchronize()
        row["device_us"] = round(e0.elapsed_time(e1) / 200 * 1e3, 2)
        rows[f"{name}_{n // 1024}Ki"] = row
    # A Reader PIPELINE at the reference's block size: Gain(ShiftReader(ConvertReader(u8 source))), 2^22 samples read in
    # 32 Ki-sample blocks through (a) the reference's own structure -- a ReadTransformer and two wrappers, three GPU calls
    # per block -- and (b) the fusing constructors (go/hip/fused.go, stream.ChainReader): one chain, one launch per 2^20
    # samples, the pinned ring reading ahead.  us per 32 Ki block, by the wall clock, Python's share included.
    S = importlib.import_module("go-sdr_amd.stream")
    nrd, blk = 1 << 22, 32 * 1024
    xr = synth_u8(5, nrd)
    row = {"samples": blk, "stream_samples": nrd}
    for label, fuse in (("host_us", False), ("fused_us", True)):
        best = None
        for _ in range(3):
            st = S.Stream(hctx, fuse=fuse)
            r = st.gain(st.shift_reader(st.convert_reader(S.BufferReader(xr, 2_400_000), hz.FMT_C64), 3.1e5), 0.5)
            buf = np.zeros(blk, np.complex64)
            t0 = time.perf_counter()
            got = 0
            try:
                while True:
                    got += r.read(buf)
            except S.EOF:
                pass
            dt = time.perf_counter() - t0
            assert got == nrd, got
            if hasattr(r, "close"):
                r.close()
            best = dt if best is None or dt < best else best
        row[label] = round(best / (nrd // blk) * 1e6, 2)
    rows["reader_chain_32Ki"] = row
    hctx.close()
    rows["note"] = ("us per call: host = HZSDR_MEM_HOST on pageable numpy buffers (copy in, kernel over the pinned "
                    "staging area, copy out; what a cgo caller with Go slices pays, plus ~2 us of ctypes), "
                    "host_pinned = the same call on hzsdr_malloc_pinned buffers (no staging), device = "
                    "HZSDR_MEM_DEVICE back to back on one stream (HIP events / 200), cpu = the oracle, one thread")
    return rows


def small_buffers_cpu(orc, table):
    """The oracle (one thread) on the same sizes, beside the GPU columns."""
    import oracle as o
    m = 0.6 + 0.8j
    for name, n in SMALL_CASES:
        if name == "convert_u8_c64":
            src, dst = synth_u8(1, n), np.zeros(n, np.complex64)
            fn = lambda: o.convert(dst, src)  # noqa: E731
        elif name == "scale_c64":
            buf = synth_c64(2, n).copy()
            fn = lambda: o.scale(buf, 0.999)  # noqa: E731
        elif name in ("rotate_c64", "multiply_c64"):
            buf = synth_c64(3, n).copy()
            fn = lambda: o.rotate(buf, m)  # noqa: E731
        elif name.startswith("add_x"):
            k = int(name[5:])
            bufs, out = [synth_c64(10 + i, n).copy() for i in range(k)], np.zeros(n, np.complex64)
            fn = lambda: o.sum_(out, bufs)  # noqa: E731
        else:
            u8 = name == "multiply_u8"
            tab = o.rotate_table_u8(m) if u8 else o.rotate_table_i8(m)
            buf = synth_u8(4, n) if u8 else synth_u8(4, n).view(np.int8)
            fn = (lambda: o.rotate_u8_apply(tab, buf)) if u8 else None
            if fn is None:
                continue
        for _ in range(5):
            fn()
        t = []
        for _ in range(50):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        table[f"{name}_{n // 1024}Ki"]["cpu_us"] = round(float(np.median(t)) * 1e6, 2)
    if "reader_chain_32Ki" in table:  # the same pipeline's arithmetic on one core: convert, Shift, Gain over a 32 Ki block
        blk = 32 * 1024
        src, buf, sh = synth_u8(5, blk), np.zeros(blk, np.complex64), o.Shifter(2_400_000)
        t = []
        for _ in range(30):
            t0 = time.perf_counter()
            o.convert(buf, src)
            sh(3.1e5, buf)
            o.scale(buf, 0.5)
            t.append(time.perf_counter() - t0)
        table["reader_chain_32Ki"]["cpu_us"] = round(float(np.median(t)) * 1e6, 2)


if __name__ == "__main__":
    sys.exit(main() or 0)

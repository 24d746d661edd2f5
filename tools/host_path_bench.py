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
    # the same chain behind the pinned ring: the producer's copy into the slot is NOT
    # timed when fill=False (a driver callback writes there directly, rtl/rx.go:49-68)
    for slot_log2, slots in ((20, 4), (22, 4), (22, 8)):
        sl = 1 << slot_log2
        ch2 = ctx.chain(hz.FMT_U8, 20_000_000).shift(-2.5e6).fir_decimate(taps, 8)
        ring = ch2.ring(sl, slots)
        for fill in (False, True):
            total = 1 << 27

            def run():
                done = 0
                sink = 0.0
                for k in range(total // sl):
                    if ring.in_flight == slots:
                        sink += float(ring.pop()[0].real)
                    slot, iq = ring.acquire()
                    if fill:
                        iq[:] = xu8[(k * sl) % n:(k * sl) % n + sl]
                    ring.submit(slot)
                    done += sl
                while ring.in_flight:
                    sink += float(ring.pop()[0].real)
                return done
            t = best(run, reps=3)
            rows.append((f"chain via ring, slot 2^{slot_log2} x{slots}{', memcpy into slot' if fill else ''}", total, t, 3))
        ring.close()
        ch2.close()
    # round 6: the ring hands its slots to the chain in GROUPS (hzsdr_ring_submit_many: one launch of the matrix kernel
    # over the group, the chain pipelined: consecutive launches overlap) -- the driver callback fills `group` slots,
    # then submits them together
    for slot_log2, slots, group in ((20, 9, 4), (22, 9, 4), (22, 9, 2), (22, 17, 8), (24, 5, 2)):
        sl = 1 << slot_log2
        ch2 = ctx.chain(hz.FMT_U8, 20_000_000).shift(-2.5e6).fir_decimate(taps, 8)
        ch2.pipeline(True)
        ring = ch2.ring(sl, slots)
        total = 1 << 28

        def run_groups():
            sink, k = 0.0, 0
            while k < total // sl:
                while ring.in_flight + group > slots:
                    sink += float(ring.pop()[0].real)
                first = None
                for _ in range(group):
                    slot, _iq = ring.acquire()
                    first = slot if first is None else first
                ring.submit_many(first, group)
                k += group
            while ring.in_flight:
                sink += float(ring.pop()[0].real)
            return sink
        t = best(run_groups, reps=3)
        rows.append((f"chain via ring, slot 2^{slot_log2} x{slots}, {group} slots per launch, overlapped", total, t, 3))
        ring.close()
        ch2.close()
    for name, n_, t, bps in rows:
        print(f"{name:74s} {t * 1e3:8.2f} ms  {n_ / t / 1e6:9.1f} Msamples/s  {bps * n_ / t / 1e9:6.1f} GB/s over PCIe")
    ctx.close()


if __name__ == "__main__":
    main()

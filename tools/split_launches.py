#!/usr/bin/env python3
"""The north-star kernel's launches in a rocprofv3 kernel trace, split by what a launch covered.

`bench.py` launches the same instantiation over ONE buffer (parity, the `kernel_ms_unpipelined` leg) and over
`--batch` buffers (the timed steps), so the per-name average of `--stats` mixes the two.  This reads the
`*_kernel_trace.csv` beside it and lists them apart (split at the widest relative gap between the sorted durations; printed).

    python3 tools/split_launches.py gpurun_out/prof_r05/bench_trace_batch [more dirs]
"""
import csv
import glob
import statistics as st
import sys


def main():
    for d in sys.argv[1:]:
        files = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)
        if not files:
            print(f"{d}: no kernel trace")
            continue
        import os
        newest = max(files, key=os.path.getmtime)  # (a directory collects one file set per profiled run)
        du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
              for r in csv.DictReader(open(newest)) if "fir_mm2_kernel" in r["Kernel_Name"]]
        # The two kinds differ by the buffers a launch covered, i.e. by a factor in duration -- not by a fixed number of
        # microseconds (round 5 cut at 70 us, right for 2^24-sample buffers on that box only): the cut is the widest
        # gap between neighbouring sorted durations, accepted when the means on its two sides differ by 1.8 x or more.
        srt = sorted(du)
        cut = None
        if len(srt) >= 2:
            gaps = [(srt[i + 1] / srt[i], i) for i in range(len(srt) - 1) if srt[i] > 0]
            ratio, at = max(gaps)
            lo, hi = srt[:at + 1], srt[at + 1:]
            if ratio >= 1.25 and st.mean(hi) >= 1.8 * st.mean(lo):
                cut = (srt[at] + srt[at + 1]) / 2
        one = [x for x in du if cut is None or x <= cut]
        many = [x for x in du if cut is not None and x > cut]
        print(f"{d.rstrip('/').split('/')[-1]}: {len(du)} launches of hz::mm2::fir_mm2_kernel"
              + (f" (one kind of launch: no gap of 1.8 x between the durations)" if cut is None else f" (split at {cut:.1f} us: the widest gap between the sorted durations)"))
        for name, xs in (("one buffer per launch", one), ("several buffers per launch", many)):
            if xs:
                print(f"    {name:28s} {len(xs):5d} launches   mean {st.mean(xs):7.2f} us   median {st.median(xs):7.2f}   "
                      f"min {min(xs):7.2f}   max {max(xs):7.2f}")


if __name__ == "__main__":
    main()

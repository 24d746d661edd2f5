#!/usr/bin/env python3
"""The north-star kernel's launches in a rocprofv3 kernel trace, split by what a launch covered.

`bench.py` launches the same instantiation over ONE buffer (parity, the `kernel_ms_unpipelined` leg) and over
`--batch` buffers (the timed steps), so the per-name average of `--stats` mixes the two.  This reads the
`*_kernel_trace.csv` beside it and lists them apart (a one-buffer launch is under 70 us, a four-buffer one over).

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
        one = [x for x in du if x <= 70]
        many = [x for x in du if x > 70]
        print(f"{d.rstrip('/').split('/')[-1]}: {len(du)} launches of hz::mm2::fir_mm2_kernel")
        for name, xs in (("one buffer per launch", one), ("several buffers per launch", many)):
            if xs:
                print(f"    {name:28s} {len(xs):5d} launches   mean {st.mean(xs):7.2f} us   median {st.median(xs):7.2f}   "
                      f"min {min(xs):7.2f}   max {max(xs):7.2f}")


if __name__ == "__main__":
    main()

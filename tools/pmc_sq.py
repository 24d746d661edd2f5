#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel: average per launch over launches 2.. .
    python3 tools/pmc_sq.py <dir> [<dir> ...]"""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            short = name.split("(")[0].replace("void hz::", "")[:60]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "fir_" not in k and "conv_" not in k and "chain" not in k and "fft2_" not in k:
                continue
            print(k)
            for c, vals in sorted(cs.items()):
                v = vals[1:] if len(vals) > 2 else vals
                print(f"    {c:28s} {sum(v) / len(v) / 1e6:10.3f} M   ({len(vals)} launches)")

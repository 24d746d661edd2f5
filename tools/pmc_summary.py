#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, csv) into
per-launch HBM traffic for the hot kernels, following MI355X_MICROARCH.md's
rocprofv3/HBM section:

  * FETCH_SIZE and WRITE_SIZE are collected in separate passes (TCC slots);
  * both are in KiB per dispatch (x 1024 for bytes);
  * on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced read,
    so the read side is doubled; other access widths are "uncalibrated", so the
    factor is CHECKED here on a kernel with a known byte count in the same access
    pattern (the u8->c64 converter: 2 B/sample read with 4-byte-per-lane loads,
    8 B/sample written with 16-byte-per-lane stores).

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [log2n]
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    # steady state: drop the first launch of each kernel, average the rest
    return {k: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for k, v in agg.items()}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    n = 1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 24)
    fetch = per_kernel(fetch_dir, "FETCH_SIZE")
    write = per_kernel(write_dir, "WRITE_SIZE")
    res = {"samples_per_launch": n, "unit_note": "counters are KiB per dispatch; read side x2 on gfx950", "kernels": {}}
    cal = None
    for k in sorted(set(fetch) | set(write)):
        if "hz::" not in k:
            continue
        short = k.split("(")[0].replace("void ", "")
        fr, wr = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        res["kernels"][short] = {"FETCH_SIZE_bytes_raw": round(fr), "WRITE_SIZE_bytes": round(wr),
                                 "read_bytes_corrected_x2": round(2 * fr), "hbm_bytes": round(2 * fr + wr)}
        if any(t in short for t in ("convert_vec_kernel<0>", "convert_vec_kernel<(hz::Conv)0>",
                                    "convert_vec_kernel<0, 0>", "convert_vec_kernel<0, 0, true>", "convert_vec_kernel<0, 0, false>",
                                    "convert_tile_kernel<0, 0, 2>")):
            cal = {"kernel": short, "known_read_bytes": 2 * n, "known_write_bytes": 8 * n,
                   "read_factor_needed": round(2 * n / fr, 3) if fr else None,
                   "write_factor_needed": round(8 * n / wr, 3) if wr else None}
    res["calibration_u8_to_c64"] = cal
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

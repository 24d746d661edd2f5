#!/usr/bin/env python3
"""Short view of a bench.py JSON line: value, dominant-kernel time, every side measurement."""
import json
import sys

r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print(f"{r['value']:.1f} {r['unit']}  ms_per_step {r['ms_per_step']}  kernel_ms {rf['kernel_ms']}  "
      f"{rf['bound']} frac {rf['frac']}  traffic {rf['traffic']}")
for k, v in r.get("extra", {}).items():
    if isinstance(v, dict) and "kernel_ms" in v:
        print(f"  {k:28s} {v['kernel_ms']:8.4f} ms  {v['GBps']:8.1f} GB/s  frac {v['hbm_frac']}")
if "cpu_baseline" in r:
    print("  cpu_baseline", r["cpu_baseline"]["value"], r["cpu_baseline"]["unit"])

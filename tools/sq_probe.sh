#!/bin/bash
# SQ counters of fir_mm_kernel (two passes), averaged per launch
export TMPDIR=/tmp
rm -rf /tmp/sq1 /tmp/sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/sq1 -- python3 tools/firmm_probe.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/sq2 -- python3 tools/firmm_probe.py > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("/tmp/sq1","/tmp/sq2"):
    tot=collections.defaultdict(float); cnt=0
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "fir_mm_kernel" not in r["Kernel_Name"]: continue
            tot[r["Counter_Name"]]+=float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES","SQ_INSTS_LDS"): cnt+=1
    print({c: round(v/max(cnt,1)) for c,v in tot.items()}, "launches", cnt)
PY

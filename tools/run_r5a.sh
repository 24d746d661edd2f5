set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5i
for b in 1 2 4 8; do
  timeout 600 python bench.py --no-extra --no-oracle --batch $b > gpurun_out/r5i/bench_b$b.json 2>/dev/null
  timeout 600 python bench.py --no-extra --no-oracle --no-pipeline --batch $b > gpurun_out/r5i/bench_plain_b$b.json 2>/dev/null
done
python - <<'PY'
import json
for b in (1,2,4,8):
    for kind in ('b','plain_b'):
        d=json.load(open('gpurun_out/r5i/bench_%s%d.json'%(kind,b))); r=d['roofline']
        print(kind, b, d['value'], d['ms_per_step'], r['kernel_ms'], r.get('kernel_ms_unpipelined'), r.get('kernel_ms_batch_unpipelined'))
PY

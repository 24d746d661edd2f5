set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5e
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5e/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r5e/tests.log
tail -6 gpurun_out/r5e/tests.log
timeout 900 python bench.py > gpurun_out/r5e/bench_full.json 2> gpurun_out/r5e/bench_full.err
tail -3 gpurun_out/r5e/bench_full.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5e/bench_full.json')); r=d['roofline']
print('headline', d['value'], d['ms_per_step'], r['kernel_ms'], r.get('kernel_ms_unpipelined'), r.get('kernel_ms_batch_unpipelined'), d.get('parity',{}).get('ok'))
for k,v in d.get('extra',{}).items():
    if isinstance(v,dict) and 'kernel_ms' in v:
        h=v.get('hbm',{})
        print('%-32s %8.4f ms %7.1f GB/s | hbm %s %s frac_copy %s' % (k, v['kernel_ms'], v.get('GBps',0), h.get('kernel_ms'), h.get('GBps'), h.get('frac_of_device_copy')))
print(d.get('cpu_baseline',{}).get('value'))
PY

set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5d
timeout 900 python -m pytest tests/test_gpu_firmm.py tests/test_gpu_fullsize.py tests/test_gpu_ring.py tests/test_gpu_fir_fuzz.py tests/test_gpu_latemix.py -x -q -m gpu > gpurun_out/r5d/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r5d/tests.log
tail -4 gpurun_out/r5d/tests.log



timeout 600 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r5d/bench_default.json 2> gpurun_out/r5d/bench_default.err
timeout 600 python bench.py --no-extra --no-oracle --no-pipeline --batch 1 > gpurun_out/r5d/bench_plain.json 2>/dev/null



python -c "
import json
for f in ('bench_default','bench_plain'):
    d=json.load(open('gpurun_out/r5d/%s.json'%f)); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['kernel_ms'], r.get('kernel_ms_unpipelined'), r.get('kernel_ms_batch_unpipelined'), d.get('parity',{}).get('ok'), d.get('parity',{}).get('rel_l2_err'))
"

set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_firmm.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r5b/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r5b/tests.log
tail -4 gpurun_out/r5b/tests.log
AB=1 timeout 300 tools/bin/mfma_fir2 > gpurun_out/r5b/ab.txt 2>&1
timeout 300 tools/bin/mfma_fir2 > gpurun_out/r5b/stamps.txt 2>&1
BATCH=4 timeout 120 tools/bin/mfma_fir2 > gpurun_out/r5b/batch4.txt 2>&1
timeout 600 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r5b/bench_default.json 2> gpurun_out/r5b/bench_default.err
timeout 600 python bench.py --no-extra --no-oracle --no-pipeline --batch 1 > gpurun_out/r5b/bench_plain.json 2>/dev/null
grep EXP gpurun_out/r5b/ab.txt | cut -c1-40,95-200
grep EXP gpurun_out/r5b/batch4.txt | cut -c1-40,95-200
grep -v "wg 10" gpurun_out/r5b/stamps.txt | cut -c1-200 | tail -60
python -c "
import json
for f in ('bench_default','bench_plain'):
    d=json.load(open('gpurun_out/r5b/%s.json'%f)); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['kernel_ms'], r.get('kernel_ms_unpipelined'), r.get('kernel_ms_batch_unpipelined'), d.get('parity',{}).get('ok'), d.get('parity',{}).get('rel_l2_err'))
"

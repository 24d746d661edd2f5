set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5h
timeout 1500 python -m pytest tests/test_gpu_mgpu.py tests/test_gpu_stream.py -x -q -m gpu > gpurun_out/r5h/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r5h/tests.log
tail -30 gpurun_out/r5h/tests.log

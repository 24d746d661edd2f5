#!/bin/bash
# What rocm-smi says about the chip while the benchmarked chain runs: board power against its cap, the shader and
# memory clocks -- sampled every 100 ms beside `python3 bench.py --no-extra --no-oracle --steps 40000` (a 1.2 s timed loop),
# once over the bench's random bytes and once with HZ_BENCH_CONSTANT_INPUT=1 (every input byte 0x80).
#   bash tools/power_watch.sh > gpurun_out/r06_power_watch.txt
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk" | head -12
for mode in random constant; do
  echo "== $mode input"
  if [ $mode = constant ]; then export HZ_BENCH_CONSTANT_INPUT=1; else unset HZ_BENCH_CONSTANT_INPUT; fi
  python3 bench.py --no-extra --no-oracle --steps 40000 --warmup 400 > /tmp/pw_$mode.json 2>/dev/null &
  pid=$!
  sleep 6
  for i in $(seq 1 40); do
    kill -0 $pid 2>/dev/null || break
    rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Average Graphics Package Power\|Current Socket Graphics Package Power\|sclk clock level\|mclk clock level" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
    echo
    sleep 0.1
  done
  wait $pid
  python3 -c "import json; d=json.loads(open('/tmp/pw_$mode.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
done

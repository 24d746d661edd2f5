for r in 1 2 3; do
for cfg in "" "--batch 8" "--batch 8 --no-pipeline" "--batch 4 --no-pipeline" "--batch 2" "--batch 1"; do
  for st in "--steps 20 --warmup 5" ""; do
    python bench.py --no-extra --no-oracle $cfg $st 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-26s %-22s ms_per_step %.4f' % ('$cfg' or 'default (batch 4, overlapped)', '$st' or '(450 steps)', d['ms_per_step']))"
  done
done
done

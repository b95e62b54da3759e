#!/bin/bash
set -o pipefail
UOCR_H3=1 timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 || exit 1
for band in 8 12 16 24; do
  echo "== h3=1 band=$band"
  timeout -k 10 200 python tools/bench_conv.py --filter line.end --option h3=1 --option pair_band=$band --reps 30 2>&1 | grep "fwd"
done
for i in 1 2 3; do
for h3 in 0 1; do
  UOCR_H3=$h3 timeout -k 10 300 python bench.py --config train-b32 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>> gpurun_out/h3_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train-b32 h3=$h3', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'], d['cpu_baseline'] if 'cpu_baseline' in d else '')" || tail -3 gpurun_out/h3_ab.err
done
done

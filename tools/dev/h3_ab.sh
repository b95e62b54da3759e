#!/bin/bash
# verdict item 7: the Line output conv forward on error-compensated binary16 MFMAs (ctx option h3) against the vector kernel
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "h3" 2>&1 | tail -5 || exit 1
for h3 in 0 1; do
  echo "== h3=$h3"
  timeout -k 10 200 python tools/bench_conv.py --filter line.end --option h3=$h3 --reps 30 2>&1 | grep -v amdgpu.ids
done
for band in 16 32 64 128; do
  echo "== h3=1 band=$band"
  timeout -k 10 200 python tools/bench_conv.py --filter line.end --option h3=1 --option pair_band=$band --reps 30 2>&1 | grep "fwd"
done
for h3 in 0 1; do
  UOCR_H3=$h3 timeout -k 10 300 python bench.py --config train-b32 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>> gpurun_out/h3_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train-b32 h3=$h3', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])" || tail -3 gpurun_out/h3_ab.err
done

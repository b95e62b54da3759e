#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py tests/test_gpu_f16.py tests/test_gpu_bounds.py -x -q -m gpu -k "loss or cross_entropy or softmax or seg" 2>&1 | tail -8 || exit 1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 || exit 1
for cfg in train-b32 highres-fp16; do
  timeout -k 10 400 python bench.py --config $cfg --no-cpu-baseline --steps 50 --warmup 10 2> gpurun_out/loss_bench_$cfg.err > gpurun_out/loss_bench_$cfg.json || { tail -5 gpurun_out/loss_bench_$cfg.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/loss_bench_$cfg.json').read().strip().splitlines()[-1])
print('$cfg', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'], 'solo', d['roofline'].get('solo_launch_us'), 'in-loop', d['roofline'].get('in_loop_launch_us'))
for r in d['roofline'].get('secondary', []):
    print('   ', r)
PY
done

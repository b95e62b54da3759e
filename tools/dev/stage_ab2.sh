#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_bounds.py tests/test_gpu_models.py tests/test_gpu_f16.py -x -q -m gpu 2>&1 | tail -2 || exit 1
timeout -k 10 300 python tools/bench_conv.py --filter "line." --reps 30 2>&1 | grep "line.end\|line.down_1\|(fused)"
timeout -k 10 200 python tools/bench_nets.py --graphs --steps 60 --only Line 2>/dev/null | grep -v "^$"
for i in 1 2; do
timeout -k 10 300 python bench.py --config train-b32 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>> gpurun_out/stage_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train-b32', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])"
done

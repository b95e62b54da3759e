#!/bin/bash
# persistent tile loops without per-tile divisions (TileWalk): tests, binary16 micro-benchmark, the two steps
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_f16.py tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_bounds.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 300 python3 tools/bench_h16.py 2>&1 | tail -24
timeout -k 10 300 python tools/bench_conv.py --filter "line.end" --reps 30 2>&1 | grep "line"
for cfg in highres-fp16 train-b32 highres-fp16 train-b32; do
timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>> gpurun_out/walk_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])"
done

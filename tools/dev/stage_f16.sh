#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f16.py tests/test_gpu_bounds.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -2 || exit 1
timeout -k 10 300 python3 tools/bench_h16.py 2>&1 | grep "par up"
for i in 1 2; do
timeout -k 10 300 python bench.py --config highres-fp16 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>> gpurun_out/stage_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('highres-fp16', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])"
done

#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_models.py tests/test_gpu_f16.py -x -q -m gpu -k "up or config or net or graph" 2>&1 | tail -3 || exit 1
bash tools/dev/quick_stats.sh train-b32 > gpurun_out/quick_stats_train.txt 2>&1 || exit 1
grep -i "finish\|upconv\|kernel time" gpurun_out/quick_stats_train.txt
timeout -k 10 200 python tools/bench_nets.py --graphs --steps 60 --only Line 2>/dev/null | grep -v "^$"
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])"; done

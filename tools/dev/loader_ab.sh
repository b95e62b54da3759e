#!/bin/bash
# conv loaders of the MFMA GEMM without integer divisions: tests, the microbenchmark rows they serve, Char alone, the step
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -x -q -m gpu -k "mfma or dense or golden or config or wide or deferred or windows" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python tools/bench_conv.py --filter "char." --reps 30 2>&1 | grep "char"
timeout -k 10 300 python tools/bench_conv.py --filter "wide" --reps 5 2>&1 | grep "wide"
timeout -k 10 200 python tools/bench_nets.py --graphs --steps 60 --only Char 2>/dev/null | grep -v "^$"
for i in 1 2; do
timeout -k 10 300 python bench.py --config train-b32 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>> gpurun_out/loader_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train-b32', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])"
done

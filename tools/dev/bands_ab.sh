#!/bin/bash
# auto band target of conv_wgrad_fast (64 for the 128-accumulator kernels) against the old 512 everywhere
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_f16.py tests/test_gpu_bounds.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -2 || exit 1
nets() { timeout -k 10 200 python tools/bench_nets.py --graphs --steps 60 --only "$2" --option wgrad_bands=$1 2>/dev/null | grep -v "^$" | sed "s/^/bands=$1  /"; }
for n in Line Char Monochrome,Paragraph; do nets 512 $n; nets 0 $n; done
for cfg in train-b32 highres-fp16; do for b in 512 0 512 0; do
UOCR_WGRAD_BANDS=$b timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>> gpurun_out/bands_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg bands=$b', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])"
done; done

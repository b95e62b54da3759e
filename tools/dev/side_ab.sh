#!/bin/bash
# A/B of weight gradients on a side stream (UOCR_SIDE_WGRAD) -- nets alone and the whole step
set -o pipefail
nets() { UOCR_SIDE_WGRAD=$1 timeout -k 10 200 python tools/bench_nets.py --graphs --steps 60 --only "$2" 2>> gpurun_out/side_ab.err | grep -v "^$" | sed "s/^/side=$1  /"; }
run() { UOCR_SIDE_WGRAD=$2 timeout -k 10 300 python bench.py --config $1 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>> gpurun_out/side_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'side=$2', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])" || tail -3 gpurun_out/side_ab.err; }
: > gpurun_out/side_ab.err
timeout -k 10 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_data_parallel.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for net in Char Line Paragraph; do
  nets none $net
  nets all $net
done
for cfg in train-b32 highres-fp16; do
  run $cfg none
  run $cfg Char
  run $cfg Char,Line
  run $cfg all
done

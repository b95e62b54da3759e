#!/bin/bash
# deferred weight-gradient groups (UOCR_GROUP_WGRAD): tests, the nets alone, the page step
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_data_parallel.py tests/test_gpu_f16.py -x -q -m gpu -k "deferred or grouped or graph or rccl or ranks or side or config4 or pair" 2>&1 | tail -5 || exit 1
nets() { UOCR_GROUP_WGRAD=$1 timeout -k 10 200 python tools/bench_nets.py --graphs --steps 60 --only "$2" 2>> gpurun_out/group_ab.err | grep -v "^$" | sed "s/^/group=$1  /"; }
run() { UOCR_GROUP_WGRAD=$2 timeout -k 10 300 python bench.py --config $1 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>> gpurun_out/group_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'group=$2', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])" || tail -3 gpurun_out/group_ab.err; }
: > gpurun_out/group_ab.err
for net in Paragraph Line Monochrome,Paragraph; do nets none $net; nets all $net; done
for cfg in train-b32 highres-fp16; do
run $cfg none
run $cfg Char
run $cfg all
run $cfg none
run $cfg all
done

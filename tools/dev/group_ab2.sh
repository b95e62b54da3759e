#!/bin/bash
set -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 || exit 1
run() { UOCR_GROUP_WGRAD=$2 timeout -k 10 300 python bench.py --config $1 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>> gpurun_out/group_ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'group=$2', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])" || tail -3 gpurun_out/group_ab.err; }
: > gpurun_out/group_ab.err
for cfg in train-b32 highres-fp16; do
run $cfg none
run $cfg all
run $cfg none
run $cfg all
done

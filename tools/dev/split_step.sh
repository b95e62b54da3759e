#!/bin/bash
# the page step under the GEMM's split options: does a cheaper (unsplit, fewer reduce launches) Char net help when the lanes share the chip?
for o in "mfma=1" "split_blocks=0" "split_min=5" "split_blocks=512" "mfma=1" "split_min=9"; do
timeout -k 10 300 python bench.py --config train-b32 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 --option $o 2>> gpurun_out/split_step.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])"
done

#!/bin/bash
set -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 || exit 1
for i in 1 2; do
for cfg in train-b32 highres-fp16; do
timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'], d['config'].get('final_losses'))"
done
done

#!/bin/bash
# lane groupings of the page step (steady-state median ms/step)
run() { timeout -k 10 300 python bench.py --config $1 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 $2 $3 2> gpurun_out/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2 $3', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])" || tail -3 gpurun_out/ab.err; }
for cfg in train-b32; do
  run $cfg
  run $cfg --lane-per-net
  run $cfg --lane-groups Monochrome+Char,Paragraph+Line
  run $cfg --lane-groups Monochrome+Line,Paragraph+Char
  run $cfg --lane-groups Monochrome+Paragraph+Char,Line
  run $cfg --lane-groups Monochrome,Paragraph+Char,Line
  run $cfg --lane-groups Monochrome+Paragraph,Line+Char
  run $cfg
done

#!/bin/bash
run() { timeout -k 10 300 python bench.py --config $1 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 $2 $3 2> gpurun_out/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2 $3', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'])" || tail -3 gpurun_out/ab.err; }
run highres-fp16
run highres-fp16 --lane-per-net
run highres-fp16 --lane-groups Monochrome+Paragraph+Line
run highres-fp16 --lane-groups Monochrome+Line,Paragraph
run highres-fp16 --lane-groups Monochrome,Paragraph+Line
run train-b32

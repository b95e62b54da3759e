#!/bin/bash
# A/B of CU-partitioned lanes against the free-for-all (steady-state median ms/step)
run() { timeout -k 10 300 python bench.py --config $1 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 $2 2> gpurun_out/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'], 'in-loop', d['roofline']['in_loop_launch_us'])" || tail -3 gpurun_out/ab.err; }
for cfg in train-b32 highres-fp16; do
  run $cfg ""
  run $cfg "--lane-xcds 0-2,3-5,6-7"
  run $cfg "--lane-xcds 0-3,4-6,7-7"
  run $cfg "--lane-xcds 0-1,2-4,5-7"
  run $cfg "--lane-xcds 0-7,0-7,0-7"
done

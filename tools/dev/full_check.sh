#!/bin/bash
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; tail -6 gpurun_out/gputests.log
for c in train-b32 highres-fp16; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-secondary > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_$c.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$c", d["value"], d["ms_per_step"], "steady", d.get("steady_state",{}).get("ms_per_step_median"), "solo", r["solo_launch_us"], "frac", r["frac"], "in-loop", r["in_loop_launch_us"])
PY
done
UOCR_BENCH_FORCE_DP=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/bench_dp1.json 2> gpurun_out/bench_dp1.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_dp1.json").read().strip().splitlines()[-1])
print("one-rank rccl", d["value"], d["ms_per_step"], d.get("dp"))
PY
tail -3 gpurun_out/bench_dp1.err

#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu  2>&1 | tail -4 || exit 1
bash tools/dev/quick_stats.sh train-b32 > gpurun_out/quick_stats_train.txt 2>&1 || exit 1
bash tools/dev/quick_stats.sh highres-fp16 > gpurun_out/quick_stats_f16.txt 2>&1 || exit 1
grep -i "seg_\|finish\|kernel time\|opt_fused" gpurun_out/quick_stats_train.txt gpurun_out/quick_stats_f16.txt
cat gpurun_out/quick_train-b32/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
cat gpurun_out/quick_highres-fp16/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"

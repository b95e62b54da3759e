#!/bin/bash
run() { UOCR_BENCH_FORCE_DP=1 UOCR_GROUP_WGRAD=$1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dp1 group=$1', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step_median'], d['dp']['lane_wait_us_mean'])"; }
run none; run Char; run all; run none; run Char; run all

#!/bin/bash
# kernel stats of the bench command (config = $1, default train-b32) -> gpurun_out/quick_stats_<config>.txt
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-train-b32}
OUT=$R/gpurun_out/quick_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT" --output-format csv -- \
    python3 "$R/bench.py" --config $CFG --steps 20 --warmup 5 --steady-steps 0 --no-cpu-baseline --no-secondary > "$OUT/bench.json" 2> "$OUT/err.txt"
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
paths = [p for p in glob.glob(f'{out}/*/*_kernel_stats.csv') if 'pair' in open(p).read()]
rows = list(csv.DictReader(open(paths[0])))
steps = 46
total = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print(f'kernel time {total/1e6/steps:.3f} ms/step, {calls/steps:.1f} kernels/step')
for r in rows[:60]:
    name = r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    print(f"{name[:90]:90s} n/step={int(r['Calls'])/steps:5.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} {float(r['Percentage']):5.1f}%")
PY

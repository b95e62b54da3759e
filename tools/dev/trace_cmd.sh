#!/bin/bash
# kernel-trace of a python command: per-kernel mean duration  ->  stdout
# usage: bash tools/dev/trace_cmd.sh <tag> <script.py> [args...]
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$SCRIPT" "$@" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt" || echo "run failed"
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(p)):
        d[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    name = k.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    print(f'{name[:100]:100s} n={len(v):4d} mean={sum(v)/len(v):9.1f} min={min(v):9.1f} max={max(v):9.1f}')
PY

#!/bin/bash
# usage: prof_net.sh <Net> : rocprofv3 kernel stats of one net trained alone (35 steps)
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for net in "$@"; do
  rm -rf "$R/gpurun_out/prof_$net"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_$net" --output-format csv -- python3 "$R/tools/bench_nets.py" --only "$net" --steps 30 > "$R/gpurun_out/prof_$net.out" 2> "$R/gpurun_out/prof_$net.err"
  python3 "$R/tools/prof_summary.py" "$R/gpurun_out/prof_$net" 35 40 > "$R/gpurun_out/prof_$net.txt" || true
done

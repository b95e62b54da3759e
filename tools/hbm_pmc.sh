#!/bin/bash
# HBM bytes of the pooling / activation / upsampling / loss kernels from the rocprofv3 counters, next to their HIP-event
# rates (run through gpurun from the repo root):  bash tools/hbm_pmc.sh  -> gpurun_out/hbm/{events.txt,fetch,write,summary.txt}
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/hbm
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$R/tools/hbm_kernels.py" > "$OUT/events.txt" 2> "$OUT/events.err"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- \
    python3 "$R/tools/hbm_kernels.py" > "$OUT/fetch.txt" 2> "$OUT/fetch.err"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- \
    python3 "$R/tools/hbm_kernels.py" > "$OUT/write.txt" 2> "$OUT/write.err"
cd "$R"
python3 tools/hbm_summary.py "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"

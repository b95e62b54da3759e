#!/bin/bash
# Regenerates the raw material of profiles/ on a GPU box (run from the repo root through gpurun):
#   bash tools/refresh_profiles.sh stats   -> kernel stats of the bench command + the plain bench line + microbenches
#   bash tools/refresh_profiles.sh pmc     -> FETCH_SIZE / WRITE_SIZE passes (counters in their own runs)
# Outputs land in gpurun_out/refresh/; copy what is to be judged into profiles/.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/refresh
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
case "$1" in
  stats)
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- \
        python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprofv3.json" 2> "$OUT/stats.err"
    cd "$R"
    timeout -k 10 400 python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench.err"
    timeout -k 10 400 python3 tools/bench_conv.py > "$OUT/conv_microbench.txt" 2>&1
    timeout -k 10 300 python3 tools/bench_membw.py > "$OUT/membw.txt" 2>&1
    ;;
  pmc)
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" --output-format csv -- \
        python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" --output-format csv -- \
        python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
    ;;
  *) echo "usage: $0 stats|pmc" >&2; exit 2 ;;
esac
echo done "$1"

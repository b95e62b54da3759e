#!/bin/bash
# Regenerates the raw material of profiles/ on a GPU box (run from the repo root through gpurun):
#   bash tools/refresh_profiles.sh stats   -> kernel stats of the bench commands (train-b32, highres-fp16) + the plain
#                                             bench lines of the three configurations + micro-benchmarks
#   bash tools/refresh_profiles.sh pmc     -> FETCH_SIZE / WRITE_SIZE passes (counters in their own runs) + the
#                                             instruction-mix passes of the dominant kernel (tools/pmc_pair.sh)
# Outputs land in gpurun_out/refresh/; tools/install_profiles.py copies what is to be judged into profiles/.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/refresh
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
case "$1" in
  stats)
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- \
        python3 "$R/bench.py" --steps 20 --warmup 5 --steady-steps 0 --no-cpu-baseline --no-secondary > "$OUT/bench_under_rocprofv3.json" 2> "$OUT/stats.err"
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_hr" --output-format csv -- \
        python3 "$R/bench.py" --config highres-fp16 --steps 20 --warmup 5 --steady-steps 0 --no-cpu-baseline --no-secondary > "$OUT/bench_hr_under_rocprofv3.json" 2> "$OUT/stats_hr.err"
    cd "$R"
    timeout -k 10 500 python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench.err"
    timeout -k 10 500 python3 bench.py --config highres-fp16 > "$OUT/bench_hr_n1.json" 2> "$OUT/bench_hr.err"
    timeout -k 10 500 python3 bench.py --config infer-b8 --no-secondary > "$OUT/bench_infer_n1.json" 2> "$OUT/bench_infer.err"
    UOCR_BENCH_FORCE_DP=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_dp1_rehearsal.json" 2> "$OUT/bench_dp1.err"
    timeout -k 10 400 python3 tools/bench_conv.py > "$OUT/conv_microbench.txt" 2>&1
    timeout -k 10 300 python3 tools/bench_membw.py > "$OUT/membw.txt" 2>&1
    timeout -k 10 300 python3 tools/bench_h16.py > "$OUT/bench_h16.txt" 2>&1
    timeout -k 10 200 python3 tools/bench_conv.py --filter mono.pair --dtype float16 --batch 8 --height 1024 --width 2048 --pair-act none > "$OUT/pair_f16_microbench.txt" 2>&1
    timeout -k 10 200 python3 tools/bench_nets.py --graphs > "$OUT/bench_nets.txt" 2>&1
    hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_coexec.hip -o /tmp/coexec 2> /dev/null && timeout -k 5 60 /tmp/coexec > "$OUT/ubench_coexec.txt"
    hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_switch.hip -o /tmp/sw 2> /dev/null && timeout -k 5 60 /tmp/sw > "$OUT/ubench_switch.txt"
    ;;
  pmc)
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" --output-format csv -- \
        python3 "$R/bench.py" --steps 3 --warmup 1 --steady-steps 0 --no-cpu-baseline --no-secondary > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" --output-format csv -- \
        python3 "$R/bench.py" --steps 3 --warmup 1 --steady-steps 0 --no-cpu-baseline --no-secondary > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
    cd "$R"
    bash tools/pmc_pair.sh refresh pair > "$OUT/pmc_pair_summary.txt" 2>&1
    ;;
  *) echo "usage: $0 stats|pmc" >&2; exit 2 ;;
esac
echo done "$1"

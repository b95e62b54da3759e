#!/bin/bash
# PMC passes over any python command (run through gpurun from the repo root):
#   bash tools/pmc_cmd.sh <tag> <kernel-name-filter> <script.py> [args...]
#   -> gpurun_out/pmc_<tag>/pass*/..._counter_collection.csv + a per-kernel summary on stdout
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; FILTER=$2; shift 2
OUT=$R/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pass$i" -- \
      python3 "$SCRIPT" "$@" > "$OUT/pass$i.txt" 2> "$OUT/pass$i.err" || echo "pass $i failed"
done
cd "$R"
python3 tools/pmc_summary.py "$OUT" "$FILTER"

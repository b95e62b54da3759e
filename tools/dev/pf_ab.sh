#!/bin/bash
# A/B of the row prefetch forms of the pair forward kernels (ctx option pair_pf) + the pair tests for each form
set -o pipefail
for pf in 0 1 2; do
  echo "== pair_pf=$pf"
  timeout -k 10 200 python tools/bench_conv.py --filter pair --option pair_pf=$pf --reps 30 2>&1 | grep -v amdgpu.ids
  timeout -k 10 200 python tools/bench_conv.py --filter pair --option pair_pf=$pf --reps 30 --dtype float16 --batch 8 --height 1024 --width 2048 2>&1 | grep -v amdgpu.ids
done
for pf in 2 1; do
  echo "== tests with UOCR_PAIR_PF=$pf"
  UOCR_PAIR_PF=$pf timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f16.py tests/test_gpu_configs.py -x -q -m gpu -k "pair or config" 2>&1 | tail -4 || exit 1
done

#!/bin/bash
# the Char net alone under the MFMA GEMM's grid options
for o in "mfma=1" "split_min=2" "split_min=4" "split_blocks=768" "split_blocks=1536" "group_blocks=768" "group_blocks=1536" "group_blocks=2048"; do
  echo -n "$o  "; timeout -k 10 200 python tools/bench_nets.py --graphs --steps 60 --only Char --option $o 2>/dev/null | grep -v "^$"
done

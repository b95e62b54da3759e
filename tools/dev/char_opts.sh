#!/bin/bash
for o in "mfma=1" "gemm_bm=128" "split_min=2" "split_min=5" "split_min=9" "split_blocks=512" "split_blocks=2048" "xcd_remap=0"; do
  echo -n "$o  "; timeout -k 10 200 python tools/bench_nets.py --graphs --steps 60 --only Char --option $o 2>/dev/null | grep -v "^$"
done

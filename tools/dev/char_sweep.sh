#!/bin/bash
for o in "mfma=1" "split_blocks=512" "split_blocks=2048" "split_blocks=0" "xcd_remap=0" "gemm_bm=128" "split_min=8"; do echo "== $o"; timeout -k 10 120 python tools/bench_conv.py --filter "char" --option $o 2>&1 | grep "char"; done

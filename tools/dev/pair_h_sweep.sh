#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_f16.py -x -q -k "pair or config4 or page_net" 2>&1 | tail -5
B="python tools/bench_conv.py --filter mono.pair --dtype float16 --batch 8 --height 1024 --width 2048"
echo "== tile kernels"; timeout -k 10 120 $B --option pair=0 2>&1 | grep "mono.pair"
echo "== strips G=4 sigmoid"; timeout -k 10 120 $B --option pair=1 2>&1 | grep "mono.pair"
echo "== strips G=4 no output activation"; timeout -k 10 120 $B --option pair=1 --pair-act none 2>&1 | grep "mono.pair"
echo "== strips G=2 sigmoid"; timeout -k 10 120 $B --option pair=1 --option pair_g=2 2>&1 | grep "mono.pair"
echo "== strips G=2 no output activation"; timeout -k 10 120 $B --option pair=1 --option pair_g=2 --pair-act none 2>&1 | grep "mono.pair"

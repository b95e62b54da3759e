#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_f16.py -x -q -k "pair or config4 or page_net" 2>&1 | tail -5
for o in "pair=0" "pair=1"; do echo "== f16 8x1024x2048 $o"; timeout -k 10 120 python tools/bench_conv.py --filter "mono.pair" --dtype float16 --batch 8 --height 1024 --width 2048 --option $o 2>&1 | grep "mono.pair"; done

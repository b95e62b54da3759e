#!/bin/bash
# correctness (debug script + kernel tests) + timing of the strip kernels under a few option sets
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bounds.py -x -q -k "pair" 2>&1 | tail -3
for o in "pair=1" "pair=1 pair_g=2"; do echo "== debug $o"; timeout -k 10 200 python tools/dev/pair_debug.py $o 2>&1 | grep -o "rel_linf [0-9.e+-]*" | sort -k2 -g | tail -1; done
for o in "pair=0" "pair=1" "pair=1 --option pair_band=16" "pair=1 --option pair_band=8" "pair=1 --option pair_band=32" "pair=1 --option pair_band=64"; do echo "== $o"; timeout -k 10 120 python tools/bench_conv.py --filter "mono.pair" --option $o 2>&1 | grep "mono.pair"; done

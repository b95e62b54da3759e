#!/bin/bash
# correctness (debug script) + timing of the strip kernels under a few option sets
for o in "pair=1" "pair=1 pair_g=2"; do echo "== debug $o"; timeout -k 10 200 python tools/dev/pair_debug.py $o 2>&1 | grep -E "shape|rel_linf" | awk '{printf "%s ", $0} END {print ""}'; done
for o in "pair=0" "pair=1" "pair=1 --option pair_g=2" "pair=1 --option pair_g=2 --option pair_band=16" "pair=1 --option pair_g=2 --option pair_band=64"; do echo "== $o"; timeout -k 10 120 python tools/bench_conv.py --filter "mono.pair" --option $o 2>&1 | grep "mono.pair"; done

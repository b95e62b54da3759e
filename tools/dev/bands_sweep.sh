#!/bin/bash
# row bands of the direct weight-gradient kernels (conv_wgrad_fast): option wgrad_bands
for b in 512 256 128 64 1024; do
  echo "== wgrad_bands=$b"
  timeout -k 10 200 python tools/bench_conv.py --reps 20 --option wgrad_bands=$b 2>&1 | grep "wgrad" | grep "para\|line.down\|char.conv_1\|line.up_2 \|line.up_1 "
done

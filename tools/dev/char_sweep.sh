#!/bin/bash
for o in "mfma=1" "split_blocks=128" "split_blocks=256" "split_blocks=384" "split_blocks=512" "split_blocks=768" "split_blocks=2048"; do echo "== $o"; timeout -k 10 120 python tools/bench_conv.py --filter "char.conv" --reps 30 --option $o 2>&1 | grep "conv_[23]"; done

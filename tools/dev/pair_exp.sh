#!/bin/bash
for e in 0 2 3 4 5; do echo "== exp $e"; timeout -k 10 120 python tools/bench_conv.py --filter "mono.pair" --option pair=1 --option pair_exp=$e 2>&1 | grep "bwd+dx"; done

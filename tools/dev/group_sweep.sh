#!/bin/bash
for gb in 0 768 1536 2048 3072; do
  echo -n "group_blocks=$gb  "; timeout -k 10 200 python tools/bench_nets.py --graphs --steps 60 --only Char --option group_blocks=$gb 2>/dev/null | grep -v "^$"
done

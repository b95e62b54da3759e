#!/bin/bash
for only in "Monochrome,Paragraph" "Line" "Char" "Monochrome,Paragraph,Line" "Monochrome,Paragraph,Char" "Line,Char" "Monochrome,Paragraph,Line,Char"; do
  timeout -k 10 200 python tools/bench_nets.py --graphs --pipelined --steps 100 --only "$only" 2>/dev/null | grep -v "^$"
done

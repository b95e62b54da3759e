#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats --output-format csv directory:
    python tools/prof_summary.py gpurun_out/profN <timed+warmup steps> [top]"""
import csv
import glob
import sys


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rows = list(csv.DictReader(open(glob.glob(f'{d}/*/*_kernel_stats.csv')[0])))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    print(f'kernel time {total / 1e6 / steps:.3f} ms/step, {calls / steps:.1f} kernels/step')
    for r in rows[:top]:
        name = r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
        print(f"{name[:88]:88s} n/step={int(r['Calls']) / steps:5.1f} avg_us={float(r['AverageNs']) / 1e3:8.1f} "
              f"ms/step={float(r['TotalDurationNs']) / 1e6 / steps:7.3f} {float(r['Percentage']):5.1f}%")


if __name__ == '__main__':
    main()

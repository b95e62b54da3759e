#!/usr/bin/env python3
"""Sum the counters of the rocprofv3 --pmc passes under a directory per kernel (mean per launch):
    python tools/pmc_summary.py gpurun_out/pmc_<tag> [name-filter]"""
import collections
import csv
import glob
import sys


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f'{root}/pass*/*/*counter_collection.csv'):
        for row in csv.DictReader(open(path)):
            name = row['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
            if flt in name:
                acc[name[:70]][row['Counter_Name']].append(float(row['Counter_Value']))
    for name, counters in sorted(acc.items()):
        print(name)
        for cname, values in sorted(counters.items()):
            print(f'    {cname:28s} {sum(values) / len(values):16.4e}   ({len(values)} launches)')


if __name__ == '__main__':
    main()

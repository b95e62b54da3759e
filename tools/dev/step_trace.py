#!/usr/bin/env python3
"""One step of the bench as the kernel trace saw it: start (us), duration, queue, kernel -- from a rocprofv3
--kernel-trace directory (tools/dev/quick_stats.sh leaves one under gpurun_out/quick_<config>)."""
import csv, glob, sys
out = sys.argv[1]
paths = sorted(glob.glob(f'{out}/*/*_kernel_trace.csv'), key=lambda p: -len(open(p).read()))
rows = list(csv.DictReader(open(paths[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
opt = [i for i, r in enumerate(rows) if 'opt_fused' in r['Kernel_Name']]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nets = 4 if any('softmax_ce' in r['Kernel_Name'] for r in rows) else 3
lo, hi = opt[-nets * back - 1] + 1, opt[-nets * (back - 1) - 1] + 1
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi]:
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:70]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} q={r['Queue_Id']} "
          f"g={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}/{r['Workgroup_Size_X']} {n}")

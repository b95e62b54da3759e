#!/bin/bash
# the Line net's output conv, backward data: 39 us in a filtered micro-benchmark run, 78 us in the full one -- which kernels?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for mode in filtered full; do
  OUT=$R/gpurun_out/t32_$mode; rm -rf "$OUT"; mkdir -p "$OUT"
  if [ $mode = filtered ]; then ARGS="--filter line.end --reps 20"; else ARGS="--reps 20"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT" --output-format csv -- python3 "$R/tools/bench_conv.py" $ARGS > "$OUT/out.txt" 2> "$OUT/err.txt"
  echo "== $mode"; grep "line.end" "$OUT/out.txt"
  python3 - "$OUT" <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'):
    rows = [r for r in csv.DictReader(open(p)) if 't32' in r['Kernel_Name'] or 't542' in r['Kernel_Name']]
    if not rows: continue
    by = {}
    for r in rows:
        key = (r['Kernel_Name'][:60], r['Grid_Size_X'], r['Workgroup_Size_X'], r['LDS_Block_Size'], r['VGPR_Count'], r['Scratch_Size'])
        by.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in by.items():
        v.sort()
        print(k, 'n', len(v), 'min %.1f med %.1f max %.1f' % (v[0], v[len(v) // 2], v[-1]))
PY
done

#!/usr/bin/env python3
"""Copies what tools/refresh_profiles.sh left in gpurun_out/refresh/ into profiles/ (kernel stats + per-step summary,
the two bench lines, micro-benchmarks, the PMC rows of the dominant kernel and the traffic JSON bench.py reads)."""
import collections
import csv
import glob
import json
import os
import shutil

import sys

R, P = 'gpurun_out/refresh', 'profiles'
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r03'


def latest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def part(name):
    for key, tag in (('pair_strip_bwd_kernel', 'main'), ('pair_wave_bwd_h_kernel', 'main'), ('pair_strip_finish', 'finish')):
        if key in name:
            return tag
    return None


def summarise(stats_dir, stem, marker, steps=46):      # 5 warm-up + 20 timed + 1 + 20 probe steps of bench.py --steps 20 --warmup 5
    """kernel stats of one profiled bench command -> profiles/<TAG>_<stem>_kernel_stats.csv + a per-step summary.
    (rocprofv3 also writes a file for the profiler's helper process: take the one that holds `marker`.)"""
    paths = [p for p in glob.glob(f'{R}/{stats_dir}/*/*_kernel_stats.csv') if marker in open(p).read()]
    shutil.copy(max(paths, key=os.path.getmtime), f'{P}/{TAG}_{stem}_kernel_stats.csv')
    rows = list(csv.DictReader(open(f'{P}/{TAG}_{stem}_kernel_stats.csv')))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    out = [f'kernel time {total / 1e6 / steps:.3f} ms/step, {calls / steps:.1f} kernels/step']
    for r in rows[:45]:
        name = r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
        out.append(f"{name[:88]:88s} n/step={int(r['Calls']) / steps:5.1f} avg_us={float(r['AverageNs']) / 1e3:8.1f} "
                   f"ms/step={float(r['TotalDurationNs']) / 1e6 / steps:7.3f} {float(r['Percentage']):5.1f}%")
    open(f'{P}/{TAG}_{stem}_summary.txt', 'w').write('\n'.join(out) + '\n')
    return rows, out


def main():
    rows, out = summarise('stats', 'rocprofv3', 'pair_strip_bwd_kernel')
    summarise('stats_hr', 'rocprofv3_highres_fp16', 'pair_wave_bwd_h_kernel')
    for src, dst in (('bench_n1.json', 'bench_n1.json'), ('bench_under_rocprofv3.json', 'bench_n1_under_rocprofv3.json'),
                     ('bench_hr_n1.json', 'bench_highres_fp16_n1.json'),
                     ('bench_hr_under_rocprofv3.json', 'bench_highres_fp16_under_rocprofv3.json'),
                     ('bench_infer_n1.json', 'bench_infer_b8_n1.json'),
                     ('bench_dp1_rehearsal.json', 'bench_one_rank_rccl_rehearsal.json')):
        shutil.copy(f'{R}/{src}', f'{P}/{TAG}_{dst}')
    for src, dst in (('conv_microbench.txt', f'{TAG}_conv_microbench.txt'), ('membw.txt', f'{TAG}_membw.txt'),
                     ('ubench_coexec.txt', f'{TAG}_ubench_mfma_valu_coexec.txt'),
                     ('ubench_switch.txt', f'{TAG}_ubench_mfma_switch.txt'),
                     ('bench_h16.txt', f'{TAG}_h16_microbench.txt'),
                     ('pair_f16_microbench.txt', f'{TAG}_pair_f16_microbench.txt'), ('bench_nets.txt', f'{TAG}_bench_nets.txt'),
                     ('pmc_pair_summary.txt', f'{TAG}_pmc_pair_kernels.txt')):
        if os.path.exists(f'{R}/{src}'):
            open(f'{P}/{dst}', 'w').write(''.join(line for line in open(f'{R}/{src}') if 'amdgpu.ids' not in line))
    picked, agg = [], {'FETCH_SIZE': collections.defaultdict(list), 'WRITE_SIZE': collections.defaultdict(list)}
    for d in ('pmc_fetch', 'pmc_write'):
        for r in csv.DictReader(open(latest(f'{R}/{d}/*/*counter_collection.csv'))):
            tag = part(r['Kernel_Name'])
            if tag is None:
                continue
            picked.append((r['Counter_Name'], r['Kernel_Name'], r['Dispatch_Id'], float(r['Counter_Value'])))
            agg[r['Counter_Name']][tag].append(float(r['Counter_Value']))
    with open(f'{P}/{TAG}_pmc_dominant_kernel.csv', 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Counter_Name', 'Kernel_Name', 'Dispatch_Id', 'Counter_Value'])
        for r in picked:
            w.writerow([r[0], r[1], r[2], f'{r[3]:.6f}'])
    traffic = json.load(open(f'{P}/dominant_kernel_traffic.json'))
    traffic['kernel'] = 'uocr_conv_pair_bwd = pair_strip_bwd_kernel<4, true, false, 0> + pair_strip_finish (csrc/conv_pair_strip.hip)'
    traffic['algorithmic_bytes_per_launch'] = 3 * 4 * 32 * 256 * 512
    traffic['source'] = f'profiles/{TAG}_pmc_dominant_kernel.csv, bench.py --steps 3 --warmup 1, batch 32, 256x512' 
    fetch = {k: sum(v) / len(v) for k, v in agg['FETCH_SIZE'].items()}
    write = {k: sum(v) / len(v) for k, v in agg['WRITE_SIZE'].items()}
    # gfx950: FETCH_SIZE tallies every 128-byte request at 64 bytes -- exactly half of the bytes read, for 2-, 4- and 16-byte
    # per-lane loads alike (profiles/r03_fetch_size_calibration.txt: 512 MiB read once with each width) -> x 2
    fb, wb = 2 * sum(fetch.values()) * 1024, sum(write.values()) * 1024
    traffic.update(FETCH_SIZE_KB_avg=fetch, WRITE_SIZE_KB_avg=write, fetch_bytes=fb, write_bytes=wb,
                   hbm_bytes_per_launch=fb + wb, algorithmic_bytes=float(traffic['algorithmic_bytes_per_launch']),
                   ratio_to_algorithmic=round((fb + wb) / traffic['algorithmic_bytes_per_launch'], 3),
                   note='Separate --pmc passes (rocprofv3 --pmc X --kernel-trace), averaged over the launches of the run, summed '
                        'over the two kernels of the entry point.  fetch_bytes = 2 x FETCH_SIZE (gfx950 counts a 128-byte request '
                        'as 64 bytes; calibrated for the 4-byte-per-lane loads of this kernel in '
                        'profiles/r03_fetch_size_calibration.txt), write_bytes = WRITE_SIZE.  Algorithmic bytes = x and g = '
                        'dLoss/d(conv_2 output) read + dx written (3 x 16.8 MB; the output Sigmoid derivative is applied by the '
                        'Dice gradient kernel).  What is measured on top: the band-overlap rows (2 of 34 row steps), 0.5 MB of '
                        'per-block partial sums written and read.')
    json.dump(traffic, open(f'{P}/dominant_kernel_traffic.json', 'w'), indent=1)
    b, u = json.load(open(f'{P}/{TAG}_bench_n1.json')), json.load(open(f'{P}/{TAG}_bench_n1_under_rocprofv3.json'))
    print('bench', b['value'], b['ms_per_step'], b['roofline']['solo_launch_us'], b['roofline']['frac'],
          b['cpu_baseline']['value'])
    print('under rocprofv3', u['value'], u['ms_per_step'], u['roofline']['solo_launch_us'], u['roofline']['in_loop_launch_us'])
    print('traffic MB', fb / 1e6, wb / 1e6, (fb + wb) / 1e6)
    for r in rows:
        if part(r['Name']):
            print(r['Name'][:70], float(r['AverageNs']) / 1e3)
    print(out[0])


if __name__ == '__main__':
    main()

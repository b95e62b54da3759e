#!/usr/bin/env python3
"""Per-kernel instruction mix of a hipcc -save-temps .s file (FMA / loads / SGPR-spill readlanes /
registers / scratch): the check that caught the SGPR-spill and scratch problems of conv_fast.hip.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iuniver-ocr_amd/csrc -c X.hip -save-temps=obj -o /tmp/x.o
    python tools/isa_stats.py /tmp/X-hip-amdgcn-amd-amdhsa-gfx950.s [name-filter]"""
import re
import sys
from collections import Counter


def meta(pattern, text):
    m = re.search(pattern, text)
    return m.group(1) if m else '?'


def main():
    text = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    for f in re.split(r'\n\s*\.globl\s+', text)[1:]:
        name = f.split('\n', 1)[0].split()[0]
        if flt not in name or 'cuid' in name:
            continue
        ops = Counter(re.findall(r'^\s+([a-z_0-9]+)', f.split('.section')[0], re.M))
        short = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', name)
        short = re.sub(r'EvPK.*', '', short).replace('ILi', '<').replace('ELi', ',')
        fma = ops['v_fmac_f32_e32'] + ops['v_fma_f32'] + ops['v_fmac_f32_e64'] + 2 * ops['v_pk_fma_f32']
        sload = sum(v for k, v in ops.items() if k.startswith('s_load'))
        gload = sum(v for k, v in ops.items() if k.startswith('global_load'))
        dsr = sum(v for k, v in ops.items() if k.startswith('ds_read'))
        print(f'{short:40s} instr={sum(ops.values()):5d} fma={fma:4d} readlane={ops["v_readlane_b32"]:4d} '
              f'sload={sload:3d} gload={gload:3d} dsread={dsr:3d} bperm={ops["ds_bpermute_b32"]:3d} '
              f'vgpr={meta(r"; NumVgprs: (.d+)".replace(".d", chr(92) + "d"), f)} '
              f'sgpr={meta(r"; NumSgprs: (.d+)".replace(".d", chr(92) + "d"), f)} '
              f'occ={meta(r"; Occupancy: (.d+)".replace(".d", chr(92) + "d"), f)} '
              f'scratch={meta(r"; ScratchSize: (.d+)".replace(".d", chr(92) + "d"), f)}')


if __name__ == '__main__':
    main()

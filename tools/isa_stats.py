#!/usr/bin/env python3
"""Per-kernel instruction statistics of one .hip translation unit compiled for gfx950
(VALU / SALU / memory instruction counts, VGPRs, scratch): the numbers DESIGN.md quotes.

    python tools/isa_stats.py proteus_amd/csrc/dswx_classify_lut.hip [name-filter]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
                        '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(ROOT, 'proteus_amd', 'csrc'),
                        '--cuda-device-only', '-S', '-o', out, src], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    names = re.findall(r'^(_Z\w+):\s+; @', txt, re.M)
    for name in names:
        if flt not in name:
            continue
        body = txt[txt.index(name + ':'):]
        body = body[:body.index('.end_amdhsa_kernel')] if '.end_amdhsa_kernel' in body else body
        code = body[:body.index('s_endpgm')] if 's_endpgm' in body else body
        ins = [l.split()[0] for l in code.split('\n')
               if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
        get = lambda k: (re.search(r'\.amdhsa_' + k + r' (\d+)', body) or [0, '?'])[1]
        print(f'{name}\n   total {len(ins)}  valu {sum(i.startswith("v_") for i in ins)}'
              f'  salu {sum(i.startswith("s_") for i in ins)}'
              f'  f64 {sum(i.startswith("v_") and "f64" in i for i in ins)}'
              f'  ds {sum(i.startswith("ds_") for i in ins)}'
              f'  global {sum(i.startswith(("global_", "buffer_")) for i in ins)}'
              f'  scratch {sum(i.startswith("scratch_") for i in ins)}'
              f'  vgpr {get("next_free_vgpr")} sgpr {get("next_free_sgpr")}'
              f'  scratch_bytes {get("private_segment_fixed_size")}')


if __name__ == '__main__':
    main()

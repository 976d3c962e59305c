#!/usr/bin/env python3
"""Instruction mix of every kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

usage: isa_mix.py file.s [substring-of-kernel-name-to-detail]
"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split('\n')
    detail = sys.argv[2] if len(sys.argv) > 2 else None
    name, counts = None, None
    for line in lines:
        m = re.match(r'^(_Z\S+):', line)
        if m:
            name, counts = m.group(1), collections.Counter()
            continue
        if name is None:
            continue
        s = line.strip()
        if s.startswith('s_endpgm'):
            tot = sum(counts.values())
            grp = lambda p: sum(v for k, v in counts.items() if k.startswith(p))
            print(f'{name[:90]}: total {tot} valu {grp("v_")} (f64 {sum(v for k, v in counts.items() if "f64" in k)}) '
                  f'salu {grp("s_")} lds {grp("ds_")} global {grp("global_")} scratch {grp("scratch_")}')
            if detail and detail in name:
                for k, v in counts.most_common(60):
                    print(f'    {k} {v}')
            name = None
            continue
        if not s or s[0] in '.;/' or s.endswith(':'):
            continue
        counts[s.split()[0]] += 1


if __name__ == '__main__':
    main()

#!/bin/bash
# round 4, GPU session 4: block order of the table-driven kernel -- the blocks of G tiles interleaved in dispatch order
# (KArgs::tile_interleave) against tile-by-tile order, on first-come arenas (three processes = three allocations)
export TMPDIR=/tmp
O=gpurun_out/r04_s4; mkdir -p $O
V="auto tune_lut_interleave=2 tune_lut_interleave=4 tune_lut_interleave=8 tune_lut_interleave=16 tune_lut_interleave=32 tune_lut_interleave=64 tune_lut_interleave=256"
for i in 1 2 3; do
( timeout 900 python3 tools/ab_variants.py --tiles 256 --rounds 3 --reps 4 $V ) > $O/ab_interleave_256_run$i.json 2>&1
python3 - <<PY
import json
d=json.load(open('$O/ab_interleave_256_run$i.json'))
print('run$i', {k.replace('tune_lut_interleave=','G'): v['GBps_median'] for k,v in d.items()})
PY
done
( timeout 900 python3 tools/ab_variants.py --tiles 256 --rounds 3 --reps 4 --masks $V ) > $O/ab_interleave_256_masks.json 2>&1
( timeout 900 python3 tools/ab_variants.py --tiles 256 --rounds 3 --reps 4 --tile-align 1 $V ) > $O/ab_interleave_256_contiguous.json 2>&1
python3 - <<PY
import json
for f in ('masks','contiguous'):
    d=json.load(open('$O/ab_interleave_256_%s.json' % f))
    print(f, {k.replace('tune_lut_interleave=','G'): v['GBps_median'] for k,v in d.items()})
PY

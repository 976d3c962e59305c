#!/bin/bash
# round 3 experiment 2: candidate placement RULES, each in fresh processes (0 probes: one layout, one measurement)
cd "$GRAFT_REPO_ROOT"
P="timeout 200 python3 tools/random_gap_probe.py --tiles 256 --reps 6"
IN=41.52; OUT=25.55
o=gpurun_out/exp2.jsonl; : > $o
for rep in 1 2 3; do
  for order in in,out out,in; do $P --sweep separate:$order >> $o 2>>gpurun_out/exp2.err; done
  # one arena = inputs + gap + outputs, outputs right after the gap
  for gap in 0 8 16 24 32 40 48 64; do
    tot=$(python3 -c "print(round(($IN+$gap+$OUT+0.1)*1.073741824,3))")
    st=$(python3 -c "print($IN+$gap)")
    $P --sweep outpos --total-gb $tot --out-from-gib $st --out-to-gib $st >> $o 2>>gpurun_out/exp2.err
  done
done
# box-to-box: the maps of experiment 1 again (coarser), and the arena size r02 used
$P --reps 4 --sweep outpos --total-gb 260 --out-step-gib 3 > gpurun_out/op2_256_260.json 2>>gpurun_out/exp2.err
$P --reps 4 --sweep outpos --total-gb 250 --out-step-gib 3 > gpurun_out/op2_256_250.json 2>>gpurun_out/exp2.err
$P --reps 4 --sweep outpos --total-gb 150 --out-step-gib 3 > gpurun_out/op2_256_150.json 2>>gpurun_out/exp2.err
echo done

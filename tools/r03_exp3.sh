#!/bin/bash
# round 3 experiment 3: write streams spread by interleaving output and input planes
cd "$GRAFT_REPO_ROOT"
P="timeout 300 python3 tools/random_gap_probe.py --reps 5 --sweep interleave"
o=gpurun_out/exp3.jsonl; : > $o
B="0,4,8,12,16,20,24,28,32,36,40,44,48,52,56,60,64"
$P --tiles 256 --total-gb 150 --points-gib $B >> $o 2>>gpurun_out/exp3.err
$P --tiles 128 --total-gb 150 --span-gib 48 --points-gib $B >> $o 2>>gpurun_out/exp3.err
$P --tiles 64 --total-gb 150 --span-gib 48 --points-gib $B >> $o 2>>gpurun_out/exp3.err
$P --tiles 128 --total-gb 150 --span-gib 32 --points-gib $B >> $o 2>>gpurun_out/exp3.err
# the product configuration: an arena of exactly the batch's size, fresh processes
for rep in 1 2 3 4 5; do
  $P --tiles 256 --total-gb 72.1 --points-gib 0 >> $o 2>>gpurun_out/exp3.err
done
for rep in 1 2 3; do
  $P --tiles 128 --total-gb 88 --span-gib 48 --points-gib 0 >> $o 2>>gpurun_out/exp3.err
  $P --tiles 512 --total-gb 144.1 --points-gib 0 >> $o 2>>gpurun_out/exp3.err
done
echo done

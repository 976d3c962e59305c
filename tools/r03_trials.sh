#!/bin/bash
# round 3: the "10 fresh processes" table: what a FIXED layout gives from one process to the next
cd "$GRAFT_REPO_ROOT"
B="timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-single-tile --realloc-repeats 0 --no-parity"
o=gpurun_out/trials.jsonl; : > $o
for i in 1 2 3 4 5 6 7 8 9 10; do $B --placement-trials 0 >> $o 2>>gpurun_out/trials.err; done
for i in 1 2 3 4 5; do $B --placement-trials 1 >> $o 2>>gpurun_out/trials.err; done
for i in 1 2 3; do $B --placement-trials 6 >> $o 2>>gpurun_out/trials.err; done
echo done

#!/bin/bash
# round 3: placement in FRESH processes.  `bash tools/lab/r03_trials.sh rules` = the ten-process table of
# fixed layouts (profiles/r03_placement_rule_trials.json); `bash tools/lab/r03_trials.sh slide` = the measured placements
cd "$GRAFT_REPO_ROOT"
B="timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-single-tile --realloc-repeats 0 --no-parity"
if [[ "${1:-rules}" == rules ]]; then
  o=gpurun_out/trials.jsonl; : > $o
  for i in 1 2 3 4 5 6 7 8 9 10; do $B --placement-trials 0 >> $o 2>>gpurun_out/trials.err; done
  for i in 1 2 3 4 5; do $B --placement-trials 1 >> $o 2>>gpurun_out/trials.err; done
  for i in 1 2 3; do $B --placement-trials 6 >> $o 2>>gpurun_out/trials.err; done
else
  o=gpurun_out/slide_trials.jsonl; : > $o
  for i in 1 2 3 4 5 6 7 8 9 10; do $B --placement slide ${SLIDE_FLAGS:-} >> $o 2>>gpurun_out/slide_trials.err; done
  for i in 1 2 3; do $B --placement search >> $o 2>>gpurun_out/slide_trials.err; done
  for i in 1 2 3; do $B --placement arena >> $o 2>>gpurun_out/slide_trials.err; done
fi
echo done

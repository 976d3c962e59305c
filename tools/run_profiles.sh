#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_*: run on the GPU box from the repository root
# (gpurun -- 'bash tools/run_profiles.sh [hot] [next]'), then `python tools/summarize_profiles.py NN 256
# [--masks --src gpurun_out/prof_masks]` in the build container digests gpurun_out/ into profiles/.
# Kernel trace and every --pmc set are SEPARATE runs; the program itself follows `--`
# (never env / bash -c / a launcher: the profiler initialises the GPU before the program starts).
export TMPDIR=/tmp
TILES=${TILES:-256}
WHAT="${*:-hot placed next}"
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES"
if [[ "$WHAT" == *hot* ]]; then
for m in "" "--masks"; do
  d=gpurun_out/prof$( [ -n "$m" ] && echo _masks )
  rm -rf "$d"; mkdir -p "$d"
  B="python3 bench.py --tiles $TILES $m --no-cpu-baseline --no-single-tile --no-host-path --realloc-repeats 0 --placement-trials 0"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -- $B --steps 20 --warmup 3 > "$d/bench_trace.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$d/pmc_fetch" -- $B --steps 3 --warmup 1 --no-parity > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$d/pmc_write" -- $B --steps 3 --warmup 1 --no-parity > /dev/null 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d "$d/pmc_sq" -- $B --steps 3 --warmup 1 --no-parity > /dev/null 2>&1
done
fi
if [[ "$WHAT" == *placed* ]]; then
# the DEFAULT bench configuration (--placement slide: dswx_batch_place_slide, ~100 probe launches of the same kernel before warm-up) under the
# kernel trace: the trace then holds the probe launches too; tools/summarize_profiles.py --placed keeps the last
# `steps` full-batch dispatches = the timed region
d=gpurun_out/prof_placed
rm -rf "$d"; mkdir -p "$d"
rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -- python3 bench.py --tiles $TILES --no-cpu-baseline --no-single-tile --no-host-path --realloc-repeats 0 --steps 20 --warmup 3 > "$d/bench_trace.log" 2>&1
fi
if [[ "$WHAT" == *chain* ]]; then
# bench.py --chain (BASELINE configs[4]'s per-pixel chain, device-resident): kernel trace of the three kernels of a step
d=gpurun_out/prof_chain
rm -rf "$d"; mkdir -p "$d"
rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -- python3 bench.py --chain --tiles $TILES --no-cpu-baseline --no-host-path --steps 20 --warmup 3 > "$d/bench_trace.log" 2>&1
fi
if [[ "$WHAT" == *next* ]]; then
# the rows next to the hot path (shadow, cover, land-cover): kernel trace + the same three counter passes
d=gpurun_out/prof_next
rm -rf "$d"; mkdir -p "$d"
N="python3 tools/next_rows_bench.py --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -- $N > "$d/next_rows.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$d/pmc_fetch" -- $N --reps 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$d/pmc_write" -- $N --reps 2 > /dev/null 2>&1
rocprofv3 --pmc $SQ --output-format csv -d "$d/pmc_sq" -- $N --reps 2 > /dev/null 2>&1
fi
ls gpurun_out/prof* 2>/dev/null

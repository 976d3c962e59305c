#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_*: run on the GPU box from the repository root
# (gpurun -- 'bash tools/run_profiles.sh'), then `python tools/summarize_profiles.py NN 256
# [--masks --src gpurun_out/prof_masks]` in the build container digests gpurun_out/ into profiles/.
# Kernel trace and every --pmc set are SEPARATE runs; the program itself follows `--`.
export TMPDIR=/tmp
TILES=${TILES:-256}
for m in "" "--masks"; do
  d=gpurun_out/prof$( [ -n "$m" ] && echo _masks )
  rm -rf "$d"; mkdir -p "$d"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -- \
      python3 bench.py --tiles $TILES $m --steps 20 --warmup 3 --no-cpu-baseline --no-single-tile > "$d/bench_trace.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$d/pmc_fetch" -- \
      python3 bench.py --tiles $TILES $m --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-single-tile > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$d/pmc_write" -- \
      python3 bench.py --tiles $TILES $m --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-single-tile > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
      SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d "$d/pmc_sq" -- \
      python3 bench.py --tiles $TILES $m --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-single-tile > /dev/null 2>&1
done
# the rows next to the hot path (shadow, cover, land-cover): kernel trace only
rm -rf gpurun_out/prof_next; mkdir -p gpurun_out/prof_next
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_next/trace -- \
    python3 tools/next_rows_bench.py --no-cpu > gpurun_out/prof_next/next_rows.log 2>&1
ls gpurun_out/prof gpurun_out/prof_masks gpurun_out/prof_next

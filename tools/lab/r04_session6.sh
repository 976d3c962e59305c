#!/bin/bash
# round 4, GPU session 6: rocprofv3 passes of the final kernel build (hot: first-come arena, trace + FETCH / WRITE / SQ;
# placed: the default bench configuration; chain; next rows)
export TMPDIR=/tmp
bash tools/run_profiles.sh hot placed chain next > gpurun_out/r04_run_profiles.log 2>&1
tail -5 gpurun_out/r04_run_profiles.log

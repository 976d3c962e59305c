export TMPDIR=/tmp
mkdir -p gpurun_out/r05
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=60 --durations-min=1.0 ) > gpurun_out/r05/gpu_suite3.log 2>&1; echo "suite rc=$?"
grep -A70 "slowest" gpurun_out/r05/gpu_suite3.log | head -90

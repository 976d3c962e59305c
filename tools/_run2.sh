export TMPDIR=/tmp
mkdir -p gpurun_out/r05
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 ) > gpurun_out/r05/gpu_suite.log 2>&1; echo "suite rc=$?"
timeout 600 python tests/helpers/trim_live_probe.py 6 > gpurun_out/r05/trim_live_probe.json 2> gpurun_out/r05/trim_live_probe.err; echo "trim probe rc=$?"
timeout 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; echo "bench rc=$?"
tail -n 25 gpurun_out/r05/gpu_suite.log; cat gpurun_out/r05/trim_live_probe.json | head -c 3000

export TMPDIR=/tmp
mkdir -p gpurun_out/r05
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 ) > gpurun_out/r05/gpu_suite5.log 2>&1; echo "suite rc=$?"
tail -n 24 gpurun_out/r05/gpu_suite5.log
python -c "import __graft_entry__ as g; g.smoke()"

export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r05/multirank3.log 2>&1; echo "multirank rc=$?"; tail -n 3 gpurun_out/r05/multirank3.log

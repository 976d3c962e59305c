export TMPDIR=/tmp
mkdir -p gpurun_out/r05
bash tools/run_profiles.sh next chain > gpurun_out/r05/run_profiles_next.log 2>&1; echo "profiles rc=$?"
for i in 1 2 3; do timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r05/bench_repeat_$i.json 2> gpurun_out/r05/bench_repeat_$i.err; echo "bench $i rc=$?"; done
timeout 900 python bench.py --also-strong --no-cpu-baseline --no-host-path --realloc-repeats 0 --no-single-tile > gpurun_out/r05/bench_weak_then_strong.json 2> gpurun_out/r05/bench_weak_then_strong.err; echo "weak+strong rc=$?"
DSWX_FORCE_DIST=1 timeout 900 python bench.py --no-cpu-baseline --realloc-repeats 0 --no-single-tile > gpurun_out/r05/bench_rccl_world1.json 2> gpurun_out/r05/bench_rccl_world1.err; echo "rccl world1 rc=$?"
DSWX_FORCE_DIST=1 timeout 300 python bench.py --preflight > gpurun_out/r05/preflight_world1.json 2> gpurun_out/r05/preflight_world1.err; echo "preflight rc=$?"; cat gpurun_out/r05/preflight_world1.json

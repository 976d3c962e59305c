export TMPDIR=/tmp
mkdir -p gpurun_out/r05
bash tools/run_profiles.sh hot placed > gpurun_out/r05/run_profiles2.log 2>&1; echo "profiles rc=$?"
d=gpurun_out/prof_scaled; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 bench.py --scaled --tiles 256 --no-cpu-baseline --no-single-tile --no-host-path --realloc-repeats 0 --placement-trials 0 --steps 20 --warmup 3 > $d/bench_trace.log 2>&1; echo "scaled trace rc=$?"
timeout 900 python bench.py > gpurun_out/r05/bench_default3.json 2> gpurun_out/r05/bench_default3.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r05/bench_default3.json

#!/bin/bash
# round 4, GPU session 10: final soak on the last build + two more kernel traces of first-come arenas
export TMPDIR=/tmp
O=gpurun_out/r04_s10; mkdir -p $O
( timeout 1500 python3 tests/fuzz_parity.py --device-batch --iters 8000 --seed 101 ) > $O/fuzz_device_batch.json 2>&1 &
( timeout 1500 python3 tests/fuzz_parity.py --iters 5000 --seed 102 ) > $O/fuzz_host.json 2>&1 &
( timeout 1500 python3 tests/fuzz_parity.py --pinned --iters 4000 --seed 103 ) > $O/fuzz_pinned.json 2>&1 &
wait
tail -c 500 $O/fuzz_device_batch.json $O/fuzz_host.json $O/fuzz_pinned.json
for i in 2 3; do
  d=$O/trace_first_come_$i; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 bench.py --tiles 256 --no-cpu-baseline --no-single-tile --no-host-path --realloc-repeats 0 --placement-trials 0 --steps 20 --warmup 3 > $d/bench_trace.log 2>&1
  grep -h "dswx_classify_lut" $d/trace/*/*kernel_stats.csv | head -2
done

#!/bin/bash
# round 4, GPU session 3: VMM accounting probe, reproducer without churn, rocprofv3 passes of the new kernel build,
# 256-tile A/B on contiguous tiles, fuzz soak, default bench line through a forced RCCL world of one
export TMPDIR=/tmp
O=gpurun_out/r04_s3; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 tools/lab/vmm_meminfo.hip -o /tmp/vmm_meminfo && timeout 600 /tmp/vmm_meminfo 64 > $O/vmm_meminfo.json 2>&1
cat $O/vmm_meminfo.json
hipcc --offload-arch=gfx950 -O2 tools/vmm_reuse_repro.hip -o /tmp/vmm_repro > $O/repro_build.log 2>&1
for m in 6 1 6; do timeout 600 /tmp/vmm_repro $m 200 2 6 >> $O/vmm_repro_churn.jsonl 2>> $O/vmm_repro_churn.err; done
cat $O/vmm_repro_churn.jsonl
( timeout 900 python3 tools/ab_variants.py --tiles 256 --rounds 3 --tile-align 1 auto LIB=proteus_amd/_lib/ab/libdswx_prev.so ) > $O/ab_contiguous_256.json 2>&1
cat $O/ab_contiguous_256.json
( timeout 900 python3 tests/fuzz_parity.py --device-batch --iters 3000 --seed 41 ) > $O/fuzz_device_batch.json 2>&1
( timeout 900 python3 tests/fuzz_parity.py --iters 2000 --seed 42 ) > $O/fuzz_host.json 2>&1
( timeout 900 python3 tests/fuzz_parity.py --pinned --iters 1500 --seed 43 ) > $O/fuzz_pinned.json 2>&1
tail -c 600 $O/fuzz_device_batch.json $O/fuzz_host.json $O/fuzz_pinned.json
( DSWX_FORCE_DIST=1 timeout 1200 python3 bench.py --no-cpu-baseline ) > $O/bench_default_rccl_world1.json 2> $O/bench_default_rccl_world1.err
cut -c1-1500 $O/bench_default_rccl_world1.json; tail -3 $O/bench_default_rccl_world1.err
bash tools/run_profiles.sh hot placed > $O/run_profiles.log 2>&1
tail -5 $O/run_profiles.log

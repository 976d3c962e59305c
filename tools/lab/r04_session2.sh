#!/bin/bash
# round 4, GPU session 2: the whole GPU suite on this round's code, the reproducer's TLB modes, the default bench line
export TMPDIR=/tmp
O=gpurun_out/r04_s2; mkdir -p $O
( timeout 2400 python3 -m pytest tests -q -m gpu --maxfail=15 ) > $O/pytest_gpu.log 2>&1
tail -40 $O/pytest_gpu.log
hipcc --offload-arch=gfx950 -O2 tools/vmm_reuse_repro.hip -o /tmp/vmm_repro > $O/repro_build.log 2>&1
for m in 1 4 5 1 4; do timeout 600 /tmp/vmm_repro $m 200 2 6 >> $O/vmm_repro_tlb.jsonl 2>> $O/vmm_repro_tlb.err; done
cat $O/vmm_repro_tlb.jsonl
( timeout 1200 python3 bench.py ) > $O/bench_default.json 2> $O/bench_default.err
cat $O/bench_default.json | cut -c1-6000

#!/bin/bash
# round 4, GPU session 1: (a) changed GPU tests, (b) the VMM address-reuse question (plain-HIP reproducer + the library
# built with each policy), (c) the per-tile lead-in of the table-driven kernel: contiguous vs padded batches, A/B against
# the round-3 build in one process
export TMPDIR=/tmp
O=gpurun_out/r04_s1; mkdir -p $O
( timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_batch_and_synth or more_tiles_than or cover_mode or sliding or kernel_variants or output_planes or unaligned" ) > $O/pytest_subset.log 2>&1
tail -3 $O/pytest_subset.log
hipcc --offload-arch=gfx950 -O2 tools/vmm_reuse_repro.hip -o /tmp/vmm_repro > $O/repro_build.log 2>&1
for m in 0 1 2 3; do timeout 300 /tmp/vmm_repro $m 200 2 6 >> $O/vmm_repro.jsonl 2>> $O/vmm_repro.err; done
for m in 0 1 3; do timeout 300 /tmp/vmm_repro $m 40 64 8 >> $O/vmm_repro.jsonl 2>> $O/vmm_repro.err; done
cat $O/vmm_repro.jsonl
( timeout 1500 python3 tests/vmm_policy_trial.py --cases 80 ) > $O/vmm_policy_trial.json 2> $O/vmm_policy_trial.err
cat $O/vmm_policy_trial.json | head -80
( timeout 600 python3 tools/ab_variants.py --tiles 64 --rounds 5 --tile-align 1 auto LIB=proteus_amd/_lib/ab/libdswx_prev.so ) > $O/ab_contiguous_64.json 2>&1
( timeout 600 python3 tools/ab_variants.py --tiles 64 --rounds 5 --tile-align 256 auto LIB=proteus_amd/_lib/ab/libdswx_prev.so ) > $O/ab_padded_64.json 2>&1
( timeout 600 python3 tools/ab_variants.py --tiles 64 --rounds 5 --tile-align 1 --masks auto LIB=proteus_amd/_lib/ab/libdswx_prev.so ) > $O/ab_contiguous_64_masks.json 2>&1
( timeout 600 python3 tools/ab_variants.py --tiles 32 --rounds 5 --tile-align 1 --masks --mode cover auto LIB=proteus_amd/_lib/ab/libdswx_prev.so ) > $O/ab_contiguous_32_cover.json 2>&1
( timeout 600 python3 tools/ab_variants.py --tiles 32 --rounds 5 --tile-align 256 --masks --mode cover auto LIB=proteus_amd/_lib/ab/libdswx_prev.so ) > $O/ab_padded_32_cover.json 2>&1
cat $O/ab_*.json

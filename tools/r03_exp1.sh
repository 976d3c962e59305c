#!/bin/bash
# round 3 experiment 1: where do the write streams run fast (fresh processes), + variant re-race at 256 tiles
cd "$GRAFT_REPO_ROOT"
T="timeout 300 python3 tools/random_gap_probe.py --sweep outpos"
$T --tiles 256 --total-gb 260 --out-step-gib 1.5 > gpurun_out/op_256_a.json 2> gpurun_out/op_256_a.err
$T --tiles 256 --total-gb 260 --out-step-gib 1.5 > gpurun_out/op_256_b.json 2> gpurun_out/op_256_b.err
$T --tiles 256 --total-gb 260 --in-base-gib 170 --out-from-gib 0 --out-to-gib 144 --out-step-gib 3 > gpurun_out/op_256_inhi.json 2> gpurun_out/op_256_inhi.err
$T --tiles 128 --total-gb 260 --out-step-gib 1.5 > gpurun_out/op_128.json 2> gpurun_out/op_128.err
$T --tiles 512 --total-gb 260 --out-step-gib 3 > gpurun_out/op_512.json 2> gpurun_out/op_512.err
$T --tiles 256 --total-gb 82 --out-step-gib 1.5 > gpurun_out/op_256_small.json 2> gpurun_out/op_256_small.err
timeout 600 python3 tools/ab_variants.py --tiles 256 --rounds 5 fused_variant=3 fused_variant=0 fused_variant=1 fused_variant=2 fused_variant=4 fused_variant=5 > gpurun_out/r03_ab256.json 2> gpurun_out/r03_ab256.err
echo done

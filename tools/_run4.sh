export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "folded or randomized or variants or cover or golden_tiles or pool_trim or sliding" > gpurun_out/r05/fold_tests2.log 2>&1; echo "fold tests rc=$?"
for t in 1 2 4 8 16; do
  echo "== tiles $t"; python tools/ab_variants.py --tiles $t --rounds 7 --reps 50 tune_fold=1 tune_fold=0 2>&1 | grep -v "^ *\"kernel\|GBps_m[ai]"
done > gpurun_out/r05/fold_ab2.txt 2>&1
for t in 1 4 16; do
  echo "== masks tiles $t"; python tools/ab_variants.py --masks --tiles $t --rounds 7 --reps 50 tune_fold=1 tune_fold=0 2>&1 | grep -v "^ *\"kernel\|GBps_m[ai]"
done >> gpurun_out/r05/fold_ab2.txt 2>&1
echo "== cover tiles 1"; python tools/ab_variants.py --masks --mode cover --tiles 1 --rounds 5 --reps 30 tune_fold=1 tune_fold=0 2>&1 | grep -v "^ *\"kernel\|GBps_m[ai]" >> gpurun_out/r05/fold_ab2.txt 2>&1
rm -rf gpurun_out/r05/single_trace; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05/single_trace -- python3 tools/ab_variants.py --tiles 1 --rounds 3 --reps 50 tune_fold=1 tune_fold=0 > /dev/null 2>&1
tail -n 5 gpurun_out/r05/fold_tests2.log; cat gpurun_out/r05/fold_ab2.txt; cat gpurun_out/r05/single_trace/*/*kernel_stats.csv

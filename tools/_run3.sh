export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "folded or randomized or variants or cover or golden_tiles" > gpurun_out/r05/fold_tests.log 2>&1; echo "fold tests rc=$?"
for t in 1 2 4 8 16; do
  echo "== tiles $t"; python tools/ab_variants.py --tiles $t --rounds 7 --reps 50 tune_fold=1 tune_fold=0 2>&1 | grep -v "^ *\"kernel\|GBps_m[ai]"
done > gpurun_out/r05/fold_ab.txt 2>&1
for t in 1 4; do
  echo "== masks tiles $t"; python tools/ab_variants.py --masks --tiles $t --rounds 7 --reps 50 tune_fold=1 tune_fold=0 2>&1 | grep -v "^ *\"kernel\|GBps_m[ai]"
done >> gpurun_out/r05/fold_ab.txt 2>&1
tail -n 5 gpurun_out/r05/fold_tests.log; cat gpurun_out/r05/fold_ab.txt

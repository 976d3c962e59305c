export TMPDIR=/tmp
mkdir -p gpurun_out/r05
for t in 1 2 4 16; do
  echo "== tiles $t"; python tools/ab_variants.py --tiles $t --rounds 9 --reps 40 tune_small_chunks=1 tune_small_chunks=2 tune_small_chunks=4 2>&1 | grep -v "GBps_m[ai]"
done > gpurun_out/r05/chunks_ab.txt 2>&1
for t in 1 4; do
  echo "== masks tiles $t"; python tools/ab_variants.py --masks --tiles $t --rounds 9 --reps 40 tune_small_chunks=1 tune_small_chunks=2 tune_small_chunks=4 2>&1 | grep -v "GBps_m[ai]"
done >> gpurun_out/r05/chunks_ab.txt 2>&1
cat gpurun_out/r05/chunks_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "folded or golden_tiles or randomized" 2>&1 | tail -3

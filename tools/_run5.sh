export TMPDIR=/tmp
mkdir -p gpurun_out/r05
for t in 1 1 4 64; do
  echo "== tiles $t"; python tools/ab_variants.py --tiles $t --rounds 9 --reps 40 auto auto,LIB=proteus_amd/_lib/ab/libdswx_prev.so 2>&1 | grep -v "^ *\"kernel"
done > gpurun_out/r05/hoist_ab.txt 2>&1
echo "== tiles 256"; python tools/ab_variants.py --tiles 256 --rounds 7 --reps 5 auto auto,LIB=proteus_amd/_lib/ab/libdswx_prev.so 2>&1 | grep -v "^ *\"kernel" >> gpurun_out/r05/hoist_ab.txt 2>&1
echo "== masks tiles 256"; python tools/ab_variants.py --masks --tiles 256 --rounds 7 --reps 5 auto auto,LIB=proteus_amd/_lib/ab/libdswx_prev.so 2>&1 | grep -v "^ *\"kernel" >> gpurun_out/r05/hoist_ab.txt 2>&1
echo "== masks tiles 1"; python tools/ab_variants.py --masks --tiles 1 --rounds 9 --reps 40 auto auto,LIB=proteus_amd/_lib/ab/libdswx_prev.so 2>&1 | grep -v "^ *\"kernel" >> gpurun_out/r05/hoist_ab.txt 2>&1
echo "== cover tiles 32"; python tools/ab_variants.py --masks --mode cover --tiles 32 --rounds 7 --reps 5 auto auto,LIB=proteus_amd/_lib/ab/libdswx_prev.so 2>&1 | grep -v "^ *\"kernel" >> gpurun_out/r05/hoist_ab.txt 2>&1
cat gpurun_out/r05/hoist_ab.txt

ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_block
mkdir -p $OUT
for args in "--lead 5" "--lead 0" "--lead 2" "--lead 5 --shards 2" "--lead 5 --steps 100"; do
  python $ROOT/tools/r05/block_anatomy.py $args >> $OUT/anatomy.txt 2>> $OUT/err.log
done
cat $OUT/anatomy.txt; grep -v amdgpu.ids $OUT/err.log | tail -5

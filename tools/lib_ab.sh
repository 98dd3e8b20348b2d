# usage: bash tools/lib_ab.sh <grep pattern> <bench_kernels args...> ; compares the in-tree library with every build_ab/*.so on one box
cd $GRAFT_REPO_ROOT
pat="$1"; shift
python tools/bench_kernels.py "$@" 2>&1 | grep "$pat" | cut -c1-62 > gpurun_out/ab_base.txt
hdr="base"; files="gpurun_out/ab_base.txt"
for so in build_ab/liboctmae_*.so; do
  tag=$(basename $so .so | sed s/liboctmae_//)
  OCTMAE_LIB=$GRAFT_REPO_ROOT/$so python tools/bench_kernels.py "$@" 2>&1 | grep "$pat" | cut -c48-57 > gpurun_out/ab_$tag.txt
  hdr="$hdr | $tag"; files="$files gpurun_out/ab_$tag.txt"
done
echo "$hdr"; paste -d"|" $files

# usage (GPU box): bash tools/lib_sweep.sh <script.py> lib1 lib2 ...   -- runs the script once per variant library (OCTMAE_LIB)
S=$1; shift
for l in "$@"; do
  echo "== $l"
  OCTMAE_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$l.so python3 $GRAFT_REPO_ROOT/$S 2>&1 | grep -v amdgpu.ids
done

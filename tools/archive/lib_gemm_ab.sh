# usage (GPU box): bash tools/lib_gemm_ab.sh <rounds> lib1 lib2 ...  -- GEMM micro-benchmarks per variant library, interleaved rounds;
# one line per shape with the microseconds of every (round, library)
R=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/gab
for r in $(seq 1 $R); do for l in "$@"; do
  OCTMAE_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$l.so python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py --batch ${B:-128} --only gemm --iters 10 2>&1 | grep "^gemm" > $GRAFT_REPO_ROOT/gpurun_out/gab/$l.$r.txt
done; done
python3 - "$R" "$@" <<'PY'
import sys, os, re
R = int(sys.argv[1]); libs = sys.argv[2:]
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/gab/"
data = {}
for l in libs:
    for r in range(1, R + 1):
        for line in open(root + f"{l}.{r}.txt"):
            m = re.match(r"(gemm .*?)\s+([\d.]+) us", line)
            name = re.sub(r"\[.*?\]x\[.*?\]\s*", "", m.group(1))
            data.setdefault(name, {}).setdefault(l, []).append(float(m.group(2)))
print(f"{'shape':34s} " + " ".join(f"{l:>10s}" for l in libs) + "   ratio vs first")
for name, d in data.items():
    med = [sorted(d[l])[len(d[l]) // 2] for l in libs]
    print(f"{name:34s} " + " ".join(f"{v:10.1f}" for v in med) + "   " + " ".join(f"{v / med[0]:.3f}" for v in med[1:]))
PY

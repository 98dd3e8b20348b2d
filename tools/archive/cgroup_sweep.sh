for r in 1 2; do for v in 0 1 2 4 8; do OCTMAE_CGROUP=$v python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py --batch 128 --only gemm --iters 10 2>&1 | grep "^gemm" | grep -v wgrad > $GRAFT_REPO_ROOT/gpurun_out/cg.$v.$r.txt; done; done
python3 - <<'PY'
import re,os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/"
d={}
vs=(0,1,2,4,8)
for v in vs:
    for r in (1,2):
        for line in open(root+f"cg.{v}.{r}.txt"):
            m=re.match(r"(gemm .*?)\s+([\d.]+) us",line); name=re.sub(r"\[.*?\]x\[.*?\]\s*","",m.group(1))
            d.setdefault(name,{}).setdefault(v,[]).append(float(m.group(2)))
print(" "*34+" ".join(f"cg{v:>7d}" for v in vs))
for n,x in d.items():
    print(f"{n:34s} "+" ".join(f"{min(x[v]):9.1f}" for v in vs))
PY

# usage (GPU box): bash tools/step_ab.sh "<switch>" "<switch>" ...   -- the whole training step on ONE box, baseline first and last, one
# bench.py run per switch set (a switch set = space-free comma list of bench.py --set arguments, e.g. attn_bwd_hd32_form=0)
cd $GRAFT_REPO_ROOT
run() {
  args=""; for kv in $(echo "$1" | tr ',' ' '); do args="$args --set $kv"; done
  python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-per-rank-proxy $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
print('%-44s %7.2f volumes/s %8.1f ms  ' % ('$1' or 'baseline', d['value'], d['ms_per_step']) + ' '.join(n.replace('attn_','a_').replace('gemm_','g_').replace('fused_','')+':'+str(int(v['avg_us'])) for n,v in k.items()))
"
}
run ""
for s in "$@"; do run "$s"; done
run ""

#!/bin/bash
# Everything the round's evidence under profiles/ comes from, in one GPU session (run from the repo root on the GPU box):
#     bash tools/round_evidence.sh r06 <git HEAD>
# Writes gpurun_out/<round>_evidence/*; copy what is judged into profiles/ (names: profiles/README.md).
ROUND=${1:-r06}; HEAD_ID=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${ROUND}_evidence
mkdir -p $O
cd $R
# 1. the GPU suite (parity ledgers land in gpurun_out/)
timeout 1800 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1
cp gpurun_out/parity_measured.json $O/${ROUND}_parity_measured.json 2>/dev/null
python - "$O" "$ROUND" <<'PY'
import json, sys
o, r = sys.argv[1], sys.argv[2]
try:
    json.dump({"bf16": json.load(open("gpurun_out/parity_ledger_bf16.json")), "f16": json.load(open("gpurun_out/parity_ledger_f16.json"))},
              open(f"{o}/{r}_f16_parity.json", "w"), indent=1)
except Exception as e:
    print("ledgers:", e)
PY
# 2. the driver's command
( time python bench.py > $O/${ROUND}_bench_final.json 2> $O/bench_final.err ) 2>&1 | grep real
# 3. the same step under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_main -- python3 $R/bench.py --no-cpu-baseline --no-per-rank-proxy \
    --no-parity-compliant --no-small-batch --steps 1 --warmup 1 > $O/${ROUND}_bench_under_rocprof.json 2> $O/prof_main.err
find $O/prof_main -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${ROUND}_bench_final_kernel_stats.csv; rm -rf $O/prof_main
# 4. one volume per step, and the per-rank shapes of 8 GPUs against the 128-volume micro-batch
for MB in 1 32 128; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mb$MB -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing \
      --no-per-rank-proxy --steps 4 --warmup 2 --global-batch $MB --micro-batch $MB > $O/bench_mb$MB.json 2> /dev/null
  find $O/prof_mb$MB -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${ROUND}_mb${MB}_kernel_stats.csv; rm -rf $O/prof_mb$MB
done
cd $R
python tools/stats_per_volume.py $O/${ROUND}_mb32_kernel_stats.csv 32 $O/${ROUND}_mb128_kernel_stats.csv 128 > $O/${ROUND}_mb32_vs_mb128_per_volume.txt 2>&1
# 5. PMC traffic of the final kernels
bash tools/collect_pmc_traffic.sh ${ROUND}_mb128 128 $HEAD_ID > $O/pmc.log 2>&1; cp gpurun_out/${ROUND}_mb128_pmc_hbm_traffic.json $O/${ROUND}_pmc_hbm_traffic_mb128.json 2>/dev/null
rm -rf gpurun_out/pmc_traffic
# 6. the cost-model fit and the other steps
timeout 900 python tools/gemm_small_fit.py 1 2 4 8 32 > $O/${ROUND}_gemm_small_fit.txt 2>&1
timeout 300 python tools/bench_finetune.py > $O/${ROUND}_finetune_step.json 2>/dev/null
timeout 300 python tools/bench_finetune.py --batch 4 > $O/${ROUND}_finetune_step_batch4.json 2>/dev/null
timeout 300 python tools/bench_coem.py 8 4 > $O/${ROUND}_coem_step.json 2>/dev/null
timeout 300 python tools/bench_joint.py > $O/${ROUND}_joint_step.json 2>/dev/null
timeout 300 python tools/graph_step_probe.py 1 2 4 8 > $O/${ROUND}_graph_probe.txt 2>&1
ls -la $O

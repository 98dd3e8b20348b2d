# usage (on the GPU box): bash tools/pmc_kernel.sh <script.py> <kernel-name substring>
# SQ counter passes (separate --pmc runs, kernel-trace only) + a --stats pass; prints per-kernel averages.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SCRIPT=$1; export KFILT=$2
rm -rf $R/gpurun_out/pmck_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmck_stats -- python3 $R/$SCRIPT > /dev/null 2>&1
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | cut -c1-14 | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmck_$tag -- python3 $R/$SCRIPT > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]; filt = os.environ["KFILT"]
for f in glob.glob(R + "/gpurun_out/pmck_stats/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print("stats:", row["Name"][:70], "calls", row["Calls"], "avg_ns", row["AverageNs"])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(R + "/gpurun_out/pmck_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        if filt not in k: continue
        k = k + "|grid" + row.get("Grid_Size", "") + "|vgpr" + row.get("VGPR_Count", "") + "|lds" + row.get("LDS_Block_Size", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:32s} {v / cnt[(k, c)]:.5g}")
PY

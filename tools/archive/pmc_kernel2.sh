# usage (GPU box): OCTMAE_LIB=... bash tools/pmc_kernel2.sh <script.py> <kernel substring>  -- instruction mix / LDS latency / fetch counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SCRIPT=$1; export KFILT=$2
rm -rf $R/gpurun_out/pmck2_*
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_MFMA" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_IOPS SQ_INSTS_VSKIPPED SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmck2_$i -- python3 $R/$SCRIPT > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]; filt = os.environ["KFILT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(R + "/gpurun_out/pmck2_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:48]
        if filt not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:32s} {v / cnt[(k, c)]:.5g}")
PY

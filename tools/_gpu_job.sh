cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/gvl_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gvl_stats -- python3 $R/tools/gemm_vs_lib_probe.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/gvl_pmc -- python3 $R/tools/gemm_vs_lib_probe.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
dur = collections.defaultdict(list)
for f in glob.glob(R + "/gpurun_out/gvl_pmc/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"][:60]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/gvl_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:60] + "|grid" + row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    if "gemm" not in k and "Cijk" not in k: continue
    kk = k.split("|")[0]
    d = sorted(dur[kk]); 
    print(k, "launches", len(agg[k]["GRBM_GUI_ACTIVE"]))
    for c, v in sorted(agg[k].items()):
        print(f"   {c:28s} " + " ".join(f"{x:.4g}" for x in v[:8]))
    print("   durations_us", [round(x / 1e3, 1) for x in dur[kk][:8]])
PY

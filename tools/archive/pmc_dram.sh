# usage (GPU box): bash tools/pmc_dram.sh <script.py> <kernel-name substring> <tag>
# Memory-side counter passes for one kernel of a script (separate --pmc runs, kernel-trace only; MI355X_MICROARCH.md, HBM section):
# FETCH_SIZE / WRITE_SIZE (all fabric requests of the L2, Infinity-Cache hits included) next to the counters that name DRAM as
# the destination, the request-size split, L2 hit / miss, and a --stats pass for the time.  tools/mall_calib.py run through the
# same script tells whether the *_DRAM counters exclude Infinity-Cache hits (they do not: see profiles/r04_dram_counters.txt).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SCRIPT=$1; export KFILT=$2; TAG=${3:-dram}
rm -rf $R/gpurun_out/pmcd_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcd_stats -- python3 $R/$SCRIPT > /dev/null 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITEBACK_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcd_$i -- python3 $R/$SCRIPT > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]; filt = os.environ["KFILT"]
for f in glob.glob(R + "/gpurun_out/pmcd_stats/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if filt in row["Name"]:
            print("stats:", row["Name"][:70], "calls", row["Calls"], "avg_ns", row["AverageNs"])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(R + "/gpurun_out/pmcd_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        if filt not in k: continue
        k = k + "|grid" + row.get("Grid_Size", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted(agg):
    print(k)
    a = {c: v / cnt[(k, c)] for c, v in agg[k].items()}
    for c, v in sorted(a.items()):
        print(f"   {c:40s} {v:.6g}")
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        print(f"   -> fabric bytes (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = {(2 * a['FETCH_SIZE'] + a['WRITE_SIZE']) * 1024 / 1e9:.3f} GB")
    if "TCC_EA0_RDREQ_sum" in a:
        r32 = a.get("TCC_EA0_RDREQ_32B_sum", 0.0); r128 = a.get("TCC_EA0_RDREQ_128B_sum", 0.0); rt = a["TCC_EA0_RDREQ_sum"]
        print(f"   -> read requests {rt:.4g}: 32 B {r32:.4g}, 128 B {r128:.4g}, 64 B {rt - r32 - r128:.4g}  = {(32 * r32 + 128 * r128 + 64 * (rt - r32 - r128)) / 1e9:.3f} GB; "
              f"DRAM-destined {a.get('TCC_EA0_RDREQ_DRAM_sum', 0.0) / max(rt, 1):.3f} of them")
    if "TCC_EA0_WRREQ_sum" in a:
        wt = a["TCC_EA0_WRREQ_sum"]; w64 = a.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        print(f"   -> write requests {wt:.4g}: 64 B {w64:.4g}, 32 B {wt - w64:.4g} = {(64 * w64 + 32 * (wt - w64)) / 1e9:.3f} GB; "
              f"DRAM-destined {a.get('TCC_EA0_WRREQ_DRAM_sum', 0.0) / max(wt, 1):.3f} of them")
PY

#!/bin/bash
# HBM-side traffic per kernel launch from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over one micro-batch of
# the benchmark step; writes gpurun_out/<round>_pmc_hbm_traffic.json (copy it to profiles/).  Run on the GPU box from the repo root:
#     bash tools/collect_pmc_traffic.sh r02_mb128 128 <git HEAD>
# Per MI355X_MICROARCH.md (HBM / rocprofv3 section): counters in their own passes, --kernel-trace only; FETCH_SIZE and
# WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports by 2x, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
set -e
ROUND=${1:-r01}; MB=${2:-64}; HEAD_ID=${3:-unknown}      # 3rd argument: the git HEAD the profiled tree was built from
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing \
      --steps 1 --warmup 0 --global-batch $MB --micro-batch $MB > $OUT/$c.log 2>&1
done
KH=$(cd $R && python3 -c "import bench; print(bench.kernels_hash())")
python3 - "$OUT" "$R/gpurun_out/${ROUND}_pmc_hbm_traffic.json" "$MB" "$HEAD_ID" "$KH" <<'PY'
import csv, glob, json, re, sys, collections
out, dst, mb, head, kh = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
acc = {c: collections.defaultdict(lambda: [0, 0.0]) for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in acc:
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != c:
                continue
            k = re.sub(r"^void ", "", row["Kernel_Name"]); k = re.sub(r"^octmae::", "", k); k = k.split("(")[0]
            a = acc[c][k]; a[0] += 1; a[1] += float(row["Counter_Value"])
res = {"_meta": {"micro_batch": mb, "head": head, "kernels_hash": kh, "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch"}}
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    nf, f = acc["FETCH_SIZE"].get(k, [0, 0.0]); nw, w = acc["WRITE_SIZE"].get(k, [0, 0.0])
    if not nf or not nw:
        continue
    res[k] = {"launches": nf, "fetch_kb_raw_per_launch": f / nf, "write_kb_per_launch": w / nw,
              "hbm_bytes_per_launch_corrected": (2 * f / nf + w / nw) * 1024}
json.dump(res, open(dst, "w"), indent=1)
print("wrote", dst, len(res) - 1, "kernels")
PY

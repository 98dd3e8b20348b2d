"""Two rocprofv3 kernel_stats.csv files of bench.py runs at different micro-batches, kernel by kernel per VOLUME: where the per-volume
time goes up when a rank has fewer volumes.   python tools/stats_per_volume.py small.csv MB_SMALL big.csv MB_BIG
(MB = the micro-batch of the run; the number of micro-batches is read off the call count of the decoder attention backward, 8 per pass)."""
import csv
import re
import sys


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return out


def short(name):
    m = re.match(r"(?:void )?(?:octmae::)?([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def volumes(stats, mb):
    calls = next(v[0] for k, v in stats.items() if "attn_bwd_fused1w_kernel" in k)
    return calls / 8 * mb


a, b = load(sys.argv[1]), load(sys.argv[3])
va, vb = volumes(a, float(sys.argv[2])), volumes(b, float(sys.argv[4]))
ta, tb = sum(v[1] for v in a.values()) / va, sum(v[1] for v in b.values()) / vb
print(f"kernel time per volume: {ta / 1e6:.3f} ms at the small micro-batch, {tb / 1e6:.3f} ms at the big one: {ta / tb - 1:+.1%}")
print(f"{'kernel':58s} {'us/vol small':>12s} {'us/vol big':>11s} {'ratio':>6s} {'of the small step':>17s} {'lost, % of step':>15s}")
rows = []
for k in a:
    if k in b:
        pa, pb = a[k][1] / va, b[k][1] / vb
        rows.append((pa - pb, k, pa, pb))
for d, k, pa, pb in sorted(rows, reverse=True)[:14] + sorted(rows)[:4]:
    print(f"{short(k):58s} {pa / 1e3:12.1f} {pb / 1e3:11.1f} {pa / pb:6.3f} {pa / ta:17.1%} {d / ta:15.2%}")

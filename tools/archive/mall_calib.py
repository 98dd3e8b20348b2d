"""Calibration target for tools/pmc_dram.sh: the same streaming read (a float32 sum) over a buffer that stays in the 256 MiB
Infinity Cache between launches (64 MiB, re-read 12 times) and over one that cannot (4 GiB, read 3 times).  If a counter reports
the same bytes per byte read for both, it counts Infinity-Cache hits; the wall-clock rates printed here tell the two regimes
apart (a resident buffer reads faster than HBM can deliver).  Grid sizes differ, so the two show up as separate rows."""
import sys, time, torch
small = torch.randn(16 * 1024 * 1024, device="cuda")           # 64 MiB
big = torch.randn(1024 * 1024 * 1024, device="cuda")           # 4 GiB
for x, reps, name in ((small, 12, "64 MiB"), (big, 3, "4 GiB")):
    x.sum(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        x.sum()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {x.numel() * 4 / dt / 1e12:.2f} TB/s per pass ({dt * 1e6:.1f} us)", flush=True)

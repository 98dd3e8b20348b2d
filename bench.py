#!/usr/bin/env python3
"""Headline benchmark: 3-D MAE pre-training volumes/s (ViT-L, 60x256x256 volumes, mask 0.75, bf16) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N = 1: plain python
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                       # N > 1: one rank per GPU over RCCL

A "step" is one optimizer step over the fixed GLOBAL batch (default 256 volumes, BASELINE config 3): every rank runs
global_batch / N volumes as micro-batches with gradient accumulation (default: 128 volumes on a single GPU -- 246 of its
268 GiB, with a fall-back to 64, agreed across ranks, should a collective-free trial step run out of memory), the
flat-arena all-reduce overlaps the last micro-batch's backward, then grad-norm + fused AdamW.  Strong scaling: total work per step is fixed.
Inputs are synthetic fp32 volumes already resident in HBM; weights are the reference's random init; masking noise comes
from the device RNG inside the timed region.  Rank 0 prints ONE JSON line.

The line also carries
  roofline     -- the dominant kernel of the timed region (by summed device time, measured with HIP events recorded on the
                  launch stream around every 5th call of each MFMA entry point): algorithmic FLOP / measured time vs the dense
                  bf16 MFMA peak.  `rocprof_kernels` names the kernel(s) one call launches (the fused attention backward is
                  three: row constants, main kernel, tail) -- `avg_launch_us` is the sum of their rocprofv3 averages and
                  `traffic` the sum of their PMC bytes; `executed_tflops` counts the recomputed matmuls as well
  roofline_hbm -- the same for the dominant BANDWIDTH-bound kernel (LayerNorm backward): algorithmic bytes / measured time vs the
                  8 TB/s HBM3E peak, `traffic` = its PMC bytes per launch
  cpu_baseline -- the CPU oracle (oracle/mae3d_ref.py, a port of the reference's non-flash model) timed on this host's
                  cores for ONE volume forward+backward (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise); the GPU boxes export it already --
# set before anything initialises HIP, and only if the launcher did not choose a value itself
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
# BASELINE.md §3 books 1636 GFLOP per forward with the patch embedding over all 5120 tokens (8.05 GF); this build embeds the
# 1280 KEPT tokens only (2.0 GF, output-identical: SURVEY §8 R1), so the model-FLOP rate counts what is executed: 1636 - 6.05.
FWD_GFLOP_PER_VOLUME = 1629.95


def cpu_baseline(timed_iters=3):
    """SURVEY 8(d) protocol: the CPU oracle (a port of the reference's non-flash model), ViT-L, ONE volume forward+backward
    per iteration, fp32, 1 warm-up + `timed_iters` timed iterations; value = volumes/s over the timed iterations.
    Threads: os.cpu_count(), capped at 32 -- all-core runs of this model on a 256-thread host are oversubscribed (356 s per
    iteration measured against ~18 s on 32 threads), which would be neither a fair baseline nor a bounded sample.  Both the
    threads used ("cores") and the host's count ("host_cores") are reported."""
    from oracle import mae3d_ref as O
    host = os.cpu_count() or 1
    cores = min(host, 32)
    torch.set_num_threads(cores)
    P = O.init_params(O.VIT_L, seed=0)
    imgs = torch.rand(1, 1, 60, 256, 256, generator=torch.Generator().manual_seed(0))
    noise = torch.rand(1, 5120, generator=torch.Generator().manual_seed(1))
    times = []
    for it in range(1 + timed_iters):
        t0 = time.time()
        loss, _, _, _, _ = O.forward_backward(P, imgs, O.VIT_L, 0.75, noise)
        times.append(time.time() - t0)
    timed = times[1:]
    dt = sum(timed) / len(timed)
    return {"value": 1.0 / dt, "unit": "volumes/s", "cores": cores, "host_cores": host, "kind": "port",
            "sample": f"1 warm-up ({times[0]:.1f} s) + {len(timed)} timed iterations of 1 volume (1x60x256x256) forward+backward, "
                      f"fp32, torch CPU {cores} of {host} host threads: " + ", ".join(f"{t:.1f}" for t in timed) +
                      f" s, loss {float(loss):.4f}"}


# bench kernel kind -> kernel name(s) in the PMC file (every 256-tile GEMM runs the phased main loop, gemm256p_kernel, since
# 993adab; the two-stage gemm256_kernel names stay listed for PMC files collected before that)
_PMC_KERNEL = {"gemm_wgrad_epi5": ["gemm256p_wgrad_pair_kernel", "gemm256p_kernel<true, true, 5, true>", "gemm256_kernel<true, true, 5, true>"],
               "gemm_dgrad_epi0": ["gemm256p_kernel<true, false, 0, false>", "gemm256_kernel<true, false, 0, false>"],
               "gemm_dgrad_epi4": ["gemm256p_kernel<true, false, 4, false>", "gemm256_kernel<true, false, 4, false>"],
               "gemm_dgrad_epi6": ["gemm256p_kernel<true, false, 6, false>"],      # proj dgrad + the attention backward's delta (round 4)
               "gemm_fwd_epi0": ["gemm256p_kernel<false, false, 0, false>", "gemm256_kernel<false, false, 0, false>"], "gemm_fwd_epi2": ["gemm256p_kernel<false, false, 2, false>", "gemm256_kernel<false, false, 2, false>"],
               "gemm_fwd_epi3": ["gemm256p_kernel<false, false, 3, false>", "gemm256_kernel<false, false, 3, false>"],
               "attn_fwd_hd32": ["attn_fwd_kernel<32, true>"], "attn_fwd_hd64": ["attn_fwd_kernel<64, true>"],
               "attn_bwd_dq_hd32": ["attn_bwd_dq_kernel<32>"], "attn_bwd_dq_hd64": ["attn_bwd_dq_kernel<64>"],
               "attn_bwd_dkv_hd32": ["attn_bwd_dkv_kernel<32>"], "attn_bwd_dkv_hd64": ["attn_bwd_dkv_kernel<64>"],
               "attn_bwd_fused_hd32": ["attn_bwd_fused1w_kernel", "attn_bwd_fused_kernel<32>"], "attn_bwd_fused_hd64": ["attn_bwd_fused1w64_kernel", "attn_bwd_fused_kernel<64>"],
               "ln_bwd_d1024": ["ln_bwd_kernel<4>"], "ln_bwd_d512": ["ln_bwd_kernel<2>"],
               "ln_fwd_d1024": ["ln_fwd_kernel<4>"], "ln_fwd_d512": ["ln_fwd_kernel<2>"]}
# A timed kind that is one C-ABI entry point but several launches: the HIP events bracket the whole entry, so `avg_launch_us`
# is the SUM of these kernels' average durations in a rocprofv3 summary, and `traffic` the sum of their bytes.
# (since round 4 the per-query constants come from the delta the proj dgrad GEMM wrote: attn_rowconst_from_delta_kernel only
# transposes and pads; PMC files taken before that list attn_rowconst_pad_kernel<HD> instead -- the second form below)
_LAUNCH_GROUP = {"attn_bwd_fused_hd32": ["attn_rowconst_from_delta_kernel", "attn_bwd_fused1w_kernel"],
                 "attn_bwd_fused_hd64": ["attn_rowconst_from_delta_kernel", "attn_bwd_fused1w64_kernel"]}
_LAUNCH_GROUP_OLD = {"attn_bwd_fused_hd32": ["attn_rowconst_pad_kernel<32>", "attn_bwd_fused1w_kernel"],
                     "attn_bwd_fused_hd64": ["attn_rowconst_pad_kernel<64>", "attn_bwd_fused1w64_kernel"]}
# (N = 5121 and 1281 leave ONE key past the last full key block: the one-wave main kernels take it, and the workspace -> bf16
# conversion, at their end -- attn_bwd_tail1.hpp; until mid round 3 that was a third launch, attn_bwd_tail1_kernel)


class BoardSampler:
    """Board power and shader clock during the timed region, read from the amdgpu hwmon files in sysfs by a side thread (plain
    file reads: no child process is started after the GPU has been initialised).  The MFMA kernels of this step run AT the package
    power cap (1400 W) with the clock pulled down to 1.65-1.95 GHz -- the same binaries run 19-26 % faster on all-zero operands at
    2.4 GHz and 1065-1235 W (tools/power_probe.py, profiles/r04_power_probe.txt) -- so the sustainable dense-bf16 rate of the part
    on real data is ~1.1-1.3 PFLOP/s, not the nominal 2.5: the line reports what the board did while the number was measured."""

    def __init__(self, device_index=0, period=0.25):
        import glob
        import threading
        self.period, self._stop = period, False
        self.cards = {}            # hwmon dir -> [(power W, sclk MHz)]
        want = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if os.path.exists(os.path.join(h, "power1_input")) and os.path.exists(os.path.join(h, "freq1_input")):
                pci = os.path.basename(os.path.realpath(os.path.join(h, "..", "..")))
                self.cards[h] = {"pci": pci, "s": []}
        match = [h for h, c in self.cards.items() if want is not None and c["pci"] == want]
        if match:                  # the card HIP device `device_index` is; otherwise every card is sampled and the busiest one reported
            self.cards = {match[0]: self.cards[match[0]]}
        self.matched = bool(match)
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop:
            for h, c in self.cards.items():
                try:
                    c["s"].append((int(open(os.path.join(h, "power1_input")).read()) * 1e-6, int(open(os.path.join(h, "freq1_input")).read()) * 1e-6))
                except Exception:
                    pass
            time.sleep(self.period)

    def start(self):
        if self.cards:
            self._t.start()

    def stop(self):
        self._stop = True
        best = None
        for h, c in self.cards.items():
            if c["s"]:
                avg = sum(p for p, _ in c["s"]) / len(c["s"])
                if best is None or avg > best[0]:
                    best = (avg, h, c)
        if best is None:
            return None
        _, h, c = best
        n = len(c["s"])
        cap = None
        try:
            cap = int(open(os.path.join(h, "power1_cap")).read()) * 1e-6
        except Exception:
            pass
        return {"power_w_avg": sum(p for p, _ in c["s"]) / n, "power_w_max": max(p for p, _ in c["s"]),
                "sclk_mhz_avg": sum(f for _, f in c["s"]) / n, "samples": n, "power_cap_w": cap, "pci": c["pci"],
                "matched_by_pci_id": self.matched,
                "source": "amdgpu hwmon (power1_input, freq1_input) in sysfs, sampled every %.2f s over the timed region" % self.period}


def kernels_hash():
    """sha1 (12 hex digits) over the kernel sources (octcubem_amd/csrc/*.hip, *.hpp, *.inc, Makefile, include/octmae.h): what the
    PMC traffic file was collected on and what this run executes are the same kernels exactly when the two hashes agree -- documentation
    and evidence commits move the git HEAD without touching a kernel ("evidence freshness")."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "octcubem_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "octcubem_amd", "csrc", "*.inc"))
                   + [os.path.join(ROOT, "octcubem_amd", "csrc", "Makefile"), os.path.join(ROOT, "include", "octmae.h")])
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def pmc_traffic(kind, micro_batch):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (tools/collect_pmc_traffic.sh,
    profiles/README.md): (2 x FETCH_SIZE + WRITE_SIZE) x 1024, FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM.  The
    profiler cannot wrap the process it runs in, so this is the offline measurement of the same per-launch shapes; None when
    the file was taken at another micro-batch (other shapes)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.json")), reverse=True):
        try:
            t = json.load(open(path))
            meta = t.get("_meta", {})
            if int(meta.get("micro_batch", 32)) != micro_batch:
                continue
            for group in (_LAUNCH_GROUP.get(kind), _LAUNCH_GROUP_OLD.get(kind)):
                if group and all(name in t for name in group):
                    return (sum(t[name]["hbm_bytes_per_launch_corrected"] for name in group), os.path.basename(path),
                            meta.get("head"), meta.get("kernels_hash"))
            for name in _PMC_KERNEL.get(kind, []):
                if name in t:
                    return t[name]["hbm_bytes_per_launch_corrected"], os.path.basename(path), meta.get("head"), meta.get("kernels_hash")
        except Exception:
            continue
    return None, None, None, None


def bring_up_native_comm(store, rank, world, local_rank, comm_cls, fail_rank=None, poll=0.5, timeout=300.0):
    """The native communicator of this rank -- or the END of the process, with status 3, on EVERY rank when ANY rank cannot create
    it: ncclCommInitRank is collective, so the peers of a rank whose creation failed would otherwise sit in it for ever.  The
    failing rank publishes a key in the control-plane store (torch.distributed's, which also carries the RCCL unique id:
    comm.NativeComm.from_store); a watchdog thread on every rank polls that key while the rank itself may be blocked inside the
    creation, and ends the process (os._exit: nothing in this process can be unwound out of a blocked RCCL call).  Ranks that
    succeeded count themselves in and wait for all `world` of them before the watchdog stands down (a peer may still fail after this
    rank succeeded).  `comm_cls`: comm.NativeComm; the CPU tests pass a gloo-backed stand-in with the same from_store protocol
    (tests/test_cpu_parallel.py).  `fail_rank`: tests -- that rank's creation fails."""
    import threading
    fail_key, ok_key, done = "octmae/bench/comm_failed", "octmae/bench/comm_ok", threading.Event()

    def _watch():
        while not done.wait(poll):
            try:
                failed = store.check([fail_key])
            except Exception:
                return
            if failed:
                print(f"[bench] rank {rank}: a peer could not create the native RCCL communicator: exiting with status 3",
                      file=sys.stderr, flush=True)
                os._exit(3)
    threading.Thread(target=_watch, daemon=True).start()
    try:
        if fail_rank is not None and str(fail_rank) == str(rank):
            raise RuntimeError("simulated communicator failure (OCTMAE_BENCH_FAIL_COMM_RANK)")
        comm = comm_cls.from_store(store, rank, world, local_rank)
    except Exception as e:
        print(f"[bench] rank {rank}: FATAL: the native RCCL communicator (octmae_comm_*) could not be created: {e!r}.  "
              "Pass --torch-nccl to measure torch.distributed's NCCL group instead.", file=sys.stderr, flush=True)
        try:
            store.set(fail_key, str(rank))
        except Exception:
            pass
        sys.exit(3)
    store.add(ok_key, 1)
    t_end = time.time() + timeout
    while int(store.add(ok_key, 0)) < world and not store.check([fail_key]) and time.time() < t_end:
        time.sleep(0.05)
    if store.check([fail_key]) or int(store.add(ok_key, 0)) < world:
        print(f"[bench] rank {rank}: not every rank created its communicator: exiting with status 3", file=sys.stderr, flush=True)
        os._exit(3)
    done.set()
    return comm


def agree_any(comm, flag: bool) -> bool:
    """True on every rank iff `flag` is true on ANY rank (one MAX all-reduce through the communicator the gradients use)."""
    return comm.all_reduce_scalar(1.0 if flag else 0.0, MAX_OP) > 0.0


def max_over_ranks(comm, value: float) -> float:
    return comm.all_reduce_scalar(float(value), MAX_OP)


MAX_OP = 2      # comm.MAX (octcubem_amd/comm.py; restated so that this file's helpers import nothing before the GPU is chosen)


def run_parity_compliant_child(args, steps=5, warmup=1):
    """The SAME workload on the library that meets the north star's 1e-3 on pred / logits / features (liboctmae_f16.so: the same kernels
    on IEEE-half operands, the reference's own default arithmetic), in a fresh process with the GPU to itself,
    through the reference's loss-scaler call -- NativeScalerWithGradNormCount() with GradScaler's dynamic loss scale
    (main_pretrain_oph_joint_2d512_flash_attn.py:456, custom_util/misc.py:311-344).  Returns the record embedded as
    `parity_compliant`; the headline value stays the bfloat16 library's (BASELINE's dtype).  A failure is recorded, never raised."""
    import subprocess
    lib = os.path.join(ROOT, "octcubem_amd", "liboctmae_f16.so")
    rec = {"dtype": "f16", "lib": os.path.basename(lib), "value": None, "unit": "volumes/s", "steps": steps, "warmup": warmup,
           "what": "this bench's workload (global batch 256, same micro-batches, fwd+bwd+AdamW) on the half-operand build of the same "
                   "kernels, dynamic loss scale as the reference constructs it; a child process run before the parent touched the GPU"}
    if not os.path.exists(lib):
        rec["error"] = "liboctmae_f16.so is missing (make -C octcubem_amd/csrc both)"
        return rec
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-kernel-timing",
           "--no-per-rank-proxy", "--no-parity-compliant", "--no-small-batch", "--dynamic-loss-scale", "--global-batch", str(args.global_batch),
           "--micro-batch", str(args.micro_batch)]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OCTMAE_LIB=lib), capture_output=True, text=True, timeout=900)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            rec["error"] = f"rc {r.returncode}: {(r.stderr or r.stdout)[-600:]}"
            return rec
        d = json.loads(lines[-1])
        assert d["dtype"] == "f16", d["dtype"]
        rec.update({"value": d["value"], "ms_per_step": d["ms_per_step"], "loss": d["loss"], "loss_scale": d["loss_scale"]["scale"],
                    "dynamic_loss_scale": d["loss_scale"]["dynamic"], "skipped_steps": d["loss_scale"]["skipped_steps"],
                    "micro_batch_per_gpu": d["config"]["micro_batch_per_gpu"], "peak_allocated_gib": d["memory"]["peak_allocated_gib"]})
    except Exception as e:      # the headline run must survive whatever happens here
        rec["error"] = repr(e)[:600]
    return rec


def small_batch_record(model, opt, scaler, params, dev):
    """The reference's SHIPPED operating point and its neighbours: 1 volume per GPU and step
    (scripts/run_chunks_pretraining_vitl_oph_joint_flash_attn.sh:25-30), 4 and 8 volumes, and the joint recipe's step -- 1 volume of
    60x256x256 + 64 B-scans of 512^2 through the high-resolution branch, mask 0.9 / 0.9, ONE optimizer step over the summed loss
    (Pre-training/engine_pretrain.py:110-127).  Same model, optimizer and kernels as the headline step; device-resident inputs."""
    import time as _t
    rec = {"what": "one optimizer step (fwd+bwd+AdamW) over B volumes, B = 1 / 4 / 8, and the shipped joint step (1 volume + 64 B-scans of "
                   "512x512 as 3-frame volumes, mask 0.9 both); ms per step = wall clock over `steps` steps after `warmup`",
           "steps": 10, "warmup": 3, "by_volumes": {}}
    g = torch.Generator(device=dev).manual_seed(4321)

    def timed(fn):
        for _ in range(rec["warmup"]):
            fn()
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        for _ in range(rec["steps"]):
            fn()
        torch.cuda.synchronize()
        return (_t.perf_counter() - t0) / rec["steps"]

    for B in (1, 4, 8):
        x = torch.rand(B, 1, 60, 256, 256, device=dev, generator=g)

        def step():
            opt.zero_grad()
            loss, _, _ = model(x, mask_ratio=0.75)
            scaler(loss, opt, parameters=params)
        dt = timed(step)
        rec["by_volumes"][str(B)] = {"ms_per_step": 1e3 * dt, "volumes_per_s": B / dt}
        del x
    try:
        vol = torch.rand(1, 1, 60, 256, 256, device=dev, generator=g)
        scans = torch.rand(64, 1, 3, 512, 512, device=dev, generator=g)

        def joint():
            opt.zero_grad()
            (l3, _frame_loss), _, _ = model(vol, mask_ratio=0.9, frame_loss=True)
            l2, _, _ = model(scans, mask_ratio=0.9)
            scaler(l3 + l2, opt, parameters=params)
        dt = timed(joint)
        rec["joint_recipe_step"] = {"ms_per_step": 1e3 * dt, "volumes": 1, "bscans_512": 64}
    except Exception as e:
        rec["joint_recipe_step"] = {"ms_per_step": None, "error": repr(e)[:300]}
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--global-batch", type=int, default=256)
    ap.add_argument("--micro-batch", type=int, default=0,
                    help="volumes per forward/backward; 0 = auto: 128 when the per-rank share is a multiple of it and it fits on "
                         "every rank (falls back to 64, agreed across ranks), min(64, global_batch / N) otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-per-rank-proxy", action="store_true")
    ap.add_argument("--clip-grad", type=float, default=None)
    ap.add_argument("--force-reducer", action="store_true", help="single rank: still create the RCCL group and run the reducer")
    ap.add_argument("--torch-nccl", action="store_true", help="exchange through torch.distributed's NCCL group instead of octmae_comm_*")
    ap.add_argument("--gloo-one-gpu", action="store_true",
                    help="diagnostic: every rank on GPU 0, exchange over gloo -- exercises the multi-rank control flow of this script on "
                         "a one-GPU box (RCCL refuses two ranks on one device); the number it prints is not a benchmark")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=INT",
                    help="kernel-selection switches for same-box A/B runs of the WHOLE step: octmae_set_option keys (attn_bwd_hd32_form, "
                         "attn_bwd_hd64_form, attn_bwd_tail_fused, gemm_small, wgrad_s1_atomic) or ops.<NAME> flags (ops.FORCE_TWO_STAGE=1, "
                         "ops.ATTN_BWD_FUSED32=0, ...); repeatable.  Isolated kernel A/Bs do not always carry over to the step "
                         "(DESIGN.md section 5, round 4), so defaults are decided here")
    ap.add_argument("--host-inputs", type=int, default=0, choices=[0, 1, 2],
                    help="0 (the contract: inputs resident in HBM when the timed region starts); 1: every micro-batch starts in pinned "
                         "host memory and is copied on the compute stream, as the engines do (samples.to(device, non_blocking=True)); "
                         "2: the same copies on a side stream, one micro-batch ahead.  For the PCIe-inclusive rate noted in DESIGN.md; "
                         "never the headline value")
    ap.add_argument("--no-parity-compliant", action="store_true",
                    help="N = 1: skip the child run of the same workload on the half-operand build (the `parity_compliant` record)")
    ap.add_argument("--no-small-batch", action="store_true", help="N = 1: skip the `small_batch` record (1 / 4 / 8 volumes per step + the joint recipe step)")
    ap.add_argument("--dynamic-loss-scale", action="store_true",
                    help="the loss scaler as the reference constructs it, NativeScalerWithGradNormCount(): GradScaler's dynamic loss scale "
                         "when the library computes on half operands (OCTMAE_LIB=liboctmae_f16.so), the identity on bfloat16")
    ap.add_argument("--same-data", action="store_true",
                    help="diagnostic: every rank draws the SAME volumes (seed without the rank) -- with identical weights and masking "
                         "noise the ranks' losses must then be bit-equal (comm.last_loss_min_max_over_ranks)")
    args = ap.parse_args()

    # ONE line on stdout: RCCL prints a version banner to stdout when a communicator is created (NCCL_DEBUG=VERSION/WARN), and
    # any library may print.  File descriptor 1 is pointed at stderr for the whole run; the JSON line goes to the saved one.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    parity_compliant = None
    if world == 1 and args.gpus == 1 and not args.no_parity_compliant and not args.force_reducer and args.global_batch == 256 \
            and os.environ.get("OCTMAE_LIB") is None:
        parity_compliant = run_parity_compliant_child(args)          # BEFORE this process touches the GPU; the child has it to itself
    local_rank = 0 if args.gloo_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("OCTMAE_BENCH_FORCE_LOCAL_RANK") is not None:      # tests: several ranks on one GPU with the NATIVE backend selected
        local_rank = int(os.environ["OCTMAE_BENCH_FORCE_LOCAL_RANK"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_reducer
    from octcubem_amd import models_mae, misc, ops, optim as foptim, comm as ocomm
    from octcubem_amd.parallel import FlatGradReducer
    for kv in args.set:
        k, v = kv.split("=")
        if k.startswith("ops.ATTN_BWD_FUSED"):
            ops.ATTN_BWD_FUSED[int(k[len("ops.ATTN_BWD_FUSED"):])] = bool(int(v))
        elif k.startswith("ops."):
            assert hasattr(ops, k[4:]), k
            old = getattr(ops, k[4:])      # the value takes the type of the switch it replaces (bool through int: "0" / "1")
            setattr(ops, k[4:], bool(int(v)) if isinstance(old, bool) else type(old)(v))
        else:
            ops.set_option(k, int(v))
    comm = None
    comm_kind = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # torch.distributed is the control plane only (its store carries the RCCL unique id): no device_id, so its own NCCL
        # communicator is created lazily -- i.e. never, unless the native communicator cannot be built.
        dist.init_process_group(backend="gloo" if args.gloo_one_gpu else "nccl", rank=rank, world_size=world)
        # ONE backend on the data path, chosen by the command line and never by what happens to work: the native communicator
        # (default), torch.distributed's NCCL group (--torch-nccl, for comparison) or gloo (--gloo-one-gpu, diagnostic).  If the
        # native communicator cannot be created the run FAILS (non-zero exit on every rank): a scaling run that silently measured
        # another backend would be worse than no number.
        if args.gloo_one_gpu:
            comm_kind = "torch.distributed gloo (--gloo-one-gpu: diagnostic, every rank on GPU 0)"
        elif args.torch_nccl:
            comm_kind = "torch.distributed nccl (--torch-nccl)"
        else:
            store = dist.distributed_c10d._get_default_store()
            comm = bring_up_native_comm(store, rank, world, local_rank, ocomm.NativeComm,
                                        fail_rank=os.environ.get("OCTMAE_BENCH_FAIL_COMM_RANK"))
            ocomm.set_default(comm)
            comm_kind = "octmae_comm (RCCL behind the C ABI)"

    assert args.global_batch % world == 0
    per_rank = args.global_batch // world
    big_ok = per_rank % 128 == 0 and torch.cuda.mem_get_info(dev)[1] >= 280e9
    if args.micro_batch > 0:
        candidates = [min(args.micro_batch, per_rank)]
    elif big_ok:
        # ~1.9 GiB of saved activations per volume: 128 volumes take 246 of the 268 GiB; +1.6 % over 64 (longer k-loops in the
        # weight-gradient GEMMs, fewer tile-round tails).  With peers the ranks must AGREE: the trial below runs without any
        # collective, then one MAX all-reduce of "it did not fit here" decides for everybody (a rank that fell back alone
        # would leave the others waiting in the gradient exchange).
        candidates = [128, 64]
    else:
        candidates = [min(64, per_rank)]

    torch.manual_seed(0)                                   # identical initial weights on every rank
    model = models_mae.octcube_vit_large_3dmae().to(dev)
    model.train()
    reducer = FlatGradReducer(model, force=args.force_reducer, comm=comm, allow_torch_nccl=args.torch_nccl) if use_dist else None
    model.prepare()
    if reducer is not None:
        reducer.broadcast_parameters(0)
    opt = foptim.FusedAdamW(misc.add_weight_decay(model, 0.05), lr=1.6e-3 * args.global_batch / 256, betas=(0.9, 0.95))
    # bfloat16 needs no loss scale (fp32=True: the identity whatever the library); --dynamic-loss-scale: the reference's own call,
    # GradScaler's state machine on the half build (main_pretrain_oph_joint_2d512_flash_attn.py:456, custom_util/misc.py:311-344)
    scaler = misc.NativeScalerWithGradNormCount(fp32=not args.dynamic_loss_scale, reducer=reducer)
    skipped_steps = [0]
    params = list(model.parameters())

    def fence():
        if comm is not None:
            comm.barrier()          # 1-element all-reduce on the communication stream + device synchronise
        elif use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def agree_failed(failed: bool) -> bool:
        if comm is not None:
            return agree_any(comm, failed)
        if use_dist:
            t = torch.tensor([1.0 if failed else 0.0], device="cpu" if args.gloo_one_gpu else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item()) > 0.0
        return failed

    def make_step(global_batch, mb):
        """One optimizer step over `global_batch` volumes (this rank's share as micro-batches of `mb`), on fresh synthetic data."""
        per = global_batch // world
        assert per % mb == 0
        accum_ = per // mb
        # rank r sees different volumes (seed + rank, main_pretrain…:306)
        g = torch.Generator(device=dev).manual_seed(1234 + (0 if args.same_data else rank))
        pool_ = [torch.rand(mb, 1, 60, 256, 256, device=dev, generator=g) for _ in range(min(accum_, 2))]
        host_ = [t.cpu().pin_memory() for t in pool_] if args.host_inputs else None
        copy_stream = torch.cuda.Stream(device=dev) if args.host_inputs == 2 else None
        ahead = {}

        def fetch(i):
            """micro-batch i of the step as a device tensor (--host-inputs: copied from pinned host memory inside the timed region)"""
            if not args.host_inputs:
                return pool_[i % len(pool_)]
            if copy_stream is None:
                return host_[i % len(host_)].to(dev, non_blocking=True)
            def start(j):
                with torch.cuda.stream(copy_stream):
                    t = host_[j % len(host_)].to(dev, non_blocking=True)
                    ev = torch.cuda.Event(); ev.record(copy_stream)
                return t, ev
            t, ev = ahead.pop(i, None) or start(i)
            ahead[(i + 1) % accum_] = start(i + 1)           # the next micro-batch (of this or the next step) under this one's compute
            torch.cuda.current_stream().wait_event(ev)
            t.record_stream(torch.cuda.current_stream())
            return t

        def step_(exchange=True):
            opt.zero_grad()
            last = None
            for i in range(accum_):
                if os.environ.get("OCTMAE_BENCH_FAKE_OOM") and mb == 128:     # exercises the fallback (tests only)
                    raise torch.OutOfMemoryError("simulated")
                loss_, _, _ = model(fetch(i), mask_ratio=0.75)
                last = loss_
                scaler(loss_ / accum_, opt, parameters=params, update_grad=(exchange and i == accum_ - 1), clip_grad=args.clip_grad)
            if exchange and scaler.enabled and scaler.last_step_skipped:
                skipped_steps[0] += 1
            return last
        return step_, accum_

    for ci, mb in enumerate(candidates):
        failed = False
        step = None
        try:
            step, accum = make_step(args.global_batch, mb)
            if len(candidates) > 1 and ci == 0:
                step(exchange=False)            # the trial: every allocation of a step, no collective, no optimizer update
                torch.cuda.synchronize()
        except torch.OutOfMemoryError:
            failed = True
        if len(candidates) > 1 and ci == 0 and agree_failed(failed):
            step = None
            torch.cuda.synchronize()
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            print(f"[bench] rank {rank}: micro-batch {mb} does not fit on every rank, falling back to {candidates[ci + 1]}",
                  file=sys.stderr, flush=True)
            continue
        if failed:
            raise torch.OutOfMemoryError(f"micro-batch {mb} does not fit")
        for _ in range(args.warmup):
            step()
        fence()
        break
    if not args.no_kernel_timing:
        ops.KTIMER = ops.KernelTimer()
    if reducer is not None:
        reducer.timing = True       # device events around every gradient chunk's exchange and around finish()'s wait (timed steps only)
    board = BoardSampler(local_rank) if rank == 0 else None
    if board is not None:
        board.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    board_stats = board.stop() if board is not None else None
    comm_timing = None
    if reducer is not None:
        reducer.timing = False
        comm_timing = reducer.timing_summary(last=1)
        # the slowest rank's wait is the step's exposed communication: MAX over ranks through the same backend
        if comm_timing is not None:
            if comm is not None:
                comm_timing["exposed_ms_per_step_max_over_ranks"] = max_over_ranks(comm, comm_timing["exposed_ms_per_step"])
            elif use_dist:
                t = torch.tensor([comm_timing["exposed_ms_per_step"]], dtype=torch.float64, device="cpu" if args.gloo_one_gpu else dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                comm_timing["exposed_ms_per_step_max_over_ranks"] = float(t.item())
    # device memory of this rank's step (caching-allocator peak since process start: the timed steps and the warm-ups are the same
    # step); what one volume of the micro-batch holds decides whether 128 fits in 288 GB
    peak_alloc, peak_reserved = torch.cuda.max_memory_allocated(dev), torch.cuda.max_memory_reserved(dev)
    memory = {"peak_allocated_gib": round(peak_alloc / 2**30, 2), "peak_reserved_gib": round(peak_reserved / 2**30, 2),
              "micro_batch": mb, "peak_allocated_gib_per_volume_of_micro_batch": round(peak_alloc / 2**30 / mb, 3)}
    kt = ops.KTIMER.summary() if ops.KTIMER is not None else {}
    ops.KTIMER = None
    if comm is not None:
        dt = max_over_ranks(comm, dt)
    elif use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.gloo_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_value = float(loss.detach())
    # every rank counts itself in through the SAME backend the gradients went through: ranks_seen must equal n_gpus
    if comm is not None:
        ranks_seen = int(round(comm.all_reduce_scalar(1.0, ocomm.SUM)))
        loss_minmax = (comm.all_reduce_scalar(-loss_value, ocomm.MAX), comm.all_reduce_scalar(loss_value, ocomm.MAX))
        loss_minmax = (-loss_minmax[0], loss_minmax[1])
    elif use_dist:
        t = torch.tensor([1.0, -loss_value, loss_value], dtype=torch.float64, device="cpu" if args.gloo_one_gpu else dev)
        t1 = t[:1].clone(); dist.all_reduce(t1, op=dist.ReduceOp.SUM)
        t2 = t[1:].clone(); dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        ranks_seen, loss_minmax = int(round(float(t1[0]))), (-float(t2[0]), float(t2[1]))
    else:
        ranks_seen, loss_minmax = 1, (loss_value, loss_value)

    # One-GPU proxy for strong scaling: what ONE rank of an N-GPU run executes per step -- global batch
    # 256 / N in one micro-batch -- timed on this GPU with the same model, optimizer and kernels; the ratio of its rate to the
    # 256-volume rate is the scaling efficiency before any exposed communication.
    proxy = None
    if world == 1 and not args.no_per_rank_proxy and args.global_batch == 256:
        proxy = {}
        main_rate = args.steps * args.global_batch / dt
        for n_ranks in (2, 4, 8):
            gb = args.global_batch // n_ranks
            pstep, _ = make_step(gb, min(gb, 128))
            for _ in range(2):
                pstep()
            torch.cuda.synchronize()
            k = 4
            t1 = time.perf_counter()
            for _ in range(k):
                pstep()
            torch.cuda.synchronize()
            pdt = (time.perf_counter() - t1) / k
            proxy[str(n_ranks)] = {"volumes_per_rank": gb, "ms_per_step": 1e3 * pdt, "volumes_per_s_per_gpu": gb / pdt,
                                   "ratio_to_256": (gb / pdt) / main_rate}
            del pstep

    small_batch = None
    if world == 1 and not args.no_small_batch and args.global_batch == 256 and not use_dist:
        try:
            small_batch = small_batch_record(model, opt, scaler, params, dev)
        except Exception as e:      # the headline number must survive
            small_batch = {"error": repr(e)[:400]}

    if rank == 0:
        vps = args.steps * args.global_batch / dt
        out = {
            "metric": "3D-MAE pretrain volumes/s (ViT-L, 60x256x256, mask 0.75)", "value": vps, "unit": "volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16" if ops.LP_IS_F16 else "bf16", "data": "synthetic",
            "config": {"workload": "OCTCube ViT-L 3D-MAE pre-train step (fwd+bwd+all-reduce+AdamW), synthetic (1,60,256,256) volumes, "
                                   "mask_ratio 0.75, decoder 512x8x16", "global_batch": args.global_batch,
                       "micro_batch_per_gpu": mb, "accum_steps": accum, "tokens_enc_dec": [1281, 5121],
                       "parallelism": f"dp{world}"},
            "loss": loss_value, "switches": args.set,
            "loss_scale": {"dynamic": scaler.enabled, "scale": float(scaler.get_scale()), "skipped_steps": skipped_steps[0]},
            "model_tflops_per_s": vps * 3 * FWD_GFLOP_PER_VOLUME / 1e3,
            "mfu_vs_dense_bf16_peak": vps * 3 * FWD_GFLOP_PER_VOLUME / 1e3 / (PEAK_BF16_TFLOPS * world),
        }
        if kt:
            hb = {k: v for k, v in kt.items() if k.startswith("ln_")}          # bandwidth-bound kinds (LayerNorm forward / backward)
            kt = {k: v for k, v in kt.items() if not k.startswith("ln_")}
            if hb:
                hdom = max(hb, key=lambda k: hb[k]["total_ms"])
                h = hb[hdom]
                gbs = h["bytes"] / (h["total_ms"] * 1e-3) / 1e9
                htraffic, hsrc, hhead, hkh = pmc_traffic(hdom, mb)
                out["roofline_hbm"] = {"kernel": hdom, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": gbs / PEAK_HBM_GBS, "traffic": htraffic, "traffic_source": hsrc, "traffic_head": hhead,
                                       "traffic_kernels_hash": hkh, "traffic_is_of_these_kernels": (hkh == kernels_hash()) if hkh else None,
                                       "avg_launch_us": h["avg_us"], "launches": h["launches"],
                                       "rocprof_kernels": _PMC_KERNEL.get(hdom, [hdom])[:1],
                                       "accounting": "algorithmic bytes: dy bf16 + x f32 + residual gradient f32 read, dx f32 + bf16 copy written",
                                       "all": {k: {"avg_us": round(v["avg_us"], 2), "launches": v["launches"],
                                                   "gbs": round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1)} for k, v in sorted(hb.items())}}
            tot = sum(v["total_ms"] for v in kt.values())
            dom = max(kt, key=lambda k: kt[k]["total_ms"])
            d = kt[dom]
            # achieved = ALGORITHMIC flop / measured time (attention forward 4 B H N^2 hd, backward 8 B H N^2 hd,
            # i.e. x3 in total; GEMMs 2 NA NB K); what the kernel EXECUTES (recomputed S, dP) is reported beside it
            ach = d["flops"] / (d["total_ms"] * 1e-3) / 1e12
            exe = d["exec_flops"] / (d["total_ms"] * 1e-3) / 1e12
            traffic, tsrc, thead, tkh = pmc_traffic(dom, mb)
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": tsrc, "traffic_head": thead,
                               # the PMC passes and this run executed the same kernel sources iff the hashes agree (kernels_hash())
                               "traffic_kernels_hash": tkh, "kernels_hash": kernels_hash(),
                               "traffic_is_of_these_kernels": (tkh == kernels_hash()) if tkh else None,
                               "executed_tflops": exe, "executed_frac": exe / PEAK_BF16_TFLOPS,
                               "avg_launch_us": d["avg_us"], "launches": d["launches"],
                               "rocprof_kernels": _LAUNCH_GROUP.get(dom, _PMC_KERNEL.get(dom, [dom])[:1]),
                               "share_of_timed_mfma_kernels": d["total_ms"] / tot,
                               "accounting": "algorithmic: attention fwd 4BHN^2hd, bwd 8BHN^2hd (x3 total), GEMM 2*NA*NB*K"}
            out["kernels"] = {k: {"ms": round(v["total_ms"], 3), "avg_us": round(v["avg_us"], 2), "launches": v["launches"],
                                  "tflops": round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 1),
                                  "executed_tflops": round(v["exec_flops"] / (v["total_ms"] * 1e-3) / 1e12, 1),
                                  # algorithmic bytes (operands, outputs, what the epilogue reads) over the same time: the
                                  # fused-epilogue GEMMs sit under both roofs
                                  "hbm_gbs_algorithmic": round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1)}
                              for k, v in sorted(kt.items())}
            att = [k for k in kt if k.startswith("attn_")]
            if att:
                fl = sum(kt[k]["flops"] for k in att); ms = sum(kt[k]["total_ms"] for k in att)
                xfl = sum(kt[k]["exec_flops"] for k in att)
                out["attention_qk_pv"] = {"tflops": fl / (ms * 1e-3) / 1e12, "frac_of_bf16_peak": fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                                          "executed_tflops": xfl / (ms * 1e-3) / 1e12, "ms": ms}
        if board_stats is not None:
            out["board"] = board_stats
        out["memory"] = memory
        if args.host_inputs:
            out["inputs"] = {1: "pinned host memory, copied on the compute stream inside the timed region (PCIe-inclusive)",
                             2: "pinned host memory, copied on a side stream one micro-batch ahead (PCIe-inclusive)"}[args.host_inputs]
        if proxy is not None:
            out["per_rank_proxy"] = {"what": "one GPU running the per-rank share of an N-GPU step (256 / N volumes, one micro-batch, "
                                             "no communication): ratio_to_256 = its volumes/s over this line's value",
                                     "by_n_gpus": proxy}
        if use_dist:
            red = None
            if reducer is not None:
                red = dict(reducer.stats)
                red["chunk_mb"] = [round(4 * (e - s_) / 1e6, 1) for s_, e in reducer.bounds]
                red["cold_chunk"] = [bool(c) for c in reducer.cold_chunk]
                red["transport"] = reducer.transport
            ct = comm_timing or {}
            out["comm"] = {"backend": comm_kind, "ranks_seen": ranks_seen, "agreed_micro_batch": mb, "reducer": red,
                           # What the optimizer waited for the exchange: the compute stream's idle time inside reducer.finish(), HIP
                           # events on the compute stream (rank 0; ..._max_over_ranks: the slowest rank), mean over the timed steps.
                           # timeline: the last timed step's chunks in launch order, ms after begin_backward() -- ready = gradients
                           # final on the compute stream, done = collective complete on the communication stream.
                           "exposed_ms_per_step": ct.get("exposed_ms_per_step"), "exposed_ms_max": ct.get("exposed_ms_max"),
                           "exposed_ms_per_step_max_over_ranks": ct.get("exposed_ms_per_step_max_over_ranks"),
                           "exposed_frac_of_step": (ct["exposed_ms_per_step_max_over_ranks"] / (1e3 * dt / args.steps))
                           if ct.get("exposed_ms_per_step_max_over_ranks") is not None else None,
                           "last_micro_step_backward_ms": ct.get("backward_ms_per_step"), "timeline": ct.get("timeline"),
                           # ranks see different volumes (seed + rank), so their last-step losses differ slightly; identical
                           # weights after the exchange keep them within sampling noise of each other
                           "last_loss_min_max_over_ranks": list(loss_minmax)}
        if parity_compliant is not None:
            if parity_compliant.get("value"):
                parity_compliant["ratio_to_headline"] = parity_compliant["value"] / vps
            out["parity_compliant"] = parity_compliant
        if small_batch is not None:
            out["small_batch"] = small_batch
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the GPU number must survive a host that cannot fit the oracle
                out["cpu_baseline"] = {"value": None, "unit": "volumes/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    if comm is not None:
        comm.barrier()
        comm.destroy()
        dist.destroy_process_group()
    elif use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which a pipe only sees when the buffer is flushed (at exit, i.e. AFTER
        # anything Python printed): push it out first so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()

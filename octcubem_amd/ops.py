"""Operators of the 3-D MAE hot path: thin wrappers over the C ABI (liboctmae.so) and the
torch.autograd.Function glue that strings them into forward/backward.

Numerics policy: fp32 residual stream / LayerNorm statistics / softmax statistics /
accumulators / master weights and weight gradients; bf16 GEMM and attention operands.

Weight gradients are accumulated by the wgrad GEMM (fp32, split-K) DIRECTLY into ``param.grad`` (the
model pre-binds ``param.grad`` to views of one flat gradient arena, see models_mae._ParamArena), so the
Functions below return ``None`` for parameters and call ``notify_grad_ready`` instead -- that is the hook
the data-parallel reducer uses to overlap the RCCL all-reduce with backward.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence

import torch

from ._lib import OctmaeError, call, load


def _lp_dtype():
    """The torch dtype of the library's 16-bit operand type (octmae_lp_dtype): bfloat16 for liboctmae.so, float16 for the
    verification build liboctmae_f16.so (OCTMAE_LIB).  A library that is missing or cannot be loaded here (a CPU-only box) is
    reported by the first compute call, not by the import; bfloat16 is assumed until then and CHECKED when the library does load."""
    try:
        return torch.float16 if load().octmae_lp_dtype() == 1 else torch.bfloat16
    except OctmaeError:
        return torch.bfloat16


BF16 = _lp_dtype()                 # named for the shipped build; every "bf16" below means "the library's 16-bit operand type"
LP_IS_F16 = BF16 == torch.float16


def _check_lp_dtype(lib):
    got = torch.float16 if lib.octmae_lp_dtype() == 1 else torch.bfloat16
    if got != BF16:
        raise OctmaeError(f"the library computes on {got} operands but octcubem_amd.ops was imported assuming {BF16} (the library was "
                          "built or selected after the import): restart the process with OCTMAE_LIB set before importing octcubem_amd")


from . import _lib as _libmod  # noqa: E402
_libmod._on_load.append(_check_lp_dtype)
F32 = torch.float32
_CHECK_IDS = os.environ.get("OCTMAE_CHECK_IDS", "0") == "1"    # verify the permutation contract of the assembly ops per call

EPI_BF16, EPI_F32, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_ACCUM = range(6)


def _stream():
    # the raw handle of the current stream of the current device: 0.3 us, against 8 us for torch.cuda.current_stream().cuda_stream
    # (a Stream object per call) -- per launch, ~240 times per forward + backward (tools/host_overhead_profile.py)
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:      # a PyTorch without these private accessors
        return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return t


# ------------------------------------------------------------------------------------------------
# grad-ready notification (used by parallel.FlatGradReducer)
# ------------------------------------------------------------------------------------------------
# Several reducers may listen during one backward (the COEM step drives one FlatGradReducer per tower): every subscriber sees
# every notification and picks out the parameters of its own arena.  Keyed by owner so that a reducer replaces / removes only
# its own entry (a single global callback let the second reducer's begin_backward() silence the first).
_grad_ready_subs: dict = {}


def add_grad_ready_callback(owner, cb: Callable[[Sequence[torch.nn.Parameter]], None]):
    _grad_ready_subs[id(owner)] = cb


def remove_grad_ready_callback(owner):
    _grad_ready_subs.pop(id(owner), None)


def set_grad_ready_callback(cb):
    """Single-subscriber form kept for callers that own the whole backward: replaces every subscription (None: clears)."""
    _grad_ready_subs.clear()
    if cb is not None:
        _grad_ready_subs[0] = cb


def notify_grad_ready(params):
    if _grad_ready_subs:
        ps = [p for p in params if p is not None]
        for cb in list(_grad_ready_subs.values()):
            cb(ps)


def grad_buf(p: torch.nn.Parameter) -> torch.Tensor:
    """The fp32 buffer weight gradients are accumulated into (``p.grad``; created zeroed if absent)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


# ------------------------------------------------------------------------------------------------
# per-launch timing (bench.py): HIP events on the stream the kernels are launched on
# ------------------------------------------------------------------------------------------------
class KernelTimer:
    """Brackets single-kernel launches with events on the launch stream; kind -> [(start, end)] plus per-kind launch counts
    and algorithmic flop / byte totals.  Only every ``stride``-th launch of a kind is bracketed (two event records per
    launch cost ~3 % of a step when every one of the ~7000 launches is timed); totals are scaled up by count / sampled.
    The stride is 5 -- coprime with the periods (2, 3, 4) in which differently-shaped GEMMs of one kind alternate inside a
    Block, so every shape is sampled in proportion (a stride of 4 only ever timed one of the four weight-gradient shapes)."""

    def __init__(self, stride: int = 5):
        self.stride = max(1, int(stride))
        self.records = {}
        self.stats = {}

    def launch(self, kind, flops, nbytes, fn, exec_flops=None):
        """flops: ALGORITHMIC (attention forward 4 B H N^2 hd, backward 8 B H N^2 hd = x3 in total);
        exec_flops: what the kernel executes (recomputed products included), defaults to flops."""
        st = self.stats.setdefault(kind, [0, 0.0, 0.0, 0.0])
        sampled = st[0] % self.stride == 0
        st[0] += 1; st[1] += flops; st[2] += nbytes; st[3] += flops if exec_flops is None else exec_flops
        if not sampled:
            fn()
            return
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records.setdefault(kind, []).append((s, e))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, recs in self.records.items():
            n, flops, nbytes, xflops = self.stats[kind]
            ms = sum(s.elapsed_time(e) for s, e in recs)
            avg_us = 1e3 * ms / len(recs)
            out[kind] = {"launches": n, "sampled": len(recs), "total_ms": avg_us * n * 1e-3, "avg_us": avg_us, "flops": flops,
                         "bytes": nbytes, "exec_flops": xflops}
        return out


KTIMER: Optional[KernelTimer] = None


def _launch(kind, flops, nbytes, fn, exec_flops=None):
    if KTIMER is None:
        fn()
    else:
        KTIMER.launch(kind, flops, nbytes, fn, exec_flops)


# ------------------------------------------------------------------------------------------------
# raw ops
# ------------------------------------------------------------------------------------------------
def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == BF16:
        return x.contiguous()
    x = x.contiguous()
    _chk(x, F32, "cast_bf16")
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    call("octmae_cast_f32_bf16", x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    return out


def cast_bf16_rowscale(x2d: torch.Tensor, rowscale: torch.Tensor, rows_per_scale: int) -> torch.Tensor:
    """bf16(rowscale[r // rows_per_scale] * x2d[r]) for fp32 x2d [R, D] (the gradient entering a stochastic-depth branch)."""
    _chk(x2d, F32, "cast_bf16_rowscale"); _chk(rowscale, F32, "rowscale")
    R, D = x2d.shape
    assert rowscale.numel() * rows_per_scale == R
    out = torch.empty((R, D), dtype=BF16, device=x2d.device)
    call("octmae_cast_rowscale_f32_bf16", x2d.data_ptr(), rowscale.data_ptr(), out.data_ptr(), R, D, rows_per_scale, _stream())
    return out


def cast_bf16_into(src: torch.Tensor, dst: torch.Tensor):
    call("octmae_cast_f32_bf16", src.data_ptr(), dst.data_ptr(), src.numel(), _stream())


def colsum_accum(x2d: torch.Tensor, out: torch.Tensor):
    """out[c] += sum_r x2d[r][c] (fp32 accumulate)."""
    M, N = x2d.shape
    call("octmae_colsum_accum", x2d.data_ptr(), 1 if x2d.dtype == BF16 else 0, out.data_ptr(), M, N, x2d.stride(0), _stream())


_GEMM_KIND = {(0, 0): "gemm_fwd", (1, 0): "gemm_dgrad", (1, 1): "gemm_wgrad"}


FORCE_SMALL_TILE = False    # tests: route every GEMM through the 128-tile register-staged kernel
FORCE_TWO_STAGE = False     # tests / A-B: the un-phased two-stage main loop of the 256-tile kernel
FORCE_PHASED = False        # tests / A-B: the phased main loop also where the two-stage one is the default
FORCE_SMALL_LAUNCH = 0      # tests / A-B: 4 or 2 = every forward / dgrad GEMM through gemm128d_kernel with that ring depth (its k split:
FORCE_SPLITK = 1            #   FORCE_SPLITK, 1 .. 4); -1 = never that kernel (the pre-round-6 choice)


# The fc1 forward's epilogue stores gelu'(pre) in place of pre and the fc2 dgrad's epilogue multiplies with it (VERDICT r05 item 3,
# route i): one more 16-bit rounding point (gelu' itself), no transcendental in the backward epilogue.  bench.py --set
# ops.GELU_PRIME_FWD=1 for the same-box A/B of the step; the flag a forward ran with travels with its saved tensor.
GELU_PRIME_FWD = os.environ.get("OCTMAE_GELU_PRIME_FWD", "0") == "1"


def _variant_bits():
    return (0x100 if FORCE_SMALL_TILE else 0) | (0x200 if FORCE_TWO_STAGE else 0) | (0x400 if FORCE_PHASED else 0) | \
        (0x1000 if FORCE_SMALL_LAUNCH == 4 else 0x2000 if FORCE_SMALL_LAUNCH == 2 else 0x4000 if FORCE_SMALL_LAUNCH < 0 else 0) | \
        (((FORCE_SPLITK & 7) << 16) if FORCE_SMALL_LAUNCH > 0 else 0)


# Split-K workspace of the small-launch forward / dgrad kernel (octmae_gemm_bf16_ws): one per (device, stream) -- launches on one
# stream run one after the other and may share it, launches on different streams may not.  64 MiB of partial-tile slots + 16 KiB of
# arrival counters, zeroed once (every launch leaves the counters zero).  ON by default since round 6: a long reduction on few
# tiles (the fc2 forward of ONE volume: 88 tiles of 128 x 128, K = 4096) is split 2-4 ways, deterministically (csrc/gemm.hip,
# gemm128d_kernel).  OCTMAE_SPLIT_WS=0 / bench.py --set ops.SPLIT_WS=0: never lend it (no launch is split).
SPLIT_WS = os.environ.get("OCTMAE_SPLIT_WS", "1") != "0"
_split_ws = {}


def _split_ws_for(st: int):
    """(pointer, bytes) of the current device's workspace for stream handle `st`, or (None, 0)."""
    if not SPLIT_WS:
        return None, 0
    key = (torch.cuda.current_device(), st)
    ws = _split_ws.get(key)
    if ws is None:
        if len(_split_ws) >= 8:                     # a process that keeps creating streams: do not hoard 64 MiB for each
            _split_ws.clear()
        ws = _split_ws[key] = torch.zeros(load().octmae_gemm_split_ws_kib() * 1024, dtype=torch.uint8, device="cuda")
    return ws.data_ptr(), ws.numel()


def _gemm(A, B, C, NA, NB, K, lda, ldb, ldc, a_ks, b_ks, epi, C2=None, bias=None, aux=None, ldaux=0, splitk=1):
    st = _stream()
    if (epi & 0xff) != EPI_ACCUM and not FORCE_SMALL_TILE:   # forward / dgrad kinds: lend the split-K workspace
        skp, skn = _split_ws_for(st)
        args = (A.data_ptr(), B.data_ptr(), C.data_ptr(), _p(C2), _p(bias), _p(aux), NA, NB, K, lda, ldb, ldc, ldaux, a_ks, b_ks,
                epi | _variant_bits(), FORCE_SPLITK if FORCE_SMALL_LAUNCH > 0 else splitk, skp, skn, st)
        fn = "octmae_gemm_bf16_ws"
    else:
        args = (A.data_ptr(), B.data_ptr(), C.data_ptr(), _p(C2), _p(bias), _p(aux), NA, NB, K, lda, ldb, ldc, ldaux, a_ks, b_ks,
                epi | _variant_bits(), splitk, st)
        fn = "octmae_gemm_bf16"
    if KTIMER is None:
        call(fn, *args)
    else:
        kind = f"{_GEMM_KIND.get((a_ks, b_ks), 'gemm')}_epi{epi & 0xff}"
        # algorithmic bytes: both operands once, the output, the second output of the GELU epilogue (2), the residual (3) or
        # pre-activation (4) the epilogue reads
        nbytes = 2.0 * (NA * K + NB * K) + C.element_size() * NA * NB
        if (epi & 0xff) == 2 and C2 is not None:
            nbytes += C2.element_size() * NA * NB
        if (epi & 0xff) in (3, 4) and aux is not None:
            nbytes += aux.element_size() * NA * NB
        KTIMER.launch(kind, 2.0 * NA * NB * K, nbytes, lambda: call(fn, *args))


def linear_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], mode: str = "bf16",
               res: Optional[torch.Tensor] = None, rowscale: Optional[torch.Tensor] = None, rows_per_scale: int = 1,
               store_dgelu: bool = False):
    """y = x @ w.T + bias.  x bf16 [M,K], w bf16 [N,K], bias f32 [N].
    mode: 'bf16' | 'f32' | 'gelu' (returns (pre, act)) | 'resid' (f32: res + y, or res + rowscale[m // rows_per_scale] * y)."""
    M, K = x.shape
    N = w.shape[0]
    dev = x.device
    if rowscale is not None:
        assert mode == "resid" and rowscale.dtype == F32 and rowscale.numel() * rows_per_scale == M
        out = torch.empty((M, N), dtype=F32, device=dev)
        st = _stream()
        args = (w.data_ptr(), x.data_ptr(), out.data_ptr(), _p(bias), res.data_ptr(), rowscale.data_ptr(), rows_per_scale, N, M, K,
                w.stride(0), x.stride(0), N, res.stride(0), _variant_bits(), *_split_ws_for(st), st)
        _launch(f"gemm_fwd_epi{EPI_RESID}", 2.0 * N * M * K, 2.0 * (N * K + M * K) + 8.0 * N * M,
                lambda: call("octmae_linear_resid_rowscale", *args))
        return out
    if mode == "bf16":
        out = torch.empty((M, N), dtype=BF16, device=dev)
        _gemm(w, x, out, N, M, K, w.stride(0), x.stride(0), N, 0, 0, EPI_BF16, bias=bias)
        return out
    if mode == "f32":
        out = torch.empty((M, N), dtype=F32, device=dev)
        _gemm(w, x, out, N, M, K, w.stride(0), x.stride(0), N, 0, 0, EPI_F32, bias=bias)
        return out
    if mode == "gelu":
        pre = torch.empty((M, N), dtype=BF16, device=dev)
        act = torch.empty((M, N), dtype=BF16, device=dev)
        # store_dgelu: `pre` receives gelu'(pre-activation) -- what linear_dgrad(..., pre_is_dgelu=True) multiplies with
        _gemm(w, x, pre, N, M, K, w.stride(0), x.stride(0), N, 0, 0, EPI_GELU | (0x8000 if store_dgelu else 0), C2=act, bias=bias)
        return pre, act
    if mode == "resid":
        out = torch.empty((M, N), dtype=F32, device=dev)
        _gemm(w, x, out, N, M, K, w.stride(0), x.stride(0), N, 0, 0, EPI_RESID, bias=bias, aux=res, ldaux=res.stride(0))
        return out
    raise ValueError(mode)


def linear_dgrad(dy: torch.Tensor, w: torch.Tensor, pre: Optional[torch.Tensor] = None,
                 colsum: Optional[torch.Tensor] = None, atomic_colsum: bool = False, pre_is_dgelu: bool = False) -> torch.Tensor:
    """dx[M,K] = dy[M,N] @ w[N,K]  (optionally * gelu'(pre[M,K])), bf16.  With ``pre``, ``colsum`` (fp32 [K]) receives
    += the column sums of dx -- the bias gradient of the Linear that produced ``pre`` -- from the same call: per-slab partial
    sums through a workspace and a folding launch (octmae_linear_dgrad_dgelu); ``atomic_colsum`` selects the
    fp32-atomic form of octmae_gemm_bf16's epilogue 4 instead (kept for comparison)."""
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty((M, K), dtype=BF16, device=dy.device)
    if pre is None:
        assert colsum is None
        _gemm(w, dy, dx, K, M, N, w.stride(0), dy.stride(0), K, 1, 0, EPI_BF16)
    elif colsum is None or atomic_colsum:
        _gemm(w, dy, dx, K, M, N, w.stride(0), dy.stride(0), K, 1, 0, EPI_DGELU | (0x8000 if pre_is_dgelu else 0), C2=colsum, aux=pre,
              ldaux=pre.stride(0))
    else:
        rows = load().octmae_dgelu_colsum_ws_rows(M)
        ws = torch.empty((rows, K), dtype=F32, device=dy.device)
        st = _stream()
        args = (w.data_ptr(), dy.data_ptr(), dx.data_ptr(), pre.data_ptr(), ws.data_ptr(), colsum.data_ptr(), M, N, K, w.stride(0),
                dy.stride(0), K, pre.stride(0), _variant_bits() | (0x8000 if pre_is_dgelu else 0), *_split_ws_for(st), st)
        if KTIMER is None:
            call("octmae_linear_dgrad_dgelu", *args)
        else:   # two launches (GEMM + the fold of the partial sums), timed together
            KTIMER.launch("gemm_dgrad_epi4", 2.0 * K * M * N, 2.0 * (K * N + M * N) + 2.0 * K * M + 2.0 * K * M,      # + the pre-activation read
                          lambda: call("octmae_linear_dgrad_dgelu", *args))
    return dx


DGRAD_DELTA = True     # the proj dgrad also produces the attention backward's delta (octmae_linear_dgrad_delta); A/B: bench.py --set ops.DGRAD_DELTA=0


def linear_dgrad_delta(dy: torch.Tensor, w: torch.Tensor, o: torch.Tensor, H: int, HD: int):
    """dx[M,K] = dy[M,N] @ w[N,K] (bf16) and delta[M,H] (fp32) = -sum over each head's HD columns of dx * o -- the attention
    backward's per-query constant, from the epilogue of the GEMM that produces dO (no separate pass over O and dO).
    Returns (dx, None) when the shape does not take the fused kernel (the caller then lets octmae_attn_bwd_fused compute it)."""
    M, N = dy.shape
    K = w.shape[1]
    if not (DGRAD_DELTA and not FORCE_SMALL_TILE and M >= 256 and K >= 256 and N % 64 == 0 and K % 8 == 0 and H * HD == K and H <= 64
            and HD in (32, 64)):
        return linear_dgrad(dy, w), None
    dx = torch.empty((M, K), dtype=BF16, device=dy.device)
    delta = torch.empty((M, H), dtype=F32, device=dy.device)
    st = _stream()
    args = (w.data_ptr(), dy.data_ptr(), dx.data_ptr(), o.data_ptr(), delta.data_ptr(), M, N, K, w.stride(0), dy.stride(0), K, o.stride(0),
            H, HD, _variant_bits(), *_split_ws_for(st), st)
    rc = [0]

    def run():
        rc[0] = load().octmae_linear_dgrad_delta(*args)

    _launch("gemm_dgrad_epi6", 2.0 * K * M * N, 2.0 * (K * N + M * N) + 2.0 * K * M + 2.0 * K * M, run)
    if rc[0] == -2:           # the library's own applicability test said no (e.g. an operand beyond a 32-bit buffer range): plain dgrad
        return linear_dgrad(dy, w), None
    if rc[0] != 0:
        raise RuntimeError(f"octmae_linear_dgrad_delta failed (rc={rc[0]})")
    return dx, delta


def _splitk_for(n_out_tiles: int, ktiles: int, target_blocks: int) -> int:
    # as many k-slices as keep tiles x slices within ONE round of workgroups over the chip (a second, partly filled
    # round costs more than the slightly lower fill), and >= 8 k-tiles (512 token rows) per slice
    s = max(1, target_blocks // n_out_tiles)
    s = max(1, min(s, ktiles // 8 if ktiles >= 8 else 1))
    if s > 1 and target_blocks == 256:
        # ... unless the split costs more than it saves (round 6, small batches): every workgroup ends with 256 KiB of fp32 atomics,
        # which the L2s retire at ~1.2 TB/s in all -- 0.22 us per tile and slice -- against ~1.4 us per k-tile of the main loop
        # (fitted on graph-replayed launches, tools/gemm_small_fit.py / profiles/r06_gemm_small_fit.txt).  One volume per step
        # (21 k-tiles): the fc1 + fc2 pair as 128 tiles x 2 slices was 68 us, unsplit (one atomic add per element) it is 54 us.
        # From 32 volumes per micro-batch on the minimum is the old choice (the most slices that fit one round).
        def cost(k):
            return 1.4 * -(-ktiles // k) + 0.22 * n_out_tiles * k
        s = min(range(1, s + 1), key=cost)
    return s


def linear_wgrad_accum(dy: torch.Tensor, x: torch.Tensor, gw: torch.Tensor, gb: Optional[torch.Tensor] = None):
    """gw[N,K] (f32) += dy[M,N].T @ x[M,K]; with ``gb`` (f32 [N]) also gb += dy.sum(0) -- the bias gradient, taken from the
    dY tiles the weight-gradient kernel stages anyway (one extra MFMA per k-step in the first column of tiles) instead of a
    separate pass over dY."""
    M, N = dy.shape
    K = x.shape[1]
    big = N >= 256 and K >= 256 and not FORCE_SMALL_TILE          # mirrors the tile choice in octmae_gemm_bf16
    t = 256 if big else 128
    tiles = ((N + t - 1) // t) * ((K + t - 1) // t)
    ktiles = (M + 63) // 64
    _gemm(dy, x, gw, N, K, M, dy.stride(0), x.stride(0), gw.stride(0), 1, 1, EPI_ACCUM, C2=gb,
          splitk=_splitk_for(tiles, ktiles, 256 if big else 1024))


# fc1 + fc2 and qkv + proj weight gradients of a Block as one launch each (octmae_wgrad_accum_pair); OCTMAE_WGRAD_PAIR=0: same-box A/B
WGRAD_PAIR = os.environ.get("OCTMAE_WGRAD_PAIR", "1") != "0"


def linear_wgrad_accum_pair(first, second):
    """Two linear_wgrad_accum calls over the same token rows -- (dy, x, gw, gb) each -- as ONE launch: the tiles of both outputs
    share the split over the rows, so there are half as many fp32-atomic epilogues and the k-loops are twice as long (measured on
    the combined shape, profiles/r04_wgrad_pair.txt: -12 ... -15 % at 32 volumes, -2 ... -7 % at 128).  Falls back to two launches when
    the library says the pair does not apply (-2) or WGRAD_PAIR is off."""
    (dy0, x0, gw0, gb0), (dy1, x1, gw1, gb1) = first, second
    M = dy0.shape[0]
    ok = (WGRAD_PAIR and not FORCE_SMALL_TILE and dy1.shape[0] == M and x0.shape[0] == M and x1.shape[0] == M
          and min(dy0.shape[1], x0.shape[1], dy1.shape[1], x1.shape[1]) >= 256
          # the library's own applicability test (each operand inside one 32-bit buffer descriptor), so that -2 is not met here
          and 2 * M * max(dy0.stride(0), x0.stride(0), dy1.stride(0), x1.stride(0)) < 0xFFF00000)
    if ok:
        tiles = sum(((dy.shape[1] + 255) // 256) * ((x.shape[1] + 255) // 256) for dy, x in ((dy0, x0), (dy1, x1)))
        rc = [0]
        args = []
        for dy, x, gw, gb in (first, second):
            for t_, dt_, nm_ in ((dy, BF16, "dy"), (x, BF16, "x"), (gw, F32, "gw")):       # rows may be slices of wider tensors
                if not (t_.is_cuda and t_.dtype == dt_ and t_.dim() == 2 and t_.stride(1) == 1):
                    raise RuntimeError(f"linear_wgrad_accum_pair: {nm_} must be a GPU {dt_} matrix with contiguous rows")
            args += [dy.data_ptr(), x.data_ptr(), gw.data_ptr(), _p(gb), dy.shape[1], x.shape[1], dy.stride(0), x.stride(0), gw.stride(0)]
        args += [M, _splitk_for(tiles, (M + 63) // 64, 256), _stream()]

        def run():
            rc[0] = load().octmae_wgrad_accum_pair(*args)

        fl = 2.0 * M * (dy0.shape[1] * x0.shape[1] + dy1.shape[1] * x1.shape[1])
        nb = 2.0 * M * (dy0.shape[1] + x0.shape[1] + dy1.shape[1] + x1.shape[1]) + 4.0 * (gw0.numel() + gw1.numel())
        _launch("gemm_wgrad_epi5", fl, nb, run)
        if rc[0] == 0:
            return
        if rc[0] != -2:
            raise RuntimeError(f"octmae_wgrad_accum_pair failed (rc={rc[0]})")
    linear_wgrad_accum(dy0, x0, gw0, gb0)
    linear_wgrad_accum(dy1, x1, gw1, gb1)


def layernorm_fwd(x: torch.Tensor, gamma, beta, eps: float):
    M, D = x.shape
    y = torch.empty((M, D), dtype=BF16, device=x.device)
    mean = torch.empty((M,), dtype=F32, device=x.device)
    rstd = torch.empty((M,), dtype=F32, device=x.device)
    # algorithmic HBM bytes: 4 B read + 2 B written per element (the row statistics are 8 B per row)
    _launch(f"ln_fwd_d{D}", 0.0, 6.0 * M * D + 8.0 * M,
            lambda: call("octmae_layernorm_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                         rstd.data_ptr(), M, D, float(eps), _stream()))
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, dres=None, want_bf16=False, dxsum=None):
    M, D = x.shape
    dx = torch.empty((M, D), dtype=F32, device=x.device)
    dxb = torch.empty((M, D), dtype=BF16, device=x.device) if want_bf16 else None
    ws = torch.empty((load().octmae_layernorm_bwd_ws_floats(M, D),), dtype=F32, device=x.device)
    # algorithmic HBM bytes: dy bf16 + x fp32 read, dx fp32 written, + the incoming residual-stream gradient (fp32) and the bf16
    # copy of dx when the fused forms are used
    nbytes = (2.0 + 4.0 + 4.0 + (4.0 if dres is not None else 0.0) + (2.0 if want_bf16 else 0.0)) * M * D + 8.0 * M
    _launch(f"ln_bwd_d{D}", 0.0, nbytes,
            lambda: call("octmae_layernorm_bwd", dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), _p(dres),
                         dx.data_ptr(), _p(dxb), _p(dgamma), _p(dbeta), _p(dxsum), ws.data_ptr(), M, D, _stream()))
    return dx, dxb


# optimistic (no running max) forward first, safe kernel as the device-side fallback.  Not in the half build: the un-normalised
# P = exp2(s) of that kernel is an MFMA operand, and half ends at 65 504 = e^11.09 where bfloat16 has fp32's range (the online-max
# kernel keeps P <= 2^8)
ATTN_OPTIMISTIC = not LP_IS_F16


def attn_fwd(qkv: torch.Tensor, B: int, N: int, H: int, HD: int, scale: float, optimistic: Optional[bool] = None):
    o = torch.empty((B * N, H * HD), dtype=BF16, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=F32, device=qkv.device)
    opt = ATTN_OPTIMISTIC if optimistic is None else optimistic
    flag = torch.empty((1,), dtype=torch.int32, device=qkv.device) if opt else None
    _launch(f"attn_fwd_hd{HD}", 4.0 * B * H * N * N * HD, 2.0 * 4 * B * N * H * HD,
            lambda: call("octmae_attn_fwd", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), _p(flag), B, N, H, HD, float(scale), _stream()))
    return o, lse


# Which backward runs per head_dim: the single-pass kernel (csrc/attn_bwd.hip) or the dQ + dK/dV kernel pair (csrc/attn.hip).
# Measured on MI355X (same process, interleaved; docs/HISTORY.md section 4 "Attention"): head_dim 32 (decoder, N = 5121) fused 11 % faster;
# head_dim 64 fused 4 % faster at N = 1281 (encoder) and 2 % at N = 5121 (fine-tune) since its LDS-DMA requests go out between
# the sub-tiles and its sub-tile addressing is hand-placed (before: the pair 10-13 % faster).
ATTN_BWD_FUSED = {32: True, 64: True}


def set_option(key: str, value: int) -> int:
    """octmae_set_option: kernel-selection switch for A/B measurements and tests ("attn_bwd_hd32_form" / "attn_bwd_hd64_form":
    1 = one wave per SIMD (default), 0 = the two-waves-per-SIMD kernel; "gemm_small": 1 (default) = forward / dgrad launches that
    would leave most CUs idle take the small-launch kernel, 0 = never; "wgrad_stagger": length step of the k slices of a
    many-way split-K weight gradient, 0 = equal slices).  Returns the previous value."""
    prev = load().octmae_set_option(key.encode(), int(value))
    if prev < 0:
        raise RuntimeError(f"octmae_set_option: unknown key {key!r}")
    return prev


_N_CU = {}
# fused backward only if its B * H workgroups fill this share of their rounds of CUs (0: always; OCTMAE_ATTN_BWD_MIN_FILL for A/B runs)
ATTN_BWD_FUSED_MIN_FILL = float(os.environ.get("OCTMAE_ATTN_BWD_MIN_FILL", "0.74"))


def attn_bwd_use_fused(B: int, H: int, HD: int, device=None) -> bool:
    """Which backward a call takes when the caller does not say.  The fused single-pass kernels run ONE workgroup per (batch, head) and CU:
    B * H workgroups, in rounds of the CU count.  Below three quarters of a round -- or of the last of a few rounds -- the idle CUs cost more
    than the dQ + dK/dV pair's two extra matrix products (it parallelises over query and key tiles as well): measured through the whole
    step on MI355X (H = 16; profiles/r04_small_batch.txt), 1 volume 42.2 vs 23.6 volumes/s for the pair, 2: 67.8 vs 43.0, 4: 98.9 vs 74.1,
    8: 126.6 vs 114.8, 12 (192 workgroups): equal, 16 (one full round): 148.3 vs 139.6 for the fused form, 24 (1.5 rounds): equal."""
    if not ATTN_BWD_FUSED[HD]:
        return False
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    ncu = _N_CU.get(idx)
    if ncu is None:
        ncu = _N_CU[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count)
    wgs = B * H
    rounds = -(-wgs // ncu)
    return wgs >= ATTN_BWD_FUSED_MIN_FILL * rounds * ncu


def attn_bwd(qkv, o, dout, lse, B, N, H, HD, scale, fused: Optional[bool] = None, delta: Optional[torch.Tensor] = None):
    """Gradient of attn_fwd w.r.t. the packed qkv.  Algorithmic work ("x3" in total): 4 matrix products =
    8 B H N^2 HD flop (dP, dV, dK, dQ); the recomputation of S is not counted.  ``delta`` (fp32 [B * N, H], from
    linear_dgrad_delta): the fused form then skips its pass over O and dO."""
    dqkv = torch.empty_like(qkv)
    st = _stream()
    unit = 2.0 * B * H * N * N * HD
    if attn_bwd_use_fused(B, H, HD, qkv.device) if fused is None else fused:
        kib = load().octmae_attn_bwd_fused_ws_kib(B, N, H, HD)
        if kib < 0:
            raise RuntimeError("octmae_attn_bwd_fused_ws_kib: unsupported shape")
        ws = torch.empty((kib * 256,), dtype=F32, device=qkv.device)     # dQ fp32 + padded row constants (contents irrelevant)
        # executes 5 products (S once); plus an fp32 read-modify-write of dQ per key block
        if delta is not None:
            _chk(delta, F32, "delta")
            _launch(f"attn_bwd_fused_hd{HD}", 4 * unit, 2.0 * 6 * B * N * H * HD,
                    lambda: call("octmae_attn_bwd_fused_delta", qkv.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                 ws.data_ptr(), dqkv.data_ptr(), B, N, H, HD, float(scale), st), exec_flops=5 * unit)
            return dqkv
        _launch(f"attn_bwd_fused_hd{HD}", 4 * unit, 2.0 * 6 * B * N * H * HD,
                lambda: call("octmae_attn_bwd_fused", qkv.data_ptr(), o.data_ptr(), dout.data_ptr(), lse.data_ptr(), ws.data_ptr(),
                             dqkv.data_ptr(), B, N, H, HD, float(scale), st), exec_flops=5 * unit)
        return dqkv
    rowc = torch.empty((2, B, H, N), dtype=F32, device=qkv.device)      # -lse*log2e | -rowsum(dO * O)
    # The two-kernel form executes 7 products: dq 3 (S, dP, dQ), dkv 4 (S, dP, dV, dK); the algorithmic 4 are booked
    # 1.6 / 2.4 in proportion.  The dQ kernel also produces the row constants (rowc) that the dK/dV kernel reads.
    _launch(f"attn_bwd_dq_hd{HD}", 1.6 * unit, 2.0 * 6 * B * N * H * HD,
            lambda: call("octmae_attn_bwd_dq_rowconst", qkv.data_ptr(), o.data_ptr(), dout.data_ptr(), lse.data_ptr(), rowc.data_ptr(),
                         dqkv.data_ptr(), B, N, H, HD, float(scale), st), exec_flops=3 * unit)
    _launch(f"attn_bwd_dkv_hd{HD}", 2.4 * unit, 2.0 * 6 * B * N * H * HD,
            lambda: call("octmae_attn_bwd_dkv", qkv.data_ptr(), dout.data_ptr(), rowc.data_ptr(), dqkv.data_ptr(), B, N, H, HD,
                         float(scale), st), exec_flops=4 * unit)
    return dqkv


def random_masking_ids(noise: torch.Tensor, len_keep: int, want_shuffle: bool = False):
    """Index part of random_masking (models_mae_joint_res_flash_attn.py:349-369): stable argsort on the GPU.
    Returns mask f32 [B,L], ids_restore i64 [B,L], ids_keep i64 [B,len_keep] (, ids_shuffle)."""
    _chk(noise, F32, "noise")
    Bn, L = noise.shape
    dev = noise.device
    ids_restore = torch.empty((Bn, L), dtype=torch.int64, device=dev)
    ids_keep = torch.empty((Bn, len_keep), dtype=torch.int64, device=dev)
    ids_shuffle = torch.empty((Bn, L), dtype=torch.int64, device=dev) if want_shuffle else None
    mask = torch.empty((Bn, L), dtype=F32, device=dev)
    call("octmae_random_masking_ids", noise.data_ptr(), ids_restore.data_ptr(), ids_keep.data_ptr() if len_keep > 0 else None, _p(ids_shuffle),
         mask.data_ptr(), Bn, L, len_keep, _stream())
    if want_shuffle:
        return mask, ids_restore, ids_keep, ids_shuffle
    return mask, ids_restore, ids_keep


def patch_gather(imgs: torch.Tensor, ids_keep: Optional[torch.Tensor], tp: int, p: int, nkeep: int) -> torch.Tensor:
    _chk(imgs, F32, "imgs")
    Bn, Cc, T, Hh, Ww = imgs.shape
    out = torch.empty((Bn * nkeep, Cc * tp * p * p), dtype=BF16, device=imgs.device)
    call("octmae_patch_gather", imgs.data_ptr(), _p(ids_keep), 1, out.data_ptr(), Bn, Cc, T, Hh, Ww, tp, p, nkeep, _stream())
    return out


# ------------------------------------------------------------------------------------------------
# autograd Functions
# ------------------------------------------------------------------------------------------------
def _as2d_bf16(g: torch.Tensor, cols: int) -> torch.Tensor:
    return cast_bf16(g.reshape(-1, cols).contiguous())


class LayerNormFn(torch.autograd.Function):
    """y(bf16) = LayerNorm(x f32) -- nn.LayerNorm(eps=1e-6) at video_vit.py:181-184 / models_mae…:489,:592."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        shp = x.shape
        x2 = _chk(x.reshape(-1, shp[-1]), F32, "layernorm input")
        y, mean, rstd = layernorm_fwd(x2, gamma, beta, eps)
        ctx.save_for_backward(x2, mean, rstd, gamma, beta)
        ctx.shp = shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, gamma, beta = ctx.saved_tensors
        D = x2.shape[1]
        dyb = _as2d_bf16(dy, D)
        colsum = torch.zeros(D, dtype=F32, device=x2.device)
        dx, dxb = layernorm_bwd(dyb, x2, mean, rstd, gamma, grad_buf(gamma), grad_buf(beta), want_bf16=True, dxsum=colsum)
        notify_grad_ready((gamma, beta))
        dx = dx.view(ctx.shp)
        _sidecar_put(dx, dxb.view(ctx.shp), colsum)      # the producing Block's backward takes these instead of redoing them
        return dx, None, None, None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b on bf16 operands.  ``w_lp`` / ``b32`` are the (possibly fused q|k|v) bf16 weight and fp32 bias
    views; ``params`` = (weight params..., bias params...) whose .grad buffers are adjacent views of the arena."""

    @staticmethod
    def forward(ctx, x, w_lp, b32, gw, gb, out_f32, *params):
        shp = x.shape
        x2 = cast_bf16(x.reshape(-1, shp[-1]))
        y = linear_fwd(x2, w_lp, b32, "f32" if out_f32 else "bf16")
        ctx.save_for_backward(x2, w_lp)
        ctx.gw, ctx.gb, ctx.params, ctx.shp = gw, gb, params, shp
        return y.view(*shp[:-1], w_lp.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w_lp = ctx.saved_tensors
        N = w_lp.shape[0]
        dyb = _as2d_bf16(dy, N)
        gw, gb = ctx.gw(), ctx.gb()
        linear_wgrad_accum(dyb, x2, gw, gb)
        notify_grad_ready(ctx.params)
        dx = linear_dgrad(dyb, w_lp).view(ctx.shp) if ctx.needs_input_grad[0] else None
        return (dx, None, None, None, None, None) + (None,) * len(ctx.params)


class PatchEmbedFn(torch.autograd.Function):
    """PatchEmbed (Conv3d k=s=(tp,p,p) as a GEMM, video_vit.py:74-83) restricted to the kept tokens."""

    @staticmethod
    def forward(ctx, imgs, ids_keep, w_lp, b32, gw, gb, tp, p, nkeep, weight, bias):
        patches = patch_gather(imgs, ids_keep, tp, p, nkeep)
        tok = linear_fwd(patches, w_lp, b32, "bf16")
        ctx.save_for_backward(patches)
        ctx.gw, ctx.gb, ctx.params = gw, gb, (weight, bias)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        (patches,) = ctx.saved_tensors
        dyb = _as2d_bf16(dtok, dtok.shape[-1])
        gw = ctx.gw()
        linear_wgrad_accum(dyb, patches, gw.view(gw.shape[0], -1), ctx.gb())
        notify_grad_ready(ctx.params)
        return (None,) * 11


class AttentionFn(torch.autograd.Function):
    """Attention (video_vit.py:112-138): fused q|k|v projection, flash attention core, output projection, and
    (inside a Block, video_vit.py:182) the residual add fused into the projection epilogue."""

    @staticmethod
    def forward(ctx, y, res, wqkv_lp, bqkv32, wproj_lp, bproj32, grads, H, *params):
        shp = y.shape
        Bn, N, Cc = shp
        HD = Cc // H
        scale = HD ** -0.5
        y2 = cast_bf16(y.reshape(-1, Cc))
        qkv = linear_fwd(y2, wqkv_lp, bqkv32, "bf16")
        o, lse = attn_fwd(qkv, Bn, N, H, HD, scale)
        if res is not None:
            out = linear_fwd(o, wproj_lp, bproj32, "resid", res=_chk(res.reshape(-1, Cc), F32, "residual"))
        else:
            out = linear_fwd(o, wproj_lp, bproj32, "bf16")
        ctx.save_for_backward(y2, qkv, o, lse, wqkv_lp, wproj_lp)
        ctx.meta = (Bn, N, H, HD, scale, res is not None)
        ctx.grads, ctx.params, ctx.shp = grads, params, shp
        return out.view(shp)

    @staticmethod
    def backward(ctx, dout):
        y2, qkv, o, lse, wqkv_lp, wproj_lp = ctx.saved_tensors
        Bn, N, H, HD, scale, has_res = ctx.meta
        Cc = H * HD
        gwqkv, gbqkv, gwproj, gbproj = ctx.grads()
        d2 = dout.reshape(-1, Cc)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        dob = cast_bf16(d2)
        if gbproj is not None:
            colsum_accum(d2 if d2.dtype in (F32, BF16) else dob, gbproj)
        linear_wgrad_accum(dob, o, gwproj)
        fused_bwd = attn_bwd_use_fused(Bn, H, HD, qkv.device)
        do, delta = linear_dgrad_delta(dob, wproj_lp, o, H, HD) if fused_bwd else (linear_dgrad(dob, wproj_lp), None)
        dqkv = attn_bwd(qkv, o, do, lse, Bn, N, H, HD, scale, fused=fused_bwd, delta=delta)
        linear_wgrad_accum(dqkv, y2, gwqkv, gbqkv)
        notify_grad_ready(ctx.params)
        dy = linear_dgrad(dqkv, wqkv_lp).view(ctx.shp)
        dres = dout if has_res else None
        return (dy, dres, None, None, None, None, None, None) + (None,) * len(ctx.params)


class MlpFn(torch.autograd.Function):
    """timm Mlp (fc1 -> exact GELU -> fc2) with the Block's residual add (video_vit.py:183) fused into fc2."""

    @staticmethod
    def forward(ctx, y, res, w1_lp, b1_32, w2_lp, b2_32, grads, *params):
        shp = y.shape
        Cc = shp[-1]
        y2 = cast_bf16(y.reshape(-1, Cc))
        ctx.pre_is_dgelu = GELU_PRIME_FWD
        pre, act = linear_fwd(y2, w1_lp, b1_32, "gelu", store_dgelu=ctx.pre_is_dgelu)
        if res is not None:
            out = linear_fwd(act, w2_lp, b2_32, "resid", res=_chk(res.reshape(-1, Cc), F32, "residual"))
        else:
            out = linear_fwd(act, w2_lp, b2_32, "bf16")
        ctx.save_for_backward(y2, pre, act, w1_lp, w2_lp)
        ctx.grads, ctx.params, ctx.shp, ctx.has_res = grads, params, shp, res is not None
        return out.view(shp)

    @staticmethod
    def backward(ctx, dout):
        y2, pre, act, w1_lp, w2_lp = ctx.saved_tensors
        Cc = ctx.shp[-1]
        gw1, gb1, gw2, gb2 = ctx.grads()
        d2 = dout.reshape(-1, Cc)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        dob = cast_bf16(d2)
        if gb2 is not None:
            colsum_accum(d2, gb2)
        linear_wgrad_accum(dob, act, gw2)
        dpre = linear_dgrad(dob, w2_lp, pre=pre, colsum=gb1, pre_is_dgelu=ctx.pre_is_dgelu)
        linear_wgrad_accum(dpre, y2, gw1)
        notify_grad_ready(ctx.params)
        dy = linear_dgrad(dpre, w1_lp).view(ctx.shp)
        return (dy, dout if ctx.has_res else None, None, None, None, None, None) + (None,) * len(ctx.params)


# Gradients handed from one fused op to the one upstream of it, keyed by the fp32 gradient tensor they describe:
#   data_ptr -> (weakref to that tensor, bf16 copy, column sums)
# The consumer (the Block whose OUTPUT gradient this is) uses the pair only if the tensor autograd hands it is that very
# tensor; if autograd summed other contributions into a new tensor the entry is ignored (and the work redone locally).
import weakref

_grad_sidecar = {}


def _sidecar_purge():
    # entries whose fp32 gradient is gone can never be claimed: the first Block of a stack hands its gradient to a consumer that is
    # not a Block (sequence assembly), and its bf16 copy -- 0.34 / 0.67 GB at 128 volumes -- would otherwise stay pinned here
    # (the allocator hands out the same addresses step after step, so in practice two to four stale entries: 1.9 of the 252 GiB
    # a 128-volume micro-batch allocated)
    for k in [k for k, ent in _grad_sidecar.items() if ent[0]() is None]:
        del _grad_sidecar[k]


def _sidecar_put(dx: torch.Tensor, dxb: torch.Tensor, colsum: torch.Tensor):
    _sidecar_purge()
    if len(_grad_sidecar) > 64:
        _grad_sidecar.clear()
    _grad_sidecar[dx.data_ptr()] = (weakref.ref(dx), dxb, colsum)


def _sidecar_take(d: torch.Tensor):
    ent = _grad_sidecar.pop(d.data_ptr(), None)
    if ent is None:
        return None
    ref, dxb, colsum = ent
    if ref() is not d or dxb.shape != d.shape:
        return None
    return dxb, colsum


class BlockFn(torch.autograd.Function):
    """One pre-norm transformer Block (video_vit.py:181-184) as a single autograd node.

    forward : x -> LN1 -> Wqkv -> attention -> proj (+x) -> LN2 -> fc1 + GELU -> fc2 (+x2)          (7 launches)
    backward: the same chain in reverse with everything that only existed to please per-op autograd fused away:
              both residual gradient adds and the bf16 gradient copy happen inside the LayerNorm backward kernels, which also
              produce the proj / fc2 bias gradients (column sums of what they write); GELU' is the fc2-dgrad epilogue."""

    @staticmethod
    def forward(ctx, x, H, eps1, eps2, lp, grads, s1, s2, final_residual, *params):
        """s1 / s2: None, or fp32 [B] per-sample stochastic-depth factors (0 or 1/keep_prob) of the attention / MLP branch.
        final_residual=False returns the MLP branch alone (flash_compat: what flash-attn's prenorm Block hands back as
        ``hidden_states`` and the reference's flash models feed to the final norm, models_mae_joint_res_flash_attn.py:480-489)."""
        shp = x.shape
        C = shp[-1]
        Bn, N = shp[0], shp[1]
        HD = C // H
        scale = HD ** -0.5
        if _grad_sidecar:
            _sidecar_purge()
        wqkv, bqkv, wproj, bproj, w1, b1, w2, b2 = lp
        g1, be1, g2, be2 = params[0], params[1], params[2], params[3]
        x2d = _chk(x.reshape(-1, C), F32, "block input")
        y1, mean1, rstd1 = layernorm_fwd(x2d, g1, be1, eps1)
        qkv = linear_fwd(y1, wqkv, bqkv, "bf16")
        o, lse = attn_fwd(qkv, Bn, N, H, HD, scale)
        x2 = linear_fwd(o, wproj, bproj, "resid", res=x2d, rowscale=s1, rows_per_scale=N)
        y2, mean2, rstd2 = layernorm_fwd(x2, g2, be2, eps2)
        ctx.pre_is_dgelu = GELU_PRIME_FWD
        pre, act = linear_fwd(y2, w1, b1, "gelu", store_dgelu=ctx.pre_is_dgelu)
        if final_residual:
            x3 = linear_fwd(act, w2, b2, "resid", res=x2, rowscale=s2, rows_per_scale=N)
        else:
            x3, s2 = linear_fwd(act, w2, b2, "f32"), None
        ctx.final_residual = final_residual
        ctx.save_for_backward(x2d, mean1, rstd1, y1, qkv, o, lse, x2, mean2, rstd2, y2, pre, act, wqkv, wproj, w1, w2, g1, g2)
        ctx.scales = (s1, s2)
        ctx.meta = (Bn, N, H, HD, scale, shp)
        ctx.grads, ctx.params = grads, params
        if not final_residual:      # (MLP branch, stream before it): the pair flash-attn's prenorm Block returns
            return x3.view(shp), x2.view(shp)
        return x3.view(shp)

    @staticmethod
    def backward(ctx, dx3, dx2_in=None):
        x2d, mean1, rstd1, y1, qkv, o, lse, x2, mean2, rstd2, y2, pre, act, wqkv, wproj, w1, w2, g1, g2 = ctx.saved_tensors
        Bn, N, H, HD, scale, shp = ctx.meta
        C = H * HD
        (gg1, gb1n, gg2, gb2n, gwqkv, gbqkv, gwproj, gbproj, gw1, gb1, gw2, gb2) = ctx.grads()
        if dx3.dtype != F32 or not dx3.is_contiguous():
            dx3 = dx3.contiguous().float()
        s1, s2 = ctx.scales
        side = _sidecar_take(dx3)
        d3 = dx3.view(-1, C)
        if s2 is not None:                       # stochastic depth: the MLP branch sees s2[b] * dx3 (the hand-off is unscaled)
            d3b = cast_bf16_rowscale(d3, s2, N)
            if gb2 is not None:
                colsum_accum(d3b, gb2)
        elif side is not None:                   # produced by the LayerNorm backward of the next Block
            d3b, colsum3 = side
            d3b = d3b.view(-1, C)
            if gb2 is not None:
                gb2.add_(colsum3)
        else:
            d3b = cast_bf16(d3)
            if gb2 is not None:
                colsum_accum(d3, gb2)
        # ---- MLP
        dpre = linear_dgrad(d3b, w2, pre=pre, colsum=gb1, pre_is_dgelu=ctx.pre_is_dgelu)          # GELU' and fc1's bias gradient in the epilogue
        linear_wgrad_accum_pair((d3b, act, gw2, None), (dpre, y2, gw1, None))
        dy2 = linear_dgrad(dpre, w1)
        # ---- LN2 backward + residual add + bf16 copy + proj bias gradient
        if ctx.final_residual:
            dres2 = d3
        else:                       # gradient arriving through the returned stream (None when only the branch was used)
            dres2 = None if dx2_in is None else dx2_in.contiguous().float().view(-1, C)
        if s1 is None:
            dx2, dx2b = layernorm_bwd(dy2, x2, mean2, rstd2, g2, gg2, gb2n, dres=dres2, want_bf16=True, dxsum=gbproj)
        else:
            dx2, _ = layernorm_bwd(dy2, x2, mean2, rstd2, g2, gg2, gb2n, dres=dres2, want_bf16=False, dxsum=None)
            dx2b = cast_bf16_rowscale(dx2, s1, N)
            if gbproj is not None:
                colsum_accum(dx2b, gbproj)
        # ---- attention
        fused_bwd = attn_bwd_use_fused(Bn, H, HD, qkv.device)
        do, delta = linear_dgrad_delta(dx2b, wproj, o, H, HD) if fused_bwd else (linear_dgrad(dx2b, wproj), None)
        dqkv = attn_bwd(qkv, o, do, lse, Bn, N, H, HD, scale, fused=fused_bwd, delta=delta)
        # the qkv bias gradient rides in the weight-gradient GEMM (column sums of its dY operand); fusing it into the attention
        # backward kernels had been measured and dropped (+10..25 % on their main loops for a 2 % pass)
        linear_wgrad_accum_pair((dx2b, o, gwproj, None), (dqkv, y1, gwqkv, gbqkv))
        dy1 = linear_dgrad(dqkv, wqkv)
        # ---- LN1 backward + residual add; its bf16 copy / column sums are what the previous Block's backward needs
        colsum = torch.zeros(C, dtype=F32, device=dx3.device)
        dx, dxb = layernorm_bwd(dy1, x2d, mean1, rstd1, g1, gg1, gb1n, dres=dx2, want_bf16=True, dxsum=colsum)
        notify_grad_ready(ctx.params)
        dx = dx.view(shp)
        _sidecar_put(dx, dxb.view(shp), colsum)
        return (dx, None, None, None, None, None, None, None, None) + (None,) * len(ctx.params)


class EncAssembleFn(torch.autograd.Function):
    """cls concat + gathered positional embedding (models_mae_joint_res_flash_attn.py:409-478).
    ``ids_restore`` (the masking's inverse permutation, [B, L]) lets the backward build the positional table's gradient as a
    deterministic gather (octmae_scatter_add_rows) instead of ATen's atomic index_add_; without it (the ViT models, where every
    token is kept in order) the ATen path is used."""

    @staticmethod
    def forward(ctx, tok, pos, cls, pos_cls, ids_keep, ids_restore=None):
        Bn, nkeep = ids_keep.shape
        D = tok.shape[-1]
        _chk(tok, BF16, "tokens"); _chk(ids_keep, torch.int64, "ids_keep")
        pos2 = _chk(pos.reshape(-1, D), F32, "pos table")
        x = torch.empty((Bn, nkeep + 1, D), dtype=F32, device=tok.device)
        call("octmae_enc_assemble", tok.data_ptr(), pos2.data_ptr(), cls.data_ptr(), pos_cls.data_ptr(), ids_keep.data_ptr(),
             x.data_ptr(), Bn, nkeep, D, _stream())
        if ids_restore is not None and (ids_restore.shape[0] != Bn or ids_restore.shape[1] != pos2.shape[0] or Bn > 1024 or D % 4):
            ids_restore = None
        if ids_restore is not None:
            # contract of octmae_scatter_add_rows: ids_restore is the inverse of the shuffle whose first nkeep entries are ids_keep
            # (source row of table row l = its rank ids_restore[b, l], when < nkeep); type / layout are checked here, the
            # permutation property by tests/test_gpu_kernels.py (and by OCTMAE_CHECK_IDS=1 on every call)
            _chk(ids_restore, torch.int64, "ids_restore")
            if _CHECK_IDS and not torch.equal(ids_restore.gather(1, ids_keep),
                                              torch.arange(nkeep, device=tok.device).expand(Bn, -1)):
                raise RuntimeError("EncAssembleFn: ids_restore is not the inverse permutation of ids_keep")
        ctx.save_for_backward(ids_keep, ids_restore if ids_restore is not None else ids_keep.new_empty(0))
        ctx.has_restore = ids_restore is not None
        ctx.pos_shape, ctx.cls_shape, ctx.pc_shape = pos.shape, cls.shape, pos_cls.shape
        return x

    @staticmethod
    def backward(ctx, dx):
        ids_keep, ids_restore = ctx.saved_tensors
        dx = dx.contiguous()
        Bn, n1, D = dx.shape
        nkeep = n1 - 1
        dtok = torch.empty((Bn * nkeep, D), dtype=BF16, device=dx.device)
        call("octmae_gather_rows_cast", dx.data_ptr(), None, dtok.data_ptr(), Bn, nkeep, n1, D, _stream())
        if ctx.has_restore:
            L = ids_restore.shape[1]
            dpos = torch.empty(ctx.pos_shape, dtype=F32, device=dx.device)
            call("octmae_scatter_add_rows", dx.data_ptr(), ids_restore.data_ptr(), dpos.data_ptr(), Bn, nkeep, L, D, n1, 1, 0, _stream())
        else:
            dpos = torch.zeros(ctx.pos_shape, dtype=F32, device=dx.device)
            dpos.view(-1, D).index_add_(0, ids_keep.reshape(-1), dx[:, 1:, :].reshape(-1, D))
        dc = dx[:, 0, :].sum(0)
        return dtok, dpos, dc.view(ctx.cls_shape), dc.view(ctx.pc_shape).clone(), None, None


class DecAssembleFn(torch.autograd.Function):
    """mask-token append, un-shuffle, cls concat, positional embedding (models_mae_joint_res_flash_attn.py:515-573).
    ``dcls is None``: the 2-D MAE form (OCTCube/models_mae.py:175-178) -- emb carries 1 + nkeep rows per sample and its
    row 0 is the cls row."""

    @staticmethod
    def forward(ctx, emb, mask_token, dpos, dcls, dpos_cls, ids_restore, ids_keep):
        Bn, L = ids_restore.shape
        nkeep = ids_keep.shape[1]
        D = emb.shape[-1]
        _chk(emb, BF16, "decoder tokens"); _chk(ids_restore, torch.int64, "ids_restore"); _chk(ids_keep, torch.int64, "ids_keep")
        if _CHECK_IDS and not torch.equal(ids_restore.gather(1, ids_keep), torch.arange(nkeep, device=emb.device).expand(Bn, -1)):
            raise RuntimeError("DecAssembleFn: ids_restore is not the inverse permutation of ids_keep")
        dpos2 = _chk(dpos.reshape(-1, D), F32, "decoder pos table")
        x = torch.empty((Bn, L + 1, D), dtype=F32, device=emb.device)
        call("octmae_dec_assemble", emb.data_ptr(), mask_token.data_ptr(), dpos2.data_ptr(), _p(dcls), dpos_cls.data_ptr(),
             ids_restore.data_ptr(), x.data_ptr(), Bn, nkeep, L, D, 1 if dcls is None else 0, _stream())
        ctx.save_for_backward(ids_restore, ids_keep)
        ctx.shapes = (mask_token.shape, dpos.shape, None if dcls is None else dcls.shape, dpos_cls.shape)
        return x

    @staticmethod
    def backward(ctx, dx):
        ids_restore, ids_keep = ctx.saved_tensors
        dx = dx.contiguous()
        Bn, L1, D = dx.shape
        nkeep = ids_keep.shape[1]
        ms, ps, cs, pcs = ctx.shapes
        if cs is None:      # cls row travels with the tokens: ids = [-1, ids_keep]  (source row = 1 + id)
            ids = torch.cat([ids_keep.new_full((Bn, 1), -1), ids_keep], dim=1).contiguous()
            demb = torch.empty((Bn * (nkeep + 1), D), dtype=BF16, device=dx.device)
            call("octmae_gather_rows_cast", dx.data_ptr(), ids.data_ptr(), demb.data_ptr(), Bn, nkeep + 1, L1, D, _stream())
        else:
            demb = torch.empty((Bn * nkeep, D), dtype=BF16, device=dx.device)
            call("octmae_gather_rows_cast", dx.data_ptr(), ids_keep.data_ptr(), demb.data_ptr(), Bn, nkeep, L1, D, _stream())
        if Bn <= 1024 and D % 4 == 0:
            # one pass over dx: positional-table gradient and, per table row, the sum over the samples that masked it
            L = L1 - 1
            ddpos = torch.empty((L, D), dtype=F32, device=dx.device)
            part = torch.empty((L, D), dtype=F32, device=dx.device)
            call("octmae_dec_assemble_bwd", dx.data_ptr(), ids_restore.data_ptr(), ddpos.data_ptr(), part.data_ptr(), Bn, nkeep, L, D,
                 _stream())
            dmask = torch.zeros(D, dtype=F32, device=dx.device)
            colsum_accum(part, dmask)
        else:
            body = dx[:, 1:, :]
            masked = (ids_restore >= nkeep).to(F32).unsqueeze(-1)
            dmask = (body * masked).sum((0, 1))
            ddpos = body.sum(0)
        dc = dx[:, 0, :].sum(0)
        return (demb, dmask.view(ms), ddpos.view(ps), None if cs is None else dc.view(cs), dc.view(pcs).clone(), None, None)


class PatchMSEFn(torch.autograd.Function):
    """Fused patchify + per-token MSE (models_mae_joint_res_flash_attn.py:289-314, :649-650).
    pred_full f32 [B, L+1, PD] (row 0 = cls) -> loss_tok f32 [B, L]."""

    @staticmethod
    def forward(ctx, pred_full, imgs, frame_idx, u_sz, p, norm_pix):
        Bn, L1, PD = pred_full.shape
        L = L1 - 1
        _, Cc, T, Hh, Ww = imgs.shape
        _chk(pred_full, F32, "pred"); _chk(imgs, F32, "imgs")
        loss_tok = torch.empty((Bn, L), dtype=F32, device=imgs.device)
        call("octmae_mse_fwd", pred_full.data_ptr(), imgs.data_ptr(), _p(frame_idx), loss_tok.data_ptr(), Bn, Cc, T, Hh, Ww, u_sz, p, L,
             int(norm_pix), _stream())
        ctx.save_for_backward(pred_full, imgs, frame_idx if frame_idx is not None else torch.empty(0, device=imgs.device))
        ctx.meta = (u_sz, p, int(norm_pix), frame_idx is not None)
        return loss_tok

    @staticmethod
    def backward(ctx, dl):
        pred_full, imgs, frame_idx = ctx.saved_tensors
        u_sz, p, norm_pix, has_fi = ctx.meta
        Bn, L1, PD = pred_full.shape
        _, Cc, T, Hh, Ww = imgs.shape
        dl = dl.contiguous().to(F32)
        coef = torch.full((1,), 2.0 / PD, dtype=F32, device=imgs.device)
        dpred = torch.empty((Bn, L1, PD), dtype=BF16, device=imgs.device)
        call("octmae_mse_bwd", pred_full.data_ptr(), imgs.data_ptr(), frame_idx.data_ptr() if has_fi else None, dl.data_ptr(),
             coef.data_ptr(), dpred.data_ptr(), Bn, Cc, T, Hh, Ww, u_sz, p, L1 - 1, norm_pix, _stream())
        return dpred, None, None, None, None, None

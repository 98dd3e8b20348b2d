"""Train-step utilities with the reference's call contracts (Pre-training/custom_util/misc.py):
NativeScalerWithGradNormCount (:308-353), get_grad_norm_ (:356-373), add_weight_decay (:678-696),
init_distributed_mode (:252-297), all_reduce_mean (:622-630), SmoothedValue / MetricLogger (:42-201).
"""
from __future__ import annotations

import datetime
import os
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist

from . import optim as _optim


# ------------------------------------------------------------------------------------------------
# distributed helpers
# ------------------------------------------------------------------------------------------------
def prefetched(data_loader, device, args=None, only=(0,)):
    """The loader the engines iterate: one batch ahead on a side stream (DevicePrefetcher) on a GPU unless ``args.prefetch_to_device``
    is False; the loader itself otherwise."""
    if torch.device(device).type == "cuda" and getattr(args, "prefetch_to_device", True):
        return DevicePrefetcher(data_loader, device, only=only)
    return data_loader


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def init_distributed_mode(args):
    """One process per GPU.  Rank discovery from the torchrun / OpenMPI / SLURM environment exactly like the
    reference; the backend is "nccl", which on ROCm IS RCCL (collectives over xGMI inside a node)."""
    if getattr(args, "no_env", False):
        pass
    elif getattr(args, "dist_on_itp", False):
        args.rank = int(os.environ["OMPI_COMM_WORLD_RANK"])
        args.world_size = int(os.environ["OMPI_COMM_WORLD_SIZE"])
        args.gpu = int(os.environ["OMPI_COMM_WORLD_LOCAL_RANK"])
        args.dist_url = "tcp://%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"])
        os.environ["LOCAL_RANK"] = str(args.gpu)
        os.environ["RANK"] = str(args.rank)
        os.environ["WORLD_SIZE"] = str(args.world_size)
    elif "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ["LOCAL_RANK"])
    elif "SLURM_PROCID" in os.environ:
        args.rank = int(os.environ["SLURM_PROCID"])
        args.gpu = args.rank % max(torch.cuda.device_count(), 1)
    else:
        print("Not using distributed mode")
        args.distributed = False
        return
    args.distributed = True
    backend = getattr(args, "dist_backend", None) or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(args.gpu)
    args.dist_backend = backend
    if not hasattr(args, "dist_url") or args.dist_url is None:
        args.dist_url = "env://"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    print(f"| distributed init (rank {args.rank}): {args.dist_url}, gpu {args.gpu}", flush=True)
    dist.init_process_group(backend=backend, init_method=args.dist_url, world_size=args.world_size, rank=args.rank,
                            timeout=datetime.timedelta(seconds=1800))
    args.comm = None
    if backend == "nccl":
        # data plane: the library's own RCCL communicator (octmae_comm_*); torch.distributed stays the control plane -- its
        # store carries the RCCL unique id, and its NCCL communicator is never created unless somebody issues a dist collective
        from . import comm as _comm
        args.comm = _comm.NativeComm.from_store(dist.distributed_c10d._get_default_store(), args.rank, args.world_size, args.gpu)
        _comm.set_default(args.comm)
        args.comm.barrier()
    else:
        dist.barrier()


def all_reduce_mean(x):
    world_size = get_world_size()
    from . import comm as _comm
    if _comm.get_default() is not None and _comm.get_default().world > 1:
        return _comm.get_default().all_reduce_scalar(float(x), _comm.AVG)
    if world_size > 1:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        x_reduce = torch.tensor(x, dtype=torch.float32, device=dev)
        dist.all_reduce(x_reduce)
        x_reduce /= world_size
        return x_reduce.item()
    return x


# ------------------------------------------------------------------------------------------------
# optimizer-side helpers
# ------------------------------------------------------------------------------------------------
def add_weight_decay(model, weight_decay=1e-5, skip_list=(), bias_wd=False):
    """No weight decay for 1-D parameters / biases / names in skip_list (same grouping rule and group order
    [no_decay, decay] as the reference)."""
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if (not bias_wd) and len(param.shape) == 1 or name.endswith(".bias") or name in skip_list:
            no_decay.append(param)
        else:
            decay.append(param)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


_norm_cache: dict = {}


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    """Global gradient 2-norm (== the reference's norm of the stack of per-tensor norms), one fused
    multi-tensor reduction instead of ~400 torch.norm launches."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    parameters = [p for p in parameters if p.grad is not None]
    if len(parameters) == 0:
        return torch.tensor(0.0)
    if float(norm_type) != 2.0:
        raise NotImplementedError("only the 2-norm is on the hot path")
    norm, _ = _optim.grad_norm_and_coef(parameters, None, _norm_cache)
    return norm


class NativeScalerWithGradNormCount:
    """``loss_scaler(loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True) -> norm``
    (custom_util/misc.py:308-353).

    The shipped compute type is bf16, which needs no loss scaling: with the default library the scale is identically 1 whatever
    ``fp32`` says.  ``dynamic_loss_scale=True`` -- the default with the half-operand build, ``OCTMAE_LIB=liboctmae_f16.so`` -- (with
    ``fp32=False``) runs the reference's fp16 machinery -- torch.cuda.amp.GradScaler as
    custom_util/misc.py:311-344 drives it -- for drop-in behaviour and checkpoint round trips (state_dict key "amp_scaler"):
    the loss is multiplied by the scale before backward; gradients are un-scaled inside the fused AdamW kernel (one device-side
    coefficient together with the clip factor, no extra pass); a non-finite gradient norm SKIPS the optimizer step and halves the
    scale (one host read of the norm per step, as GradScaler.step has); ``growth_interval`` consecutive good steps double it.
    A ``reducer`` (parallel.FlatGradReducer) -- if given -- is flushed before the norm is taken, which is where the
    data-parallel all-reduce that overlapped with backward is waited for.
    """
    state_dict_key = "amp_scaler"

    def __init__(self, fp32=False, reducer=None, dynamic_loss_scale=None, init_scale=65536.0, growth_factor=2.0,
                 backoff_factor=0.5, growth_interval=2000):
        # dynamic_loss_scale=None (default): what the reference's ``GradScaler(enabled=not fp32)`` means for the library in use --
        # on for the half-operand build (liboctmae_f16.so: gradients underflow without it), off for bfloat16 (fp32's exponent range)
        if dynamic_loss_scale is None:
            from . import ops as _ops
            dynamic_loss_scale = _ops.LP_IS_F16
        self.enabled = bool(dynamic_loss_scale) and not fp32
        self._scale = float(init_scale) if self.enabled else 1.0
        self._growth_factor, self._backoff_factor, self._growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self._growth_tracker = 0
        self.reducer = reducer
        self.last_step_skipped = False

    def get_scale(self):
        return self._scale

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True,
                 cancel_last_layer_grad=False, named_parameters=None, epoch_and_freeze_last_layer_gradient_epoch=None):
        if self.reducer is not None:
            self.reducer.begin_backward(sync=update_grad)
        (loss * self._scale if self.enabled else loss).backward(create_graph=create_graph)
        if not update_grad:
            return None
        if self.reducer is not None:
            self.reducer.finish()
        params = list(parameters) if parameters is not None else [p for g in optimizer.param_groups for p in g["params"]]
        if clip_grad is not None:
            assert parameters is not None
        if not self.enabled:
            if clip_grad is None and isinstance(optimizer, _optim.FusedAdamW) and optimizer._grad_scale is None \
                    and frozenset(id(p) for p in params if p.grad is not None) == optimizer.grads_key():
                # no coefficient has to be known before the update and the norm is asked of exactly the gradients the optimizer
                # consumes: the AdamW kernels accumulate it in the pass they make over the gradients anyway
                _, norm = optimizer.step(want_norm=True)
                return norm
            norm, coef = _optim.grad_norm_and_coef(params, clip_grad, _norm_cache)
            if isinstance(optimizer, _optim.FusedAdamW):
                optimizer.set_grad_scale(coef if clip_grad is not None else None)
            elif clip_grad is not None:
                for p in params:
                    if p.grad is not None:
                        p.grad.mul_(coef)
            optimizer.step()
            return norm
        # ---- fp16-style dynamic loss scale (GradScaler.unscale_ / step / update)
        scaled_norm, _ = _optim.grad_norm_and_coef(params, None, _norm_cache)
        norm = scaled_norm / self._scale
        found_inf = not bool(torch.isfinite(norm).item())
        self.last_step_skipped = found_inf
        if not found_inf:
            inv = 1.0 / self._scale
            if clip_grad is not None:
                coef = torch.clamp(float(clip_grad) / (norm + 1e-6), max=1.0) * inv      # clip_grad_norm_'s coefficient x 1 / scale
            else:
                coef = torch.full((), inv, dtype=torch.float32, device=norm.device)
            if isinstance(optimizer, _optim.FusedAdamW):
                optimizer.set_grad_scale(coef)
            else:
                for p in params:
                    if p.grad is not None:
                        p.grad.mul_(coef)
            optimizer.step()
            if isinstance(optimizer, _optim.FusedAdamW):
                optimizer.set_grad_scale(None)
            self._growth_tracker += 1
            if self._growth_tracker >= self._growth_interval:
                self._scale *= self._growth_factor
                self._growth_tracker = 0
        else:
            self._scale *= self._backoff_factor
            self._growth_tracker = 0
        return norm

    def state_dict(self):
        return {"scale": self._scale, "growth_factor": self._growth_factor, "backoff_factor": self._backoff_factor,
                "growth_interval": self._growth_interval, "_growth_tracker": self._growth_tracker}

    def load_state_dict(self, state_dict):
        if self.enabled:
            self._scale = float(state_dict.get("scale", self._scale))
            self._growth_factor = float(state_dict.get("growth_factor", self._growth_factor))
            self._backoff_factor = float(state_dict.get("backoff_factor", self._backoff_factor))
            self._growth_interval = int(state_dict.get("growth_interval", self._growth_interval))
        self._growth_tracker = int(state_dict.get("_growth_tracker", 0))


# ------------------------------------------------------------------------------------------------
# logging
# ------------------------------------------------------------------------------------------------
class SmoothedValue:
    """Windowed median / average plus a global average of a scalar series."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total = 0.0
        self.count = 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / max(self.count, 1)

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(v)

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join(f"{n}: {m}" for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def log_every(self, iterable, print_freq, header=""):
        start = time.time()
        end = time.time()
        iter_time = SmoothedValue(fmt="{avg:.4f}")
        n = len(iterable) if hasattr(iterable, "__len__") else -1
        for i, obj in enumerate(iterable):
            yield obj
            iter_time.update(time.time() - end)
            if is_main_process() and (i % print_freq == 0 or i == n - 1):
                mem = torch.cuda.max_memory_allocated() / 2 ** 20 if torch.cuda.is_available() else 0.0
                print(f"{header} [{i}/{n}] {self} time: {iter_time} max mem: {mem:.0f}", flush=True)
            end = time.time()
        if is_main_process():
            print(f"{header} Total time: {datetime.timedelta(seconds=int(time.time() - start))}", flush=True)


# ------------------------------------------------------------------------------------------------
# host -> device input pipeline
# ------------------------------------------------------------------------------------------------
class DevicePrefetcher:
    """Iterates a data loader one batch ahead: batch i + 1 is copied host -> device on a side stream while batch i is computed.

    The reference's loops copy on the compute stream (``samples.to(device, non_blocking=True)``, engine_pretrain.py:44 /
    Pre-training/engine_pretrain.py:108-109): with pinned memory that is asynchronous for the HOST only -- the copy still sits in the
    stream ahead of the forward.  A 256-volume step moves 4 GB of fp32 volumes: 76 ms at the 53 GB/s this box's PCIe link gives, 5 % of
    the step (bench.py --host-inputs 1: 167.8 against 176.6 volumes/s resident); one batch ahead on a side stream it is hidden (176.2,
    --host-inputs 2).  Same batches in the same order; tensors inside tuples / lists / dicts are moved (or only the listed positions of the batch),
    everything else is passed through; the engines' own ``.to(device)`` then finds the tensor already there.  On a CPU device this is a plain pass-through."""

    def __init__(self, loader, device, only=None):
        """only: None = every tensor of the batch; a tuple of positions = just those elements of a tuple / list batch (the volume,
        the target), so that bookkeeping the loops read on the host -- ids, names, per-sample dicts -- stays where it was."""
        self.loader = loader
        self.device = torch.device(device)
        self.only = None if only is None else tuple(only)

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):              # sampler, dataset, batch_size, ... of the wrapped loader
        if name in ("loader", "device", "only"):
            raise AttributeError(name)
        return getattr(self.loader, name)

    def _move(self, obj):
        if torch.is_tensor(obj):
            return obj.to(self.device, non_blocking=True)
        if isinstance(obj, tuple):
            items = [self._move(o) for o in obj]
            return type(obj)(*items) if hasattr(obj, "_fields") else tuple(items)        # namedtuples keep their type
        if isinstance(obj, list):
            return [self._move(o) for o in obj]
        if isinstance(obj, dict):
            return {k: self._move(v) for k, v in obj.items()}
        return obj

    def _record(self, obj, stream):
        if torch.is_tensor(obj):
            if obj.is_cuda:
                obj.record_stream(stream)
        elif isinstance(obj, (tuple, list)):
            for o in obj:
                self._record(o, stream)
        elif isinstance(obj, dict):
            for o in obj.values():
                self._record(o, stream)

    def __iter__(self):
        if self.device.type != "cuda":
            yield from self.loader
            return
        side = torch.cuda.Stream(device=self.device)

        def start(batch):
            with torch.cuda.stream(side):
                if self.only is not None and isinstance(batch, (tuple, list)):
                    items = [self._move(o) if i in self.only else o for i, o in enumerate(batch)]
                    moved = type(batch)(*items) if hasattr(batch, "_fields") else (tuple(items) if isinstance(batch, tuple) else items)
                else:
                    moved = self._move(batch)
                ev = torch.cuda.Event()
                ev.record(side)
            return moved, ev

        it = iter(self.loader)
        try:
            nxt = start(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = start(next(it))
            except StopIteration:
                nxt = None
            main = torch.cuda.current_stream(self.device)
            main.wait_event(ev)
            self._record(cur, main)              # the side stream allocated it; the compute stream uses it
            yield cur


# ------------------------------------------------------------------------------------------------
# checkpoint positional tables, as OCTCube/util/misc.py has them (the functions inference_utils.py and the fine-tune drivers import
# from there; util/pos_embed.py holds a DIFFERENT interpolate_pos_embed, which also resizes ``pos_embed_spatial`` -- pos_embed.py here)
# ------------------------------------------------------------------------------------------------
def interpolate_pos_embed(model, checkpoint_model):
    """OCTCube/util/misc.py:1159-1222: bicubic resize of ``pos_embed`` and ``decoder_pos_embed`` (extra tokens kept) to the model's
    patch grid.  ``pos_embed_spatial`` is NOT touched by this variant: a separable-table checkpoint of another spatial grid then fails
    the strict load that follows, exactly as in the reference."""
    from .pos_embed import _resize_square
    for name in ("pos_embed", "decoder_pos_embed"):
        if name not in checkpoint_model:
            continue
        ck = checkpoint_model[name]
        num_patches = model.patch_embed.num_patches
        num_extra = getattr(model, name).shape[-2] - num_patches
        orig_size = int((ck.shape[-2] - num_extra) ** 0.5)
        new_size = int(num_patches ** 0.5)
        if orig_size != new_size:
            print("Position interpolate from %dx%d to %dx%d" % (orig_size, orig_size, new_size, new_size))
            checkpoint_model[name] = torch.cat((ck[:, :num_extra], _resize_square(ck[:, num_extra:], orig_size, new_size)), dim=1)


def interpolate_temporal_pos_embed(model, checkpoint_model, smaller_interpolate_type="interp"):
    """OCTCube/util/misc.py:1225-1258 (the same function as util/pos_embed.py's)."""
    from .pos_embed import interpolate_temporal_pos_embed as f
    return f(model, checkpoint_model, smaller_interpolate_type)

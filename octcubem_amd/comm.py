"""Host side of the RCCL layer (``octmae_comm_*`` in include/octmae.h): one communicator per process, created from the
launcher's key-value store, used by ``parallel.FlatGradReducer`` for the gradient exchange, by ``bench.py`` for its barrier /
max-over-ranks timing and by ``coem.gather_features`` for the feature all-gather.

Replaces ``init_process_group("nccl")`` + ``DistributedDataParallel`` of the reference
(Pre-training/custom_util/misc.py:283-296, main_pretrain_oph_joint_2d512_flash_attn.py:434-439) on the data path.
``torch.distributed`` remains the control plane only: its store carries the 128-byte RCCL unique id from rank 0 to the others
(the same MASTER_ADDR / MASTER_PORT rendezvous torchrun sets up for the reference).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from ._lib import call, load

F32, BF16, F64, F16 = 0, 1, 2, 3
SUM, AVG, MAX = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float64: F64, torch.float16: F16}      # F16: collectives of the half-operand build
ID_BYTES = 128

_default: Optional["NativeComm"] = None


def available() -> bool:
    return bool(load().octmae_comm_available())


class NativeComm:
    """An RCCL communicator behind the C ABI.  Collectives are enqueued on the communicator's own stream, ordered behind the
    current torch stream at the time of the call; ``wait()`` makes the current torch stream wait for them."""

    def __init__(self, id_bytes: bytes, rank: int, world: int, device: int):
        assert len(id_bytes) == ID_BYTES
        h = C.c_void_p()
        buf = C.create_string_buffer(id_bytes, ID_BYTES)
        call("octmae_comm_init", C.byref(h), buf, int(rank), int(world), int(device))
        self._h = h
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self._keep = []          # tensors that must outlive the collectives reading / writing them (released by wait())

    # ------------------------------------------------------------------ construction
    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(ID_BYTES)
        call("octmae_comm_unique_id", buf)
        return buf.raw

    @classmethod
    def from_store(cls, store, rank: int, world: int, device: int, key: str = "octmae/comm_id/0") -> "NativeComm":
        """Rank 0 creates the RCCL unique id and publishes it under ``key`` of a torch.distributed store; everyone joins.
        The published key carries a GENERATION: every rank counts itself in with ``store.add(key + "/arrivals", 1)`` and the
        generation is (arrival - 1) // world -- communicator creation is collective (ncclCommInitRank returns only when all
        ranks have joined), so the ``world`` arrivals of one creation are complete before the first of the next.  A second
        communicator on the same store (re-initialisation after destroy(), a library and a benchmark both bootstrapping) can
        therefore never read the previous, stale id and hang with mismatched ids."""
        cls._set_device(device)
        gen = (int(store.add(key + "/arrivals", 1)) - 1) // world
        skey = f"{key}#{gen}"
        if rank == 0:
            store.set(skey, cls.unique_id())
        id_bytes = bytes(store.get(skey))
        return cls(id_bytes, rank, world, device)

    @staticmethod
    def _set_device(device: int):
        torch.cuda.set_device(device)

    @classmethod
    def from_env(cls, device: Optional[int] = None, tag: str = "0") -> "NativeComm":
        """From torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK) and the default process group's store when
        torch.distributed is initialised (any backend; no collective of that group is issued), else a TCPStore of its own
        on MASTER_ADDR : MASTER_PORT + 1."""
        import torch.distributed as dist
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        if world == 1:
            torch.cuda.set_device(dev)
            return cls(cls.unique_id(), 0, 1, dev)
        if dist.is_available() and dist.is_initialized():
            store = dist.distributed_c10d._get_default_store()
        else:
            from datetime import timedelta
            store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29531")) + 1,
                                  world, rank == 0, timeout=timedelta(seconds=300))
        return cls.from_store(store, rank, world, dev, key=f"octmae/comm_id/{tag}")

    # ------------------------------------------------------------------ collectives
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    @staticmethod
    def _check(t: torch.Tensor):
        if not (t.is_cuda and t.is_contiguous() and t.dtype in _DT):
            raise RuntimeError("NativeComm: contiguous fp32 / bf16 / fp16 / fp64 GPU tensors only")

    def all_reduce_async(self, t: torch.Tensor, op: int = SUM):
        self._check(t)
        call("octmae_comm_allreduce_async", self._h, t.data_ptr(), t.numel(), _DT[t.dtype], op, self._stream())
        self._keep.append(t)

    def broadcast_async(self, t: torch.Tensor, root: int = 0):
        self._check(t)
        call("octmae_comm_broadcast_async", self._h, t.data_ptr(), t.numel(), _DT[t.dtype], int(root), self._stream())
        self._keep.append(t)

    def all_gather_async(self, send: torch.Tensor, recv: torch.Tensor):
        self._check(send); self._check(recv)
        assert recv.numel() == send.numel() * self.world and recv.dtype == send.dtype
        call("octmae_comm_allgather_async", self._h, send.data_ptr(), recv.data_ptr(), send.numel(), _DT[send.dtype], self._stream())
        self._keep += [send, recv]

    def reduce_scatter_async(self, send: torch.Tensor, recv: torch.Tensor, op: int = SUM):
        self._check(send); self._check(recv)
        assert send.numel() == recv.numel() * self.world and recv.dtype == send.dtype
        call("octmae_comm_reduce_scatter_async", self._h, send.data_ptr(), recv.data_ptr(), recv.numel(), _DT[send.dtype], op,
             self._stream())
        self._keep += [send, recv]

    def torch_stream(self):
        """The communication stream as a torch stream object, for timing events only (octmae_comm_stream)."""
        if getattr(self, "_tstream", None) is None:
            h = C.c_void_p()
            call("octmae_comm_stream", self._h, C.byref(h))
            self._tstream = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", self.device))
        return self._tstream

    def wait(self):
        """The current torch stream waits for every collective enqueued so far."""
        call("octmae_comm_wait", self._h, self._stream())
        # buffers were kept alive until here; freed now they return to the CURRENT stream's pool, whose later work is ordered
        # behind the wait just enqueued
        self._keep = []

    # ------------------------------------------------------------------ conveniences for bench / engines
    def barrier(self):
        """All ranks reach this point and the device is idle (a 1-element all-reduce + a host synchronise)."""
        t = torch.zeros(1, dtype=torch.float32, device=torch.device("cuda", self.device))
        self.all_reduce_async(t, SUM)
        self.wait()
        torch.cuda.synchronize(self.device)

    def all_reduce_scalar(self, value: float, op: int = AVG) -> float:
        """misc.all_reduce_mean (custom_util/misc.py:622-630) / max over ranks of a host float."""
        t = torch.tensor([value], dtype=torch.float64, device=torch.device("cuda", self.device))
        self.all_reduce_async(t, op)
        self.wait()
        return float(t.item())

    def destroy(self):
        if self._h is not None and self._h.value:
            call("octmae_comm_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def set_default(comm: Optional[NativeComm]):
    global _default
    _default = comm


def get_default() -> Optional[NativeComm]:
    return _default

"""Multi-GPU sharding: one process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over
xGMI).  Environments are independent replicas, so the batch is cut into contiguous blocks
``[rank*B/W, (rank+1)*B/W)`` with no collective on the stepping path; the only exchange is an
all-gather of the flat observation block.  Random ctrl is keyed by the GLOBAL
environment index (``env0``), so results do not depend on the number of GPUs.
"""

from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass


def world() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(global_batch: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous block of environments owned by ``rank``: (env0, count); remainders go to the low ranks."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    base, rem = divmod(int(global_batch), world_size)
    count = base + (1 if rank < rem else 0)
    env0 = rank * base + min(rank, rem)
    return env0, count


def init_process_group(backend: str | None = None, single_rank: bool = False) -> bool:
    """Initialise torch.distributed from the env when WORLD_SIZE > 1.  Returns True if initialised.

    ``single_rank=True`` also initialises a ONE-rank group when the torchrun environment is present (RANK set): the rehearsal of the
    ``nccl`` code path on a one-GPU box (communicator setup, the collective on the launch stream, the barrier / reductions of the bench).
    """
    import torch
    import torch.distributed as dist

    rank, ws, local = world()
    if ws <= 1 and not (single_rank and "RANK" in os.environ):
        return False
    if dist.is_initialized():
        return True
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return True


def all_gather_obs(local_obs, global_batch: int | None = None, counts=None, single_rank: bool = False):
    """All-gather ``[..., b_local, dim]`` blocks along the batch axis (axis -2) into ``[..., B, dim]``.

    Equal shards use one ``all_gather_into_tensor`` (a single RCCL collective); ragged shards
    fall back to ``all_gather`` with padding.  Without an initialised process group this is the identity.
    ``counts`` (per-rank shard sizes, e.g. from ``shard_range``) skips the size exchange and its host sync.
    With the ``gloo`` backend CUDA blocks are staged through the host (rehearsals on one GPU).
    ``single_rank=True`` issues the collective even in a one-rank group (one-GPU rehearsal of the ``nccl`` path) instead of returning early.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not single_rank):
        return local_obs
    ws = dist.get_world_size()
    x = local_obs.movedim(-2, 0).contiguous()                  # [b_local, ..., dim]
    home = x.device
    if x.is_cuda and dist.get_backend() == "gloo":
        x = x.cpu()
    if counts is None:
        sizes = [torch.zeros(1, dtype=torch.int64, device=x.device) for _ in range(ws)]
        dist.all_gather(sizes, torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device))
        counts = [int(s.item()) for s in sizes]
    elif len(counts) != ws or counts[dist.get_rank()] != x.shape[0]:
        raise ValueError("counts does not match the process group / the local block")
    if len(set(counts)) == 1:
        out = torch.empty((ws * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x)
    else:
        mx = max(counts)
        pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        pad[: x.shape[0]] = x
        parts = [torch.empty_like(pad) for _ in range(ws)]
        dist.all_gather(parts, pad)
        out = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    return out.to(home).movedim(0, -2)


@dataclass(frozen=True)
class ShardPlan:
    """Which contiguous block of a GLOBAL batch this process owns: what ``Env.from_xml_path(..., batch=GLOBAL,
    shard=True)`` derives from the torchrun environment and keeps as ``env.shard``.  ``gather`` is the path's one collective."""

    rank: int
    world_size: int
    local_rank: int
    global_batch: int
    env0: int
    count: int
    counts: tuple

    @classmethod
    def from_environment(cls, global_batch: int, rank: int | None = None, world_size: int | None = None, local_rank: int | None = None) -> "ShardPlan":
        r, ws, lr = world()
        rank = r if rank is None else int(rank)
        world_size = ws if world_size is None else int(world_size)
        local_rank = lr if local_rank is None else int(local_rank)
        if int(global_batch) < world_size:
            raise ValueError(f"global batch {global_batch} is smaller than the number of ranks {world_size}")
        env0, count = shard_range(global_batch, rank, world_size)
        counts = tuple(shard_range(global_batch, k, world_size)[1] for k in range(world_size))
        return cls(rank, world_size, local_rank, int(global_batch), env0, count, counts)

    @property
    def equal_shards(self) -> bool:
        return len(set(self.counts)) == 1

    def gather(self, local_obs, comm: "RcclCommunicator | None" = None, stream=None):
        """``[..., count, dim]`` of this rank -> ``[..., global_batch, dim]`` on every rank (rank order = environment order).  With an
        ``RcclCommunicator`` and equal shards the collective is the library's own C-ABI entry point ``mjb_allgather_obs``
        (ncclAllGather on the given HIP stream); otherwise ``torch.distributed`` (``all_gather_obs``)."""
        if self.world_size == 1 and comm is None:
            return local_obs
        if comm is not None and self.equal_shards and getattr(local_obs, "is_cuda", False):
            return comm.all_gather_obs(local_obs, stream=stream)
        return all_gather_obs(local_obs, counts=list(self.counts))


class _NcclUniqueId(ctypes.Structure):                             # ncclUniqueId: 128 opaque bytes, passed BY VALUE
    _fields_ = [("internal", ctypes.c_char * 128)]


class RcclCommunicator:
    """An ``ncclComm_t`` of this process' own, so that the observation all-gather runs through the library's C-ABI collective
    (``mjb_allgather_obs`` -> ``ncclAllGather``, include/mjbatch.h) instead of ``torch.distributed``: the unique id is made on rank 0,
    carried to the other ranks over the already initialised ``torch.distributed`` group (any backend), and every rank calls
    ``ncclCommInitRank`` on the RCCL that torch loaded.  One rank per GPU (RCCL refuses two ranks on one device)."""

    def __init__(self, rank: int, world_size: int, device: int):
        import torch
        import torch.distributed as dist

        self._rccl = None
        for name in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so"):
            try:
                self._rccl = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
                break
            except OSError:
                continue
        if self._rccl is None:
            raise RuntimeError("no loadable librccl")
        R = self._rccl
        R.ncclGetUniqueId.argtypes = [ctypes.POINTER(_NcclUniqueId)]
        R.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
        R.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        uid = _NcclUniqueId()
        if rank == 0 and R.ncclGetUniqueId(ctypes.byref(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        if world_size > 1:
            box = [bytes(uid.internal) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctypes.memmove(ctypes.byref(uid), box[0], 128)
        torch.cuda.set_device(device)
        self.comm = ctypes.c_void_p()
        rc = R.ncclCommInitRank(ctypes.byref(self.comm), int(world_size), uid, int(rank))
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank failed with ncclResult {rc}")
        self.rank, self.world_size, self.device = int(rank), int(world_size), int(device)

    def all_gather_obs(self, local_obs, stream=None):
        """``[..., b, dim]`` -> ``[..., world_size * b, dim]`` (equal shards) through ``mjb_allgather_obs`` on ``stream`` (default: torch's current)."""
        import torch

        from ._capi import _check, load_library

        L = load_library()
        L.mjb_allgather_obs.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
        L.mjb_allgather_obs.restype = ctypes.c_int
        x = local_obs.movedim(-2, 0).contiguous()                  # [b, ..., dim]: rank blocks are contiguous along the batch axis
        out = torch.empty((self.world_size * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if stream is None:
            stream = torch.cuda.current_stream(x.device).cuda_stream
        dtype = {torch.float32: 0, torch.float64: 1}[x.dtype]
        _check(L.mjb_allgather_obs(self.comm, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), x.numel(), dtype, ctypes.c_void_p(stream)))
        return out.movedim(0, -2)

    def close(self) -> None:
        if getattr(self, "comm", None) and self._rccl is not None:
            self._rccl.ncclCommDestroy(self.comm)
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


__all__ = ["world", "shard_range", "init_process_group", "all_gather_obs", "ShardPlan", "RcclCommunicator"]

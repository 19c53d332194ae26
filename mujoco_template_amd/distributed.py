"""Multi-GPU sharding: one process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over
xGMI).  Environments are independent replicas, so the batch is cut into contiguous blocks
``[rank*B/W, (rank+1)*B/W)`` with no collective on the stepping path; the only exchange is an
all-gather of the flat observation block.  Random ctrl is keyed by the GLOBAL
environment index (``env0``), so results do not depend on the number of GPUs.
"""

from __future__ import annotations

import os


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


__all__ = ["world", "shard_range", "init_process_group", "all_gather_obs"]

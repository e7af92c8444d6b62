"""Multi-GPU sharding of the CLOUDSC2 hot path.

Columns are independent, so the 8 GPUs of a node each own a contiguous sub-range of the global columns -- the
reference's own MPI split (src/cloudsc2_nl/dwarf_cloudsc.F90:66-69) -- and the data path needs no collective.
The only exchange is the verdict of the two self-tests: the reference max-reduces ZNORMG over OpenMP threads
(cloudsc_driver_tl_mod.F90:125, cloudsc_driver_ad_mod.F90:107); across GPUs that becomes one all-reduce(MAX) of
10 doubles (TL) or 1 double (AD) -- RCCL over xGMI on the GPU box (backend "nccl"), gloo in CPU tests.
"""
from __future__ import annotations

import os

import numpy as np

from .state import column_range


def init_process_group(backend: str | None = None):
    """One process per GPU; reads RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment (torchrun)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend is None:
            # CLOUDSC2_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("CLOUDSC2_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            if torch.cuda.device_count() < world and int(os.environ.get("LOCAL_WORLD_SIZE", world)) > torch.cuda.device_count():
                raise RuntimeError(f"{world} ranks need {world} GPUs, this node has {torch.cuda.device_count()} "
                                   "(CLOUDSC2_DIST_BACKEND=gloo rehearses the multi-rank path on fewer GPUs)")
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            local = local % max(torch.cuda.device_count(), 1)  # rehearsal: ranks share the GPUs that exist
        # explicit deadline: a rank that never shows up fails the rendezvous (and every later collective) after this long
        # instead of torch's default half hour -- the launcher then sees a non-zero exit and stops the job
        import datetime

        timeout = datetime.timedelta(seconds=float(os.environ.get("CLOUDSC2_DIST_TIMEOUT_S", "120")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, local, world


def shard(ngptotg: int, rank: int, world: int) -> tuple[int, int]:
    """(first global column, number of columns) of this rank."""
    a, b = column_range(ngptotg, rank, world)
    return a, b - a


def allreduce_max(values, device=None) -> np.ndarray:
    """Element-wise MAX over all ranks of a small vector of doubles (the test verdict norms)."""
    import torch
    import torch.distributed as dist

    v = np.atleast_1d(np.asarray(values, dtype=np.float64))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return v
    if dist.get_backend() != "nccl":
        device = torch.device("cpu")  # gloo (CPU tests, rehearsals): reduce on the host
    elif device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.from_numpy(v.copy()).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.cpu().numpy()


def allgather_scalar(value: float, device=None) -> list:
    """One double from every rank, in rank order (per-rank kernel times of the bench line)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    v = np.zeros(dist.get_world_size(), dtype=np.float64)
    v[dist.get_rank()] = value
    # slots of the other ranks are zero: MAX would drop negative values, so reduce |v| and sign separately via SUM
    import torch

    dev = torch.device("cpu") if dist.get_backend() != "nccl" else (device or torch.device("cuda", torch.cuda.current_device()))
    t = torch.from_numpy(v).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.cpu().numpy()]


def allreduce_validation(stats: dict, device=None) -> dict:
    """NL validation statistics across ranks with the reference's reductions (validate_mod.F90:197-199):
    min of minima, max of maxima / max abs error, sum of the L1 sums."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return stats
    if dist.get_backend() != "nccl":
        device = torch.device("cpu")
    elif device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    names = sorted(stats)
    mn = torch.tensor([stats[n]["min"] for n in names], dtype=torch.float64, device=device)
    mx = torch.tensor([[stats[n]["max"], stats[n]["maxabserr"]] for n in names], dtype=torch.float64, device=device)
    sm = torch.tensor([[stats[n]["sumabserr"], stats[n]["sumabsref"]] for n in names], dtype=torch.float64, device=device)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    out = {}
    for k, n in enumerate(names):
        out[n] = {"min": float(mn[k]), "max": float(mx[k, 0]), "maxabserr": float(mx[k, 1]),
                  "sumabserr": float(sm[k, 0]), "sumabsref": float(sm[k, 1])}
    return out

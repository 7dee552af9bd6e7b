"""torch.distributed plumbing for the multi-GPU path (one process per GPU).

Tracks are sharded over the ranks inside the C library (slices of the
length-sorted track order, dealt longest-work-first).  The only data-path
exchange is the sum of the reduced camera system built from each rank's
tracks -- once per LM iteration -- plus a few scalars; the engine calls the
hook below on its own HIP stream with a device pointer, and the hook runs an
RCCL all-reduce (backend "nccl" is RCCL on ROCm) over xGMI, stream-ordered
with the engine's kernels.  With backend "gloo" (CPU tests) the buffer is a
host pointer and the reduction goes through a CPU tensor.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can alias a raw device pointer."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {
            "shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2,
            "strides": None,
        }


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment.  Returns
    (rank, world, local_rank).  World size 1 needs no process group."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def make_device_allreduce(collective=None, stats=None):
    """Hook for lib.Solver.set_allreduce on GPU: RCCL sum over the default group,
    enqueued on the engine's stream (no host synchronisation).

    collective: the in-place sum of a device tensor over the group (default: torch.distributed.all_reduce, i.e. RCCL
    with backend "nccl").  tests/test_gpu_bench_multirank.py passes gloo_staged_sum so that THIS hook -- the pointer
    aliasing through __cuda_array_interface__, the per-buffer tensor cache, the ExternalStream of the engine -- runs
    with two real processes on a box whose ranks share one GPU (where RCCL itself cannot).
    stats: optional dict, counts calls and bytes (bench.py prints them for N > 1)."""
    import torch
    import torch.distributed as dist

    streams = {}
    tensors = {}  # the engine's buffers are persistent: alias each (pointer, count) once
    if collective is None:
        def collective(t):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def hook(ptr: int, count: int, stream: int) -> int:
        if count <= 0:
            return 0
        t = tensors.get((ptr, count))
        if t is None:
            t = torch.as_tensor(_DevArray(ptr, count), device="cuda")
            tensors[(ptr, count)] = t
        ext = streams.get(stream)
        if ext is None:
            ext = torch.cuda.ExternalStream(stream)
            streams[stream] = ext
        with torch.cuda.stream(ext):
            collective(t)
        if stats is not None:
            stats["calls"] = stats.get("calls", 0) + 1
            stats["bytes"] = stats.get("bytes", 0) + 8 * int(count)
        return 0

    return hook


def gloo_staged_sum(t):
    """In-place sum of a device tensor over the default group through host memory (gloo): the collective
    make_device_allreduce takes when the ranks share one GPU.  Runs under the caller's stream context; the copies
    order themselves on it."""
    import torch.distributed as dist
    h = t.cpu()
    dist.all_reduce(h, op=dist.ReduceOp.SUM)
    t.copy_(h)


def make_host_allreduce():
    """Same contract for a host buffer (gloo, used by the CPU tests of the
    sharding logic): sums `count` doubles at address `ptr` in place."""
    import torch
    import torch.distributed as dist

    def hook(ptr: int, count: int, stream: int) -> int:
        if count <= 0:
            return 0
        buf = (C.c_double * count).from_address(ptr)
        a = np.frombuffer(buf, dtype=np.float64)
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        a[:] = t.numpy()
        return 0

    return hook


def init_native_rccl(solver, rank: int, world: int) -> bool:
    """Give `solver` its own RCCL communicator (the engine then calls ncclAllReduce
    itself, no Python in the loop).  The 128-byte ncclUniqueId travels from rank 0
    through the already initialised torch.distributed group.  Returns False (and leaves
    the solver untouched) when RCCL cannot be bound, so callers can fall back to the
    torch.distributed hook."""
    import torch.distributed as dist
    from . import lib
    payload = [None]
    if rank == 0:
        try:
            payload[0] = lib.rccl_unique_id()
        except lib.EngineError as exc:  # librccl not resolvable
            payload[0] = repr(exc)
    dist.broadcast_object_list(payload, src=0)
    if not isinstance(payload[0], (bytes, bytearray)):
        return False
    ok = 1
    try:
        solver.init_rccl(bytes(payload[0]))
    except lib.EngineError as exc:
        print(f"[theiasfm_amd] rank {rank}: native RCCL transport unavailable ({exc}); "
              "falling back to the torch.distributed hook", file=sys.stderr, flush=True)
        ok = 0
    # every rank must take the same decision
    import torch
    flag = torch.tensor([ok], dtype=torch.int32,
                        device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        solver.init_rccl(None)
        return False
    return True


def make_staged_allreduce():
    """Device buffer, host transport: copies the buffer to the host, all-reduces it over
    the default group (e.g. gloo) and copies it back.  For testing the multi-process path
    on a box where the ranks have to share one GPU (RCCL refuses that)."""
    import torch
    import torch.distributed as dist

    def hook(ptr: int, count: int, stream: int) -> int:
        if count <= 0:
            return 0
        ext = torch.cuda.ExternalStream(stream)
        with torch.cuda.stream(ext):
            t = torch.as_tensor(_DevArray(ptr, count), device="cuda")
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
            ext.synchronize()
        return 0

    return hook

"""Multi-GPU plumbing: one process per GPU, games sharded by contiguous game-id ranges, ONE exchange per update —
an all-reduce (sum) of the flat gradient buffer over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

The reference is single-process / single-GPU; this is build-defined.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_games(rank, games_per_rank):
    """Contiguous game-id range of a rank: ids double as Philox stream ids, so shards never share a stream."""
    first = rank * games_per_rank
    return first, first + games_per_rank


def init_process_group(backend=None):
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class _DevPtr:
    """Zero-copy view of a device allocation owned by libxqhip (float32[n]) for torch."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def wrap_device_floats(ptr, n, device="cuda"):
    return torch.as_tensor(_DevPtr(ptr, n), device=device)


def allreduce_gradients(flat, world_size=None):
    """Sum the flat gradient buffer over all ranks, in place.  `flat` is any torch tensor (HBM view of the
    xq_dqn gradient buffer on the GPU path, a CPU tensor under gloo in the tests).  The mean over the global batch
    is applied afterwards by xq_trainer_learn_apply(world_size) / the caller's grad_scale."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if world_size > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def max_over_ranks(value, device="cpu"):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

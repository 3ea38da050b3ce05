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


class Comm:
    """xq_comm: the RCCL communicator behind the C ABI (include/xq_capi.h).  The 128-byte id is drawn on rank 0 and shipped
    through torch.distributed's control plane (any backend) — the gradient traffic itself never goes through torch."""

    def __init__(self, rank=None, world=None, id_bytes=None, path=None, timeout_s=120.0):
        import ctypes as C
        import numpy as np
        from . import _capi
        r, _, w = env_rank()
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        h = C.c_void_p()
        if path is not None:
            _capi.call("xq_comm_create_from_file", self.rank, self.world, str(path).encode(), float(timeout_s), C.byref(h))
        else:
            if id_bytes is None:
                buf = np.zeros(128, dtype=np.uint8)
                if self.rank == 0:
                    _capi.call("xq_comm_unique_id", buf.ctypes.data_as(C.POINTER(C.c_uint8)))
                if self.world > 1:
                    box = [buf.tobytes()]
                    dist.broadcast_object_list(box, src=0)
                    buf = np.frombuffer(box[0], dtype=np.uint8).copy()
                id_bytes = buf.tobytes()
            arr = np.frombuffer(id_bytes, dtype=np.uint8).copy()
            _capi.call("xq_comm_create", self.rank, self.world, arr.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(h))
        self._h = h

    @property
    def handle(self):
        return self._h

    def info(self):
        import ctypes as C
        from . import _capi
        r, w, n, f = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64()
        _capi.call("xq_comm_info", self._h, C.byref(r), C.byref(w), C.byref(n), C.byref(f))
        return dict(rank=r.value, world=w.value, collectives=n.value, floats=f.value)

    def close(self):
        from . import _capi
        if self._h is not None:
            _capi.call("xq_comm_destroy", self._h)
        self._h = None


def max_over_ranks(value, device="cpu"):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

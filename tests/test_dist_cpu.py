"""Multi-GPU path on CPU (not gpu): world_size-2 gloo run of the sharding + gradient all-reduce plumbing.

Each rank owns a contiguous range of game ids, computes the TD gradients of ITS transitions with the fp64 oracle
(standing in for xq_dqn_td_grads, whose output is the same flat buffer), all-reduces the flat buffer through
cn_chess_ai_amd.dist, applies the mean — the result must equal the single-process update on the union."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import xqoracle as xo

SIZES = [40, 16, 24, 32]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(n, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, size=(n, SIZES[0])), rng.uniform(-0.9, 0.9, size=(n, SIZES[-1]))


def _worker(rank, world, port, per_rank, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cn_chess_ai_amd import dist as xd
    r, _, w = xd.init_process_group("gloo")
    assert (r, w) == (rank, world)
    first, last = xd.shard_games(rank, per_rank)
    assert (first, last) == (rank * per_rank, (rank + 1) * per_rank)
    wts, bias = xo.init_weights(SIZES, 3)
    X, T = _data(per_rank * world, 9)
    gw, gb = np.zeros_like(wts), np.zeros_like(bias)
    for i in range(first, last):
        assert xo.nn_accum_grad(SIZES, wts, bias, X[i], T[i], 1, gw, gb) == 0
    flat = torch.from_numpy(np.concatenate([gw, gb]))
    xd.allreduce_gradients(flat, world)
    t = xd.max_over_ranks(float(rank + 1))
    assert t == float(world)
    lr, scale = 0.05, 1.0 / (per_rank * world)
    new = np.concatenate([wts, bias]) - lr * scale * flat.numpy()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), new)
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_process(tmp_path):
    world, per_rank = 2, 5
    mp.spawn(_worker, args=(world, _free_port(), per_rank, str(tmp_path)), nprocs=world, join=True)
    wts, bias = xo.init_weights(SIZES, 3)
    X, T = _data(per_rank * world, 9)
    gw, gb = np.zeros_like(wts), np.zeros_like(bias)
    for x, t in zip(X, T):
        xo.nn_accum_grad(SIZES, wts, bias, x, t, 1, gw, gb)
    want = np.concatenate([wts, bias]) - 0.05 / (per_rank * world) * np.concatenate([gw, gb])
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1)                         # replicas stay bit-identical
    assert np.abs(r0 - want).max() < 1e-15


def test_single_process_helpers():
    from cn_chess_ai_amd import dist as xd
    assert xd.shard_games(3, 8192) == (24576, 32768)
    t = torch.ones(4)
    assert xd.allreduce_gradients(t, 1) is t and xd.max_over_ranks(2.5) == 2.5

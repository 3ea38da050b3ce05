"""GPU-side plumbing of the multi-GPU path that can be checked on one device — `pytest -m gpu`.

The N > 1 collective itself is covered by the gloo test (tests/test_dist_cpu.py) and by the driver's scaling run."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gradient_buffer_is_visible_to_torch_zero_copy():
    import torch
    import cn_chess_ai_amd as xq
    from cn_chess_ai_amd import dist as xd
    s = torch.cuda.Stream()
    torch.cuda.set_stream(s)
    cfg = xq.TrainerConfig(n_games=256, layer_sizes=(1260, 128, 8100), replay_capacity=0, minibatch=256, td_net=0)
    t = xq.Trainer(cfg, stream=C.c_void_p(s.cuda_stream))
    ptr, n = t.dqn.grad_buffer()
    g = xd.wrap_device_floats(ptr, n)
    assert g.is_cuda and g.numel() == n and g.data_ptr() == ptr and g.dtype == torch.float32
    t.collect()
    t.learn_grads()
    torch.cuda.synchronize()
    before = g.clone()
    assert torch.isfinite(before).all() and before.abs().sum().item() > 0
    # an in-place torch op on the view is what RCCL's all_reduce does: the library must see it
    w0, b0 = t.dqn.get_params()
    g.mul_(2.0)                                   # "sum over 2 identical ranks"
    xd.allreduce_gradients(g, 1)
    t.learn_apply(world_size=2)                   # mean over batch * 2 ranks -> same update as one rank, unscaled
    w1, b1 = t.dqn.get_params()
    t2 = xq.Trainer(cfg, stream=C.c_void_p(s.cuda_stream))
    t2.collect(); t2.learn_grads(); t2.learn_apply(world_size=1)
    w2, b2 = t2.dqn.get_params()
    assert np.array_equal(w1, w2) and np.array_equal(b1, b2) and not np.array_equal(w1, w0)
    t.close(); t2.close()
    torch.cuda.set_stream(torch.cuda.default_stream())


def _rehearse(extra):
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, XQ_FORCE_DEVICE="0", XQ_DIST_BACKEND="gloo")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                           "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"] + extra,
                          capture_output=True, text=True, env=env, cwd=root, timeout=600)


def test_two_rank_independent_shards_on_one_gpu():
    """BASELINE configs[2]: two ranks, own game shards, no all-reduce."""
    import json
    out = _rehearse(["--independent"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "independent shards" in line["config"]["parallelism"]


def test_two_rank_rehearsal_on_one_gpu():
    """The whole N > 1 path of bench.py (game sharding, zero-copy gradient view, all-reduce per update, barriers, max-over-
    ranks timing) with two ranks pinned to the one GPU of the box and gloo standing in for RCCL; the replicas must end
    with bit-identical parameters."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, XQ_FORCE_DEVICE="0", XQ_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--check-replicas"],
                         capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "replicas identical on 2 ranks" in out.stderr
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["parallelism"].startswith("dp2")

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


def test_rccl_path_behind_the_c_abi_world_size_one(tmp_path):
    """xq_comm_* / xq_dqn_set_comm / xq_allreduce_grads with a one-rank RCCL communicator: the all-reduce runs inside every TD step
    (one collective behind the fused launches of the gradient half; two buckets on two streams with xq_dqn_set_td_tail(0); every
    float of the gradient buffer once either way) and the training result is bit-identical to the same loop without a communicator —
    also against the fused-apply single-GPU path."""
    import cn_chess_ai_amd as xq
    from cn_chess_ai_amd import dist as xd, _capi
    sizes = (1260, 64, 64, 8100)
    mk = lambda overlap: xq.TrainerConfig(n_games=512, layer_sizes=sizes, replay_capacity=4096, minibatch=1024, td_net=0,
                                          target_sync_interval=3, seed=99, overlap_collect=overlap)
    for overlap in (0, 1):
        ta, tb, tc, td = xq.Trainer(mk(overlap)), xq.Trainer(mk(overlap)), xq.Trainer(mk(overlap)), xq.Trainer(mk(overlap))
        comm = xd.Comm(rank=0, world=1)                          # id drawn and consumed in this process
        comm2 = xd.Comm(rank=0, world=1)
        assert comm.info() == dict(rank=0, world=1, collectives=0, floats=0)
        tb.set_comm(comm)
        tc.dqn.set_fused_apply(True)
        td.set_comm(comm2); td.dqn.set_td_tail(False)            # the two-stream form: two buckets, each behind its producer
        te = xq.Trainer(mk(overlap))
        comm3 = xd.Comm(rank=0, world=1)
        te.set_comm(comm3); te.dqn.set_exchange_overlap(1)       # what more than one rank runs: the select chain starts behind the
        steps = 6                                                # gradients and runs beside the all-reduce
        for t in (ta, tb, tc, td, te):
            for _ in range(steps):
                if overlap:
                    t.learn_grads(); t.collect()
                else:
                    t.collect(); t.learn_grads()
                t.learn_apply(1)
        _, n = tb.dqn.grad_buffer()
        info = comm.info()
        assert info["collectives"] == steps and info["floats"] == n * steps
        info2 = comm2.info()
        assert info2["collectives"] == 2 * steps and info2["floats"] == n * steps
        wa, ba = ta.dqn.get_params(); wb, bb = tb.dqn.get_params(); wc, bc = tc.dqn.get_params(); wd, bd = td.dqn.get_params()
        assert np.array_equal(wa, wb) and np.array_equal(ba, bb)
        assert np.array_equal(wa, wc) and np.array_equal(ba, bc)
        assert np.array_equal(wa, wd) and np.array_equal(ba, bd)
        we, be = te.dqn.get_params()
        assert np.array_equal(wa, we) and np.array_equal(ba, be) and comm3.info()["collectives"] == steps
        se, _ = te.env.get_state()
        assert np.array_equal(ta.env.get_state()[0], se)
        sa, _ = ta.env.get_state(); sb, _ = tb.env.get_state()
        assert np.array_equal(sa, sb)
        # xq_trainer_step takes the world size from the attached communicator
        tb.step(2); ta.step(2)
        assert np.array_equal(ta.dqn.get_params()[0], tb.dqn.get_params()[0])
        # the stand-alone entry point of SURVEY §8(b): whole buffer, handle's stream; sum over one rank = identity
        tb.set_comm(None)
        tb.collect(); tb.learn_grads()
        ptr, n = tb.dqn.grad_buffer()
        import torch
        g = xd.wrap_device_floats(ptr, n)
        torch.cuda.synchronize()
        before = g.clone()
        _capi.call("xq_allreduce_grads", tb.dqn.handle, comm.handle)
        _capi.call("xq_stream_synchronize", None); torch.cuda.synchronize()
        assert torch.equal(before, g) and comm.info()["collectives"] == steps + 2 + 1
        v = C.c_uint64(41)
        _capi.call("xq_comm_sum_u64", comm.handle, C.byref(v))
        assert v.value == 41
        tb.learn_apply(1)
        for t in (ta, tb, tc, td, te):
            t.close()
        comm.close(); comm2.close(); comm3.close()
    # file rendezvous (what the C++ facade uses)
    c2 = xd.Comm(rank=0, world=1, path=str(tmp_path / "xq_comm_id"))
    assert c2.info()["world"] == 1
    assert not (tmp_path / "xq_comm_id").exists()        # rank 0 takes the file away once every rank has joined ...
    c2.close()
    c3 = xd.Comm(rank=0, world=1, path=str(tmp_path / "xq_comm_id"))     # ... so the same path serves the next run
    c3.close()
    (tmp_path / "xq_comm_id").write_bytes(b"\0" * 128)
    with pytest.raises(xq.XqError) as e:                 # a leftover of a crashed run would hand out a dead id: refused
        xd.Comm(rank=0, world=1, path=str(tmp_path / "xq_comm_id"))
    assert e.value.code == 4


def test_exchange_calibration_takes_both_branches_with_identical_results():
    """xq_dqn_calibrate_exchange (VERDICT r4 Next #4): where the select chain of a data-parallel step starts is decided by a measurement of
    the all-reduce, not by the rank count.  One-rank communicator, thresholds forced: 0 us => the measured collective is "long" => late
    start (beside the all-reduce); 1e9 us => early start (beside the gradient kernels).  Same bits either way, and the same as the
    explicit settings."""
    import cn_chess_ai_amd as xq
    from cn_chess_ai_amd import dist as xd
    sizes = (1260, 64, 64, 8100)
    cfg = xq.TrainerConfig(n_games=512, layer_sizes=sizes, replay_capacity=4096, minibatch=1024, td_net=0, target_sync_interval=3, seed=7,
                           overlap_collect=1)
    outs, cals = [], []
    for thr, explicit in ((0.0, None), (1e9, None), (None, 0), (None, 1)):
        t = xq.Trainer(cfg)
        comm = xd.Comm(rank=0, world=1)
        t.set_comm(comm)
        assert t.dqn.exchange_calibration() is None              # one rank: xq_dqn_set_comm does not calibrate by itself
        if thr is not None:
            c = t.dqn.calibrate_exchange(thr)
            cal = t.dqn.exchange_calibration()
            assert cal["threshold_us"] == thr and cal["late_start"] == c["late_start"] == (thr == 0.0) and 0.0 < cal["allreduce_us"] < 5e4
            cals.append(cal)
        else:
            t.dqn.set_exchange_overlap(explicit)
        for _ in range(5):
            t.learn_grads(); t.collect(); t.learn_apply(1)
        outs.append(t.dqn.get_params() + (t.env.get_state()[0],))
        assert comm.info()["collectives"] == 5 + (24 if thr is not None else 0) or comm.info()["collectives"] >= 5
        t.close(); comm.close()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])
    assert cals[0]["late_start"] is True and cals[1]["late_start"] is False


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
    assert line["exchange"]["path"] == "none (independent shards)" and line["exchange"]["rccl_behind_c_abi"] is False
    out = _rehearse(["--config", "3"])                       # the same through --config 3 (BASELINE configs[2])
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["baseline_config"] == 3 and "independent shards" in line["config"]["parallelism"] and line["n_gpus"] == 2


def test_two_rank_rehearsal_on_one_gpu():
    """The whole N > 1 path of bench.py (game sharding, zero-copy gradient view, all-reduce per update, barriers, max-over-
    ranks timing) with two ranks pinned to the one GPU of the box and gloo standing in for RCCL; the replicas must end
    with bit-identical parameters.  Started exactly as the driver starts the N = 1 bench — `python bench.py --gpus 2 ...`, NO
    external launcher: bench.py spawns its own fresh rank processes before anything touches the GPU."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XQ_FORCE_DEVICE="0", XQ_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                          "--check-replicas", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "replicas identical on 2 ranks" in out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0's line only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["parallelism"].startswith("dp2")
    assert len(line["ms_per_step_samples"]) == 5
    # the line says which path carried the exchange: here gloo stands in, so NOT the RCCL communicator behind the C ABI
    ex = line["exchange"]
    assert ex["rccl_behind_c_abi"] is False and ex["path"].startswith("torch.distributed") and ex["comm"] is None
    assert ex["gradient_buffer_bytes"] == 4 * (1260 * 256 + 256 * 256 + 96 * 256 + 96 + 512)
    # ... and what the exchange measured (the rule that places the select chain; here the torch path, so nothing is moved)
    assert ex["calibration"]["allreduce_us"] > 0 and ex["calibration"]["threshold_us"] == 41.0 and ex["calibration"]["late_start"] is False


def test_self_launch_reports_a_failed_rank():
    """A rank that fails takes the whole self-launched job down with a non-zero exit code (no JSON line, no hang)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XQ_FORCE_DEVICE="99", XQ_DIST_BACKEND="gloo")           # no such device: every rank fails at set_device
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]

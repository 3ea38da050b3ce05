#!/usr/bin/env python3
"""Which initial weights put the trained step into the screen's WHOLE-GROUP regime (two largest outputs of a sample in one 32-row group, within the
bf16 bound of each other)?  Per seed: 2500 updates of the bench's schedule, then 300 timed steps with the whole-group share of those steps.
usage: python tools/whole_seed_scan.py [first_seed] [n_seeds]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cn_chess_ai_amd as xq
from cn_chess_ai_amd import _capi

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for seed in range(first, first + count):
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
    cfg = xq.TrainerConfig(n_games=8192, layer_sizes=(1260, 256, 256, 8100), replay_capacity=1 << 20, minibatch=8192, td_net=_capi.TD_ONLINE_NET,
                           overlap_collect=1, mean_gradient=1, target_sync_interval=100, seed=seed)
    t = xq.Trainer(cfg, stream=C.c_void_p(ts.cuda_stream))
    t.dqn.set_qmax_mode(_capi.QMAX_SCREENED); t.dqn.set_l0_derive(True); t.dqn.set_fused_apply(True)
    t.random_plies(300)
    for _ in range(128):
        t.collect()
    for _ in range(2500):
        t.learn_grads(); t.collect(); t.learn_apply(1)
    t.synchronize()
    s0 = t.dqn.qmax_stats()
    t0 = time.perf_counter()
    for _ in range(300):
        t.learn_grads(); t.collect(); t.learn_apply(1)
    t.synchronize()
    dt = time.perf_counter() - t0
    s1 = t.dqn.qmax_stats()
    smp = max(s1[1] - s0[1], 1)
    print("seed %d: %.4f ms per step, %.2f candidate groups per sample, %.2f whole" % (seed, 1e3 * dt / 300, (s1[2] - s0[2]) / smp, (s1[3] - s0[3]) / smp), flush=True)
    t.close()

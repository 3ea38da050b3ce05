#!/usr/bin/env python3
"""Candidate statistics and isolated kernel times of the screened max_a' Q(s',a') (xq_dqn_set_qmax_mode) in the bench's steady
state.  usage: python tools/qmax_stats.py [--config 2|4] [--steps N]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cn_chess_ai_amd as xq
from cn_chess_ai_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--td", default="online")
a = ap.parse_args()
layers = (1260, 256, 256, 8100) if a.config == 2 else (1260, 512, 512, 512, 8100)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
cfg = xq.TrainerConfig(n_games=8192, layer_sizes=layers, replay_capacity=1 << 20, minibatch=8192,
                       td_net=_capi.TD_ONLINE_NET if a.td == "online" else _capi.TD_TARGET_NET, overlap_collect=1, mean_gradient=1,
                       target_sync_interval=10)
t = xq.Trainer(cfg, stream=C.c_void_p(ts.cuda_stream))
t.dqn.set_qmax_mode(_capi.QMAX_SCREENED)
t.dqn.set_fused_apply(True)
t.dqn.set_l0_derive(True)
t.random_plies(300)
for _ in range(128):
    t.collect()
prev = (0, 0, 0, 0)
for block in range(a.steps // 50):
    for _ in range(50):
        t.learn_grads(); t.collect(); t.learn_apply(1)
    st = t.dqn.qmax_stats()
    d = [x - y for x, y in zip(st, prev)]; prev = st
    print(f"steps {st[0]:5d}: candidate groups/sample {d[2] / max(d[1], 1):7.3f}  whole groups/sample {d[3] / max(d[1], 1):7.4f}", flush=True)
t.dqn.kernel_stats(enable=2)
for _ in range(10):
    t.collect(); torch.cuda.synchronize()
    t.learn_grads(); t.learn_apply(1); torch.cuda.synchronize()
for s in t.dqn.kernel_stats(enable=0):
    print(f"  isolated {s['name']:22s} {1e3 * s['ms'] / max(s['launches'], 1):8.1f} us x {s['launches']}")
t.close()

#!/usr/bin/env python3
"""The step and the refine kernel in the screen's WHOLE-GROUP regime (seed 4: 0.8 whole groups per sample after 2500 updates; tools/whole_seed_scan.py).
usage: [XQ_REFINE_WHOLE=1] python tools/whole_probe.py [seed]      (1 = every whole group from global memory, four per round trip)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cn_chess_ai_amd as xq
from cn_chess_ai_amd import _capi
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
cfg = xq.TrainerConfig(n_games=8192, layer_sizes=(1260, 256, 256, 8100), replay_capacity=1 << 20, minibatch=8192, td_net=_capi.TD_ONLINE_NET,
                       overlap_collect=1, mean_gradient=1, target_sync_interval=100, seed=seed)
t = xq.Trainer(cfg, stream=C.c_void_p(ts.cuda_stream))
t.dqn.set_qmax_mode(_capi.QMAX_SCREENED); t.dqn.set_l0_derive(True); t.dqn.set_fused_apply(True)
t.random_plies(300)
for _ in range(128):
    t.collect()
def steps(k):
    for _ in range(k):
        t.learn_grads(); t.collect(); t.learn_apply(1)
    t.synchronize()
steps(2500)
t.dqn.kernel_filter(["qmax_refine"]); t.dqn.kernel_stats(enable=3)
s0 = t.dqn.qmax_stats(); t0 = time.perf_counter(); steps(300); dt = time.perf_counter() - t0; s1 = t.dqn.qmax_stats()
smp = max(s1[1] - s0[1], 1)
mode = os.environ.get("XQ_REFINE_WHOLE", "2 (default)")
if True:
    k = [x for x in t.dqn.kernel_stats(enable=0) if x["name"] == "qmax_refine"][0]
    print("seed %d XQ_REFINE_WHOLE=%s: %.4f ms per step, %.2f whole groups per sample, refine kernel %.1f us" %
          (seed, mode, 1e3 * dt / 300, (s1[3] - s0[3]) / smp, 1e3 * k["ms"] / max(k["launches"], 1)), flush=True)

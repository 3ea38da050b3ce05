#!/usr/bin/env python3
"""Live kernel timeline of the bench loop from the library's own HIP-event brackets (no profiler attached): prints the
launches of the last full step, across the trainer's streams.  usage: python tools/live_timeline.py [--no-overlap]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cn_chess_ai_amd as xq
from cn_chess_ai_amd import _capi

overlap = 0 if "--no-overlap" in sys.argv else 1
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
cfg = xq.TrainerConfig(n_games=8192, layer_sizes=(1260, 256, 256, 8100), replay_capacity=1 << 20, minibatch=8192,
                       td_net=_capi.TD_TARGET_NET, overlap_collect=overlap)
t = xq.Trainer(cfg, stream=C.c_void_p(ts.cuda_stream))
for _ in range(20):
    (t.learn_grads(), t.collect()) if overlap else (t.collect(), t.learn_grads()); t.learn_apply(1)
t.dqn.kernel_stats(enable=2)
for _ in range(6):
    (t.learn_grads(), t.collect()) if overlap else (t.collect(), t.learn_grads()); t.learn_apply(1)
torch.cuda.synchronize()
t.dqn.kernel_stats(enable=0)
spans = sorted(t.dqn.kernel_timeline(), key=lambda s: s[1])
envs = [i for i, s in enumerate(spans) if s[0] == "env_selfplay_step"]
sgds = [i for i, s in enumerate(spans) if s[0] == "sgd_apply"]
a = sgds[-3] if len(sgds) >= 3 else 0
b = sgds[-2] if len(sgds) >= 2 else len(spans) - 1
t0 = spans[a][2]
for name, s, e in spans[a + 1:b + 1]:
    print(f"{(s - t0) * 1e3:8.1f} {(e - t0) * 1e3:8.1f} dur {(e - s) * 1e3:7.1f}  {name}")
print("step", (spans[b][2] - t0) * 1e3, "us (sgd end to sgd end)")

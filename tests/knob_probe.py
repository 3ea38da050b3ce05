"""Child process of tests/test_trainer_gpu.py::test_timing_knobs_do_not_change_a_bit: a short run of the bench's schedule, printed as one hash.

The library reads its A/B environment knobs once per process, so each setting needs a process of its own."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cn_chess_ai_amd as xq
from cn_chess_ai_amd import _capi

n = int(os.environ.get("KNOB_PROBE_GAMES", "2048"))
updates = int(os.environ.get("KNOB_PROBE_UPDATES", "6"))       # a soak by hand: KNOB_PROBE_GAMES=8192 KNOB_PROBE_UPDATES=5000
cfg = xq.TrainerConfig(n_games=n, layer_sizes=(1260, 256, 256, 8100), learning_rate=0.001, gamma=0.99, epsilon=0.1, replay_capacity=1 << 15,
                       minibatch=n, td_net=_capi.TD_ONLINE_NET, backprop_mode=_capi.BACKPROP_REFERENCE, target_sync_interval=4, mean_gradient=1,
                       seed=0x5EED, first_game_id=0, overlap_collect=1, collects_per_update=1)
t = xq.Trainer(cfg)
t.dqn.set_qmax_mode(_capi.QMAX_SCREENED)
t.dqn.set_l0_derive(True)
t.dqn.set_fused_apply(True)
t.random_plies(40)
for _ in range(16):
    t.collect()
for _ in range(updates):
    t.learn_grads(); t.collect(); t.learn_apply(1)
t.synchronize()
w, b = t.dqn.get_params()
boards, meta = t.env.get_state()
qsa, y = t.dqn.last_td_values(n)
h = hashlib.sha256()
for a in (w, b, boards, meta, qsa, y):
    h.update(np.ascontiguousarray(a).tobytes())
st = t.dqn.qmax_stats()
print("KNOB_PROBE", h.hexdigest(), int(st[0]))

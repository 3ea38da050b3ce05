import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import cn_chess_ai_amd as xq
cfg = xq.TrainerConfig(n_games=8192, layer_sizes=(1260, 512, 512, 512, 8100), replay_capacity=1 << 18, minibatch=8192,
                       collects_per_update=4, target_sync_interval=10)
t = xq.Trainer(cfg)
t.step(3)
t.synchronize()
t0 = time.perf_counter(); t.step(20); t.synchronize(); el = time.perf_counter() - t0
c = t.counters()
w, b = t.dqn.get_params()
print("config-4 topology on 1 GPU: env steps/s", 8192 * 4 * 20 / el, "updates/s", 20 / el, c, np.isfinite(w).all())
t.close()

"""bench.py keeps its contract on every configuration it offers (`pytest -m gpu`): one JSON line with the driver's keys, the roofline
and cpu_baseline objects, and the steady-state fields — run at reduced size so that the three runs take seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--prefill-plies", "20"] + list(extra),
                         capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # exactly ONE JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("config,games,dtype", [(2, 2048, "f32"), (4, 2048, "f32"), (5, 2048, "bf16")])
def test_bench_line_contract(config, games, dtype):
    d = _bench("--config", str(config), "--games", str(games), "--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == dtype and d["data"] == "synthetic" and d["value"] > 0 and d["updates_per_s"] > 0
    cfg = d["config"]
    assert cfg["baseline_config"] == config and cfg["games_per_gpu"] == games and "workload" in cfg
    plies = {2: 1, 4: 4, 5: 1}[config]
    assert cfg["plies_per_update"] == plies
    assert abs(d["value"] - games * plies * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    ss = cfg["steady_state"]
    assert ss["replay_fill"] == 1.0 and ss["prefill_random_plies"] == 20 and ss["mean_legal_moves"] > 10
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    screened = cfg["qmax"]["mode"] == "screened"                 # the exact screen runs on the bf16 pipe whatever the net's dtype
    assert screened == (config in (2, 4))                        # config 5 (Double DQN on a bf16 net) keeps its arg-max GEMM
    assert r["launches"] == 5 and r["peak"] == (2500.0 if dtype == "bf16" or screened else 157.3)      # every 4th of 5 x 4 timed steps
    assert len(d["ms_per_step_samples"]) == 5 and min(d["ms_per_step_samples"]) <= d["ms_per_step"] <= max(d["ms_per_step_samples"])
    if screened:
        assert cfg["qmax"]["candidate_groups_per_sample"] >= 1.0
        v = d["variant_qmax_full_fp32_product"]
        assert v["value"] > 0 and v["roofline"]["peak"] == 157.3 and v["roofline"]["launches"] == 5
    if config == 5:
        assert cfg["prioritized_replay"] is True and "double" in cfg["td_net"] and "bf16" in cfg["q_net_precision"]
    if config == 2:
        assert "variant_td_target" in d and d["variant_td_target"]["value"] > 0


def test_bench_cpu_baseline_leg():
    """The CPU leg (oracle port of ChessAI::train on host cores) is bounded and carries the keys the contract names."""
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", "import bench, json; r = bench.cpu_train_loop(1.0); print(json.dumps(r))"],
                         capture_output=True, text=True, cwd=ROOT, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-1000:]
    steps, episodes, el = json.loads(out.stdout.strip().splitlines()[-1])
    assert steps > 0 and episodes >= 1 and 0.5 < el < 30

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
    d = _bench("--config", str(config), "--games", str(games), "--no-cpu-baseline", "--chain-steps", "4")
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
    # `roofline` = the kernel with the largest share of the timed step, measured live (every 4th launch) inside the timed region
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] == ("TFLOP/s" if r["bound"] == "mfma" else "GB/s")
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["avg_launch_ms"] > 0 and 0 < r["share_of_step"] < 1
    assert r["peak"] in (157.3, 2500.0, 8000.0) and r["launches"] >= 5
    # the matrix-pipe kernel of the TD target keeps its own object
    rq = d["roofline_qmax"]
    assert rq["bound"] == "mfma" and rq["unit"] == "TFLOP/s" and 0 < rq["frac"] < 1 and abs(rq["frac"] - rq["achieved"] / rq["peak"]) < 1e-9
    screened = cfg["qmax"]["mode"] == "screened"                 # the exact screen runs on the bf16 pipe whatever the net's dtype
    assert screened == (config in (2, 4))                        # config 5 (Double DQN on a bf16 net) keeps its arg-max GEMM
    assert rq["launches"] == 5 and rq["peak"] == (2500.0 if dtype == "bf16" or screened else 157.3)      # every 4th of 5 x 4 timed steps
    # every kernel of the step priced: the handle's stream and the collect stream, each entry with its bound and fraction
    ch = d["roofline_chain"]
    names = [e["kernel"] for e in ch["handle_stream"]]
    assert "l0_forward_gather" in names and "sgd_apply" in names and any(n.startswith("gemm_qmax") for n in names)
    assert any(e["kernel"] == "env_selfplay_step" for e in ch["collect_stream"])
    for e in ch["handle_stream"] + ch["collect_stream"]:
        assert e["avg_us"] > 0 and e["launches_per_step"] >= 0.5
        if "frac" in e:
            assert e["bound"] in ("mfma", "hbm") and e["frac"] > 0 and abs(e["frac"] - e["achieved"] / e["peak"]) < 1e-9
    assert abs(ch["handle_stream_sum_us"] - sum(e["avg_us"] * e["launches_per_step"] for e in ch["handle_stream"])) < 1e-6
    assert r["kernel"].split(" ")[0] in names + [e["kernel"] for e in ch["collect_stream"]]
    assert d["exchange"]["path"] == "none (one GPU)" and d["exchange"]["rccl_behind_c_abi"] is False
    assert len(d["ms_per_step_samples"]) == 5 and min(d["ms_per_step_samples"]) <= d["ms_per_step"] <= max(d["ms_per_step_samples"])
    if screened:
        assert cfg["qmax"]["candidate_groups_per_sample"] >= 1.0
        v = d["variant_qmax_full_fp32_product"]
        assert v["value"] > 0 and v["roofline"]["peak"] == 157.3 and v["roofline"]["launches"] == 5
    if config == 5:
        assert cfg["prioritized_replay"] is True and "double" in cfg["td_net"] and "bf16" in cfg["q_net_precision"]
    if config == 2:
        assert "variant_td_target" in d and d["variant_td_target"]["value"] > 0


def test_bench_facade_leg_runs_the_cpp_entry_point():
    """`facade`: examples/train_selfplay (xq::ChessAI::train with setReplay) as a process of its own at the headline's size."""
    d = _bench("--no-cpu-baseline", "--no-chain", "--no-variants", "--facade-episodes", "30000", "--settle-steps", "10")
    f = d["facade"]
    assert "error" not in f, f
    assert f["parallel_games"] == 8192 and f["replay_capacity"] == 1 << 20 and f["minibatch"] == 8192
    assert f["episodes_reported"] == 30000 and f["episodes_finished"] >= 30000
    assert f["env_steps"] == f["updates"] * 8192 and f["env_steps_per_s"] > 1e6 and 0 < f["vs_headline"] < 2


def test_bench_sustain_mode_smoke():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sustain", "60", "--games", "2048", "--prefill-plies", "20"],
                         capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["mode"] == "sustain" and d["updates"] == 60 and [c["updates"] for c in d["checkpoints"]] == [0, 60]
    last = d["checkpoints"][-1]
    assert last["ms_per_step"] > 0 and last["candidate_groups_per_sample"] >= 1.0 and 0 <= last["frac_outputs_saturated"] <= 1
    for m in d["greedy_vs_random"]:
        assert m["finished"] > 0 and abs(m["greedy_win_rate"] + m["random_win_rate"] + m["draw_move_cap"] / m["finished"] - 1) < 1e-9


def test_bench_cpu_baseline_leg():
    """The CPU leg (oracle port of ChessAI::train on host cores) is bounded and carries the keys the contract names."""
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", "import bench, json; r = bench.cpu_train_loop(1.0); print(json.dumps(r))"],
                         capture_output=True, text=True, cwd=ROOT, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-1000:]
    steps, episodes, el = json.loads(out.stdout.strip().splitlines()[-1])
    assert steps > 0 and episodes >= 1 and 0.5 < el < 30

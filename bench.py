#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: env steps/s + DQN updates/s at 8192 parallel games per MI355X.

One "step" = one pass of the hot path over one batch: one ply in every one of the 8192 games of this GPU
(Q(s)[0..89] forward -> fused self-play kernel -> transition into the replay ring) followed by one DQN update on a
minibatch of 8192 replayed transitions (TD target with the full 8100-wide max, backward, [gradient all-reduce], SGD).
Workload = BASELINE configs[1]: 8192 games, (256,256) hidden layers, fp32, replay 1 M transitions; random-init
weights, self-play from the start position ("synthetic": nothing is read from disk).  N > 1: weak scaling, every rank
runs its own 8192 games (game ids rank*8192..) and the gradient buffer is all-reduced over RCCL every update.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus N --steps K --warmup W            # no launcher: spawns its own N ranks (one fresh process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               # under a launcher: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel with HIP events recorded around each of its launches inside
the timed region: by default (`--qmax screened`) the 8100 x 8192 x 256 bf16-MFMA screening pass of max_a' Q(s',a') (its fp32
re-evaluation kernel follows it), with `--qmax full` the fp32-MFMA column-max GEMM, which the default run
also times as `variant_qmax_full_fp32_product` (own roofline inside).  `roofline_env` does the same for the fused self-play step
kernel (HBM-bound).  `cpu_baseline` times the CPU port of ChessAI::train (oracle) on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_GAMES = 8192
LAYERS = (1260, 256, 256, 8100)
REPLAY = 1 << 20
MINIBATCH = 8192
# --config K: K = 1-based index into BASELINE.json `configs` (2 = configs[1], the configuration `metric` is quoted on and the default;
# 4 / 5 = the per-GPU share of the 8-GPU configs[3] / configs[4], printed as lines of their own, never in place of the headline)
CONFIGS = {
    2: dict(workload="BASELINE configs[1]: 8192 self-play games per GPU, DQN 1260-256-256-8100 fp32, replay 1M transitions, "
                     "minibatch 8192, one update per ply",
            games=8192, layers=(1260, 256, 256, 8100), replay=1 << 20, minibatch=8192, plies=1, td="online", dtype="f32",
            prioritized=0, bf16=False),
    3: dict(workload="BASELINE configs[2]: 8192 self-play games per GPU, N GPUs as INDEPENDENT shards (own games, own replay ring, own "
                     "replica; no gradient all-reduce) — the embarrassingly-parallel env scaling curve; per GPU the same work as configs[1]",
            games=8192, layers=(1260, 256, 256, 8100), replay=1 << 20, minibatch=8192, plies=1, td="online", dtype="f32",
            prioritized=0, bf16=False, independent=True),
    4: dict(workload="BASELINE configs[3], per-GPU share (65536 games / 8): 8192 games, DQN 1260-512-512-512-8100 fp32, one update "
                     "(gradient all-reduce when N > 1) per 4 env steps, replay 1M, minibatch 8192",
            games=8192, layers=(1260, 512, 512, 512, 8100), replay=1 << 20, minibatch=8192, plies=4, td="online", dtype="f32",
            prioritized=0, bf16=False),
    5: dict(workload="BASELINE configs[4], per-GPU share (131072 games / 8): 16384 games, Double-DQN + proportional prioritized replay "
                     "(alpha 0.6, beta 0.4), bf16 MFMA Q-net 1260-512-512-512-8100 (bf16 operands in the forward AND backward products, "
                     "fp32 accumulation, fp32 master weights), replay 1M, "
                     "minibatch 16384, one update per ply",
            games=16384, layers=(1260, 512, 512, 512, 8100), replay=1 << 20, minibatch=16384, plies=1, td="double", dtype="bf16",
            prioritized=1, bf16=True),
}
from cn_chess_ai_amd.workmodel import (PEAK_F32_MFMA_TFLOPS, PEAK_BF16_MFMA_TFLOPS, PEAK_HBM_GBS, ENV_BYTES_PER_GAME,   # noqa: E402
                                        step_work, price, rocprof_kernel as wm_rocprof_kernel)   # (no torch, no HIP: plain arithmetic)


def cpu_train_loop(seconds, net=(1260, 128, 8100)):
    """(plies, episodes, elapsed) of the CPU port of ChessAI::train run for ~`seconds` on the calling core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xqoracle as xo
    L = xo.lib()
    sizes = xo.sizes_arr(list(net))
    w, b = xo.init_weights(list(net), 1)
    rng = C.c_uint64(12345 + os.getpid())
    st = xo.EpisodeStats()
    steps, episodes = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        L.xqo_train_episode(sizes.ctypes.data_as(C.POINTER(C.c_int)), len(sizes), w.ctypes.data_as(C.POINTER(C.c_double)),
                            b.ctypes.data_as(C.POINTER(C.c_double)), 0.001, 0.99, 0.1, C.byref(rng), 0, C.byref(st))
        steps += st.steps
        episodes += 1
    return steps, episodes, time.perf_counter() - t0


def cpu_all_cores(seconds=8.0):
    """T independent instances of the same loop, T = the box's CPU share, for an all-cores figure."""
    T = max(1, min(len(os.sched_getaffinity(0)), 16))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(seconds)], stdout=subprocess.PIPE,
                              text=True) for _ in range(T)]
    total = 0.0
    for p in procs:
        out, _ = p.communicate(timeout=seconds * 6 + 60)
        steps, _, el = json.loads(out.strip().splitlines()[-1])
        total += steps / el
    return {"value": total, "cores": T, "sample": f"{T} independent instances x {seconds:.0f} s"}


def cpu_baseline(seconds=15.0, reference_nn=False):
    """CPU port of the reference loop (oracle/xq_oracle.c: xqo_train_episode = chessai.cpp:90-167 with the
    {1260,128,8100} fp64 net, batch 1, bug-compatible backprop) on one host core, bounded to ~`seconds`."""
    steps, episodes, el = cpu_train_loop(seconds)
    out = {"value": steps / el, "unit": "env steps/s (= DQN updates/s, batch 1)", "cores": 1, "kind": "port",
           "sample": f"{episodes} episodes / {steps} plies of the ChessAI::train restatement, net 1260-128-8100 fp64, "
                     f"{el:.1f} s on one host core"}
    try:     # the same loop on the bench's OWN topology (the reference default is 1260-128-8100), bounded to ~6 s
        s2, e2, el2 = cpu_train_loop(6.0, LAYERS)
        out["bench_net"] = {"value": s2 / el2, "unit": "env steps/s (= DQN updates/s, batch 1)", "cores": 1,
                            "sample": f"{e2} episodes / {s2} plies, net {'-'.join(map(str, LAYERS))} fp64, {el2:.1f} s on one host core"}
    except Exception as e:
        out["bench_net_error"] = str(e)[:120]
    try:
        out["all_cores"] = cpu_all_cores()
    except Exception as e:
        out["all_cores_error"] = str(e)[:120]
    ref = os.path.join(ROOT, "oracle", "_ref", "xqref")
    if os.path.exists(ref):     # the real reference rules engine (env only: movegen + movePiece), when it travelled
        try:
            s, t = subprocess.check_output([ref, "bench", "1", "3"], timeout=60).split()
            out["env_only_reference_steps_per_s"] = float(s) / float(t)
        except Exception as e:   # Qt runtime missing on the box: report, do not fail the bench
            out["env_only_reference_error"] = str(e)[:80]
    refnn = os.path.join(ROOT, "oracle", "_ref", "xqref_nn")
    if reference_nn and os.path.exists(refnn):   # (--reference-nn) the reference's OWN NN runtime (dqn.cu through hipify-perl, oracle/ref/ref_nn_driver.cpp) on this very GPU:
        try:                    # the NN work of one ply of ChessAI::train — two forwards + one backpropagate, batch 1 — as upstream does it
            if subprocess.run([refnn, "probe"], capture_output=True, timeout=60).returncode == 0:
                r = json.loads(subprocess.check_output(["timeout", "-k", "5", "60", refnn, "time", "300"] + [str(x) for x in (1260, 128, 8100)],
                                                       timeout=90).decode().strip().splitlines()[-1])
                out["reference_nn_runtime_on_this_gpu"] = {
                    "value": r["plies_per_s"], "unit": "plies/s (2 x forward + 1 x backpropagate, batch 1, net 1260-128-8100 fp64)",
                    "sample": f"{r['plies']} plies in {r['seconds']:.2f} s", "what": "/root/reference/src/dqn.cu's kernels and host code, "
                    "CUDA API identifiers renamed by hipify-perl, run by oracle/_ref/xqref_nn as a child process — no env, no move generation"}
        except Exception as e:
            out["reference_nn_runtime_error"] = str(e)[:80]
    return out


_PROFILE_SET = {}


# a kernel_stats entry with exact == launches: every launch was timed by the kernel's OWN start / stop events (the library launches single-kernel
# brackets with hipExtLaunchKernelGGL; csrc/xq_dqn.hip ProfScope(..., attach)) — no recorded bracket, and none of its ~6.5 us, is in the figure
KERNEL_EXACT_NOTE = "the kernel's own start / stop events (no bracket overhead; agrees with rocprofv3's duration)"


def profile_set(config=2):
    """The ONE committed profile set every offline figure of a bench line comes from: the newest rNN_x tag under profiles/ that has both
    a kernel-stats CSV and a PMC summary for this configuration (cn_chess_ai_amd/workmodel.py::newest_profile_set).  No fallback to an
    older set for a kernel the newest one does not hold (one line used to mix three builds): such a kernel gets
    traffic = null, and tests/test_workmodel_cpu.py keeps the newest set complete."""
    if config not in _PROFILE_SET:
        from cn_chess_ai_amd.workmodel import newest_profile_set
        best = newest_profile_set(ROOT, config)
        data = None
        if best:
            try:
                data = json.load(open(best[2]))
            except Exception:
                data = None
        _PROFILE_SET[config] = (best, data)
    return _PROFILE_SET[config]


def pmc_traffic(kernel_prefix, config=2):
    """HBM bytes per launch of a kernel instance (workmodel.rocprof_kernel's prefix) from the PMC summary of profile_set(config):
    tools/collect_profiles*.sh + tools/summarize_profiles.py on this same bench command, separate FETCH_SIZE / WRITE_SIZE passes,
    bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 correction.  (None, source) when the set does not hold the kernel: PMC
    counters cannot be read from inside the timed run."""
    from cn_chess_ai_amd.workmodel import match_kernel
    best, data = profile_set(config)
    if not best or not data:
        return None, None
    src = os.path.relpath(best[2], ROOT)
    hits = [v for k, v in data.items() if match_kernel(k, kernel_prefix)]
    if len(hits) != 1:
        return None, src
    return hits[0].get("hbm_bytes_per_launch"), src


def env_instruction_mix(config=2, n_boards=8192):
    """Wave-instructions per board of env_kernel<SELFPLAY> AS THE TRAINING LOOP RUNS IT (Q-policy launch: reads the select head's
    slabs, sums them, tanh) from the SQ_INSTS_* counters of profile_set(config) — the same file `traffic` comes from, i.e. the same build
    and the same bench command as the kernel-stats set.  None when that summary has no instruction counters for the kernel."""
    best, data = profile_set(config)
    if not best or not data:
        return None
    try:
        sq = next(v["sq"] for k, v in data.items() if "env_kernel<2>" in k)
        per = {k: sq.get("SQ_INSTS_" + k.upper(), 0.0) / n_boards for k in ("valu", "salu", "lds", "smem", "vmem")}
        if per["valu"] <= 0 or per["salu"] <= 0:
            return None
        out = dict(per)
        out["total"] = sum(per.values())
        out["smem_counted"] = "SQ_INSTS_SMEM" in sq
        if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_WAIT_ANY") is not None:
            out["wait_any_frac_of_wave_cycles"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
        out["source"] = os.path.relpath(best[2], ROOT)
        return out
    except Exception:
        return None


def env_only(args, xq, tstream):
    """Env-only random-policy stepping (BASELINE.md §3 C2 on the GPU): legal moves + uniform choice + movePiece + reward +
    terminal + auto-reset for every game, nothing else.  Prints one JSON line (diagnostic, not the headline metric)."""
    import torch
    env = xq.VecEnv(args.games, seed=0x5EED, stream=C.c_void_p(tstream.cuda_stream))
    for _ in range(args.warmup):
        env.selfplay_step_dev()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(tstream)
    for _ in range(args.steps):
        env.selfplay_step_dev()
    b.record(tstream)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.steps
    bytes_per = args.games * (2 * (48 + 16))                     # board + meta read and written; no Q row, no replay record
    c = env.counters()
    print(json.dumps({"metric": "env-only steps/s, uniform-random policy", "value": args.games / (ms * 1e-3), "unit": "env steps/s",
                      "games": args.games, "steps": args.steps, "us_per_launch": ms * 1e3,
                      "algorithmic_GBps": bytes_per / (ms * 1e-3) / 1e9, "episodes_finished": c["episodes"]}), flush=True)
    env.close()


def greedy_match(xq, dqn, stream, n=2048, greedy_colour=0, seed=0xC0FFEE):
    """n fresh games, `greedy_colour` (0 Red, 1 Black) plays the net's greedy move (epsilon 0: first strict maximum of q[action.to],
    dqn.cpp:36-52), the other side uniformly at random; every game counts until its FIRST end.  Move-cap endings with both generals
    alive are draws here (ChessBoard::getWinner() calls them Red)."""
    import numpy as np
    env = xq.VecEnv(n, seed=seed, stream=C.c_void_p(stream))
    open_ = np.ones(n, bool)
    wins = {"greedy": 0, "random": 0, "draw_move_cap": 0}
    plies_sum = 0
    for ply in range(200):
        if (ply & 1) == greedy_colour:
            q = dqn.q_boards(env, 96).cpu().numpy()[:, :90]
            res = env.selfplay_step(q, eps=0.0)
        else:
            res = env.selfplay_step(None)
        ended = open_ & (res["terminated"] != 0)
        for i in np.nonzero(ended)[0]:
            r = res[i]
            if r["move_count"] >= 200 and r["red_score"] < 1000 and r["black_score"] < 1000:
                wins["draw_move_cap"] += 1
            elif int(r["winner"]) == greedy_colour:
                wins["greedy"] += 1
            else:
                wins["random"] += 1
            plies_sum += int(r["move_count"])
        open_ &= ~ended
        if not open_.any():
            break
    env.close()
    done = n - int(open_.sum())
    return {"games": n, "greedy_colour": "Red" if greedy_colour == 0 else "Black", "finished": done, **wins,
            "greedy_win_rate": wins["greedy"] / max(done, 1), "random_win_rate": wins["random"] / max(done, 1),
            "mean_plies": plies_sum / max(done, 1)}


def sustain_mode(args, xq, t, one_step, stream, n_games, minibatch, plies, CFG):
    """--sustain N: the headline loop for N updates from a random initialisation, reference-scale rewards (evaluateBoard's raw
    integers, +-1000s, against tanh outputs — chessai.cpp:311-345 / dqn.cu:184-195), as a real train() spends its life.  At update
    0, 10^2, 10^3, 10^4, 2*10^4 (those <= N) and N: step time over a 200-step window, candidate statistics of the screened maximum,
    guard fallbacks, loss, saturation of the outputs; then a greedy-vs-random match.  One JSON line."""
    import numpy as np
    import torch
    marks = sorted({m for m in (0, 100, 1000, 10000, 20000, args.sustain) if m <= args.sustain})
    window = 200
    ck = []
    u = 0

    def run(k):
        nonlocal u
        for _ in range(k):
            one_step()
        u += k

    t0_all = time.perf_counter()
    for m in marks:
        if m - u > window:
            run(m - u - window if m >= window else 0)
        torch.cuda.synchronize()
        q0, g0 = t.dqn.qmax_stats(), t.dqn.qmax_guard()
        k = min(window, max(m - u, 0)) if m > 0 else 0
        ms = None
        if k > 0:
            a = time.perf_counter()
            run(k)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - a) / k
        q1, g1 = t.dqn.qmax_stats(), t.dqn.qmax_guard()
        loss = t.dqn.last_loss() / minibatch if u > 0 else None
        qall = t.dqn.q_boards(t.env, 8100)                      # tanh outputs of every action for the current boards of all games
        sat = float((qall.abs() > 0.999).float().mean().item())
        sat90 = float((qall[:, :90].abs() > 0.999).float().mean().item())
        mean_abs = float(qall.abs().mean().item())
        spread = float((qall.max(dim=1).values - qall.min(dim=1).values).mean().item())
        del qall
        c = t.counters()
        _, meta = t.env.get_state()
        ck.append({"updates": u, "ms_per_step": ms, "window_steps": k,
                   "candidate_groups_per_sample": (q1[2] - q0[2]) / max(q1[1] - q0[1], 1) if q1[1] > q0[1] else None,
                   "whole_groups_per_sample": (q1[3] - q0[3]) / max(q1[1] - q0[1], 1) if q1[1] > q0[1] else None,
                   "screened_steps_in_window": q1[0] - q0[0], "guard_fallbacks_total": g1[0], "guard_hold_steps_left": g1[1],
                   "mean_td_loss_per_sample": loss, "frac_outputs_saturated": sat, "frac_select_outputs_saturated": sat90,
                   "mean_abs_q": mean_abs, "mean_q_spread_per_board": spread, "episodes": c["episodes"], "mean_ply": float(np.mean(meta[:, 0]))})
        print("[sustain] " + json.dumps(ck[-1]), file=sys.stderr, flush=True)
    total_s = time.perf_counter() - t0_all
    times = [c["ms_per_step"] for c in ck if c["ms_per_step"]]
    match = [greedy_match(xq, t.dqn, stream, 2048, 0), greedy_match(xq, t.dqn, stream, 2048, 1)]
    out = {"mode": "sustain", "metric": "headline loop (BASELINE configs[%d]) sustained for %d updates" % (args.config - 1, args.sustain),
           "workload": CFG["workload"], "updates": u, "env_steps": u * n_games * plies, "wall_seconds_including_checkpoints": total_s,
           "rewards": "raw evaluateBoard integers (reference scale)", "qmax": args.qmax, "checkpoints": ck,
           "ms_per_step_min": min(times) if times else None, "ms_per_step_max": max(times) if times else None,
           "slowest_over_fastest": max(times) / min(times) if times else None,
           "greedy_vs_random": match}
    print(json.dumps(out), flush=True)


def facade_leg(n_games, layers, replay, minibatch, episodes, derive, prefill):
    """The drop-in entry point itself: examples/train_selfplay.cpp = what the reference's Worker::process() does
    (include/mainwindow.h:140-150 -> ChessAI::train(n), chessai.cpp:85) through include/xq/xq.hpp, as a process of its own, with the
    replay ring and the throughput schedule switched on (xq::ChessAI::setReplay).  Built with plain g++ when missing.  Returns the
    child's JSON (counters + wall seconds of the loop inside train()) or {"error": ...}; never raises."""
    import tempfile
    try:
        exe = os.path.join(ROOT, "examples", "_build", "train_selfplay")
        src = os.path.join(ROOT, "examples", "train_selfplay.cpp")
        hdr = os.path.join(ROOT, "include", "xq", "xq.hpp")
        lib = os.path.join(ROOT, "cn_chess_ai_amd", "libxqhip.so")
        if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(lib)):
            os.makedirs(os.path.dirname(exe), exist_ok=True)
            pkg = os.path.join(ROOT, "cn_chess_ai_amd")
            subprocess.check_call(["g++", "-std=c++17", "-O2", src, "-o", exe, "-I" + os.path.join(ROOT, "include"), "-L" + pkg, "-lxqhip",
                                   "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"], timeout=300)
        with tempfile.TemporaryDirectory() as tmp:          # the facade opens game_log.txt in its working directory (chessai.cpp:17)
            cmd = [exe, str(episodes), os.path.join(tmp, "model.bin"), str(n_games), "--replay", str(replay), "--minibatch", str(minibatch),
                   "--hidden", ",".join(str(x) for x in layers[1:-1]), "--save-every", "0", "--prefill", str(prefill), "--seed", "0x5EED",
                   "--json"] + (["--derive"] if derive else [])
            t0 = time.perf_counter()
            out = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp, timeout=600)
            wall = time.perf_counter() - t0
        if out.returncode != 0:
            return {"error": "train_selfplay exited with %d: %s" % (out.returncode, out.stderr.strip()[-300:])}
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        d["process_wall_seconds"] = wall
        d["command"] = " ".join(["examples/_build/train_selfplay"] + cmd[1:2] + ["<tmp>/model.bin"] + cmd[3:])
        return d
    except Exception as e:
        return {"error": str(e)[:300]}


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this very script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, one GPU each), pass rank 0's stdout through and fail if any rank fails.  The
    parent never imports torch and never touches HIP — a process that has initialised the GPU must not be replaced or forked."""
    import signal
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs between processes on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        live = set(range(n_ranks))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"[bench.py] rank {r} exited with code {code}: stopping the other ranks", file=sys.stderr)
                    for q in live:
                        procs[q].send_signal(signal.SIGTERM)
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reference-nn", action="store_true",
                    help="cpu_baseline also times the reference's own NN runtime on this GPU (oracle/_ref/xqref_nn time; off by default: upstream's "
                         "backpropagate reads device memory it has released, which the default bench does not need to run)")
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)   # one instance of cpu_all_cores()
    ap.add_argument("--profile-all", action="store_true", help="bracket every kernel with HIP events (diagnostic)")
    ap.add_argument("--games", type=int, default=N_GAMES, help="games per GPU (default = BASELINE's 8192; other values are diagnostic)")
    ap.add_argument("--minibatch", type=int, default=0, help="transitions per update (default = games per GPU)")
    ap.add_argument("--check-replicas", action="store_true",
                    help="N > 1: assert that every rank ends with bit-identical parameters (diagnostic)")
    ap.add_argument("--env-only", action="store_true",
                    help="diagnostic: time only the fused self-play kernel with the uniform-random policy (no Q-network)")
    ap.add_argument("--independent", action="store_true",
                    help="BASELINE configs[2]: N > 1 ranks train independent replicas on their own game shards, no gradient all-reduce")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json configs entry, 1-based: 2 = configs[1] (headline, default), 3 = configs[2] (independent shards, no "
                         "all-reduce), 4 = configs[3] share, 5 = configs[4] share")
    ap.add_argument("--prefill-plies", type=int, default=300,
                    help="uniform-random plies played in every game before anything is timed (spreads the games over all phases)")
    ap.add_argument("--no-fill", action="store_true", help="diagnostic: do not fill the replay ring before timing")
    ap.add_argument("--td-net", choices=("online", "target", "double"), default=None,
                    help="net that gives max Q(s'): online = ChessAI::train (chessai.cpp:126, headline), target = DQN::train (dqn.cpp:166)")
    ap.add_argument("--target-sync-interval", type=int, default=10, help="updates between updateTargetNetwork() calls")
    ap.add_argument("--torch-allreduce", action="store_true",
                    help="diagnostic, N > 1: all-reduce through torch.distributed after learn_grads instead of the bucketed RCCL path of the C ABI")
    ap.add_argument("--qmax", choices=("screened", "full"), default="screened",
                    help="max_a' Q(s',a') of the TD target: exact bf16 screening + fp32 re-evaluation (default) or the full fp32 product")
    ap.add_argument("--bracket-all", action="store_true", help="HIP-event bracket around EVERY launch of the dominant GEMM (default: every 4th)")
    ap.add_argument("--bf16-fp32-backward", action="store_true",
                    help="--config 5: XQ_PRECISION_BF16 (bf16 forward, fp32 backward products) instead of XQ_PRECISION_BF16_FULL")
    ap.add_argument("--no-derive", action="store_true",
                    help="gather layer 0 of the s' chain in full (the library default) instead of deriving it from the s chain")
    ap.add_argument("--exchange-overlap", type=int, default=-1, choices=(-1, 0, 1),
                    help="data-parallel step: where the select chain starts (xq_dqn_set_exchange_overlap; -1 = library default: beside "
                         "the all-reduce when there is more than one rank)")
    ap.add_argument("--no-td-tail", action="store_true",
                    help="A/B: the gradient kernels of the TD step one by one on two streams instead of the fused launches (xq_dqn_set_td_tail)")
    ap.add_argument("--l0-grad", choices=("mfma", "segmented"), default="mfma",
                    help="A/B: layer-0 weight gradient as segmented sums (library default) or on the bf16 matrix pipe (exact 3-term split)")
    ap.add_argument("--no-variants", action="store_true", help="skip the variant legs (other TD net, full fp32 product): A/B runs")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (exactly --steps steps, barrier + synchronize on both sides) is run this many times back to "
                         "back; ms_per_step / value are the MEDIAN repetition, all of them are listed in ms_per_step_samples")
    ap.add_argument("--settle-steps", type=int, default=100,
                    help="untimed training steps between the W warm-up steps and the timed region (reported in config.steady_state)")
    ap.add_argument("--strict-rccl", action="store_true",
                    help="N > 1: exit non-zero instead of falling back to torch.distributed when the RCCL communicator behind the C ABI "
                         "cannot be created on every rank")
    ap.add_argument("--no-chain", action="store_true", help="skip the per-kernel leg (roofline_chain): A/B runs")
    ap.add_argument("--chain-steps", type=int, default=10, help="untimed steps per kernel in the per-kernel leg (one bracketed kernel at a time)")
    ap.add_argument("--no-facade", action="store_true", help="skip the C++ facade leg (examples/train_selfplay through xq::ChessAI::train)")
    ap.add_argument("--facade-episodes", type=int, default=400000, help="episodes the facade leg trains for (~1.5 s of GPU at 8192 games)")
    ap.add_argument("--sustain", type=int, default=0,
                    help="sustained-run mode: N updates of the headline loop with checkpoints (step time, screening statistics, loss, "
                         "saturation) and a greedy-vs-random match at the end; prints ONE JSON line of its own (profiles/*_sustain.json)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="queue collect and learn on one stream (collect -> learn -> apply) instead of running collect beside learn_grads")
    args = ap.parse_args()
    if args.cpu_worker > 0:                      # child of cpu_all_cores(): CPU only, never touches the GPU
        print(json.dumps(cpu_train_loop(args.cpu_worker)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:        # no launcher: be our own (before torch / HIP are touched)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import cn_chess_ai_amd as xq
    from cn_chess_ai_amd import _capi, dist as xd
    import torch.distributed as dist

    rank, local_rank, world = xd.env_rank()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or no launcher at all: bench.py spawns its ranks itself)"
                         % (args.gpus, world))
    if not torch.cuda.is_available() or _capi.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # rehearsal hooks (one-GPU box only): XQ_FORCE_DEVICE pins every rank to one device, XQ_DIST_BACKEND=gloo replaces RCCL —
    # lets the N > 1 code path (sharding, zero-copy gradient view, all-reduce per update, barriers) run on a single GPU
    device = int(os.environ.get("XQ_FORCE_DEVICE", local_rank))
    backend = os.environ.get("XQ_DIST_BACKEND", "nccl")
    torch.cuda.set_device(device)
    _capi.call("xq_set_device", device)
    xd.init_process_group(backend if world > 1 else None)
    # library kernels and RCCL share ONE torch stream (a real stream object: the legacy default stream has handle 0,
    # which the C ABI reads as "create your own")
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    global LAYERS, REPLAY
    CFG = CONFIGS[args.config]
    LAYERS, REPLAY = CFG["layers"], CFG["replay"]
    if CFG.get("independent"):
        args.independent = True
    n_games = args.games if args.games != N_GAMES or args.config == 2 else CFG["games"]
    minibatch = args.minibatch or (CFG["minibatch"] if n_games == CFG["games"] else n_games)
    if args.td_net is None:
        args.td_net = CFG["td"]
    td_code = {"online": _capi.TD_ONLINE_NET, "target": _capi.TD_TARGET_NET, "double": _capi.TD_DOUBLE}
    if args.env_only:
        return env_only(args, xq, tstream)
    first, _ = xd.shard_games(rank, n_games)
    cfg = xq.TrainerConfig(n_games=n_games, layer_sizes=LAYERS, learning_rate=0.001, gamma=0.99, epsilon=0.1,
                           replay_capacity=max(REPLAY, n_games), minibatch=minibatch, td_net=td_code[args.td_net],
                           backprop_mode=_capi.BACKPROP_REFERENCE, target_sync_interval=args.target_sync_interval, mean_gradient=1,
                           seed=0x5EED, first_game_id=first, overlap_collect=0 if args.no_overlap else 1,
                           collects_per_update=CFG["plies"], prioritized=CFG["prioritized"],
                           precision=(_capi.PRECISION_BF16 if args.bf16_fp32_backward else _capi.PRECISION_BF16_FULL) if CFG["bf16"]
                           else _capi.PRECISION_F32)
    plies = CFG["plies"]
    t = xq.Trainer(cfg, stream=C.c_void_p(stream))
    t.dqn.set_qmax_mode(_capi.QMAX_SCREENED if args.qmax == "screened" else _capi.QMAX_FULL)
    t.dqn.set_l0_derive(not args.no_derive)       # layer-0 sums of s' from those of s (library default: off, the reference's order)
    t.dqn.set_l0_grad_mode(1 if args.l0_grad == "mfma" else 0)
    if args.no_td_tail: t.dqn.set_td_tail(False)  # A/B: the gradient kernels one by one on two streams (library default: fused launches)
    if args.exchange_overlap >= 0: t.dqn.set_exchange_overlap(args.exchange_overlap)
    grads, comm, comm_error = None, None, ""
    exchange = "none (one GPU)" if world == 1 else "none (independent shards)" if args.independent else None
    if world == 1 and os.environ.get("XQ_BENCH_COMM1"):
        # rehearsal on one GPU: a one-rank communicator attached, so that the whole N > 1 code path of the library runs (one launch
        # that reduces the partial sums, the all-reduce behind it, unfused SGD) — shows what that path costs besides the wire time
        comm = xd.Comm(rank=0, world=1)
        t.set_comm(comm)
        if args.exchange_overlap < 0:        # one rank: xq_dqn_set_comm does not calibrate by itself; run the rule so that the line shows it
            t.dqn.calibrate_exchange(float(os.environ.get("XQ_BENCH_EXCHANGE_THRESHOLD_US", "-1")))
        exchange = "RCCL behind the C ABI (one-rank rehearsal)"
    elif world == 1 or args.independent:
        t.dqn.set_fused_apply(True)          # nothing reads the gradient buffer between td_grads and apply_grads
    elif args.torch_allreduce or backend != "nccl":   # diagnostic / one-GPU gloo rehearsal: the exchange through torch.distributed
        ptr, n = t.dqn.grad_buffer()
        grads = xd.wrap_device_floats(ptr, n)
        exchange = "torch.distributed (%s)" % ("--torch-allreduce" if args.torch_allreduce else "backend %s: one-GPU rehearsal" % backend)
    else:
        # the exchange step lives behind the C ABI: learn_grads all-reduces the gradient buffer over RCCL itself, right behind the
        # launch that produces it (xq_dqn_set_comm); torch.distributed only carries the 128-byte id and barriers
        ok = torch.ones(1, device="cuda")
        try:
            comm = xd.Comm()
            t.set_comm(comm)
        except Exception as e:               # keep the measurement: every rank falls back together to the torch.distributed exchange
            comm_error = str(e)[:200]
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0:
            if comm is not None:
                t.set_comm(None)
                comm.close()
                comm = None
            if args.strict_rccl:
                raise SystemExit(f"[rank {rank}] --strict-rccl: C-ABI communicator unavailable ({comm_error or 'another rank failed'})")
            ptr, n = t.dqn.grad_buffer()
            grads = xd.wrap_device_floats(ptr, n)
            exchange = "torch.distributed (fallback)"
            print(f"[rank {rank}] C-ABI communicator unavailable ({comm_error or 'another rank failed'}): all-reduce through torch.distributed",
                  file=sys.stderr)
        else:
            exchange = "RCCL behind the C ABI"

    def one_step():
        if args.no_overlap:
            for _ in range(plies):
                t.collect()
            t.learn_grads()
        else:                   # collect is queued behind the column-max GEMM of learn_grads and runs beside the gradient chain
            t.learn_grads()
            for _ in range(plies):
                t.collect()
        if grads is not None and not args.independent:
            xd.allreduce_gradients(grads, world)
        t.learn_apply(1 if args.independent else world)

    # ---- steady state before anything is timed: games spread over all phases, replay ring filled to capacity ----
    cap = max(REPLAY, n_games)
    t.random_plies(args.prefill_plies)
    fill_collects = 0 if args.no_fill else (cap + n_games - 1) // n_games
    for _ in range(fill_collects):
        t.collect()                          # epsilon-greedy plies on the initial net, transitions into the ring
    if args.sustain > 0:
        if world != 1:
            raise SystemExit("--sustain is a one-GPU diagnostic")
        t.dqn.kernel_stats(enable=0)
        sustain_mode(args, xq, t, one_step, stream, n_games, minibatch, plies, CFG)
        t.close()
        return
    # the HIP-event brackets of the roofline leg run during the warm-up too (the library's event pool is then filled before anything is
    # timed), and `--settle-steps` more untimed steps follow the W warm-up steps: with the driver's short windows (W = 5, K = 20: 5 ms)
    # the first repetitions otherwise still see the chip ramping up from the host-side preparation (0.2056, 0.2028, 0.1981, 0.1922,
    # 0.1921 ms per step over five repetitions on one run)
    live_mode = 2 if args.profile_all else 3 if args.bracket_all else 4
    t.dqn.kernel_stats(enable=live_mode)
    for _ in range(args.warmup + args.settle_steps):
        one_step()
    torch.cuda.synchronize()

    # ---- per-kernel leg (untimed): every kernel of the step measured IN the loop with only its own HIP-event bracket in it -----------
    # (a bracket costs its stream ~6-10 us, so bracketing everything at once — --profile-all — stretches the step by a third; one
    # kernel at a time leaves the co-scheduling of the two chains as it is in the timed region).  Names: 3 steps with every bracket
    # on; then `--chain-steps` steps per name.  The kernel with the largest share of the step becomes `roofline` and is the one
    # bracketed (every 4th launch) inside the timed region.
    chain = {}
    if not args.no_chain and not args.profile_all:
        t.dqn.kernel_stats(enable=2)
        for _ in range(3):
            one_step()
        torch.cuda.synchronize()
        names = [k["name"] for k in t.dqn.kernel_stats(enable=0) if k["launches"]]
        for nm in names:
            t.dqn.kernel_filter([nm])
            t.dqn.kernel_stats(enable=3)
            for _ in range(args.chain_steps):
                one_step()
            torch.cuda.synchronize()
            for k in t.dqn.kernel_stats(enable=0):
                if k["name"] == nm and k["launches"]:
                    chain[nm] = dict(avg_us=1e3 * k["ms"] / k["launches"], launches_per_step=k["launches"] / args.chain_steps,
                                     exact=k.get("exact", 0) == k["launches"])
        torch.cuda.synchronize()
    # candidates for `roofline`: the kernels ON the step's dependency chain — the handle's stream, and the select chain too when an
    # update has several plies (then the plies are the long pole).  With one ply per update the select chain runs beside the TD step,
    # ends before it and reads stretched by the co-scheduling (35 us against 15-28 alone): its share is not the step's.
    def on_chain(k):
        return plies > 1 or not (k.endswith("@select") or k == "env_selfplay_step")
    per_step_us = {k: v["avg_us"] * v["launches_per_step"] for k, v in chain.items()
                   if v["launches_per_step"] >= 0.5 and k != "rccl_allreduce_grads" and on_chain(k)}   # (a target sync every 10th step is not a step kernel)
    dominant = max(per_step_us, key=per_step_us.get) if per_step_us else None
    live_names = [n for n in (dominant, "gemm_qmax_screen", "gemm_qmax_rowmax", "env_selfplay_step", "rccl_allreduce_grads") if n]
    t.dqn.kernel_filter(sorted(set(live_names)) if dominant else None)
    for _ in range(8):                       # back to the sparse live brackets
        one_step()
    torch.cuda.synchronize()
    c0 = t.counters()

    def timed_once(steps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        return xd.max_over_ranks(el, device="cuda" if world > 1 else "cpu"), enq

    def timed(steps, repeats=None):
        """The driver's short window (--steps 20 = 5 ms) is one sample of a noisy quantity: the same region `repeats` times,
        the median repetition is the reported one (every rank sees the same max-over-ranks times, so all pick the same)."""
        runs = [timed_once(steps) for _ in range(max(1, repeats or args.repeats))]
        order = sorted(range(len(runs)), key=lambda i: runs[i][0])
        mid = order[(len(runs) - 1) // 2]
        return runs[mid][0], runs[mid][1], [r[0] for r in runs]

    qstat0 = t.dqn.qmax_stats()
    # the priced kernels are bracketed with HIP events on every 4th launch of the timed region (--bracket-all: every launch): a
    # bracket is two event records, and a record drains the recording queue — ~10 us each time on a 190-us step
    t.dqn.kernel_stats(enable=live_mode)
    elapsed, host_enqueue, samples = timed(args.steps)
    stats = {s["name"]: s for s in t.dqn.kernel_stats(enable=0)}
    c1 = t.counters()
    comm_info = comm.info() if comm is not None else None
    # the other TD rule on the same trainer, same steady state, timed the same way (reported beside the headline)
    other, el_other = None, None
    if args.config == 2 and args.td_net in ("online", "target") and not args.no_variants:
        other = "target" if args.td_net == "online" else "online"
        t.set_td_net(td_code[other])
        one_step()
        el_other, _, _ = timed(args.steps)
        t.set_td_net(td_code[args.td_net])
    # the same loop with the full fp32 column-max product in place of the exact screen (bracketed the same way)
    qmax_info, full_variant = None, None
    screened_live = "gemm_qmax_screen" in stats
    if args.qmax == "screened":
        st = t.dqn.qmax_stats()
        qmax_info = {"mode": "screened" if screened_live else "full (screen not applicable to this configuration)",
                     "what": "max_a' Q(s',a') = maximum of fp32-evaluated outputs; candidates found by a bf16 MFMA pass with a "
                             "rigorous error bound (DESIGN.md section 4)",
                     "candidate_groups_per_sample": (st[2] - qstat0[2]) / max(st[1] - qstat0[1], 1),
                     "whole_groups_per_sample": (st[3] - qstat0[3]) / max(st[1] - qstat0[1], 1)}
        if screened_live and world == 1 and not args.no_variants:
            t.dqn.set_qmax_mode(_capi.QMAX_FULL)
            one_step()
            t.dqn.kernel_stats(enable=3 if args.bracket_all else 4)
            el_full, _, samples_full = timed(args.steps)
            full_variant = (el_full, {s["name"]: s for s in t.dqn.kernel_stats(enable=0)})
            t.dqn.set_qmax_mode(_capi.QMAX_SCREENED)
    else:
        qmax_info = {"mode": "full fp32 product"}
    # workload statistics of the state the numbers were taken in (host reads, outside every timed region)
    import numpy as np
    _, meta = t.env.get_state()
    _, legal_counts = t.env.legal_moves(-1)
    rp_size, rp_cap, rp_total = t.replay.stats()
    iso = {}
    if not args.no_overlap and world == 1 and args.config == 2:
        # the same two kernels with nothing beside them (device-wide sync between collect and learn), outside the timed region:
        # in the overlapped loop the env kernel shares the chip with the TD step, so its live duration is not its own
        t.dqn.kernel_stats(enable=3)
        for _ in range(10):
            t.collect(); torch.cuda.synchronize()
            t.learn_grads(); t.learn_apply(world); torch.cuda.synchronize()
        iso = {s["name"]: s for s in t.dqn.kernel_stats(enable=0)}
    if args.check_replicas and world > 1:
        import numpy as np
        w, b = t.dqn.get_params()
        digest = torch.tensor([float(np.abs(w).sum()), float(np.abs(b).sum()), float(w[::997].sum())], dtype=torch.float64,
                              device="cuda")
        lo, hi = digest.clone(), digest.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), f"replicas diverged: {lo.tolist()} vs {hi.tolist()}"
        if rank == 0:
            print(f"replicas identical on {world} ranks: {digest.tolist()}", file=sys.stderr)

    torch_exchange_calibration = None
    if comm is None and grads is not None and not args.independent:
        # the torch.distributed path (gloo rehearsal / fallback): the same measurement the library makes behind the C ABI, on every rank
        # (a collective), AFTER the replicas were compared — it sums the gradient buffer into itself
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        grads.zero_()
        for _ in range(4):
            xd.allreduce_gradients(grads, world)
        ev0.record()
        for _ in range(20):
            xd.allreduce_gradients(grads, world)
        ev1.record(); torch.cuda.synchronize()
        torch_exchange_calibration = {
            "allreduce_us": 1e3 * ev0.elapsed_time(ev1) / 20, "threshold_us": 41.0, "late_start": False,
            "rule": "torch.distributed path: the caller issues the collective between learn_grads and learn_apply, the select chain's "
                    "start is not moved; behind the C ABI the same measurement decides (xq_dqn_calibrate_exchange)"}

    if rank == 0:
        env_steps = world * n_games * plies * args.steps
        td_text = {"online": "online (max Q(s') from the online net, chessai.cpp:126)",
                   "target": "target (max Q(s') from the target net, dqn.cpp:166)",
                   "double": "double (a* = argmax_a Q_online(s',a), y = r + gamma Q_target(s',a*); build-defined)"}[args.td_net]
        line = {
            "metric": "env steps/sec + DQN updates/sec at 8192 parallel games, 1/2/4/8 MI355X",
            "value": env_steps / elapsed, "unit": "env steps/s",
            "updates_per_s": args.steps / elapsed,
            "transitions_trained_per_s": world * minibatch * args.steps / elapsed,
            "metric_config": "BASELINE.json configs[%d]%s" % (args.config - 1, "" if args.config == 2 else " (NOT the headline configuration)"),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_samples": [1e3 * x / args.steps for x in samples],
            "timing": "the timed region (exactly `steps` steps between barrier + synchronize) run %d times back to back; value, "
                      "updates_per_s and ms_per_step are the median repetition" % len(samples),
            "host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": CFG["dtype"], "data": "synthetic",
            "dtype_note": ("weights, activations, Q-values, TD targets and gradients fp32; with config.qmax.mode = screened the candidates for "
                           "max_a' Q(s',a') are found on bf16 MFMA under a rigorous bound and re-evaluated in fp32 (same maximum)"
                           if CFG["dtype"] == "f32" else "bf16 operands on the matrix pipe (forward and backward products), fp32 accumulation, "
                           "fp32 master weights, fp32 layer-0 / output-layer / bias gradients"),
            "config": {"workload": CFG["workload"], "baseline_config": args.config,
                       "games_per_gpu": n_games, "layer_sizes": list(LAYERS), "replay_capacity": max(REPLAY, n_games),
                       "minibatch": minibatch, "plies_per_update": plies, "epsilon": 0.1, "backprop": "reference-compatible",
                       "td_net": td_text, "prioritized_replay": bool(CFG["prioritized"]),
                       "layer0_next_state": "derived from the state's sums (xq_dqn_set_l0_derive)" if not args.no_derive else "gathered in full",
                       "q_net_precision": ("bf16 forward (fp32 master weights, fp32 backward)" if args.bf16_fp32_backward else
                                           "bf16 forward and bf16 operands in the backward products (fp32 accumulation, fp32 master weights)")
                                          if CFG["bf16"] else "fp32",
                       "target_sync_interval": args.target_sync_interval,
                       "target_syncs_in_timed_region": (c1["updates"] // max(args.target_sync_interval, 1)
                                                        - c0["updates"] // max(args.target_sync_interval, 1)) / len(samples)
                                                       if args.target_sync_interval else 0,
                       "steady_state": {"prefill_random_plies": args.prefill_plies, "fill_collects": fill_collects,
                                        "settle_steps_untimed": args.settle_steps,
                                        "replay_fill": rp_size / rp_cap, "replay_total_pushed": rp_total,
                                        "mean_ply": float(np.mean(meta[:, 0])), "max_ply": int(np.max(meta[:, 0])),
                                        "mean_legal_moves": float(np.mean(legal_counts)), "max_legal_moves": int(np.max(legal_counts)),
                                        "episodes_finished_in_timed_region": (c1["episodes"] - c0["episodes"]) / len(samples)},
                       "schedule": "collect -> learn -> apply on one stream" if args.no_overlap else
                                   "collect(t) on its own stream beside learn_grads(t), both on theta_t; minibatch from the ring minus "
                                   "the slots collect(t) writes; apply joins both",
                       "parallelism": ("1 GPU" if world == 1 else
                                       f"{world} independent shards, no all-reduce (BASELINE configs[2])" if args.independent else
                                       f"dp{world} (games sharded; per update one RCCL sum all-reduce of the 1.65 MB gradient buffer "
                                       f"behind the C ABI, on the handle's stream right behind the launch that reduces the partial sums)" if comm is not None else
                                       f"dp{world} (games sharded, gradient all-reduce per update through torch.distributed)")},
        }
        if other is not None:
            line["variant_td_" + other] = {"value": world * n_games * plies * args.steps / el_other, "unit": "env steps/s",
                                           "updates_per_s": args.steps / el_other, "ms_per_step": 1e3 * el_other / args.steps}
        def empty_bracket_ms(reps=48):
            """What a HIP-event bracket reads around a kernel that does nothing (a 64-element add on the bench's stream, behind another
            kernel like every bracket of the loop): dispatch + a ~2 us kernel + completion + the two records.  The bracket around the
            dominant kernel contains the same overheads, which is why it reads 3-4 us more than rocprofv3's kernel duration."""
            x = torch.zeros(64, device="cuda")
            evs = []
            with torch.cuda.stream(tstream):
                for _ in range(reps):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    x.add_(1.0)
                    a.record(tstream); x.add_(1.0); b.record(tstream)
                    evs.append((a, b))
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            return ts[len(ts) // 2]

        bracket_ms = empty_bracket_ms()

        def gemm_roofline(st, iso_st, screened_kernel):
            ms = st["ms"] / st["launches"]
            fl = st["flops"] / st["launches"]
            ach = fl / (ms * 1e-3) / 1e12
            bf16_pipe = screened_kernel or CFG["bf16"]
            peak = PEAK_BF16_MFMA_TFLOPS if bf16_pipe else PEAK_F32_MFMA_TFLOPS
            # kernel instance as rocprofv3 prints it.  bf16 pipe, last hidden width 256 / 512: screen_top2_kernel<KU, NS, MODE, 0>
            # (xq_screen.hip.h; MODE 0 = top-2 screen, 1 = arg-max, 2 = max); fp32: gemm_colmax_persistent_kernel<2, 2, 0, MODE>
            if bf16_pipe and LAYERS[-2] in (256, 512):
                ku = LAYERS[-2] // 256
                inst = "screen_top2_kernel<%d, %d, %d, 0>(" % (ku, 2 // ku, 0 if screened_kernel else 1 if args.td_net == "double" else 2)
            else:
                inst = "gemm_colmax_persistent_kernel<2, 2, %d, %d>(" % (1 if bf16_pipe else 0,
                                                                          2 if screened_kernel else 1 if args.td_net == "double" else 0)
            tr, src = pmc_traffic(inst, args.config)
            what = ("exact screen of max_a' Q(s'): all outputs once on bf16 MFMA, top-2 per 32-output group" if screened_kernel else
                    "%s_a' Q(s')" % ("argmax" if args.td_net == "double" else "max"))
            r = {"kernel": "%s (%s: 8100 x %d x %d, %s MFMA)" % (inst.rstrip("("), what, minibatch, LAYERS[-2], "bf16" if bf16_pipe else "f32"),
                 "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                 "frac": ach / peak, "traffic": tr, "traffic_unit": "HBM bytes/launch (rocprofv3 PMC, offline)",
                 "traffic_source": src, "avg_launch_ms": ms, "flops_per_launch": fl, "launches": st["launches"],
                 "launches_note": (KERNEL_EXACT_NOTE if st.get("exact", 0) == st["launches"] else "HIP-event brackets") + " inside the timed region, on " +
                                  ("every launch" if args.bracket_all or args.profile_all else
                                   "every 4th launch (--bracket-all for every launch)")}
            # context for the difference between avg_launch_ms and rocprofv3's kernel duration (profiles/): a pair of RECORDED events around a
            # kernel that does nothing reads this much — it is in every bracketed figure and NOT subtracted; the screening pass carries its own
            # start / stop events instead (hipExtLaunchKernelGGL), so its figure IS the kernel's duration
            r["hip_event_bracket_of_an_empty_kernel_ms"] = 0.0 if st.get("exact", 0) == st["launches"] else bracket_ms
            if iso_st and iso_st["launches"]:
                r["isolated_avg_launch_ms"] = iso_st["ms"] / iso_st["launches"]
                r["isolated_frac"] = fl / (iso_st["ms"] / iso_st["launches"] * 1e-3) / 1e12 / peak
            return r
        g = stats.get("gemm_qmax_screen") or stats.get("gemm_qmax_rowmax")
        if g and g["launches"]:
            scr = "gemm_qmax_screen" in stats
            # the matrix-pipe kernel of the TD target (rounds 1-3 reported it as `roofline`; it is no longer the longest kernel)
            line["roofline_qmax"] = gemm_roofline(g, iso.get("gemm_qmax_screen" if scr else "gemm_qmax_rowmax"), scr)
        # ---- every kernel of the step priced against its own bound ------------------------------------------------
        work = step_work(LAYERS, minibatch, n_games, plies, bf16=CFG["bf16"], bf16_bwd=CFG["bf16"] and not args.bf16_fp32_backward,
                         td=args.td_net, screened=screened_live, derive=not args.no_derive, prioritized=bool(CFG["prioritized"]),
                         l0_mfma=args.l0_grad == "mfma")

        def rocprof_kernel(name):
            """the kernel instance(s) rocprofv3 prints for a bracket name (cn_chess_ai_amd/workmodel.py::rocprof_kernel)"""
            return wm_rocprof_kernel(name, LAYERS, minibatch, td=args.td_net, bf16=CFG["bf16"], l0_mfma=args.l0_grad == "mfma", n_games=n_games)

        def chain_entry(name, avg_us, lps, live=None):
            e = {"kernel": name, "avg_us": avg_us, "launches_per_step": lps,
                 "stream": "collect (select chain)" if name.endswith("@select") or name == "env_selfplay_step" else "handle (TD step)"}
            wk = work.get(name)
            if wk:
                ach, frac = price(wk, avg_us)
                e.update(bound=wk["bound"], achieved=ach, peak=wk["peak"], unit=wk["peak_unit"], frac=frac, flops_per_launch=wk["flops"],
                         hbm_bytes_per_launch=wk["hbm_bytes"], what=wk["what"])
            rk = rocprof_kernel(name)
            e["rocprof_kernel"] = rk
            if rk:
                if isinstance(rk, list):
                    got = [pmc_traffic(x, args.config) for x in rk]
                    vals = [t_ for t_, _ in got if t_]
                    tr, src = (sum(vals) / len(vals) if vals else None), got[0][1]
                else:
                    tr, src = pmc_traffic(rk, args.config)
                e["traffic"] = tr
                e["traffic_source"] = src
                if tr and wk and wk["hbm_bytes"]:
                    e["traffic_over_algorithmic"] = tr / wk["hbm_bytes"]
            if live:
                e.update(live)
            return e

        if chain:
            step_us = 1e3 * line["ms_per_step"]
            entries = [chain_entry(k, v["avg_us"], v["launches_per_step"]) for k, v in chain.items()]
            crit = [e for e in entries if e["stream"].startswith("handle") and e["launches_per_step"] >= 0.5]
            sel = [e for e in entries if e["stream"].startswith("collect")]
            rare = [e for e in entries if e["stream"].startswith("handle") and e["launches_per_step"] < 0.5]
            crit_us = sum(e["avg_us"] * e["launches_per_step"] for e in crit)
            sel_us = sum(e["avg_us"] * e["launches_per_step"] for e in sel)
            for e in entries:
                e["share_of_step"] = e["avg_us"] * e["launches_per_step"] / step_us
                if chain.get(e["kernel"], {}).get("exact"):
                    e["timing"] = KERNEL_EXACT_NOTE
            line["roofline_chain"] = {
                "how": "each kernel measured IN the training loop with nothing else being timed (%d untimed steps per kernel, between the settle steps "
                       "and the timed region).  Entries with `timing`: the launch carries its own start / stop events, avg_us IS the kernel's duration; "
                       "the others are pairs of recorded events around the launch(es), which read ~%.1f us more than the kernel "
                       "(hip_event_bracket_of_an_empty_kernel_us); algorithmic FLOPs / compulsory HBM bytes: cn_chess_ai_amd/workmodel.py = "
                       "DESIGN.md section 5; traffic: rocprofv3 PMC of the same command (profiles/)" % (args.chain_steps, 1e3 * bracket_ms),
                "hip_event_bracket_of_an_empty_kernel_us": 1e3 * bracket_ms,
                "step_us": step_us,
                "handle_stream": crit, "handle_stream_sum_us": crit_us,
                "handle_stream_unaccounted_us": step_us - crit_us,
                "handle_stream_unaccounted_note": "step time minus the sum of the kernels queued on the handle's stream: cross-stream waits "
                                                  "(collect fork / join) and kernel boundaries (minus the bracket overhead of entries without `timing`)",
                "collect_stream": sel, "collect_stream_sum_us": sel_us,
                "collect_stream_note": "runs beside the TD step (one ply per update: ends before the gradients do) — or IS the long pole of "
                                       "the step when an update has several plies",
                "less_than_once_per_step": rare,
            }
            if dominant:
                dom = next(e for e in entries if e["kernel"] == dominant)
                st = stats.get(dominant)
                r = {"kernel": "%s = %s (%s)" % (dominant, dom.get("rocprof_kernel"), dom.get("what", "")),
                     "why_this_kernel": "largest share of the timed step among the kernels on its dependency chain (%.0f %%; the handle's "
                                        "stream%s)" % (100 * dom["share_of_step"], " and the select chain: several plies per update" if plies > 1 else
                                                       "; the select chain runs beside it with slack"),
                     "bound": dom.get("bound"), "peak": dom.get("peak"), "unit": dom.get("unit"), "traffic": dom.get("traffic"),
                     "traffic_unit": "HBM bytes/launch (rocprofv3 PMC, offline)", "traffic_source": dom.get("traffic_source"),
                     "share_of_step": dom["share_of_step"], "launches_per_step": dom["launches_per_step"]}
                wk = work.get(dominant)
                if st and st["launches"] and wk:         # measured live inside the timed region (every 4th launch)
                    ms = st["ms"] / st["launches"]
                    ach, frac = price(wk, 1e3 * ms)
                    r.update(achieved=ach, frac=frac, avg_launch_ms=ms, launches=st["launches"],
                             launches_note=(KERNEL_EXACT_NOTE if st.get("exact", 0) == st["launches"] else "HIP-event brackets") + " inside the timed region, on " +
                                           ("every launch" if args.bracket_all or args.profile_all else "every 4th launch"),
                             flops_per_launch=wk["flops"], hbm_bytes_per_launch=wk["hbm_bytes"],
                             hip_event_bracket_of_an_empty_kernel_ms=0.0 if st.get("exact", 0) == st["launches"] else bracket_ms)
                elif wk:
                    r.update(achieved=dom.get("achieved"), frac=dom.get("frac"), avg_launch_ms=1e-3 * dom["avg_us"],
                             launches_note="from the per-kernel leg (no live bracket landed in the timed region)")
                if "achieved" in r:
                    line["roofline"] = r
        if "roofline" not in line and "roofline_qmax" in line:       # --no-chain / --profile-all: the matrix-pipe kernel as before
            line["roofline"] = line["roofline_qmax"]
        a = stats.get("rccl_allreduce_grads")
        line["exchange"] = {"path": exchange, "rccl_behind_c_abi": comm is not None, "comm": comm_info,
                            "gradient_buffer_bytes": 4 * t.dqn.grad_buffer()[1]}
        # where the select chain starts in a data-parallel step is decided by a measurement of the exchange (xq_dqn_calibrate_exchange:
        # 20 all-reduces of the gradient buffer at xq_dqn_set_comm time, mean over the ranks against 41 us), not by the rank count
        if comm is not None:
            line["exchange"]["calibration"] = t.dqn.exchange_calibration()
            if args.exchange_overlap >= 0 and line["exchange"]["calibration"] is not None:
                line["exchange"]["calibration"]["overridden_by"] = "--exchange-overlap %d" % args.exchange_overlap
        elif torch_exchange_calibration is not None:
            line["exchange"]["calibration"] = torch_exchange_calibration
        if qmax_info is not None:
            line["config"]["qmax"] = qmax_info
        if full_variant is not None:
            el_full, st_full = full_variant
            line["variant_qmax_full_fp32_product"] = {"value": world * n_games * plies * args.steps / el_full, "unit": "env steps/s",
                                                      "updates_per_s": args.steps / el_full, "ms_per_step": 1e3 * el_full / args.steps,
                                                      "ms_per_step_samples": [1e3 * x / args.steps for x in samples_full]}
            gf = st_full.get("gemm_qmax_rowmax")
            if gf and gf["launches"]:
                line["variant_qmax_full_fp32_product"]["roofline"] = gemm_roofline(gf, None, False)
        e = stats.get("env_selfplay_step")
        if e and e["launches"]:
            ms = e["ms"] / e["launches"]
            by = e["bytes"] / e["launches"]
            ach = by / (ms * 1e-3) / 1e9
            line["roofline_env"] = {"kernel": "env_kernel<SELFPLAY> (movegen+select+move+reward+reset, %d boards)" % n_games,
                                    "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": ach / PEAK_HBM_GBS, "traffic": pmc_traffic("env_kernel<2>(", args.config)[0], "avg_launch_ms": ms,
                                    "bytes_per_launch": by, "launches": e["launches"],
                                    "launches_note": (KERNEL_EXACT_NOTE if e.get("exact", 0) == e["launches"] else "HIP-event brackets") + " on " +
                                                     ("every launch" if args.bracket_all or args.profile_all else "every 5th launch")}
            mix = env_instruction_mix(args.config, n_games) if n_games == CFG["games"] else None
            if mix:
                # the bound that actually applies: instruction issue.  Floor = the larger of the VALU issue time (a wave64 VALU
                # instruction occupies its SIMD-32 for 2 cycles; 4 SIMDs per CU) and the scalar issue time (one scalar unit per
                # CU, one instruction per cycle) at the 2.4 GHz peak clock, every CU busy, nothing else stalling.
                valu_us = mix["valu"] * n_games * 2.0 / (4 * 256) / 2400.0
                salu_us = (mix["salu"] + mix["smem"]) * n_games / 256.0 / 2400.0
                floor_us = max(valu_us, salu_us)
                line["roofline_env"]["issue"] = {
                    "bound": "issue", "wave_instructions_per_board": mix["total"], "valu": mix["valu"], "salu": mix["salu"],
                    "lds": mix["lds"], "smem": mix["smem"], "vmem": mix["vmem"], "source": mix["source"],
                    "wait_any_frac_of_wave_cycles": mix.get("wait_any_frac_of_wave_cycles"),
                    "floor_us": floor_us, "valu_issue_us": valu_us, "scalar_issue_us": salu_us,
                    "frac": floor_us / (ms * 1e3),
                    "note": "one wavefront per board: ~%d wave-instructions per board-ply (move generation alone ~400); BASELINE's "
                            ">= 40 %% of HBM peak would need ~100 — the HBM figure above is reported, but it is not the bound of this kernel; "
                            "wait_any_frac_of_wave_cycles says how much of a wave's life is dependent latency rather than issue"
                            % round(mix["total"])}
            ei = iso.get("env_selfplay_step")
            if ei and ei["launches"]:
                line["roofline_env"]["co_scheduled"] = "runs on its own stream beside the TD step; avg_launch_ms is its stretched live duration"
                line["roofline_env"]["isolated_avg_launch_ms"] = ei["ms"] / ei["launches"]
                line["roofline_env"]["isolated_achieved"] = by / (ei["ms"] / ei["launches"] * 1e-3) / 1e9
                if "issue" in line["roofline_env"]:
                    line["roofline_env"]["issue"]["isolated_frac"] = line["roofline_env"]["issue"]["floor_us"] / (ei["ms"] / ei["launches"] * 1e3)
        if args.profile_all:
            line["kernels"] = {k: {"ms_per_launch": v["ms"] / max(v["launches"], 1), "launches": v["launches"]}
                               for k, v in stats.items()}
        if world == 1 and not args.no_facade and args.config == 2 and n_games == CFG["games"]:
            # the trainer of this process is idle now (everything synchronised); the facade process gets the GPU to itself
            f = facade_leg(n_games, LAYERS, max(REPLAY, n_games), minibatch, args.facade_episodes, not args.no_derive, args.prefill_plies)
            if "env_steps_per_s" in f:
                f["vs_headline"] = f["env_steps_per_s"] / line["value"]
                f["note"] = ("the C++ entry point a maintainer calls (mainwindow.h:140-150 -> ChessAI::train) on the same schedule as the "
                             "ctypes loop above; its clock starts with an EMPTY ring and includes the ring fill, the episode drains (one device "
                             "synchronisation per 64 iterations) and a gameCompleted callback per episode")
            line["facade"] = f
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(15.0 if args.config == 2 else 6.0, args.reference_nn)
            except Exception as e:           # never lose the measured line to the reporting leg
                line["cpu_baseline"] = {"value": None, "unit": "env steps/s", "cores": 0, "kind": "port", "sample": "failed: " + str(e)[:160]}
        print(json.dumps(line), flush=True)
    t.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

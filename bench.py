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
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 matrix peak (the 5 PF headline figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak
ENV_BYTES_PER_GAME = 2 * (48 + 16) + 360 + 105   # board+meta r/w, Q row, transition record


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


def cpu_baseline(seconds=15.0):
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
    return out


def pmc_traffic(kernel_substring, config=2):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 --pmc summary of this configuration
    (profiles/*_pmc_hbm_traffic.json for configs[1], profiles/*_config<K>_pmc_hbm_traffic.json for the others; made by
    tools/collect_profiles*.sh + tools/summarize_profiles.py on this same bench command: separate FETCH_SIZE / WRITE_SIZE passes,
    bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 correction).  None if absent: PMC counters cannot be read from inside
    the timed run."""
    import glob
    if config == 2:
        files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")) if "_config" not in os.path.basename(f))
    else:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_config%d_pmc_hbm_traffic.json" % config)))
    for path in reversed(files):
        try:
            data = json.load(open(path))
        except Exception:
            continue
        for k, v in data.items():
            if kernel_substring in k:
                return v.get("hbm_bytes_per_launch"), os.path.relpath(path, ROOT)
    return None, None


def env_instruction_mix():
    """Wave-instructions per board of env_kernel<SELFPLAY> from the newest committed PMC summary (profiles/*env_kernel_instruction*.json,
    tools/env_pmc.sh: SQ_INSTS_* of the env-only launch at 8192 boards).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*env_kernel_instruction*.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        m = d.get("after", d)["instruction_mix_per_launch_8192_boards"]
        return {"total": m["wave_instructions_per_board"], "valu": m["SQ_INSTS_VALU_per_board"], "salu": m["SQ_INSTS_SALU_per_board"],
                "lds": m["SQ_INSTS_LDS_per_board"], "smem": m["SQ_INSTS_SMEM_per_board"], "vmem": m["SQ_INSTS_VMEM_per_board"],
                "source": os.path.relpath(files[-1], ROOT)}
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
                    help="BASELINE.json configs entry, 1-based: 2 = configs[1] (headline, default), 4 = configs[3] share, 5 = configs[4] share")
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
    ap.add_argument("--no-variants", action="store_true", help="skip the variant legs (other TD net, full fp32 product): A/B runs")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (exactly --steps steps, barrier + synchronize on both sides) is run this many times back to "
                         "back; ms_per_step / value are the MEDIAN repetition, all of them are listed in ms_per_step_samples")
    ap.add_argument("--settle-steps", type=int, default=100,
                    help="untimed training steps between the W warm-up steps and the timed region (reported in config.steady_state)")
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
    if args.no_td_tail: t.dqn.set_td_tail(False)  # A/B: the gradient kernels one by one on two streams (library default: fused launches)
    if args.exchange_overlap >= 0: t.dqn.set_exchange_overlap(args.exchange_overlap)
    grads, comm, comm_error = None, None, ""
    if world == 1 and os.environ.get("XQ_BENCH_COMM1"):
        # rehearsal on one GPU: a one-rank communicator attached, so that the whole N > 1 code path of the library runs (one launch
        # that reduces the partial sums, the all-reduce behind it, unfused SGD) — shows what that path costs besides the wire time
        comm = xd.Comm(rank=0, world=1)
        t.set_comm(comm)
    elif world == 1 or args.independent:
        t.dqn.set_fused_apply(True)          # nothing reads the gradient buffer between td_grads and apply_grads
    elif args.torch_allreduce or backend != "nccl":   # diagnostic / one-GPU gloo rehearsal: the exchange through torch.distributed
        ptr, n = t.dqn.grad_buffer()
        grads = xd.wrap_device_floats(ptr, n)
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
            ptr, n = t.dqn.grad_buffer()
            grads = xd.wrap_device_floats(ptr, n)
            print(f"[rank {rank}] C-ABI communicator unavailable ({comm_error or 'another rank failed'}): all-reduce through torch.distributed",
                  file=sys.stderr)

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
    # the HIP-event brackets of the roofline leg run during the warm-up too (the library's event pool is then filled before anything is
    # timed), and `--settle-steps` more untimed steps follow the W warm-up steps: with the driver's short windows (W = 5, K = 20: 5 ms)
    # the first repetitions otherwise still see the chip ramping up from the host-side preparation (0.2056, 0.2028, 0.1981, 0.1922,
    # 0.1921 ms per step over five repetitions on one run)
    t.dqn.kernel_stats(enable=2 if args.profile_all else 3 if args.bracket_all else 4)
    for _ in range(args.warmup + args.settle_steps):
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
    # the dominant GEMM is bracketed with HIP events on every 4th launch of the timed region (--bracket-all: every launch): a
    # bracket is two event records, and a record drains the recording queue — ~10 us each time on a 250-us step
    t.dqn.kernel_stats(enable=2 if args.profile_all else 3 if args.bracket_all else 4)
    elapsed, host_enqueue, samples = timed(args.steps)
    stats = {s["name"]: s for s in t.dqn.kernel_stats(enable=0)}
    c1 = t.counters()
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
                inst = "screen_top2_kernel<%d, %d, %d, 0>" % (ku, 2 // ku, 0 if screened_kernel else 1 if args.td_net == "double" else 2)
            else:
                inst = "gemm_colmax_persistent_kernel<2, 2, %d, %d>" % (1 if bf16_pipe else 0,
                                                                         2 if screened_kernel else 1 if args.td_net == "double" else 0)
            tr, src = pmc_traffic(inst, args.config)
            what = ("exact screen of max_a' Q(s'): all outputs once on bf16 MFMA, top-2 per 32-output group" if screened_kernel else
                    "%s_a' Q(s')" % ("argmax" if args.td_net == "double" else "max"))
            r = {"kernel": "%s (%s: 8100 x %d x %d, %s MFMA)" % (inst, what, minibatch, LAYERS[-2], "bf16" if bf16_pipe else "f32"),
                 "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                 "frac": ach / peak, "traffic": tr, "traffic_unit": "HBM bytes/launch (rocprofv3 PMC, offline)",
                 "traffic_source": src, "avg_launch_ms": ms, "flops_per_launch": fl, "launches": st["launches"],
                 "launches_note": "HIP-event brackets inside the timed region, on " + ("every launch" if args.bracket_all or args.profile_all else
                                  "every 4th launch (two event records per bracket drain the stream's queue, ~10 us; --bracket-all for every launch)")}
            # context for the difference between avg_launch_ms and rocprofv3's kernel duration (profiles/): the same bracket around a
            # kernel that does nothing — the raw figure above contains up to this much that is not the kernel; it is NOT subtracted
            r["hip_event_bracket_of_an_empty_kernel_ms"] = bracket_ms
            if iso_st and iso_st["launches"]:
                r["isolated_avg_launch_ms"] = iso_st["ms"] / iso_st["launches"]
                r["isolated_frac"] = fl / (iso_st["ms"] / iso_st["launches"] * 1e-3) / 1e12 / peak
            return r
        g = stats.get("gemm_qmax_screen") or stats.get("gemm_qmax_rowmax")
        if g and g["launches"]:
            scr = "gemm_qmax_screen" in stats
            line["roofline"] = gemm_roofline(g, iso.get("gemm_qmax_screen" if scr else "gemm_qmax_rowmax"), scr)
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
                                    "frac": ach / PEAK_HBM_GBS, "traffic": pmc_traffic("env_kernel<2>", args.config)[0], "avg_launch_ms": ms,
                                    "bytes_per_launch": by, "launches": e["launches"],
                                    "launches_note": "HIP-event brackets on " + ("every launch" if args.bracket_all or args.profile_all else
                                                     "every 5th launch (the brackets sit on the collect stream: ~12 us per ply, which "
                                                     "is the critical chain when an update has several plies)")}
            mix = env_instruction_mix() if args.config == 2 else None
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
                    "floor_us": floor_us, "valu_issue_us": valu_us, "scalar_issue_us": salu_us,
                    "frac": floor_us / (ms * 1e3),
                    "note": "one wavefront per board: ~%d wave-instructions per board-ply (move generation alone ~400); BASELINE's "
                            ">= 40 %% of HBM peak would need ~100 — the HBM figure above is reported, but it is not the bound of this kernel"
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
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline() if args.config == 2 else cpu_baseline(6.0)
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

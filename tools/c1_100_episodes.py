#!/usr/bin/env python3
"""BASELINE.md C1 / BASELINE.json configs[0] AS WRITTEN: 1 game, 100 calls of trainEpisode() (chessai.cpp:85-170, net 1260-128-8100, lr 0.001,
gamma 0.99, epsilon 0.1) — (a) the CPU restatement of the reference loop (oracle/xq_oracle.c: xqo_train_episode, fp64, batch 1, bug-compatible
backprop) on one host core, all 100 episodes, not time-bounded; (b) the same loop through the C++ facade on the HIP path with ONE game
(examples/train_selfplay.cpp, `ai.setParallelGames(1)`: the reference's sequential loop, a handful of launches and a host round trip per
ply — the shape the library is NOT built for, reported for completeness).  Prints one JSON object.  Runs on the GPU box (tools/)."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cpu_100():
    import xqoracle as xo
    L = xo.lib()
    net = [1260, 128, 8100]
    sizes = xo.sizes_arr(net)
    w, b = xo.init_weights(net, 1)
    rng = C.c_uint64(12345)
    st = xo.EpisodeStats()
    plies = 0
    t0 = time.perf_counter()
    for _ in range(100):
        L.xqo_train_episode(sizes.ctypes.data_as(C.POINTER(C.c_int)), len(sizes), w.ctypes.data_as(C.POINTER(C.c_double)),
                            b.ctypes.data_as(C.POINTER(C.c_double)), 0.001, 0.99, 0.1, C.byref(rng), 0, C.byref(st))
        plies += st.steps
    el = time.perf_counter() - t0
    return {"episodes": 100, "plies": plies, "seconds": el, "plies_per_s": plies / el, "cores": 1,
            "what": "CPU restatement of ChessAI::train (xqo_train_episode), net 1260-128-8100 fp64, batch 1"}


def hip_100():
    import test_facade_gpu as tf
    tf.build_example()
    t0 = time.perf_counter()
    out = subprocess.run([tf.EXAMPLE_BIN, "100", "/tmp/c1_model.bin", "1"], capture_output=True, text=True, timeout=1200)
    el = time.perf_counter() - t0
    line = [l for l in out.stdout.splitlines() if "episodes in" in l]
    return {"episodes": 100, "process_seconds": el, "report": line[-1] if line else out.stdout[-300:] + out.stderr[-300:],
            "what": "xq::ChessAI::train(100) with setParallelGames(1) on the HIP path (examples/train_selfplay.cpp)"}


if __name__ == "__main__":
    res = {"config": "BASELINE.json configs[0]: 1 game, 100 train episodes", "cpu_port": cpu_100()}
    try:
        res["hip_one_game"] = hip_100()
    except Exception as e:
        res["hip_one_game"] = {"error": str(e)[:300]}
    print(json.dumps(res, indent=1))

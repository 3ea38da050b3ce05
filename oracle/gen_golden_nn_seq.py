#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — regenerates tests/golden/ref_nn_seq.npz: a K-ply TRAJECTORY of the reference's own NN runtime on the
reference's own net {1260, 128, 8100}, run on an MI355X (Next #3).

tests/golden/ref_nn.npz pins single calls from fixed parameters; this fixture pins a sequence: per ply the NN half of ChessAI::train's
loop body (chessai.cpp:121-133) — getQValues(state), getQValues(nextState), the target entry, backpropagate — with the parameters
carried from ply to ply inside the reference's NeuralNetwork object (`xqref_nn seq`, oracle/ref/ref_nn_driver.cpp: /root/reference/src/
dqn.cu through hipify-perl, API identifiers only).  The states are 17 consecutive positions of one random-play game of
tests/golden/ref_trace.npz (the real rules engine's output), rewards are evaluateBoard's integers (oracle restatement of chessai.cpp:311-345).

backpropagate updates the output layer from a hidden activation it has ALREADY RELEASED (dqn.cu:371 / :441).  After every ply the
generator compares 64 sampled output-layer weights with the "released block still holds the activation" model (the oracle's): `intact[t]`
says whether the allocator left the block alone at ply t; `first_bad_ply` = K when it always did.  Tests assert the trajectory up to there.

    gpurun -- 'python oracle/gen_golden_nn_seq.py --out gpurun_out/ref_nn_seq.npz'      # then: cp gpurun_out/ref_nn_seq.npz tests/golden/
"""
import argparse
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_golden_nn as gg          # noqa: E402  (parse(), BIN)
import refnn                        # noqa: E402
import xqoracle as xo               # noqa: E402

SIZES = [1260, 128, 8100]
SEED, LR, GAMMA, K = 21, 0.001, 0.99, 16


def pick_run(tr, k):
    """first run of k consecutive valid, in-turn moves of one game that contains a capture (so that rewards differ from ply to ply)"""
    n = len(tr["valid"])
    for i in range(n - k - 1):
        ok = all(tr["valid"][i + j] and not tr["over"][i + j] and tr["moveCount"][i + j + 1] == tr["moveCount"][i + j] + 1 for j in range(k))
        if ok and tr["captured"][i:i + k].astype(bool).sum() >= 2 and tr["moveCount"][i] >= 6:
            return i
    raise SystemExit("no suitable run in ref_trace.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    probe = subprocess.run([gg.BIN, "probe"], capture_output=True, text=True)
    print("probe:", probe.stdout.strip(), "rc", probe.returncode, flush=True)
    if probe.returncode != 0:
        sys.exit("allocator probe failed: not running the reference's backpropagate (its stale reads must stay in mapped memory)")
    tr = np.load(os.path.join(ROOT, "tests", "golden", "ref_trace.npz"))
    i0 = pick_run(tr, K)
    S = tr["board"][i0:i0 + K].astype(np.uint8)
    S2 = tr["board"][i0 + 1:i0 + K + 1].astype(np.uint8)
    A = (tr["move"][i0:i0 + K, 2].astype(np.int32) * 9 + tr["move"][i0:i0 + K, 3].astype(np.int32))          # action.to
    R = np.zeros(K)
    D = np.zeros(K, dtype=np.int32)
    for t in range(K):
        j = i0 + t + 1
        b = xo.board_from(tr["board"][j], int(tr["moveCount"][j]), int(tr["player"][j]), int(tr["redScore"][j]), int(tr["blackScore"][j]))
        R[t] = xo.lib().xqo_evaluate_board(b, int(tr["player"][i0 + t]), int(tr["moveCount"][j]))           # evaluateBoard(mover, post-move count)
        D[t] = int(tr["over"][j])
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<i", K))
            for t in range(K):
                f.write(S[t].tobytes()); f.write(S2[t].tobytes()); f.write(struct.pack("<idi", int(A[t]), float(R[t]), int(D[t])))
        cmd = [gg.BIN, "seq", fin, fout, str(SEED), repr(LR), repr(GAMMA)] + [str(s) for s in SIZES]
        subprocess.run(["timeout", "-k", "10", "120"] + cmd, check=True)
        rec = gg.parse(fout)
    # the oracle's trajectory beside it: where does the released block stop holding the activation?
    w, b = refnn.params(SEED, SIZES)
    pos = refnn.sample_positions(SEED, 1, SIZES)[:64]
    off1 = SIZES[0] * SIZES[1]
    intact = np.zeros(K, dtype=np.uint8)
    worst_q = 0.0
    for t in range(K):
        x = xo.state_repr(xo.board_from(S[t]))
        q = xo.nn_forward(SIZES, w, b, x)
        worst_q = max(worst_q, float(np.abs(q[:96] - rec[f"ply{t}_q"]).max()))
        tq = xo.td_target(SIZES, w, b, x, xo.state_repr(xo.board_from(S2[t])), int(A[t]), float(R[t]), int(D[t]), GAMMA)
        assert xo.nn_backprop(SIZES, w, b, x, tq, LR, 0) == 0
        intact[t] = np.abs(w[off1 + pos] - rec[f"ply{t}_ub_w1"]).max() <= 1e-12
        if not intact[t]:
            break                                   # from here on the reference's net is no longer the model's: nothing later is comparable
    first_bad = int(np.argmin(intact)) if not intact.all() else K
    print(f"trajectory of {K} plies from trace row {i0}: released block intact at plies {intact.tolist()}, first_bad_ply {first_bad}, "
          f"oracle vs reference Q up to there: {worst_q:.3g}", flush=True)
    arrays = {"sizes": np.array(SIZES, dtype=np.int32), "seed_lr_gamma": np.array([SEED, LR, GAMMA]), "trace_row": np.array([i0], dtype=np.int64),
              "states": S, "next_states": S2, "action_to": A.astype(np.int32), "reward": R, "done": D, "intact": intact,
              "first_bad_ply": np.array([first_bad], dtype=np.int32), "allocator_probe": np.frombuffer(probe.stdout.strip().encode(), dtype=np.uint8)}
    for k, v in rec.items():
        arrays[k] = v
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, **arrays)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()

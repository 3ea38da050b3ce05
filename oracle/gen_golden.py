#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — regenerates tests/golden/*.npz from the REAL reference rules engine.

Runs only in the authoring container (needs /root/reference + conda QtCore): `make -C oracle ref` builds
oracle/_ref/xqref from /root/reference/src/chessboard.cpp unmodified, this script runs it and stores the
OUTPUTS (boards, ordered move lists, move results) as compact fixtures.  No reference source is copied.

    python oracle/gen_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import xqoracle as xo  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def run_trace(seed, ngames):
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        subprocess.check_call([xo.REF_BIN, "trace", str(seed), str(ngames), f.name])
        return np.fromfile(f.name, dtype=xo.REF_RECORD)


def run_validmat(seed, npos):
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        subprocess.check_call([xo.REF_BIN, "validmat", str(seed), str(npos), f.name])
        raw = np.fromfile(f.name, dtype=np.uint8).reshape(npos, 90 + 8100)
    return raw[:, :90].copy(), np.packbits(raw[:, 90:], axis=1)


def run_tracebig(seed, nrec):
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        subprocess.check_call([xo.REF_BIN, "tracebig", str(seed), str(nrec), f.name])
        return np.fromfile(f.name, dtype=xo.REF_RECORD)


def run_rulemat(seed, npos):
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        subprocess.check_call([xo.REF_BIN, "rulemat", str(seed), str(npos), f.name])
        raw = np.fromfile(f.name, dtype=np.uint8).reshape(npos, 90 + 7 * 8100 + 64 * 6)
    q = raw[:, 90 + 7 * 8100:].reshape(npos, 64, 6).view(np.int8)
    return raw[:, :90].copy(), np.packbits(raw[:, 90:90 + 7 * 8100], axis=1), q[:, :, :5].copy(), q[:, :, 5].astype(np.uint8)


def save_trace(name, rec):
    # ragged move lists -> flat arrays + offsets (keeps the fixture small)
    def flat(field, nfield):
        n = rec[nfield].astype(np.int64)
        assert n.max() <= 128
        off = np.concatenate([[0], np.cumsum(n)])
        out = np.concatenate([rec[field][i, :n[i]] for i in range(len(rec))]).astype(np.uint16)
        return out, off.astype(np.int32)
    red, red_off = flat("red", "nRed")
    black, black_off = flat("black", "nBlack")
    np.savez_compressed(
        os.path.join(GOLD, name),
        board=rec["board"], moveCount=rec["moveCount"], player=rec["player"], redScore=rec["redScore"],
        blackScore=rec["blackScore"], over=rec["over"], winner=rec["winner"],
        red=red, red_off=red_off, black=black, black_off=black_off,
        move=np.stack([rec["fr"], rec["fc"], rec["tr"], rec["tc"]], axis=1), valid=rec["valid"],
        captured=rec["captured"])


def main():
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
    os.makedirs(GOLD, exist_ok=True)
    big = run_tracebig(0xB16, 160)           # positions where a side has > 64 moves (second half of the 128-entry lists)
    save_trace("ref_bigmoves.npz", big)
    print("bigmoves records:", len(big), "max moves:", int(max(big["nRed"].max(), big["nBlack"].max())))
    rb, rbits, rq, rres = run_rulemat(0xFACE, 24)
    np.savez_compressed(os.path.join(GOLD, "ref_rulemat.npz"), board=rb, rule_bits=rbits, query=rq, query_result=rres)
    print("rulemat positions:", len(rb), "queries true:", int(rres.sum()), "of", rres.size)
    rec = run_trace(0x5EED, 20)
    def flat(field, nfield):
        n = rec[nfield].astype(np.int64)
        assert n.max() <= 128
        off = np.concatenate([[0], np.cumsum(n)])
        out = np.concatenate([rec[field][i, :n[i]] for i in range(len(rec))]).astype(np.uint16)
        return out, off.astype(np.int32)
    red, red_off = flat("red", "nRed")
    black, black_off = flat("black", "nBlack")
    np.savez_compressed(
        os.path.join(GOLD, "ref_trace.npz"),
        board=rec["board"], moveCount=rec["moveCount"], player=rec["player"], redScore=rec["redScore"],
        blackScore=rec["blackScore"], over=rec["over"], winner=rec["winner"],
        red=red, red_off=red_off, black=black, black_off=black_off,
        move=np.stack([rec["fr"], rec["fc"], rec["tr"], rec["tc"]], axis=1), valid=rec["valid"],
        captured=rec["captured"])
    boards, mats = run_validmat(0xC0FFEE, 64)
    np.savez_compressed(os.path.join(GOLD, "ref_validmat.npz"), board=boards, valid_bits=mats)
    print("records:", len(rec), "positions with both-side lists; max moves:",
          int(max(rec["nRed"].max(), rec["nBlack"].max())),
          "| validmat positions:", len(boards))
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — regenerates tests/golden/ref_nn.npz from the REAL reference NN runtime, run on an MI355X.

`make -C oracle refnn` (authoring container) builds oracle/_ref/xqref_nn from /root/reference/src/dqn.cu + include/dqn.h
(hipify-perl: API identifiers only, see oracle/ref/ref_nn_driver.cpp); the binary travels to the GPU box, this script
runs it there and stores its OUTPUTS: Q-values of NeuralNetwork::forward, biases and layer-0 weights after one
NeuralNetwork::backpropagate, the constructor's counts / offsets / init range.  No reference source is copied.

    gpurun -- 'python oracle/gen_golden_nn.py --out gpurun_out/ref_nn.npz'      # then: cp gpurun_out/ref_nn.npz tests/golden/
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "_ref", "xqref_nn")

TOPOLOGIES = [  # (seed, sizes): the three BASELINE topology classes at full size + small ones of each depth
    (11, [1260, 128, 8100]),
    (12, [1260, 256, 256, 8100]),
    (13, [1260, 512, 512, 512, 8100]),
    (14, [12, 4, 20]),
    (15, [30, 16, 16, 40]),
    (16, [24, 8, 8, 8, 96]),
    (17, [20, 30]),
]
DTYPES = {0: "<f8", 1: "<i4", 2: "<i8"}


def parse(path):
    out = {}
    with open(path, "rb") as f:
        raw = f.read()
    p = 0
    while p < len(raw):
        (nl,) = struct.unpack_from("<I", raw, p); p += 4
        name = raw[p:p + nl].decode(); p += nl
        dt, cnt = struct.unpack_from("<IQ", raw, p); p += 12
        a = np.frombuffer(raw, dtype=DTYPES[dt], count=cnt, offset=p).copy(); p += a.nbytes
        out[name] = a
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    probe = subprocess.run([BIN, "probe"], capture_output=True, text=True)
    print("probe:", probe.stdout.strip(), "rc", probe.returncode, flush=True)
    if probe.returncode != 0:
        sys.exit("allocator probe failed: not running the reference's backpropagate (its stale reads must stay in mapped memory)")
    arrays = {"allocator_probe": np.frombuffer(probe.stdout.strip().encode(), dtype=np.uint8)}
    for seed, sizes in TOPOLOGIES:
        with tempfile.NamedTemporaryFile(suffix=".bin") as f:
            cmd = [BIN, "nn", f.name, str(seed)] + [str(s) for s in sizes]
            subprocess.run(["timeout", "-k", "10", "120"] + cmd, check=True)
            rec = parse(f.name)
        key = "-".join(str(s) for s in sizes)
        rec["seed"] = np.array([seed], dtype=np.int64)
        for k, v in rec.items():
            if k.endswith("_pos"):                   # positions of the ub_* samples: a formula (tests/refnn.py::sample_positions), checked here
                continue
            if "_ub_w" in k:
                v = v[:64]
            arrays[f"{key}/{k}"] = v
        print(key, "records:", len(rec), flush=True)
    arrays["topologies"] = np.frombuffer(json.dumps([[s, t] for s, t in TOPOLOGIES]).encode(), dtype=np.uint8)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, **arrays)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()

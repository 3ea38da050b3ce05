"""Helpers for tests/golden/ref_nn.npz — outputs of the REAL reference NN runtime (/root/reference/src/dqn.cu through
hipify-perl, run on an MI355X; oracle/ref/ref_nn_driver.cpp, oracle/gen_golden_nn.py).  The driver filled the
net's parameters and inputs from a counter-based generator; this module restates that generator with numpy so that a
test can rebuild the very same parameters and feed them to the oracle or to the HIP path."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_nn.npz")
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def u01(seed, stream, i):
    """ref_nn_driver.cpp::u01, vectorised over i."""
    with np.errstate(over="ignore"):
        base = _mix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        return (_mix64(base + np.asarray(i, dtype=np.uint64)) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def counts(sizes):
    nw = sum(int(a) * int(b) for a, b in zip(sizes[:-1], sizes[1:]))
    nb = sum(int(b) for b in sizes[1:])
    return nw, nb


def params(seed, sizes):
    """ref_nn_driver.cpp::fill_params: weights in (-0.05, 0.05), biases in (-0.01, 0.01); reference flat layout."""
    nw, nb = counts(sizes)
    w = (u01(seed, 1, np.arange(nw)) - 0.5) * 0.1
    b = (u01(seed, 2, np.arange(nb)) - 0.5) * 0.02
    return w, b


def dense_input(seed, stream, width):
    return u01(seed, stream, np.arange(width)) * 2.0 - 1.0


def sample_positions(seed, l, sizes):
    """positions of the `ub_w<l>` samples inside layer l (rows 0..95 x any column)."""
    i = np.arange(256)
    row = (u01(seed, 80 + l, 2 * i) * min(sizes[l + 1], 96)).astype(np.int64)
    col = (u01(seed, 80 + l, 2 * i + 1) * sizes[l]).astype(np.int64)
    return row * sizes[l] + col


class Golden:
    def __init__(self, path=GOLDEN):
        self.z = np.load(path)
        self.topologies = [(int(s), [int(x) for x in t]) for s, t in json.loads(bytes(self.z["topologies"]).decode())]
        self.probe = json.loads(bytes(self.z["allocator_probe"]).decode())

    def get(self, sizes, name):
        return self.z["-".join(str(s) for s in sizes) + "/" + name]

    def has(self, sizes, name):
        return "-".join(str(s) for s in sizes) + "/" + name in self.z.files

    def input(self, sizes, tag, seed):
        """the input vector of record `tag` (fwd0.. / bp0..), rebuilt from the one-hot indices or the dense values."""
        x = np.zeros(sizes[0])
        if self.has(sizes, tag + "_onehot"):
            x[self.get(sizes, tag + "_onehot")] = 1.0
        else:
            x[:] = self.get(sizes, tag + "_dense")
        return x

"""TEST INFRASTRUCTURE — ctypes loader for the CPU oracle (oracle/xq_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (cn_chess_ai_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libxqoracle.so")
REF_BIN = os.path.join(HERE, "_ref", "xqref")

MAX_MOVES = 128


class Board(C.Structure):
    _fields_ = [("sq", C.c_uint8 * 90), ("moveCount", C.c_int32), ("currentPlayer", C.c_int32),
                ("redScore", C.c_int32), ("blackScore", C.c_int32)]

    def squares(self):
        return np.frombuffer(bytes(self.sq), dtype=np.uint8).copy()

    def set_squares(self, arr):
        arr = np.asarray(arr, dtype=np.uint8)
        for i in range(90):
            self.sq[i] = int(arr[i])


class StepOut(C.Structure):
    _fields_ = [("action_code", C.c_int32), ("n_moves", C.c_int32), ("reward", C.c_int32),
                ("done", C.c_uint8), ("terminated", C.c_uint8), ("winner", C.c_uint8), ("explored", C.c_uint8),
                ("redScore", C.c_int32), ("blackScore", C.c_int32), ("moveCount", C.c_int32)]


class EpisodeStats(C.Structure):
    _fields_ = [("steps", C.c_int32), ("winner", C.c_int32), ("redScore", C.c_int32),
                ("blackScore", C.c_int32), ("moveCount", C.c_int32), ("target_syncs", C.c_int32)]


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < max(os.path.getmtime(os.path.join(HERE, f))
                                             for f in ("xq_oracle.c", "xq_oracle_ext.c", "xq_oracle.h")):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    pB = C.POINTER(Board)
    pd = C.POINTER(C.c_double)
    pi = C.POINTER(C.c_int)
    pu16 = C.POINTER(C.c_uint16)
    L.xqo_reset.argtypes = [pB]
    L.xqo_is_valid_move.argtypes = [pB, C.c_int, C.c_int, C.c_int, C.c_int]
    L.xqo_get_valid_moves.argtypes = [pB, C.c_int, C.c_int, pi]
    L.xqo_piece_rule.argtypes = [pB, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.xqo_move_piece.argtypes = [pB, C.c_int, C.c_int, C.c_int, C.c_int]
    L.xqo_check_game_over.argtypes = [pB]
    L.xqo_get_winner.argtypes = [pB]
    L.xqo_all_valid_actions.argtypes = [pB, C.c_int, pu16]
    L.xqo_state_indices.argtypes = [pB, pi]
    L.xqo_state_repr.argtypes = [pB, pd]
    L.xqo_evaluate_board.argtypes = [pB, C.c_int, C.c_int]
    L.xqo_select_action.argtypes = [pd, C.c_int, pu16, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
    L.xqo_nn_num_weights.argtypes = [pi, C.c_int]
    L.xqo_nn_num_weights.restype = C.c_size_t
    L.xqo_nn_num_biases.argtypes = [pi, C.c_int]
    L.xqo_nn_num_biases.restype = C.c_size_t
    L.xqo_nn_forward.argtypes = [pi, C.c_int, pd, pd, pd, pd]
    L.xqo_nn_backprop.argtypes = [pi, C.c_int, pd, pd, pd, pd, C.c_double, C.c_int]
    L.xqo_nn_accum_grad.argtypes = [pi, C.c_int, pd, pd, pd, pd, C.c_int, pd, pd]
    L.xqo_nn_forward_all.argtypes = [pi, C.c_int, pd, pd, pd, pd]
    L.xqo_td_target.argtypes = [pi, C.c_int, pd, pd, pd, pd, C.c_int, C.c_double, C.c_int, C.c_double, pd]
    L.xqo_train_episode.argtypes = [pi, C.c_int, pd, pd, C.c_double, C.c_double, C.c_double,
                                    C.POINTER(C.c_uint64), C.c_int, C.POINTER(EpisodeStats)]
    L.xqo_rand.argtypes = [C.POINTER(C.c_uint64)]
    L.xqo_philox4x32.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.xqo_selfplay_step.argtypes = [pB, C.POINTER(C.c_float), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.POINTER(StepOut)]
    # build-defined extensions (xq_oracle_ext.c)
    pf, pi32 = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.xqo_bf16_round.argtypes = [C.c_float]
    L.xqo_bf16_round.restype = C.c_float
    L.xqo_ext_forward.argtypes = [pi, C.c_int, pd, pd, pd, C.c_int, pd, pd]
    L.xqo_ext_td_accum.argtypes = [pi, C.c_int, pd, pd, pd, pd, pd, pd, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int,
                                   C.c_int, C.c_int, C.c_double, pd, pd, pd, pd, pi]
    L.xqo_per_tree_floats.argtypes = [C.c_int]
    L.xqo_per_tree_floats.restype = C.c_size_t
    L.xqo_per_build.argtypes = [pf, C.c_int, pf]
    L.xqo_per_total.argtypes = [pf, C.c_int]
    L.xqo_per_total.restype = C.c_float
    L.xqo_per_descend.argtypes = [pf, C.c_int, C.c_float]
    L.xqo_per_sample.argtypes = [pf, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.c_float, pi32, pf]
    L.xqo_per_sample.restype = C.c_float
    L.xqo_per_priority.argtypes = [C.c_float, C.c_float, C.c_float]
    L.xqo_per_priority.restype = C.c_float
    _lib = L
    return L


# ---------------------------------------------------------------- convenience wrappers
def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def new_board():
    b = Board()
    lib().xqo_reset(C.byref(b))
    return b


def board_from(squares, move_count=0, player=0, red=0, black=0):
    b = Board()
    b.set_squares(squares)
    b.moveCount, b.currentPlayer, b.redScore, b.blackScore = int(move_count), int(player), int(red), int(black)
    return b


def all_valid_actions(b, player):
    codes = (C.c_uint16 * MAX_MOVES)()
    n = lib().xqo_all_valid_actions(C.byref(b), int(player), codes)
    return np.array(codes[:min(n, MAX_MOVES)], dtype=np.uint16), n


def valid_matrix(b):
    L = lib()
    m = np.zeros(8100, dtype=np.uint8)
    for f in range(90):
        for t in range(90):
            m[f * 90 + t] = L.xqo_is_valid_move(C.byref(b), f // 9, f % 9, t // 9, t % 9)
    return m


def state_indices(b):
    idx = (C.c_int * 90)()
    n = lib().xqo_state_indices(C.byref(b), idx)
    return np.array(idx[:n], dtype=np.int32)


def state_repr(b):
    s = np.zeros(1260, dtype=np.float64)
    lib().xqo_state_repr(C.byref(b), _pd(s))
    return s


def sizes_arr(sizes):
    return np.ascontiguousarray(np.asarray(sizes, dtype=np.int32))


def nn_counts(sizes):
    s = sizes_arr(sizes)
    return int(lib().xqo_nn_num_weights(_pi(s), len(s))), int(lib().xqo_nn_num_biases(_pi(s), len(s)))


def nn_forward(sizes, w, b, x):
    s = sizes_arr(sizes)
    out = np.zeros(int(s[-1]), dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    rc = lib().xqo_nn_forward(_pi(s), len(s), _pd(w), _pd(b), _pd(x), _pd(out))
    assert rc == 0
    return out


def nn_forward_all(sizes, w, b, x):
    s = sizes_arr(sizes)
    out = np.zeros(int(sum(s[1:])), dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    rc = lib().xqo_nn_forward_all(_pi(s), len(s), _pd(w), _pd(b), _pd(x), _pd(out))
    assert rc == 0
    return out


def nn_backprop(sizes, w, b, x, target, lr, mode=0):
    """In place on w, b (float64, C-contiguous)."""
    s = sizes_arr(sizes)
    x = np.ascontiguousarray(x, dtype=np.float64)
    target = np.ascontiguousarray(target, dtype=np.float64)
    return lib().xqo_nn_backprop(_pi(s), len(s), _pd(w), _pd(b), _pd(x), _pd(target), float(lr), int(mode))


def nn_accum_grad(sizes, w, b, x, target, mode, gw, gb):
    s = sizes_arr(sizes)
    x = np.ascontiguousarray(x, dtype=np.float64)
    target = np.ascontiguousarray(target, dtype=np.float64)
    return lib().xqo_nn_accum_grad(_pi(s), len(s), _pd(w), _pd(b), _pd(x), _pd(target), int(mode), _pd(gw), _pd(gb))


def td_target(sizes, w, b, state, next_state, action_to, reward, done, gamma):
    s = sizes_arr(sizes)
    tq = np.zeros(int(s[-1]), dtype=np.float64)
    state = np.ascontiguousarray(state, dtype=np.float64)
    next_state = np.ascontiguousarray(next_state, dtype=np.float64)
    rc = lib().xqo_td_target(_pi(s), len(s), _pd(w), _pd(b), _pd(state), _pd(next_state), int(action_to),
                             float(reward), int(done), float(gamma), _pd(tq))
    assert rc == 0
    return tq


# ---------------------------------------------------------------- build-defined extensions (configs[4])
def ext_forward(sizes, w, b, x, bf16=False):
    """(hidden activations concat, z of the output layer) in fp64 or in the bf16 Q-net arithmetic."""
    s = sizes_arr(sizes)
    acts = np.zeros(int(sum(s[1:-1])), dtype=np.float64)
    z = np.zeros(int(s[-1]), dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    assert lib().xqo_ext_forward(_pi(s), len(s), _pd(w), _pd(b), _pd(x), int(bf16), _pd(acts), _pd(z)) == 0
    return acts, z


def ext_td_accum(sizes, w, b, wt, bt, state, next_state, action_to, reward, done, gamma, td_rule, mode, bf16, weight, gw, gb):
    """Accumulates the weighted TD gradient of one transition; returns (Q(s,a), y, a*)."""
    s = sizes_arr(sizes)
    q, y, a = C.c_double(), C.c_double(), C.c_int()
    state = np.ascontiguousarray(state, dtype=np.float64)
    next_state = np.ascontiguousarray(next_state, dtype=np.float64)
    rc = lib().xqo_ext_td_accum(_pi(s), len(s), _pd(w), _pd(b), _pd(wt), _pd(bt), _pd(state), _pd(next_state), int(action_to),
                                float(reward), int(done), float(gamma), int(td_rule), int(mode), int(bf16), float(weight),
                                _pd(gw), _pd(gb), C.byref(q), C.byref(y), C.byref(a))
    assert rc == 0
    return q.value, y.value, a.value


def per_build(prio):
    prio = np.ascontiguousarray(prio, dtype=np.float32)
    tree = np.zeros(lib().xqo_per_tree_floats(len(prio)), dtype=np.float32)
    pf = C.POINTER(C.c_float)
    lib().xqo_per_build(prio.ctypes.data_as(pf), len(prio), tree.ctypes.data_as(pf))
    return tree


def per_sample(tree, capacity, batch, seed, call, n_eligible, beta):
    slots = np.zeros(batch, dtype=np.int32)
    w = np.zeros(batch, dtype=np.float32)
    pf = C.POINTER(C.c_float)
    wmax = lib().xqo_per_sample(tree.ctypes.data_as(pf), int(capacity), int(batch), C.c_uint64(int(seed)), int(call),
                                int(n_eligible), float(beta), slots.ctypes.data_as(C.POINTER(C.c_int32)), w.ctypes.data_as(pf))
    return slots, w, float(wmax)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*[int(x) & 0xFFFFFFFF for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) & 0xFFFFFFFF for x in key])
    o = (C.c_uint32 * 4)()
    lib().xqo_philox4x32(c, k, o)
    return [int(x) for x in o]


def selfplay_step(b, q90, seed, game_id, step_id, eps_u32):
    out = StepOut()
    if q90 is None:
        qp = None
    else:
        q90 = np.ascontiguousarray(q90, dtype=np.float32)
        qp = q90.ctypes.data_as(C.POINTER(C.c_float))
    lib().xqo_selfplay_step(C.byref(b), qp, C.c_uint64(int(seed)), int(game_id), int(step_id), int(eps_u32),
                            C.byref(out))
    return out


def eps_to_u32(eps):
    return int(min(max(float(eps), 0.0) * 4294967296.0, 4294967295.0))


def init_weights(sizes, seed=1):
    """U(-0.05, 0.05) weights, zero biases (dqn.cu:96-123 distribution; the reference's own stream is
    random_device-seeded, so only the distribution is reproducible).  float64, reference flat layout."""
    nw, nb = nn_counts(sizes)
    rng = np.random.default_rng(seed)
    return rng.uniform(-0.05, 0.05, size=nw), np.zeros(nb)


# record layout of oracle/ref/ref_driver.cpp (struct Record, packed)
REF_RECORD = np.dtype([
    ("board", "u1", (90,)), ("moveCount", "<i4"), ("player", "u1"), ("redScore", "<i4"), ("blackScore", "<i4"),
    ("over", "u1"), ("winner", "u1"), ("nRed", "<u2"), ("nBlack", "<u2"),
    ("red", "<u2", (128,)), ("black", "<u2", (128,)),
    ("fr", "i1"), ("fc", "i1"), ("tr", "i1"), ("tc", "i1"), ("valid", "u1"), ("captured", "u1")])
assert REF_RECORD.itemsize == 627

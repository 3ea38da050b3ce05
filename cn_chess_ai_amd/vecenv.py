"""VecEnv / ReplayBuffer — batched ChessBoard (reference include/chessboard.h:35-78) resident in HBM.

VecEnv is the build-defined batched form of the env half of ChessAI::train (reference chessai.cpp:90-119):
legal_moves() = getAllValidActions, step(actions) = movePiece + evaluateBoard + checkGameOver,
selfplay_step(q90, eps) = the whole ply on device with epsilon-greedy DQN::selectAction.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import StepResult, EpisodeRecord, call

RED, BLACK, NONE = 0, 1, 2

# reference chessboard.cpp:8-29
START_BOARD = np.zeros(90, dtype=np.uint8)
START_BOARD[0:9] = [5, 4, 3, 2, 1, 2, 3, 4, 5]
START_BOARD[81:90] = START_BOARD[0:9] + 7
START_BOARD[[19, 25]] = 6
START_BOARD[[64, 70]] = 13
START_BOARD[27:36:2] = 7
START_BOARD[54:63:2] = 14

STEP_DTYPE = np.dtype([("action", "<i4"), ("n_moves", "<i4"), ("reward", "<i4"), ("captured", "u1"), ("valid", "u1"),
                       ("done", "u1"), ("terminated", "u1"), ("winner", "u1"), ("explored", "u1"),
                       ("move_count", "<u2"), ("red_score", "<i2"), ("black_score", "<i2")])
EPISODE_DTYPE = np.dtype([("game_id", "<u4"), ("episode", "<u4"), ("red_score", "<i2"), ("black_score", "<i2"),
                          ("move_count", "<u2"), ("winner", "u1"), ("reserved", "u1")])
assert STEP_DTYPE.itemsize == 24 and EPISODE_DTYPE.itemsize == 16


def eps_to_u32(eps):
    """explore iff philox_word < eps_u32  (replaces `rand()/RAND_MAX < epsilon`, reference dqn.cpp:30-31)."""
    return int(min(max(float(eps), 0.0) * 4294967296.0, 4294967295.0))


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class ReplayBuffer:
    """ReplayBuffer{capacity; push(s,a,r,s',done); sample(B)} — element = the 5-tuple of DQN::train (dqn.cpp:157)."""

    def __init__(self, capacity, seed=0, stream=None, _handle=None):
        self._own = _handle is None
        if _handle is None:
            h = C.c_void_p()
            call("xq_replay_create", int(capacity), int(seed), stream, C.byref(h))
            _handle = h
        self._h = _handle

    def close(self):
        if self._h is not None and self._own:
            call("xq_replay_destroy", self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def stream(self):
        """The HIP stream the ring runs on (its own when created without one), as an int."""
        s = C.c_void_p()
        call("xq_replay_stream", self._h, C.byref(s))
        return s.value

    def stats(self):
        size, cap, tot = C.c_int32(), C.c_int32(), C.c_uint64()
        call("xq_replay_size", self._h, C.byref(size), C.byref(cap), C.byref(tot))
        return size.value, cap.value, tot.value

    def __len__(self):
        return self.stats()[0]

    def push(self, boards, action_to, reward, done, next_boards):
        boards = np.ascontiguousarray(boards, dtype=np.uint8).reshape(-1, 90)
        next_boards = np.ascontiguousarray(next_boards, dtype=np.uint8).reshape(-1, 90)
        n = len(boards)
        a = np.ascontiguousarray(action_to, dtype=np.int32).reshape(n)
        r = np.ascontiguousarray(reward, dtype=np.float32).reshape(n)
        d = np.ascontiguousarray(done, dtype=np.uint8).reshape(n)
        call("xq_replay_push_host", self._h, n, _ptr(boards, C.c_uint8), _ptr(a, C.c_int32), _ptr(r, C.c_float),
             _ptr(d, C.c_uint8), _ptr(next_boards, C.c_uint8))

    def sample(self, batch, host=True):
        """host=False: the draw is only queued (no copy of the slots back, no synchronisation)."""
        if not host:
            call("xq_replay_sample", self._h, int(batch), None)
            return None
        slots = np.zeros(batch, dtype=np.int32)
        call("xq_replay_sample", self._h, int(batch), _ptr(slots, C.c_int32))
        return slots

    def sample_window(self, batch, start, count, host=True):
        """sample(B) from the `count` ring slots that start at `start` (wrapping)."""
        if not host:
            call("xq_replay_sample_window", self._h, int(batch), int(start), int(count), None)
            return None
        slots = np.zeros(batch, dtype=np.int32)
        call("xq_replay_sample_window", self._h, int(batch), int(start), int(count), _ptr(slots, C.c_int32))
        return slots

    # ---- prioritized replay (build-defined, BASELINE configs[4]) ----
    def enable_per(self, alpha=0.6, beta=0.4, eps=1e-3):
        call("xq_replay_enable_per", self._h, float(alpha), float(beta), float(eps))

    def per_rebuild(self, retire_start=0, retire_count=0):
        call("xq_replay_per_rebuild", self._h, int(retire_start), int(retire_count))

    def sample_prioritized(self, batch, host=True):
        """(slots, normalised importance weights) of a stratified proportional draw from the tree as of the last rebuild."""
        if not host:
            call("xq_replay_sample_prioritized", self._h, int(batch), None, None)
            return None
        slots = np.zeros(batch, dtype=np.int32)
        w = np.zeros(batch, dtype=np.float32)
        call("xq_replay_sample_prioritized", self._h, int(batch), _ptr(slots, C.c_int32), _ptr(w, C.c_float))
        return slots, w

    def set_priorities(self, prio, first=0):
        p = np.ascontiguousarray(prio, dtype=np.float32)
        call("xq_replay_set_priorities", self._h, int(first), len(p), _ptr(p, C.c_float))

    def get_priorities(self, first=0, n=None):
        n = self.stats()[1] - first if n is None else n
        p = np.zeros(n, dtype=np.float32)
        call("xq_replay_get_priorities", self._h, int(first), int(n), _ptr(p, C.c_float))
        return p

    def per_stats(self):
        t, m, k = C.c_float(), C.c_float(), C.c_int32()
        call("xq_replay_per_stats", self._h, C.byref(t), C.byref(m), C.byref(k))
        return dict(total=t.value, max_priority=m.value, n_eligible=k.value)

    def get(self, slot):
        b, nb = np.zeros(90, np.uint8), np.zeros(90, np.uint8)
        a, r, d = C.c_int32(), C.c_float(), C.c_uint8()
        call("xq_replay_get", self._h, int(slot), _ptr(b, C.c_uint8), C.byref(a), C.byref(r), C.byref(d),
             _ptr(nb, C.c_uint8))
        return b, a.value, r.value, d.value, nb


class VecEnv:
    def __init__(self, n_games, seed=0x5EED, first_game_id=0, stream=None, _handle=None):
        self._own = _handle is None
        if _handle is None:
            h = C.c_void_p()
            call("xq_env_create", int(n_games), int(seed), int(first_game_id), stream, C.byref(h))
            _handle = h
        self._h = _handle
        n = C.c_int32()
        call("xq_env_num_games", self._h, C.byref(n))
        self.n_games = n.value

    def close(self):
        if self._h is not None and self._own:
            call("xq_env_destroy", self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def stream(self):
        s = C.c_void_p()
        call("xq_env_stream", self._h, C.byref(s))
        return s.value

    def reset(self):
        call("xq_env_reset", self._h)

    def set_state(self, boards, meta=None, first=0):
        boards = np.ascontiguousarray(boards, dtype=np.uint8).reshape(-1, 90)
        n = len(boards)
        mp = None
        if meta is not None:
            meta = np.ascontiguousarray(meta, dtype=np.int32).reshape(n, 4)
            mp = _ptr(meta, C.c_int32)
        call("xq_env_set_state", self._h, int(first), n, _ptr(boards, C.c_uint8), mp)

    def get_state(self, first=0, n=None):
        n = self.n_games - first if n is None else n
        boards = np.zeros((n, 90), dtype=np.uint8)
        meta = np.zeros((n, 4), dtype=np.int32)
        call("xq_env_get_state", self._h, int(first), int(n), _ptr(boards, C.c_uint8), _ptr(meta, C.c_int32))
        return boards, meta

    def legal_moves(self, player=-1):
        """-> (codes [n][128] u16 in canonical order, counts [n])"""
        codes = np.zeros((self.n_games, _capi.MAX_MOVES), dtype=np.uint16)
        counts = np.zeros(self.n_games, dtype=np.int32)
        call("xq_env_legal_moves", self._h, int(player), _ptr(codes, C.c_uint16), _ptr(counts, C.c_int32))
        return codes, counts

    def valid_matrix(self, game):
        m = np.zeros(8100, dtype=np.uint8)
        call("xq_env_valid_matrix", self._h, int(game), _ptr(m, C.c_uint8))
        return m

    def rule_matrix(self, game):
        """[7][8100] results of isValid{General..Soldier}Move (chessboard.h:50-56) for every in-board (from, to) of a game."""
        m = np.zeros((7, 8100), dtype=np.uint8)
        call("xq_env_rule_matrix", self._h, int(game), _ptr(m, C.c_uint8))
        return m

    def rule_query(self, game, piece_type, fr, fc, tr, tc):
        ok = C.c_int32()
        call("xq_env_rule_query", self._h, int(game), int(piece_type), int(fr), int(fc), int(tr), int(tc), C.byref(ok))
        return bool(ok.value)

    def get_winner(self, first=0, n=None):
        """ChessBoard::getWinner() per game (colour of the first general in index order)."""
        n = self.n_games - first if n is None else n
        w = np.zeros(n, dtype=np.uint8)
        call("xq_env_get_winner", self._h, int(first), int(n), _ptr(w, C.c_uint8))
        return w

    def step(self, actions, auto_reset=True):
        actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.n_games)
        res = np.zeros(self.n_games, dtype=STEP_DTYPE)
        call("xq_env_step", self._h, _ptr(actions, C.c_int32), int(bool(auto_reset)),
             res.ctypes.data_as(C.POINTER(StepResult)))
        return res

    def selfplay_step(self, q90=None, eps=0.1):
        """One ply everywhere.  q90: [n][90] float32 Q-values of outputs 0..89 (None = uniform random policy)."""
        res = np.zeros(self.n_games, dtype=STEP_DTYPE)
        qp = None
        if q90 is not None:
            q90 = np.ascontiguousarray(q90, dtype=np.float32).reshape(self.n_games, 90)
            qp = _ptr(q90, C.c_float)
        call("xq_env_selfplay_step_host", self._h, qp, eps_to_u32(eps), res.ctypes.data_as(C.POINTER(StepResult)))
        return res

    def selfplay_step_dev(self, q90_dev=0, q_stride=96, eps=0.1, results_dev=None, replay=None):
        """Asynchronous, device pointers only (ints): the hot-loop form."""
        call("xq_env_selfplay_step", self._h, C.c_void_p(q90_dev) if q90_dev else None, int(q_stride), eps_to_u32(eps),
             C.c_void_p(results_dev) if results_dev else None, replay.handle if replay is not None else None)

    def drain_episodes(self, max_records=65536):
        rec = np.zeros(max_records, dtype=EPISODE_DTYPE)
        n, total = C.c_int32(), C.c_uint64()
        call("xq_env_drain_episodes", self._h, rec.ctypes.data_as(C.POINTER(EpisodeRecord)), int(max_records),
             C.byref(n), C.byref(total))
        return rec[:n.value], total.value

    def counters(self):
        c = (C.c_uint64 * 6)()
        call("xq_env_counters", self._h, c)
        return dict(zip(("plies", "episodes", "red_wins", "black_wins", "captures", "explored"), [int(x) for x in c]))

    def boards_dev(self):
        return _capi.load().xq_env_boards_dev(self._h)

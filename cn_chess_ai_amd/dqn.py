"""DQN / Trainer — host-side mirror of the reference's DQN (include/dqn.h:97-116) and ChessAI::train
(src/chessai.cpp:85-170) over the C ABI.  Same method names and argument meaning as upstream; all arithmetic
runs in libxqhip.so on the GPU (fp32 MFMA), fp64 at the boundary like upstream's std::vector<double>.
"""
import ctypes as C
import random

import numpy as np

from . import _capi
from ._capi import call, KernelStat, KernelSpan, TrainerConfig as _CConfig
from .vecenv import VecEnv, ReplayBuffer, _ptr

REFERENCE_LAYERS = (1260, 128, 8100)     # ChessAI::initializeDQN, chessai.cpp:395-404


class DQN:
    def __init__(self, layer_sizes=REFERENCE_LAYERS, learning_rate=0.001, gamma=0.99, seed=1, stream=None, _handle=None):
        self.layer_sizes = tuple(int(x) for x in layer_sizes)
        self.learning_rate, self.gamma = float(learning_rate), float(gamma)
        self._own = _handle is None
        if _handle is None:
            sizes = (C.c_int32 * len(self.layer_sizes))(*self.layer_sizes)
            h = C.c_void_p()
            call("xq_dqn_create", sizes, len(self.layer_sizes), self.learning_rate, self.gamma, int(seed), stream,
                 C.byref(h))
            _handle = h
        self._h = _handle
        nw, nb = C.c_size_t(), C.c_size_t()
        call("xq_dqn_num_params", self._h, C.byref(nw), C.byref(nb))
        self.n_weights, self.n_biases = nw.value, nb.value
        self._rng = random.Random(seed)

    def close(self):
        if self._h is not None and self._own:
            call("xq_dqn_destroy", self._h)
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
        call("xq_dqn_stream", self._h, C.byref(s))
        return s.value

    # ---- parameters in the reference flat layout (host_weights / host_biases, dqn.h:47-50) ----
    def set_params(self, weights, biases, net=_capi.NET_ONLINE):
        w = np.ascontiguousarray(weights, dtype=np.float64).reshape(self.n_weights)
        b = np.ascontiguousarray(biases, dtype=np.float64).reshape(self.n_biases)
        call("xq_dqn_set_params", self._h, int(net), _ptr(w, C.c_double), _ptr(b, C.c_double))

    def get_params(self, net=_capi.NET_ONLINE):
        w = np.zeros(self.n_weights, dtype=np.float64)
        b = np.zeros(self.n_biases, dtype=np.float64)
        call("xq_dqn_get_params", self._h, int(net), _ptr(w, C.c_double), _ptr(b, C.c_double))
        return w, b

    # ---- DQN API (dqn.h:102-109) ----
    def getQValues(self, state, net=_capi.NET_ONLINE):
        """state: [L0] or [n][L0] doubles -> [nout] or [n][nout]."""
        x = np.ascontiguousarray(state, dtype=np.float64)
        single = x.ndim == 1
        x = x.reshape(-1, self.layer_sizes[0])
        q = np.zeros((len(x), self.layer_sizes[-1]), dtype=np.float64)
        call("xq_dqn_forward", self._h, int(net), _ptr(x, C.c_double), len(x), _ptr(q, C.c_double))
        return q[0] if single else q

    def selectAction(self, state, epsilon, valid_actions, rand1=None, rand2=None):
        """DQN::selectAction (dqn.cpp:24-56).  valid_actions: sequence of (from, to).  rand1 in [0,1) and rand2
        (non-negative int) replace the two rand() draws when given (parity tests inject them)."""
        if len(valid_actions) == 0:
            raise RuntimeError("No valid actions available.")
        r1 = self._rng.random() if rand1 is None else rand1
        if r1 < epsilon:
            r2 = self._rng.randrange(1 << 31) if rand2 is None else rand2
            return valid_actions[r2 % len(valid_actions)]
        q = self.getQValues(state)
        best, maxq = valid_actions[0], -np.inf
        for a in valid_actions:
            if a[1] >= len(q):
                continue
            if q[a[1]] > maxq:
                maxq, best = q[a[1]], a
        return best

    def backpropagate(self, state, target, learning_rate=None, grad_scale=1.0, mode=_capi.BACKPROP_REFERENCE):
        x = np.ascontiguousarray(state, dtype=np.float64).reshape(-1, self.layer_sizes[0])
        t = np.ascontiguousarray(target, dtype=np.float64).reshape(-1, self.layer_sizes[-1])
        assert len(x) == len(t)
        lr = self.learning_rate if learning_rate is None else learning_rate
        call("xq_dqn_backpropagate", self._h, _ptr(x, C.c_double), _ptr(t, C.c_double), len(x), float(lr),
             float(grad_scale), int(mode))

    def updateTargetNetwork(self):
        call("xq_dqn_update_target", self._h)

    def saveModel(self, filename):
        call("xq_dqn_save_model", self._h, str(filename).encode())

    def loadModel(self, filename):
        call("xq_dqn_load_model", self._h, str(filename).encode())

    def train(self, state, action, reward, next_state, done, mode=_capi.BACKPROP_REFERENCE):
        """DQN::train (dqn.cpp:157-172): single transition, target network for max Q(s')."""
        q = self.getQValues(state)
        if done:
            q[action] = reward
        else:
            q[action] = reward + self.gamma * self.getQValues(next_state, net=_capi.NET_TARGET).max()
        self.backpropagate(state, q, self.learning_rate, 1.0, mode)

    # ---- batched device-side TD step on packed boards ----
    def td_update(self, boards, next_boards, action_to, reward, done, td_net=_capi.TD_ONLINE_NET,
                  mode=_capi.BACKPROP_REFERENCE, learning_rate=None, grad_scale=1.0):
        b = np.ascontiguousarray(boards, dtype=np.uint8).reshape(-1, 90)
        nb = np.ascontiguousarray(next_boards, dtype=np.uint8).reshape(-1, 90)
        n = len(b)
        a = np.ascontiguousarray(action_to, dtype=np.int32).reshape(n)
        r = np.ascontiguousarray(reward, dtype=np.float32).reshape(n)
        d = np.ascontiguousarray(done, dtype=np.uint8).reshape(n)
        qsa, y = np.zeros(n, np.float32), np.zeros(n, np.float32)
        lr = self.learning_rate if learning_rate is None else learning_rate
        call("xq_dqn_td_update_host", self._h, n, _ptr(b, C.c_uint8), _ptr(nb, C.c_uint8), _ptr(a, C.c_int32),
             _ptr(r, C.c_float), _ptr(d, C.c_uint8), int(td_net), int(mode), float(lr), float(grad_scale),
             _ptr(qsa, C.c_float), _ptr(y, C.c_float))
        return qsa, y

    def q_boards(self, env, n_out=96, net=_capi.NET_ONLINE):
        """Q-values of the first n_out outputs for every board of a VecEnv (device -> host copy via torch-free path)."""
        import torch
        q = torch.empty((env.n_games, n_out), dtype=torch.float32, device="cuda")
        call("xq_dqn_forward_boards_dev", self._h, int(net), C.c_void_p(env.boards_dev()), env.n_games, int(n_out),
             C.c_void_p(q.data_ptr()), int(n_out))
        call("xq_stream_synchronize", None)
        torch.cuda.synchronize()
        return q

    def select_q(self, env):
        """Q(s)[0..95] of every board of a VecEnv through the select chain of the self-play loop (xq_dqn_select_q_dev)."""
        import torch
        q = torch.empty((env.n_games, 96), dtype=torch.float32, device="cuda")
        call("xq_dqn_select_q_dev", self._h, C.c_void_p(env.boards_dev()), env.n_games, C.c_void_p(q.data_ptr()))
        call("xq_stream_synchronize", None)
        torch.cuda.synchronize()
        return q

    def td_grads_replay(self, replay, batch, td_net=_capi.TD_ONLINE_NET, mode=_capi.BACKPROP_REFERENCE):
        """Gradients of one TD minibatch taken from a replay ring (batch = 0: the whole filled ring, in order)."""
        call("xq_dqn_td_grads_replay", self._h, replay.handle, int(batch), int(td_net), int(mode))

    def apply_grads(self, learning_rate=None, grad_scale=1.0):
        lr = self.learning_rate if learning_rate is None else learning_rate
        call("xq_dqn_apply_grads", self._h, float(lr), float(grad_scale))

    def grad_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        call("xq_dqn_grad_buffer", self._h, C.byref(p), C.byref(n))
        return p.value, n.value

    def last_loss(self):
        v = C.c_double()
        call("xq_dqn_last_loss", self._h, C.byref(v))
        return v.value

    def last_td_values(self, n):
        """(Q(s,a), y) of the first n samples of the last TD step."""
        q, y = np.zeros(n, np.float32), np.zeros(n, np.float32)
        call("xq_dqn_last_td_values", self._h, int(n), _ptr(q, C.c_float), _ptr(y, C.c_float))
        return q, y

    def kernel_stats(self, enable=-1):
        """Returns the HIP-event timings collected so far; enable: 1 on, 0 off, 2 on + clear, -1 leave."""
        arr = (KernelStat * 64)()
        n = C.c_int32()
        call("xq_dqn_kernel_stats", self._h, int(enable), arr, 64, C.byref(n))
        return [dict(name=arr[i].name.decode(), ms=arr[i].ms, launches=arr[i].launches, flops=arr[i].flops,
                     bytes=arr[i].bytes, exact=arr[i].exact_launches) for i in range(n.value)]


def _kernel_filter(self, names=None):
    """Which launches kernel_stats(enable=3 / 4) brackets: an iterable of bracket names (None = the default three)."""
    call("xq_dqn_kernel_filter", self._h, ",".join(names).encode() if names else None)


DQN.kernel_filter = _kernel_filter


def _set_precision(self, precision):
    """_capi.PRECISION_F32 / PRECISION_BF16: arithmetic of the forward passes on packed boards (bf16 MFMA Q-net)."""
    call("xq_dqn_set_precision", self._h, int(precision))


DQN.set_precision = _set_precision


def _set_qmax_mode(self, mode):
    """_capi.QMAX_FULL / QMAX_SCREENED: how the TD step finds max_a' Q(s', a') (exact bf16 screening + fp32 re-evaluation)."""
    call("xq_dqn_set_qmax_mode", self._h, int(mode))


def _set_l0_derive(self, on):
    """Layer-0 sums of s' derived from those of s (online TD rule, fp32 net): one gather instead of two, a different summation order."""
    call("xq_dqn_set_l0_derive", self._h, 1 if on else 0)


DQN.set_l0_derive = _set_l0_derive


def _set_td_tail(self, on):
    """Gradient half of a TD step as fused launches on one stream (default) or kernel by kernel on two streams; same bits."""
    call("xq_dqn_set_td_tail", self._h, 1 if on else 0)


DQN.set_td_tail = _set_td_tail


def _set_l0_grad_mode(self, mode):
    """Layer-0 weight gradient: 1 = exact dense product on the bf16 matrix pipe (library default, where the shape allows), 0 = segmented sums."""
    call("xq_dqn_set_l0_grad_mode", self._h, int(mode))


DQN.set_l0_grad_mode = _set_l0_grad_mode


def _set_refine_stage(self, mode):
    """Exact screening, whole groups that many samples of a block share: -1 = through LDS while the screen's counters show many of them (default),
    0 = never, 1 = whenever the launch has room.  Same bits either way."""
    call("xq_dqn_set_refine_stage", self._h, int(mode))


DQN.set_refine_stage = _set_refine_stage


def _set_exchange_overlap(self, mode):
    """Data-parallel step: -1 auto, 0 select chain beside the gradient kernels, 1 beside the all-reduce (hides the exchange)."""
    call("xq_dqn_set_exchange_overlap", self._h, int(mode))


DQN.set_exchange_overlap = _set_exchange_overlap


def _calibrate_exchange(self, threshold_us=-1.0):
    """COLLECTIVE: time 20 all-reduces of the gradient buffer on the attached communicator; the auto setting of set_exchange_overlap
    then starts the select chain late iff their mean over the ranks exceeds threshold_us (< 0: the library's 41 us)."""
    us, late = C.c_double(0.0), C.c_int(0)
    call("xq_dqn_calibrate_exchange", self._h, float(threshold_us), C.byref(us), C.byref(late))
    return {"allreduce_us": us.value, "late_start": bool(late.value)}


def _exchange_calibration(self):
    """what the last calibration (explicit, or xq_dqn_set_comm's with more than one rank) measured and chose; None: none yet"""
    cal, late, us, thr = C.c_int(0), C.c_int(0), C.c_double(0.0), C.c_double(0.0)
    call("xq_dqn_exchange_calibration", self._h, C.byref(cal), C.byref(us), C.byref(thr), C.byref(late))
    if not cal.value:
        return None
    return {"allreduce_us": us.value, "threshold_us": thr.value, "late_start": bool(late.value),
            "rule": "select chain starts behind the gradient kernels (beside the all-reduce) iff the measured all-reduce exceeds the threshold"}


DQN.calibrate_exchange = _calibrate_exchange
DQN.exchange_calibration = _exchange_calibration


def _qmax_stats(self):
    """(TD steps screened, samples, candidate (sample, group) pairs, pairs re-evaluated as whole groups); synchronises."""
    st = (C.c_uint64 * 4)()
    call("xq_dqn_qmax_stats", self._h, st)
    return tuple(int(x) for x in st)


def _qmax_guard(self):
    """(fallbacks to the full product so far, TD steps of the current fallback still to run)."""
    f, h = C.c_uint64(), C.c_int32()
    call("xq_dqn_qmax_guard", self._h, C.byref(f), C.byref(h))
    return f.value, h.value


DQN.set_qmax_mode = _set_qmax_mode
DQN.qmax_stats = _qmax_stats
DQN.qmax_guard = _qmax_guard


def _set_comm(self, comm):
    """Attach an xq_comm (dist.Comm) or None: td_grads then all-reduces the gradient buffer itself."""
    call("xq_dqn_set_comm", self._h, comm.handle if comm is not None else None)


DQN.set_comm = _set_comm


def _set_fused_apply(self, on=True):
    """apply_grads sums the layer-0 gradient partials itself (single GPU: nothing reads the gradient buffer in between)."""
    call("xq_dqn_set_fused_apply", self._h, 1 if on else 0)


DQN.set_fused_apply = _set_fused_apply


def _timeline(self, max_spans=8192):
    """(name, start_ms, end_ms) of every launch bracketed in the session closed by the last kernel_stats() call."""
    arr = (KernelSpan * max_spans)()
    n = C.c_int32()
    call("xq_dqn_kernel_timeline", self._h, arr, max_spans, C.byref(n))
    return [(arr[i].name.decode(), arr[i].start_ms, arr[i].end_ms) for i in range(n.value)]


DQN.kernel_timeline = _timeline


def TrainerConfig(n_games=8192, layer_sizes=(1260, 256, 256, 8100), learning_rate=0.001, gamma=0.99, epsilon=0.1,
                  replay_capacity=1 << 20, minibatch=8192, td_net=_capi.TD_TARGET_NET,
                  backprop_mode=_capi.BACKPROP_REFERENCE, target_sync_interval=100, mean_gradient=1, seed=0x5EED,
                  first_game_id=0, collects_per_update=1, overlap_collect=0, prioritized=0, per_alpha=0.0, per_beta=0.0,
                  per_eps=0.0, precision=_capi.PRECISION_F32):
    c = _CConfig()
    c.n_games = n_games
    for i, s in enumerate(layer_sizes):
        c.layer_sizes[i] = int(s)
    c.n_sizes = len(layer_sizes)
    c.learning_rate, c.gamma, c.epsilon = learning_rate, gamma, epsilon
    c.replay_capacity, c.minibatch = replay_capacity, minibatch
    c.td_net, c.backprop_mode = td_net, backprop_mode
    c.target_sync_interval, c.mean_gradient = target_sync_interval, mean_gradient
    c.seed, c.first_game_id = seed, first_game_id
    c.collects_per_update = collects_per_update
    c.overlap_collect = overlap_collect
    c.prioritized, c.per_alpha, c.per_beta, c.per_eps = prioritized, per_alpha, per_beta, per_eps
    c.precision = precision
    return c


class Trainer:
    """ChessAI::train for thousands of games at once: collect() = one ply everywhere, learn_* = one minibatch update."""

    def __init__(self, config, stream=None):
        self.config = config
        h = C.c_void_p()
        call("xq_trainer_create", C.byref(config), stream, C.byref(h))
        self._h = h
        e, d, r = C.c_void_p(), C.c_void_p(), C.c_void_p()
        call("xq_trainer_env", self._h, C.byref(e))
        call("xq_trainer_dqn", self._h, C.byref(d))
        call("xq_trainer_replay", self._h, C.byref(r))
        self.env = VecEnv(0, _handle=e)
        self.dqn = DQN([config.layer_sizes[i] for i in range(config.n_sizes)], config.learning_rate, config.gamma,
                       _handle=d)
        self.replay = ReplayBuffer(0, _handle=r)

    def close(self):
        if self._h is not None:
            call("xq_trainer_destroy", self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def random_plies(self, n):
        """n uniform-random plies in every game, outside the replay ring and the step counters (desynchronises the games)."""
        call("xq_trainer_random_plies", self._h, int(n))

    def set_comm(self, comm):
        call("xq_trainer_set_comm", self._h, comm.handle if comm is not None else None)

    def set_td_net(self, td_net):
        call("xq_trainer_set_td_net", self._h, int(td_net))

    def collect(self):
        call("xq_trainer_collect", self._h)

    def learn_grads(self):
        call("xq_trainer_learn_grads", self._h)

    def learn_apply(self, world_size=1):
        call("xq_trainer_learn_apply", self._h, int(world_size))

    def step(self, n=1):
        call("xq_trainer_step", self._h, int(n))

    def synchronize(self):
        self.env.get_state(0, 1)

    def counters(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        call("xq_trainer_counters", self._h, C.byref(a), C.byref(b), C.byref(c))
        return dict(env_steps=a.value, updates=b.value, episodes=c.value)

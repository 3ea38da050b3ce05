"""Cross-stream ordering of handles that exchange device data (`pytest -m gpu`).  One case per row of the "Stream ordering" table in
include/xq_capi.h.  No upstream analogue: the reference runs one stream and synchronises the device after every launch
(dqn.cu:233-236), so its results are those of the fully synchronised leg below.

Every case runs the same call sequence with env, replay ring and Q-net each on a stream of its own:
  sync      : a device synchronisation after every call — the definition;
  ordered   : no host synchronisation; the producer is queued behind a GATE — a kernel that waits until the host releases it
              (xq_debug_stream_gate) — and a host timer releases the gate 0.1 ms / 30 ms after the producer was queued: until then the
              producer CANNOT have run, whatever the kernels' durations.  Both must give the bits of `sync`;
  unordered : the same (release after 50 ms) with the ordering class under test switched off (xq_debug_set_stream_ordering); must NOT give
              them — the case goes red if the ordering is removed from the library.  This negative control needs producer and consumer on
              different hardware queues (two streams that share one run in submission order: the consumer would sit behind the gate by
              accident).  Every rig PROBES that (gate one stream, watch the others drain) and the negative control is SKIPPED, not
              failed, where it does not hold.
The unordered legs only ever read initialised, in-range data (stale slot lists, old transitions), never wild pointers.
"""
import ctypes as C
import os
import threading
import time

os.environ.setdefault("XQ_DEBUG_API", "1")          # the gate / ordering switches are refused without it (include/xq_capi.h)

import numpy as np
import pytest

import xqoracle as xo
from test_dqn_gpu import CFG2_NET

pytestmark = pytest.mark.gpu

N, CAP = 1024, 4096
GATE_TIMEOUT_MS = 1500                               # the gate kernel's own exit, should the host never release it
LEG = {"name": "sync", "release_after_s": 0.0, "separate": True}


@pytest.fixture(scope="module")
def xq():
    import cn_chess_ai_amd as m
    assert m._capi.device_count() > 0
    return m


def dsync():
    import torch
    torch.cuda.synchronize()


class Gate:
    """a kernel on `stream` that ends when release() is called (or after GATE_TIMEOUT_MS); release is armed on a host timer"""

    def __init__(self, xq, stream, release_after_s):
        self.xq, self.h, self.done = xq, C.c_void_p(), False
        xq._capi.call("xq_debug_stream_gate", C.c_void_p(stream), GATE_TIMEOUT_MS, C.byref(self.h))
        self.timer = threading.Timer(release_after_s, self.release)
        self.timer.daemon = True
        self.timer.start()

    def release(self):
        if not self.done:
            self.done = True
            self.xq._capi.call("xq_debug_gate_release", self.h)

    def destroy(self):
        self.timer.cancel()
        self.release()
        dsync()
        self.xq._capi.call("xq_debug_gate_destroy", self.h)


def hold(xq, stream, gates):
    """queue a gate on `stream` in the unsynchronised legs (nothing in the `sync` leg: its every call is followed by a device
    synchronisation, which a closed gate would block)"""
    if LEG["name"] != "sync":
        gates.append(Gate(xq, stream, LEG["release_after_s"]))


def idle(xq, stream):
    v = C.c_int(0)
    xq._capi.call("xq_stream_query", C.c_void_p(stream), C.byref(v))
    return bool(v.value)


def queues_are_separate(xq, streams):
    """gate each stream in turn and watch a trivial kernel on the other two complete while the gate is closed"""
    ok = True
    for i, s in enumerate(streams):
        g = Gate(xq, s, 0.2)
        others = [t for j, t in enumerate(streams) if j != i]
        for t in others:
            xq._capi.call("xq_debug_stream_delay", C.c_void_p(t), 1)
        t0 = time.time()
        while time.time() - t0 < 0.1 and not all(idle(xq, t) for t in others):
            time.sleep(0.0005)
        ok = ok and all(idle(xq, t) for t in others) and not g.done
        g.destroy()
    return ok


class Rig:
    """env + ring + Q-net, each on a stream of its own — of three different PRIORITIES, which in practice puts them on different
    hardware queues (streams of one priority are multiplexed onto a few queues); whether it did is probed, see the module docstring."""

    def __init__(self, xq, per=False, cap=CAP, n=N, prefill=2, env_nonblocking=False):
        self.xq = xq
        self.streams, self.gates = [], []
        for prio, nb in ((-1, 1 if env_nonblocking else 0), (1, 0), (0, 0)):
            h = C.c_void_p()
            xq._capi.call("xq_stream_create", prio, nb, C.byref(h))
            self.streams.append(h)
        self.env = xq.VecEnv(n, seed=3, stream=self.streams[0])
        self.rp = xq.ReplayBuffer(cap, seed=11, stream=self.streams[1])
        if per:
            self.rp.enable_per(0.6, 0.4, 1e-3)
        self.d = xq.DQN(CFG2_NET, 0.001, 0.99, seed=1, stream=self.streams[2])
        w, b = xo.init_weights(CFG2_NET, 2)
        self.d.set_params(w, b)
        self.d.updateTargetNetwork()
        assert len({self.env.stream(), self.rp.stream(), self.d.stream()}) == 3
        for _ in range(prefill):
            self.env.selfplay_step_dev(0, 96, 0.1, replay=self.rp)
            dsync()
        if per:
            self.rp.per_rebuild()
            dsync()
        if prefill:
            # first calls allocate workspaces (hipMalloc / hipFree synchronise): make them now, so that the sequences under test only
            # ever queue work.  A TD step with learning rate 0 changes no parameter (its priorities, with `per`, equally in every leg).
            if per:
                self.rp.sample_prioritized(n, host=False)
            else:
                self.rp.sample_window(n, 0, n, host=False)
            self.d.td_grads_replay(self.rp, n, td_net=0, mode=0)
            self.d.apply_grads(0.0, 1.0)
            dsync()
            if per:
                self.rp.per_rebuild()
                dsync()
        if LEG["name"] == "unordered":
            LEG["separate"] = LEG["separate"] and queues_are_separate(xq, [self.env.stream(), self.rp.stream(), self.d.stream()])

    def delay(self, stream):
        hold(self.xq, stream, self.gates)

    def close(self):
        for g in self.gates:
            g.destroy()
        dsync()
        self.d.close(); self.rp.close(); self.env.close()
        for h in self.streams:
            self.xq._capi.call("xq_stream_destroy", h)


def run_leg(xq, name, mask, release_after_s, body, arg):
    LEG.update(name=name, release_after_s=release_after_s)
    xq._capi.call("xq_debug_set_stream_ordering", mask)
    try:
        return body(arg)
    finally:
        xq._capi.call("xq_debug_set_stream_ordering", xq._capi.ORDER_ALL)
        LEG.update(name="sync")


def legs(xq, cls, body):
    """body(sync: bool) -> comparable result; returns (sync, [ordered at 0.1 ms, ordered at 30 ms], unordered, queues separate?)."""
    ALL = xq._capi.ORDER_ALL
    LEG["separate"] = True
    return (run_leg(xq, "sync", ALL, 0.0, body, True),
            [run_leg(xq, "ordered", ALL, 0.0001, body, False), run_leg(xq, "ordered", ALL, 0.03, body, False)],
            run_leg(xq, "unordered", ALL & ~cls, 0.05, body, False), LEG["separate"])


def same(a, b):
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    return np.array_equal(np.asarray(a), np.asarray(b))


def check(res):
    sync, ordered, unordered, separate = res
    for o in ordered:
        assert same(sync, o), "unsynchronised loop differs from the synchronised one"
    if not separate:
        pytest.skip("producer and consumer streams share a hardware queue on this box: the ordering-off leg would pass by accident")
    assert not same(sync, unordered), "the case does not exercise the ordering it is named after"


# ----------------------------------------------------------------------------------------------- ring contents
def test_env_step_then_td_step_reads_the_new_slots(xq):
    """xq_env_selfplay_step(replay) on the env's stream -> xq_dqn_td_grads_replay on the Q-net's: RAW on the ring slots."""
    def body(sync):
        r = Rig(xq)
        s = (lambda: dsync()) if sync else (lambda: None)
        r.delay(r.env.stream())
        r.env.selfplay_step_dev(0, 96, 0.1, replay=r.rp); s()          # writes slots [2N, 3N)
        r.rp.sample_window(N, 2 * N, N, host=False); s()                # a minibatch of exactly those
        r.d.td_grads_replay(r.rp, N, td_net=0, mode=0); s()
        r.d.apply_grads(0.05, 1.0 / N); s()
        dsync()
        out = r.d.get_params()
        r.close()
        return out
    check(legs(xq, xq._capi.ORDER_RING_CONTENTS, body))


def test_td_step_then_env_step_overwrites_its_slots(xq):
    """xq_dqn_td_grads_replay still reading slots -> xq_env_selfplay_step that overwrites them (ring wrapped): WAR."""
    def body(sync):
        r = Rig(xq, cap=2 * N)                                          # full after the two prefill plies: the next ply writes [0, N)
        s = (lambda: dsync()) if sync else (lambda: None)
        r.rp.sample_window(N, 0, N, host=False); s()
        r.delay(r.d.stream())
        r.d.td_grads_replay(r.rp, N, td_net=0, mode=0); s()             # reads slots [0, N) ...
        r.env.selfplay_step_dev(0, 96, 0.1, replay=r.rp); s()           # ... which this ply overwrites
        r.d.apply_grads(0.05, 1.0 / N); s()
        dsync()
        out = r.d.get_params()
        r.close()
        return out
    check(legs(xq, xq._capi.ORDER_RING_CONTENTS, body))


def test_env_step_then_replay_get(xq):
    """xq_env_selfplay_step(replay) -> xq_replay_get of a slot it wrote (host read on the ring's stream).  The env runs on a NON-BLOCKING
    stream here: a stream created with default flags is ordered against the null-stream copy of the host read by
    the runtime itself."""
    def body(sync):
        r = Rig(xq, env_nonblocking=True)
        r.delay(r.env.stream())
        r.env.selfplay_step_dev(0, 96, 0.1, replay=r.rp)
        if sync:
            dsync()
        out = r.rp.get(2 * N + 5)
        r.close()
        return [np.asarray(x) for x in out]
    check(legs(xq, xq._capi.ORDER_RING_CONTENTS, body))


# ----------------------------------------------------------------------------------------------- priorities / tree
def test_env_step_then_rebuild_sees_the_new_priorities(xq):
    """xq_env_selfplay_step(replay) writes the new transitions' priorities -> xq_replay_per_rebuild reads the table."""
    def body(sync):
        r = Rig(xq, per=True)
        r.delay(r.env.stream())
        r.env.selfplay_step_dev(0, 96, 0.1, replay=r.rp)
        if sync:
            dsync()
        r.rp.per_rebuild()
        st = r.rp.per_stats()                                           # synchronises the device
        r.close()
        return [st["total"], st["n_eligible"]]
    check(legs(xq, xq._capi.ORDER_RING_PRIORITIES, body))


def test_td_step_then_rebuild_sees_the_td_error_priorities(xq):
    """The record lost in round 3: xq_dqn_td_grads_replay (prioritized) writes TD-error priorities on the Q-net's stream ->
    xq_replay_per_rebuild on the ring's stream builds the tree the next minibatch is drawn from."""
    def body(sync):
        r = Rig(xq, per=True)
        s = (lambda: dsync()) if sync else (lambda: None)
        outs = []
        for it in range(3):
            r.rp.sample_prioritized(N, host=False); s()
            r.delay(r.d.stream())
            r.d.td_grads_replay(r.rp, N, td_net=0, mode=0); s()
            r.d.apply_grads(0.05, 1.0 / N); s()
            r.rp.per_rebuild(); s()
        dsync()
        st = r.rp.per_stats()
        outs = [r.d.get_params(), r.rp.get_priorities(0, 2 * N), st["total"]]
        r.close()
        return outs
    check(legs(xq, xq._capi.ORDER_RING_PRIORITIES, body))


def test_rebuild_then_env_step_reads_the_maximum_snapshot(xq):
    """xq_replay_per_rebuild snapshots the running maximum -> xq_env_selfplay_step gives it to the new transitions."""
    def body(sync):
        r = Rig(xq, per=True)
        r.rp.set_priorities(np.full(8, 50.0, np.float32), first=0)       # raises the running maximum (host call, synchronous)
        r.delay(r.rp.stream())
        r.rp.per_rebuild()                                              # snapshot: 50
        if sync:
            dsync()
        r.env.selfplay_step_dev(0, 96, 0.1, replay=r.rp)                # new slots [2N, 3N) enter with the snapshot
        out = r.rp.get_priorities(2 * N, N)                             # synchronises the device
        r.close()
        return out
    res = legs(xq, xq._capi.ORDER_RING_PRIORITIES, body)
    check(res)
    assert (res[0][res[0] > 0] == np.float32(50.0)).all()


def test_rebuild_between_a_draw_and_its_td_step_keeps_the_batch_maximum(xq):
    """sample_prioritized -> per_rebuild -> td_grads_replay used to hand the TD step a zeroed batch-maximum weight
    (inf / NaN deltas).  The rebuild now leaves the slot to the unconsumed draw."""
    r = Rig(xq, per=True)
    r.rp.sample_prioritized(N)
    r.rp.per_rebuild()
    r.d.td_grads_replay(r.rp, N, td_net=0, mode=0)
    r.d.apply_grads(0.05, 1.0 / N)
    dsync()
    w, b = r.d.get_params()
    p = r.rp.get_priorities(0, 2 * N)
    assert np.isfinite(w).all() and np.isfinite(b).all() and np.isfinite(p).all()
    r.close()


# ----------------------------------------------------------------------------------------------- the draw
def test_draw_then_td_step_reads_the_new_list(xq):
    """xq_replay_sample on the ring's stream -> xq_dqn_td_grads_replay reads the slot list on the Q-net's stream."""
    def body(sync):
        r = Rig(xq)
        s = (lambda: dsync()) if sync else (lambda: None)
        r.rp.sample_window(N, 0, N, host=False); dsync()                # a valid older list for the unordered leg to fall back on
        r.delay(r.rp.stream())
        r.rp.sample_window(N, N, N, host=False); s()
        r.d.td_grads_replay(r.rp, N, td_net=0, mode=0); s()
        r.d.apply_grads(0.05, 1.0 / N)
        dsync()
        out = r.d.get_params()
        r.close()
        return out
    check(legs(xq, xq._capi.ORDER_RING_DRAW, body))


def test_td_step_then_next_draw_overwrites_its_list(xq):
    """xq_dqn_td_grads_replay still reading the slot list -> the next xq_replay_sample overwrites it (the first hazard of round 3,
    here made deterministic)."""
    def body(sync):
        r = Rig(xq)
        s = (lambda: dsync()) if sync else (lambda: None)
        r.rp.sample_window(N, 0, N, host=False); s()
        r.delay(r.d.stream())
        r.d.td_grads_replay(r.rp, N, td_net=0, mode=0); s()
        r.rp.sample_window(N, N, N, host=False); s()                    # overwrites the list the queued step reads
        r.d.apply_grads(0.05, 1.0 / N)
        dsync()
        out = r.d.get_params()
        r.close()
        return out
    check(legs(xq, xq._capi.ORDER_RING_DRAW, body))


# ----------------------------------------------------------------------------------------------- trainer: parameters vs the collect stream
def test_parameters_rewritten_between_trainer_iterations(xq):
    """with several plies per update the trainer's collects start behind an event recorded at the last
    learn_apply; an asynchronous parameter update of the caller's own queued on the handle's stream after that (here a TD step on the
    trainer's Q-net) was not ordered in front of the select chain."""
    def body(sync):
        cfg = xq.TrainerConfig(n_games=256, layer_sizes=[1260, 64, 64, 8100], learning_rate=0.05, epsilon=0.0, replay_capacity=4096,
                               minibatch=256, td_net=0, target_sync_interval=0, seed=5, collects_per_update=2, overlap_collect=1)
        t = xq.Trainer(cfg)
        t.step(3)
        dsync()
        t.replay.sample(256, host=False)                                # (allocates the slot list: not inside the sequence under test)
        dsync()
        stream = t.dqn.stream()
        gates = []
        for _ in range(2):
            hold(xq, stream, gates)
            t.replay.sample(256, host=False)                            # the caller's own update, asynchronous on the handle's stream
            t.dqn.td_grads_replay(t.replay, 256, td_net=0, mode=0)
            t.dqn.apply_grads(0.5, 1.0)                                 # a large step: the greedy moves change
            if sync:
                dsync()
            t.step(1)
            if sync:
                dsync()
        for g in gates:
            g.destroy()
        dsync()
        boards, meta = t.env.get_state()
        w, b = t.dqn.get_params()
        t.close()
        return [boards, meta, w, b]
    check(legs(xq, xq._capi.ORDER_TRAINER_PARAMS, body))


# ----------------------------------------------------------------------------------------------- raw device pointers: the caller orders
@pytest.mark.parametrize("direction", ["q_to_env", "boards_to_q"])
def test_stream_wait_stream_orders_raw_pointer_exchanges(xq, direction):
    """Device pointers handed from one handle to another are the caller's to order; xq_stream_wait_stream does it without a host
    synchronisation: Q-values from xq_dqn_select_q_dev into xq_env_selfplay_step, env boards into xq_dqn_select_q_dev."""
    import torch

    def body(mode):                      # "sync" | "wait" | "nothing"
        r = Rig(xq, prefill=0)
        q = torch.zeros((N, 96), dtype=torch.float32, device="cuda")
        xq._capi.call("xq_dqn_select_q_dev", r.d.handle, C.c_void_p(r.env.boards_dev()), N, C.c_void_p(q.data_ptr()))   # allocates
        dsync()
        q.zero_()
        dsync()
        es, ds = r.env.stream(), r.d.stream()
        if direction == "q_to_env":
            r.delay(ds)
            xq._capi.call("xq_dqn_select_q_dev", r.d.handle, C.c_void_p(r.env.boards_dev()), N, C.c_void_p(q.data_ptr()))
            if mode == "sync":
                dsync()
            elif mode == "wait":
                xq._capi.call("xq_stream_wait_stream", C.c_void_p(es), C.c_void_p(ds))
            r.env.selfplay_step_dev(q.data_ptr(), 96, 0.0)
            dsync()
            out = list(r.env.get_state())
        else:
            r.delay(es)
            r.env.selfplay_step_dev(0, 96, 0.1)
            if mode == "sync":
                dsync()
            elif mode == "wait":
                xq._capi.call("xq_stream_wait_stream", C.c_void_p(ds), C.c_void_p(es))
            xq._capi.call("xq_dqn_select_q_dev", r.d.handle, C.c_void_p(r.env.boards_dev()), N, C.c_void_p(q.data_ptr()))
            dsync()
            out = [q.cpu().numpy()]
        r.close()
        return out
    ALL = xq._capi.ORDER_ALL
    LEG["separate"] = True
    check((run_leg(xq, "sync", ALL, 0.0, body, "sync"),
           [run_leg(xq, "ordered", ALL, 0.0001, body, "wait"), run_leg(xq, "ordered", ALL, 0.03, body, "wait")],
           run_leg(xq, "unordered", ALL, 0.05, body, "nothing"), LEG["separate"]))

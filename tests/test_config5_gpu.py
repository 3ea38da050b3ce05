"""GPU parity tests of the build-defined modes of BASELINE configs[4] — Double DQN, proportional prioritized replay, bf16 MFMA
Q-net — against their definition in oracle/xq_oracle_ext.c (`pytest -m gpu`).  Nothing upstream to pin them to ("parity
unpinned"): the reference has no such code.

Tolerances: fp32 modes as everywhere (Q 1e-4, parameters 2e-5); sampled replay slots and the sum tree bit-exact; the bf16 Q-net
within 1e-2 on Q (one bf16 ulp of a hidden activation is 2^-8 relative; device accumulates in fp32, the oracle in fp64, so a
tanh that lands on a rounding boundary may round the other way).
"""
import ctypes as C
import os

import numpy as np
import pytest

import xqoracle as xo
from test_dqn_gpu import QTOL, PTOL, CFG2_NET, make_net, transitions, valid_indices

pytestmark = pytest.mark.gpu

CFG4_NET = [1260, 512, 512, 512, 8100]
BF16_QTOL = 1e-2


@pytest.fixture(scope="module")
def xq():
    import cn_chess_ai_amd as m
    assert m._capi.device_count() > 0
    return m


@pytest.fixture(scope="module")
def trace(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_trace.npz"))


def oracle_update(sizes, w, b, wt, bt, S, A, R, D, S2, gamma, lr, scale, mode, td_rule, bf16=False, weights=None):
    gw, gb = np.zeros_like(w), np.zeros_like(b)
    qs, ys, stars = [], [], []
    for i, (s, a, r, dn, s2) in enumerate(zip(S, A, R, D, S2)):
        q, y, star = xo.ext_td_accum(sizes, w, b, wt, bt, xo.state_repr(xo.board_from(s)), xo.state_repr(xo.board_from(s2)), int(a),
                                     float(r), int(dn), gamma, td_rule, mode, bf16, 1.0 if weights is None else float(weights[i]),
                                     gw, gb)
        qs.append(q); ys.append(y); stars.append(star)
    return w - lr * scale * gw, b - lr * scale * gb, np.array(qs), np.array(ys), np.array(stars)


# ------------------------------------------------------------------------------------------------ Double DQN (fp32)
@pytest.mark.parametrize("sizes,mode,n", [(CFG2_NET, 0, 48), (CFG2_NET, 1, 48), (CFG4_NET, 0, 12)])
def test_double_dqn_td_update_matches_oracle(xq, trace, sizes, mode, n):
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, n, seed=2))
    d, w, b = make_net(xq, sizes, seed=8)
    wt, bt = xo.init_weights(sizes, 77)
    bt = np.random.default_rng(5).uniform(-0.05, 0.05, size=len(bt))
    d.set_params(wt, bt, net=1)
    R = R / 1000.0
    lr, scale = 0.05, 1.0 / n
    want_w, want_b, want_q, want_y, stars = oracle_update(sizes, w, b, wt, bt, S, A, R, D, S2, 0.99, lr, scale, mode, 2)
    qsa, y = d.td_update(S, S2, A, R, D, td_net=xq._capi.TD_DOUBLE, mode=mode, learning_rate=lr, grad_scale=scale)
    assert np.abs(qsa - want_q).max() < QTOL and np.abs(y - want_y).max() < QTOL
    got_w, got_b = d.get_params()
    assert np.abs(got_w - want_w).max() < PTOL and np.abs(got_b - want_b).max() < PTOL
    # the rule really differs from both single-net rules on this data
    _, _, _, y_online, _ = oracle_update(sizes, w, b, wt, bt, S, A, R, D, S2, 0.99, lr, scale, mode, 0)
    _, _, _, y_target, _ = oracle_update(sizes, w, b, wt, bt, S, A, R, D, S2, 0.99, lr, scale, mode, 1)
    assert np.abs(want_y - y_online).max() > 1e-3 and np.abs(want_y - y_target).max() > 1e-3
    assert len(set(stars[D == 0].tolist())) > 1
    d.close()


def test_double_dqn_persistent_argmax_kernel(xq, trace):
    """n = 1100 samples: the persistent column-ARG-max GEMM (128x128 tiles, 2 blocks per CU) instead of the plain kernel.
    y must be the target net's value at the online net's first maximum (near-ties within 2e-6 of the maximum accepted)."""
    sizes, n = CFG2_NET, 1100
    idx = valid_indices(trace, n, seed=9)
    S, A, R, D, S2 = transitions(trace, idx)
    d, w, b = make_net(xq, sizes, seed=11)
    wt, bt = xo.init_weights(sizes, 12)
    d.set_params(wt, bt, net=1)
    R = R / 1000.0
    qsa, y = d.td_update(S, S2, A, R, D, td_net=xq._capi.TD_DOUBLE, mode=0, learning_rate=0.0, grad_scale=1.0)
    exact = 0
    for i in range(0, n, 7):
        if D[i]:
            assert abs(y[i] - R[i]) < 1e-6
            continue
        x2 = xo.state_repr(xo.board_from(S2[i]))
        _, zo = xo.ext_forward(sizes, w, b, x2)
        _, zt = xo.ext_forward(sizes, wt, bt, x2)
        star = int(np.argmax(zo))
        cand = np.nonzero(zo >= zo[star] - 2e-6)[0]
        ys = R[i] + 0.99 * np.tanh(zt[cand])
        assert np.abs(ys - y[i]).min() < QTOL, i
        exact += abs(R[i] + 0.99 * np.tanh(zt[star]) - y[i]) < QTOL
    assert exact > 100
    d.close()


# ------------------------------------------------------------------------------------------------ bf16 Q-net
@pytest.mark.parametrize("sizes", [CFG2_NET, CFG4_NET, [1260, 128, 8100]])
def test_bf16_forward_matches_oracle(xq, trace, sizes):
    d, w, b = make_net(xq, sizes, seed=3)
    d.set_precision(xq._capi.PRECISION_BF16)
    n = 64
    boards = trace["board"][np.random.default_rng(1).choice(len(trace["board"]), n, replace=False)]
    env = xq.VecEnv(n)
    env.set_state(boards)
    q = d.q_boards(env, 8100).cpu().numpy()
    q96 = d.q_boards(env, 96).cpu().numpy()
    assert np.array_equal(q96, q[:, :96])                      # same arithmetic for the select head
    want = np.stack([np.tanh(xo.ext_forward(sizes, w, b, xo.state_repr(xo.board_from(bd)), bf16=True)[1]) for bd in boards[:24]])
    err = np.abs(q[:24] - want)
    assert err.max() < BF16_QTOL
    assert np.median(err) < 2e-4                                # most outputs agree far better: only boundary roundings differ
    # and it is a different function from the fp32 net (the rounding is really applied)
    d.set_precision(xq._capi.PRECISION_F32)
    q32 = d.q_boards(env, 8100).cpu().numpy()
    assert 1e-5 < np.abs(q32 - q).max() < 5e-2
    env.close(); d.close()


@pytest.mark.parametrize("sizes,td_rule,n", [(CFG2_NET, 2, 48), (CFG2_NET, 0, 48), (CFG4_NET, 2, 12)])
def test_bf16_td_update_matches_oracle(xq, trace, sizes, td_rule, n):
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, n, seed=4))
    d, w, b = make_net(xq, sizes, seed=9)
    wt, bt = xo.init_weights(sizes, 78)
    d.set_params(wt, bt, net=1)
    d.set_precision(xq._capi.PRECISION_BF16)
    R = R / 1000.0
    lr, scale = 0.05, 1.0 / n
    want_w, want_b, want_q, want_y, _ = oracle_update(sizes, w, b, wt, bt, S, A, R, D, S2, 0.99, lr, scale, 0, td_rule, bf16=True)
    qsa, y = d.td_update(S, S2, A, R, D, td_net=td_rule, mode=0, learning_rate=lr, grad_scale=scale)
    assert np.abs(qsa - want_q).max() < BF16_QTOL
    same_action = np.abs(y - want_y) < 2 * BF16_QTOL
    if td_rule == 2:
        # an arg-max is not continuous: where the two best online outputs lie within bf16 noise of each other the device may pick
        # the other one — its y must then be the target net's value at one of those near-maximal actions
        for i in np.nonzero(~same_action)[0]:
            x2 = xo.state_repr(xo.board_from(S2[i]))
            zo, zt = xo.ext_forward(sizes, w, b, x2, bf16=True)[1], xo.ext_forward(sizes, wt, bt, x2, bf16=True)[1]
            cand = np.nonzero(zo >= zo.max() - 5e-3)[0]
            assert np.abs(R[i] + 0.99 * np.tanh(zt[cand]) - y[i]).min() < 2 * BF16_QTOL, i
        assert same_action.mean() > 0.7
    else:
        assert same_action.all()
    got_w, got_b = d.get_params()
    dw = np.abs(want_w - w).max()
    assert dw > 1e-4
    if same_action.all():
        assert np.abs(got_w - want_w).max() < 0.05 * dw + 1e-6 and np.abs(got_b - want_b).max() < 0.05 * np.abs(want_b - b).max() + 1e-6
    # the bf16 shadow follows the master weights: Q after the update matches the oracle's bf16 forward of the updated parameters
    env = xq.VecEnv(4)
    env.set_state(S[:4])
    q = d.q_boards(env, 96).cpu().numpy()
    gw, gb = d.get_params()
    want = np.stack([np.tanh(xo.ext_forward(sizes, gw, gb, xo.state_repr(xo.board_from(s)), bf16=True)[1][:96]) for s in S[:4]])
    assert np.abs(q - want).max() < BF16_QTOL
    env.close(); d.close()


# ------------------------------------------------------------------------------------------------ prioritized replay
def _filled_ring(xq, trace, cap, n, seed, per=(0.6, 0.4, 1e-3)):
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, n, seed=seed))
    rp = xq.ReplayBuffer(cap, seed=0xFEED + seed)
    rp.enable_per(*per)
    rp.push(S, A, R / 1000.0, D, S2)
    return rp, (S, A, (R / 1000.0).astype(np.float32), D, S2)


def test_sum_tree_and_sampler_are_bit_exact(xq, trace):
    cap, n = 5000, 700                                  # capacity that is no power of 32: padded levels 5024 / 160 / 32 / 32
    rp, _ = _filled_ring(xq, trace, cap, n, seed=1)
    assert np.array_equal(rp.get_priorities(0, n), np.ones(n, np.float32))          # new transitions: the maximum so far (1 at start)
    assert not rp.get_priorities(n, cap - n).any()
    rng = np.random.default_rng(0)
    prio = rng.uniform(0.0, 3.0, size=cap).astype(np.float32)
    prio[rng.choice(cap, 900, replace=False)] = 0.0
    prio[4000:] = 0.0
    rp.set_priorities(prio)
    rp.per_rebuild()
    tree = xo.per_build(prio)
    st = rp.per_stats()
    assert st["total"] == float(xo.lib().xqo_per_total(tree.ctypes.data_as(C.POINTER(C.c_float)), cap))     # same bits
    assert st["n_eligible"] == int((prio > 0).sum()) and st["max_priority"] == float(prio.max())
    for call, batch in enumerate((512, 64, 1000)):
        slots, w = rp.sample_prioritized(batch)
        want_slots, want_w, wmax = xo.per_sample(tree, cap, batch, 0xFEED + 1, call, st["n_eligible"], 0.4)
        assert np.array_equal(slots, want_slots)                    # every descent takes the same branch
        assert (prio[slots] > 0).all()
        assert np.allclose(w, want_w / wmax, rtol=2e-6, atol=0) and abs(w.max() - 1.0) < 1e-6
    # retiring a window: zero priorities, never sampled, eligible count drops
    rp.per_rebuild(3990, 25)
    p2 = rp.get_priorities()
    want = prio.copy(); want[3990:4015] = 0.0
    assert np.array_equal(p2, want)
    slots, _ = rp.sample_prioritized(2000)
    assert not np.isin(slots, np.arange(3990, 4015)).any() and rp.per_stats()["n_eligible"] == int((want > 0).sum())
    assert np.array_equal(slots, xo.per_sample(xo.per_build(want), cap, 2000, 0xFEED + 1, 3, int((want > 0).sum()), 0.4)[0])
    rp.close()


def test_prioritized_td_step_weights_and_priorities(xq, trace):
    """sample -> TD step with importance weights -> priorities of the sampled slots written back, against the oracle."""
    sizes, cap, n, batch = CFG2_NET, 256, 200, 96
    rp, (S, A, R, D, S2) = _filled_ring(xq, trace, cap, n, seed=2, per=(0.6, 0.4, 1e-3))
    prio = np.random.default_rng(3).uniform(0.05, 2.0, size=n).astype(np.float32)
    rp.set_priorities(prio)
    rp.per_rebuild()
    d, w, b = make_net(xq, sizes, seed=13)
    wt, bt = xo.init_weights(sizes, 14)
    d.set_params(wt, bt, net=1)
    slots, wts = rp.sample_prioritized(batch)
    d.td_grads_replay(rp, batch, td_net=xq._capi.TD_DOUBLE, mode=0)
    lr = 0.05
    d.apply_grads(lr, 1.0 / batch)
    want_w, want_b, want_q, want_y, _ = oracle_update(sizes, w, b, wt, bt, S[slots], A[slots], R[slots], D[slots], S2[slots], 0.99, lr,
                                                      1.0 / batch, 0, 2, weights=wts)
    got_w, got_b = d.get_params()
    assert np.abs(got_w - want_w).max() < PTOL and np.abs(got_b - want_b).max() < PTOL
    assert np.abs(want_w - w).max() > 1e-5
    # new priorities: (|Q(s,a) - y| + eps)^alpha for the sampled slots, untouched elsewhere; running maximum follows
    p2 = rp.get_priorities(0, n)
    want_p = prio.copy()
    for s, q, y in zip(slots, want_q, want_y):
        want_p[s] = (abs(q - y) + 1e-3) ** 0.6
    assert np.allclose(p2, want_p, rtol=2e-3, atol=1e-5)
    untouched = np.setdiff1d(np.arange(n), slots)
    assert np.array_equal(p2[untouched], prio[untouched])
    rp.per_rebuild()
    assert abs(rp.per_stats()["max_priority"] - max(prio.max(), want_p.max())) < 1e-3
    d.close(); rp.close()


def test_new_transitions_enter_with_the_maximum_priority(xq):
    n, cap = 64, 256
    env = xq.VecEnv(n, seed=5)
    rp = xq.ReplayBuffer(cap, seed=5)
    rp.enable_per(0.6, 0.4, 1e-3)
    env.selfplay_step_dev(replay=rp)
    import torch; torch.cuda.synchronize()
    assert np.array_equal(rp.get_priorities(0, n), np.ones(n, np.float32)) and not rp.get_priorities(n, cap - n).any()
    rp.set_priorities(np.full(3, 7.5, np.float32), first=10)          # a TD step found a larger error
    env.selfplay_step_dev(replay=rp)
    torch.cuda.synchronize()
    assert np.array_equal(rp.get_priorities(n, n), np.ones(n, np.float32))       # still the snapshot of the last rebuild
    rp.per_rebuild()
    env.selfplay_step_dev(replay=rp)
    torch.cuda.synchronize()
    assert np.array_equal(rp.get_priorities(2 * n, n), np.full(n, 7.5, np.float32))
    env.close(); rp.close()


# ------------------------------------------------------------------------------------------------ the whole configs[4] loop
def _cfg5(xq, n, cap, minibatch, sizes, overlap, plies=1, seed=31, precision=1):
    return xq.TrainerConfig(n_games=n, layer_sizes=sizes, learning_rate=0.01, gamma=0.99, epsilon=0.2, replay_capacity=cap,
                            minibatch=minibatch, td_net=xq._capi.TD_DOUBLE, backprop_mode=0, target_sync_interval=3,
                            mean_gradient=1, seed=seed, first_game_id=11, collects_per_update=plies, overlap_collect=overlap,
                            prioritized=1, per_alpha=0.6, per_beta=0.4, per_eps=1e-3, precision=precision)


@pytest.mark.parametrize("overlap,n,cap,minibatch,plies,iters", [(0, 64, 256, 48, 1, 9), (1, 64, 256, 48, 1, 9), (1, 32, 200, 64, 2, 8)])
def test_config5_trainer_equals_its_composition(xq, overlap, n, cap, minibatch, plies, iters):
    """Double DQN + prioritized replay + bf16 through xq_trainer == the same iteration composed from synchronous C-ABI calls,
    bit for bit: [first iteration: collect, rebuild] sample from the tree -> TD grads -> collect -> apply -> retire the next
    collect's slots + rebuild.  (Non-overlapped trainers call collect first; the minibatch is the same either way because
    the tree is only rebuilt at learn_apply.)"""
    import torch
    sizes = [1260, 64, 64, 8100]
    cfg = _cfg5(xq, n, cap, minibatch, sizes, overlap, plies)
    t = xq.Trainer(cfg)
    w0, b0 = t.dqn.get_params()
    t.step(iters)
    tw, tb = t.dqn.get_params()
    tboards, tmeta = t.env.get_state()
    tprio = t.replay.get_priorities()

    env = xq.VecEnv(n, seed=31, first_game_id=11)
    d = xq.DQN(sizes, 0.01, 0.99, seed=1)
    d.set_params(w0, b0); d.updateTargetNetwork()
    d.set_precision(xq._capi.PRECISION_BF16)
    rp = xq.ReplayBuffer(cap, seed=31 + 0x1234567 + 11)
    rp.enable_per(0.6, 0.4, 1e-3)
    m = n * plies

    def collect():
        for _ in range(plies):
            q = d.q_boards(env, 96)
            env.selfplay_step_dev(q.data_ptr(), 96, 0.2, replay=rp)
            torch.cuda.synchronize()

    for it in range(iters):
        if it == 0:
            collect()
            rp.per_rebuild()
        elif not overlap:
            collect()
        rp.sample_prioritized(minibatch)
        d.td_grads_replay(rp, minibatch, td_net=xq._capi.TD_DOUBLE, mode=0)
        if overlap and it > 0:
            collect()
        d.apply_grads(0.01, 1.0 / minibatch)
        if (it + 1) % 3 == 0:
            d.updateTargetNetwork()
        size, _, total = rp.stats()
        rp.per_rebuild(total % cap, min(m, cap) if size + m > cap else 0)
    w, b = d.get_params()
    boards, meta = env.get_state()
    assert np.array_equal(boards, tboards) and np.array_equal(meta, tmeta)
    assert np.array_equal(w, tw) and np.array_equal(b, tb)
    assert np.array_equal(rp.get_priorities(), tprio)
    assert np.abs(w - w0).max() > 0
    t.close(); env.close(); d.close(); rp.close()


@pytest.mark.parametrize("precision", [1, 2])       # XQ_PRECISION_BF16 (fp32 backward products) / XQ_PRECISION_BF16_FULL (what bench.py --config 5 runs)
def test_config5_full_size_properties(xq, precision):
    """BASELINE configs[4] per GPU: 16384 games (131072 / 8), (512,512,512) bf16 Q-net, Double DQN, prioritized replay from a 1 M
    ring, minibatch 16384.  Size-independent properties: counters, finite parameters, priorities of sampled slots positive, the
    tree total equals the sequential sums of the priority table, and a bit-identical rerun."""
    def run():
        cfg = _cfg5(xq, 16384, 1 << 20, 16384, CFG4_NET, 1, seed=0x5EED, precision=precision)
        t = xq.Trainer(cfg)
        t.random_plies(60)
        t.step(4)
        c = t.counters()
        w, b = t.dqn.get_params()
        prio = t.replay.get_priorities()
        st = t.replay.per_stats()
        boards, meta = t.env.get_state()
        loss = t.dqn.last_loss()
        t.close()
        return c, w, b, prio, st, boards, meta, loss
    c, w, b, prio, st, boards, meta, loss = run()
    assert c["env_steps"] == 4 * 16384 and c["updates"] == 4
    assert np.isfinite(w).all() and np.isfinite(b).all() and np.isfinite(loss) and loss > 0
    assert (prio >= 0).all() and (prio[:4 * 16384] > 0).mean() > 0.99 and not prio[4 * 16384:].any()
    tree = xo.per_build(prio)
    assert st["total"] == float(xo.lib().xqo_per_total(tree.ctypes.data_as(C.POINTER(C.c_float)), 1 << 20))
    assert st["n_eligible"] == int((prio > 0).sum()) and st["max_priority"] >= prio.max()
    c2, w2, b2, prio2, st2, boards2, meta2, loss2 = run()
    assert c2 == c and np.array_equal(w, w2) and np.array_equal(b, b2) and np.array_equal(prio, prio2) and st == st2
    assert np.array_equal(boards, boards2) and np.array_equal(meta, meta2) and loss == loss2


def test_bf16_dense_backpropagate_refreshes_the_shadow(xq, trace):
    """DQN::backpropagate (dense path, dqn.cu:323-467) updates the fp32 master weights; every later bf16 forward
    must read the UPDATED bf16 shadow — the same numbers a fresh set_params of the new weights gives — and so must a target sync."""
    sizes = [1260, 128, 8100]
    d, w, b = make_net(xq, sizes, seed=5)
    d.set_precision(xq._capi.PRECISION_BF16)
    n = 32
    boards = trace["board"][np.random.default_rng(3).choice(len(trace["board"]), n, replace=False)]
    env = xq.VecEnv(n)
    env.set_state(boards)
    q_before = d.q_boards(env, 8100).cpu().numpy()
    x = xo.state_repr(xo.board_from(boards[0]))
    target = np.tanh(np.random.default_rng(4).uniform(-1, 1, size=8100))
    d.backpropagate(x, target, learning_rate=0.5)
    q_after = d.q_boards(env, 8100).cpu().numpy()
    assert np.abs(q_after - q_before).max() > 1e-3              # the update is visible to the bf16 forward at all
    w1, b1 = d.get_params()
    d.updateTargetNetwork()
    q_target = d.q_boards(env, 8100, net=1).cpu().numpy()
    d2 = xq.DQN(sizes, 0.001, 0.99, seed=1)
    d2.set_params(w1, b1)
    d2.set_precision(xq._capi.PRECISION_BF16)
    q_fresh = d2.q_boards(env, 8100).cpu().numpy()
    assert np.array_equal(q_after, q_fresh)
    assert np.array_equal(q_target, q_fresh)
    env.close(); d.close(); d2.close()


@pytest.mark.parametrize("td_rule,mode", [(2, 0), (0, 1)])
def test_bf16_full_td_update_matches_oracle(xq, trace, td_rule, mode):
    """XQ_PRECISION_BF16_FULL at a batch the bf16 GEMM loop of its own takes (n = 256 = one 256-row tile; widths 512): forward chains,
    hidden deltas and hidden weight gradients on gemm_bf16_kernel (xq_gemm_bf16.hip.h), against the oracle's bf16 = 2 definition."""
    sizes = CFG4_NET
    n = 256
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, n, seed=6))
    d, w, b = make_net(xq, sizes, seed=9)
    wt, bt = xo.init_weights(sizes, 78)
    d.set_params(wt, bt, net=1)
    d.set_precision(xq._capi.PRECISION_BF16_FULL)
    R = R / 1000.0
    lr, scale = 0.5, 1.0 / n
    want_w, want_b, want_q, want_y, _ = oracle_update(sizes, w, b, wt, bt, S, A, R, D, S2, 0.99, lr, scale, mode, td_rule, bf16=2)
    qsa, y = d.td_update(S, S2, A, R, D, td_net=td_rule, mode=mode, learning_rate=lr, grad_scale=scale)
    assert np.abs(qsa - want_q).max() < BF16_QTOL
    same_action = np.abs(y - want_y) < 2 * BF16_QTOL
    assert same_action.mean() > (0.7 if td_rule == 2 else 0.999)
    got_w, got_b = d.get_params()
    # per layer: the update agrees with the oracle's within 5 % of its largest entry (bf16 noise of the forward moves the deltas a
    # little; samples whose arg-max flipped move them more — their share is bounded above)
    off = 0
    for l in range(len(sizes) - 1):
        cnt = sizes[l] * sizes[l + 1]
        dw = (want_w - w)[off:off + cnt]
        err = np.abs((got_w - want_w)[off:off + cnt]).max()
        assert np.abs(dw).max() > 0
        tol = 0.05 if same_action.all() else 0.25
        assert err <= tol * np.abs(dw).max() + 1e-7, (l, err, np.abs(dw).max())
        off += cnt
    assert np.abs(got_b - want_b).max() <= (0.05 if same_action.all() else 0.25) * np.abs(want_b - b).max() + 1e-7
    # and the plain bf16 mode (fp32 backward) gives a DIFFERENT update on the same data: the mode is really in effect
    d2, _, _ = make_net(xq, sizes, seed=9)
    d2.set_params(wt, bt, net=1)
    d2.set_precision(xq._capi.PRECISION_BF16)
    d2.td_update(S, S2, A, R, D, td_net=td_rule, mode=mode, learning_rate=lr, grad_scale=scale)
    w2, _ = d2.get_params()
    h1 = slice(sizes[0] * sizes[1], sizes[0] * sizes[1] + sizes[1] * sizes[2])
    assert np.abs(w2[h1] - got_w[h1]).max() > 0
    d.close(); d2.close()


@pytest.mark.parametrize("sizes,td_rule", [(CFG4_NET, 2), (CFG4_NET, 0), (CFG2_NET, 2)])
def test_bf16_next_state_chain_derived_and_direct(xq, trace, sizes, td_rule):
    """xq_dqn_set_l0_derive on a bf16 net (what bench.py --config 5 runs): layer 0 of the online net's s' chain comes from the s chain's
    sums (rows out / rows in, fp32) inside the same wave; Double DQN's third chain (target net) is still gathered.  Same targets as the
    direct gather up to bf16 noise of a summation-order change, on real transitions and on unrelated board pairs (full re-gather)."""
    n = 64
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, n, seed=7))
    R = R / 1000.0
    d, w, b = make_net(xq, sizes, seed=9)
    wt, bt = xo.init_weights(sizes, 78)
    d.set_params(wt, bt, net=1)
    d.set_precision(xq._capi.PRECISION_BF16)
    for nxt in (S2, np.roll(S2, 5, axis=0)):
        d.set_l0_derive(False)
        q0, y0 = d.td_update(S, nxt, A, R, D, td_net=td_rule, mode=0, learning_rate=0.0, grad_scale=1.0)
        d.set_l0_derive(True)
        q1, y1 = d.td_update(S, nxt, A, R, D, td_net=td_rule, mode=0, learning_rate=0.0, grad_scale=1.0)
        assert np.array_equal(q0, q1)                               # the s chain itself is untouched
        close = np.abs(y0 - y1) < 2 * BF16_QTOL
        assert close.mean() > (0.8 if td_rule == 2 else 0.999)     # (an arg-max may flip between near-equal outputs)
    d.close()

"""GPU tests of the whole loop (collect -> learn) — `pytest -m gpu`.

(1) the loop composed from C-ABI calls vs the CPU oracle's restatement of chessai.cpp:96-143, same Q inputs;
(2) xq_trainer_step == that composition, bit for bit (the trainer adds no arithmetic of its own);
(3) BASELINE config 2 shape (8192 games, (256,256) net) runs and stays finite;
(4) overlap_collect (collect on its own stream beside learn_grads) == its sequential definition, bit for bit — a race between
    the two streams would show up as a difference.
"""
import ctypes as C
import os

import numpy as np
import pytest

import xqoracle as xo
from test_dqn_gpu import oracle_td_update, REF_NET, CFG2_NET

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def xq():
    import cn_chess_ai_amd as m
    assert m._capi.device_count() > 0
    return m


def test_loop_matches_oracle(xq):
    """16 games x 8 plies, reference net, on-policy batch update with upstream hyper-parameters (lr 1e-3, gamma .99)."""
    n, iters, seed, first, eps, lr = 16, 8, 77, 5, 0.1, 0.001
    sizes = REF_NET
    env = xq.VecEnv(n, seed=seed, first_game_id=first)
    w, b = xo.init_weights(sizes, 21)
    d = xq.DQN(sizes, lr, 0.99, seed=1)
    d.set_params(w, b)
    boards = [xo.new_board() for _ in range(n)]
    plies = np.zeros(n, dtype=np.int64)
    for it in range(iters):
        q = d.q_boards(env, 96).cpu().numpy()[:, :90]
        # the oracle's own Q agrees with the device Q to far better than the tolerance
        want_q = np.stack([xo.nn_forward(sizes, w, b, xo.state_repr(bd))[:90] for bd in boards])
        assert np.abs(q - want_q).max() < 1e-4
        S, _ = env.get_state()
        res = env.selfplay_step(q, eps)
        S2, _ = env.get_state()
        for g in range(n):
            o = xo.selfplay_step(boards[g], q[g], seed, first + g, int(plies[g]), xo.eps_to_u32(eps))
            assert (res[g]["action"], res[g]["reward"], res[g]["done"], res[g]["terminated"]) == \
                (o.action_code, o.reward, o.done, o.terminated)
            plies[g] += 1
        A = (res["action"] % 90).astype(np.int32)
        R = res["reward"].astype(np.float32)
        D = res["done"]
        w, b, _, _ = oracle_td_update(sizes, w, b, w, b, S, A, R, D, S2, 0.99, lr, 1.0 / n, 0)
        d.td_update(S, S2, A, R, D, td_net=0, mode=0, learning_rate=lr, grad_scale=1.0 / n)
        gw, gb = d.get_params()
        assert np.abs(gw - w).max() < 1e-4 and np.abs(gb - b).max() < 1e-4, it
    env.close(); d.close()


@pytest.mark.parametrize("replay_capacity,minibatch", [(0, 0), (256, 48)])
def test_trainer_equals_composition(xq, replay_capacity, minibatch):
    n, iters, seed, first = 64, 7, 4242, 100
    sizes = CFG2_NET
    cfg = xq.TrainerConfig(n_games=n, layer_sizes=sizes, learning_rate=0.01, gamma=0.99, epsilon=0.2,
                           replay_capacity=replay_capacity, minibatch=max(minibatch, 1), td_net=1, backprop_mode=0,
                           target_sync_interval=3, mean_gradient=1, seed=seed, first_game_id=first)
    t = xq.Trainer(cfg)
    w0, b0 = t.dqn.get_params()
    t.step(iters)
    tw, tb = t.dqn.get_params()
    ttw, ttb = t.dqn.get_params(net=1)
    tboards, tmeta = t.env.get_state()
    assert t.counters()["env_steps"] == n * iters and t.counters()["updates"] == iters

    env = xq.VecEnv(n, seed=seed, first_game_id=first)
    d = xq.DQN(sizes, 0.01, 0.99, seed=1)
    d.set_params(w0, b0); d.updateTargetNetwork()
    cap = replay_capacity if replay_capacity else n
    rp = xq.ReplayBuffer(cap, seed=seed + 0x1234567 + first)
    import torch
    for it in range(iters):
        q = d.q_boards(env, 96)
        if replay_capacity == 0:
            rp.close(); rp = xq.ReplayBuffer(cap, seed=0)          # on-policy: ring refilled from slot 0 every ply
        env.selfplay_step_dev(q.data_ptr(), 96, 0.2, replay=rp)
        torch.cuda.synchronize()
        if replay_capacity:
            rp.sample(minibatch)
            d.td_grads_replay(rp, minibatch, td_net=1, mode=0)
            d.apply_grads(0.01, 1.0 / minibatch)
        else:
            d.td_grads_replay(rp, 0, td_net=1, mode=0)
            d.apply_grads(0.01, 1.0 / n)
        if (it + 1) % 3 == 0:
            d.updateTargetNetwork()
    w, b = d.get_params()
    wt, bt = d.get_params(net=1)
    boards, meta = env.get_state()
    assert np.array_equal(boards, tboards) and np.array_equal(meta, tmeta)
    assert np.array_equal(w, tw) and np.array_equal(b, tb)
    assert np.array_equal(wt, ttw) and np.array_equal(bt, ttb)
    assert np.abs(w - w0).max() > 0
    t.close(); env.close(); d.close(); rp.close()


def test_trainer_baseline_config_shape(xq):
    cfg = xq.TrainerConfig(n_games=8192, layer_sizes=CFG2_NET, replay_capacity=1 << 16, minibatch=8192,
                           target_sync_interval=2)
    t = xq.Trainer(cfg)
    t.dqn.kernel_stats(enable=2)
    t.step(4)
    c = t.counters()
    assert c["env_steps"] == 4 * 8192 and c["updates"] == 4
    w, b = t.dqn.get_params()
    assert np.isfinite(w).all() and np.isfinite(b).all()
    stats = {s["name"]: s for s in t.dqn.kernel_stats(enable=0)}
    assert stats["gemm_qmax_rowmax"]["launches"] == 4 and stats["env_selfplay_step"]["launches"] == 4
    assert stats["gemm_qmax_rowmax"]["ms"] > 0
    assert np.isfinite(t.dqn.last_loss())
    size, cap, tot = t.replay.stats()
    assert size == 4 * 8192 and tot == 4 * 8192
    t.close()


def test_collects_per_update(xq):
    """BASELINE configs[3] cadence: 4 plies per update."""
    cfg = xq.TrainerConfig(n_games=128, layer_sizes=REF_NET, replay_capacity=2048, minibatch=256, collects_per_update=4)
    t = xq.Trainer(cfg)
    t.step(3)
    c = t.counters()
    assert c["env_steps"] == 128 * 12 and c["updates"] == 3 and t.replay.stats()[0] == 128 * 12
    bad = xq.Trainer(xq.TrainerConfig(n_games=16, layer_sizes=REF_NET, replay_capacity=0, minibatch=16, collects_per_update=2))
    with pytest.raises(xq.XqError):
        bad.step(1)
    t.close(); bad.close()


def overlap_window(size, total, cap, m):
    """Eligible ring slots while a collect of m transitions is in flight: (start, count) from the pre-collect state."""
    w = total % cap
    if size + m < cap:
        return 0, size
    return (w + m) % cap, cap - m


CFG4_NET = (1260, 512, 512, 512, 8100)          # BASELINE configs[3]: (512,512,512) hidden layers, 4 plies per update


@pytest.mark.parametrize("n,cap,minibatch,plies,iters,sizes", [
    (64, 256, 48, 1, 9, CFG2_NET), (32, 200, 64, 2, 8, CFG2_NET), (2048, 1 << 15, 2048, 1, 6, CFG2_NET),
    (256, 4096, 512, 4, 5, CFG4_NET)])           # configs[3] topology and cadence at a size the composition finishes quickly
def test_overlapped_trainer_equals_its_sequential_definition(xq, n, cap, minibatch, plies, iters, sizes):
    """Iteration t: minibatch drawn from the ring minus the slots collect(t) writes, gradients with theta_t; collect(t) acts with
    theta_t; then apply.  First iteration (nothing older in the ring): collect, then learn on what was just played."""
    import torch
    seed, first = 99, 7
    cfg = xq.TrainerConfig(n_games=n, layer_sizes=sizes, learning_rate=0.01, gamma=0.99, epsilon=0.2, replay_capacity=cap,
                           minibatch=minibatch, td_net=1, backprop_mode=0, target_sync_interval=3, mean_gradient=1, seed=seed,
                           first_game_id=first, collects_per_update=plies, overlap_collect=1)
    t = xq.Trainer(cfg)
    w0, b0 = t.dqn.get_params()
    t.step(iters)
    tw, tb = t.dqn.get_params()
    tboards, tmeta = t.env.get_state()
    c = t.counters()
    assert c["env_steps"] == n * plies * iters and c["updates"] == iters

    env = xq.VecEnv(n, seed=seed, first_game_id=first)
    d = xq.DQN(sizes, 0.01, 0.99, seed=1)
    d.set_params(w0, b0); d.updateTargetNetwork()
    rp = xq.ReplayBuffer(cap, seed=seed + 0x1234567 + first)
    m = n * plies

    def collect():
        for _ in range(plies):
            q = d.q_boards(env, 96)
            env.selfplay_step_dev(q.data_ptr(), 96, 0.2, replay=rp)
            torch.cuda.synchronize()

    for it in range(iters):
        size, _, total = rp.stats()
        start, count = overlap_window(size, total, cap, m)
        if count <= 0:
            collect()
            rp.sample(minibatch)
            d.td_grads_replay(rp, minibatch, td_net=1, mode=0)
        else:
            slots = rp.sample_window(minibatch, start, count)
            inflight = {(total + i) % cap for i in range(m)}
            assert not inflight.intersection(slots.tolist())
            d.td_grads_replay(rp, minibatch, td_net=1, mode=0)
            collect()
        d.apply_grads(0.01, 1.0 / minibatch)
        if (it + 1) % 3 == 0:
            d.updateTargetNetwork()
    w, b = d.get_params()
    boards, meta = env.get_state()
    assert np.array_equal(boards, tboards) and np.array_equal(meta, tmeta)
    assert np.array_equal(w, tw) and np.array_equal(b, tb)
    assert np.abs(w - w0).max() > 0
    t.close(); env.close(); d.close(); rp.close()


def test_baseline_config4_full_size(xq):
    """BASELINE configs[3] per GPU at FULL size: 8192 games, (512,512,512), 4 plies per update, collect overlapped with the TD
    step, minibatch 8192 from a 256 k ring.  Size-independent properties: counters, finite parameters and loss, every
    sampled transition well-formed, and a bit-identical rerun (no race between the three streams, no float atomics)."""
    def run():
        cfg = xq.TrainerConfig(n_games=8192, layer_sizes=CFG4_NET, replay_capacity=1 << 18, minibatch=8192, collects_per_update=4,
                               target_sync_interval=2, td_net=0, overlap_collect=1, seed=0x5EED, first_game_id=8192)
        t = xq.Trainer(cfg)
        w0, _ = t.dqn.get_params()
        t.random_plies(40)
        t.step(5)
        c = t.counters()
        w, b = t.dqn.get_params()
        wt, _ = t.dqn.get_params(net=1)
        boards, meta = t.env.get_state()
        loss = t.dqn.last_loss()
        size, cap, tot = t.replay.stats()
        acts = [t.replay.get(s)[1] for s in range(0, size, 4099)]
        t.close()
        return c, w, b, wt, boards, meta, loss, (size, cap, tot), (acts, w0)
    c, w, b, wt, boards, meta, loss, rp, (acts, w0) = run()
    assert c["env_steps"] == 8192 * 4 * 5 and c["updates"] == 5
    assert rp == (8192 * 20, 1 << 18, 8192 * 20)
    assert np.isfinite(w).all() and np.isfinite(b).all() and np.isfinite(loss) and loss > 0
    assert not np.array_equal(wt, w0) and not np.array_equal(wt, w)      # target = the weights after update 4 (sync every 2)
    assert all(-1 <= a < 90 for a in acts)
    assert meta[:, 0].max() <= 200 and (boards <= 14).all()
    c2, w2, b2, wt2, boards2, meta2, loss2, rp2, _ = run()
    assert c2 == c and np.array_equal(w, w2) and np.array_equal(b, b2) and np.array_equal(wt, wt2)
    assert np.array_equal(boards, boards2) and np.array_equal(meta, meta2) and loss == loss2


def test_overlap_needs_a_replay_ring(xq):
    with pytest.raises(xq.XqError):
        xq.Trainer(xq.TrainerConfig(n_games=16, layer_sizes=REF_NET, replay_capacity=0, minibatch=16, overlap_collect=1))


def test_sample_window_matches_the_oracle_philox(xq):
    rp = xq.ReplayBuffer(50, seed=0xABCDEF0123)
    b = np.tile(xq.START_BOARD, (40, 1))
    rp.push(b, np.arange(40) % 90, np.zeros(40), np.zeros(40), b)
    for call, (start, count) in enumerate([(0, 40), (35, 5), (10, 17)]):
        got = rp.sample_window(64, start, count)
        want = [(start + xo.philox((i, 0, call, 1), (0xCDEF0123, 0xAB))[0] % count) % 50 for i in range(64)]
        assert got.tolist() == want
    with pytest.raises(xq.XqError):
        rp.sample_window(8, 0, 41)
    rp.close()


def test_overlapped_trainer_mixed_call_order(xq):
    """collect, learn_grads, collect, learn_apply: the second collect overwrites ring slots the queued minibatch may contain
    (the ring is small and full), so the trainer has to hold it back until learn_grads has read them.  Compared bit for bit
    with the same sequence composed from synchronous C-ABI calls."""
    import torch
    n, cap, minibatch, iters, seed, first, sizes = 64, 192, 96, 8, 5, 3, CFG2_NET
    cfg = xq.TrainerConfig(n_games=n, layer_sizes=sizes, learning_rate=0.01, gamma=0.99, epsilon=0.2, replay_capacity=cap,
                           minibatch=minibatch, td_net=1, backprop_mode=0, target_sync_interval=0, mean_gradient=1, seed=seed,
                           first_game_id=first, overlap_collect=1)
    t = xq.Trainer(cfg)
    w0, b0 = t.dqn.get_params()
    for _ in range(iters):
        t.collect(); t.learn_grads(); t.collect(); t.learn_apply(1)
    tw, tb = t.dqn.get_params()
    tboards, tmeta = t.env.get_state()

    env = xq.VecEnv(n, seed=seed, first_game_id=first)
    d = xq.DQN(sizes, 0.01, 0.99, seed=1)
    d.set_params(w0, b0); d.updateTargetNetwork()
    rp = xq.ReplayBuffer(cap, seed=seed + 0x1234567 + first)

    def collect():
        q = d.q_boards(env, 96)
        env.selfplay_step_dev(q.data_ptr(), 96, 0.2, replay=rp)
        torch.cuda.synchronize()

    for it in range(iters):
        size0, _, total0 = rp.stats()
        collect()
        start, count = overlap_window(size0, total0, cap, n)      # the ring minus the first collect's slots
        if count <= 0:
            rp.sample(minibatch)
        else:
            rp.sample_window(minibatch, start, count)
        d.td_grads_replay(rp, minibatch, td_net=1, mode=0)
        torch.cuda.synchronize()
        collect()
        d.apply_grads(0.01, 1.0 / minibatch)
    w, b = d.get_params()
    boards, meta = env.get_state()
    assert np.array_equal(boards, tboards) and np.array_equal(meta, tmeta)
    assert np.array_equal(w, tw) and np.array_equal(b, tb)
    t.close(); env.close(); d.close(); rp.close()


def test_screened_qmax_in_the_overlapped_trainer(xq):
    """xq_dqn_set_qmax_mode(XQ_QMAX_SCREENED) inside the real loop (BASELINE configs[1] at full size: 8192 games, three streams,
    replay sampling, target syncs): (1) a rerun is bit-identical — the screen has no float atomics and no order dependence;
    (2) the trained weights stay within float rounding of the trainer that runs the full fp32 product (the TD targets are the
    same fp32 maxima up to summation order); (3) the screen really prunes (a handful of candidate groups per sample)."""
    from cn_chess_ai_amd import _capi

    def run(mode, iters=6):
        cfg = xq.TrainerConfig(n_games=8192, layer_sizes=CFG2_NET, replay_capacity=1 << 16, minibatch=8192, target_sync_interval=3,
                               td_net=0, overlap_collect=1, seed=0x5EED, mean_gradient=1)
        t = xq.Trainer(cfg)
        t.dqn.set_qmax_mode(mode)
        t.random_plies(60)
        for _ in range(8):
            t.collect()
        t.step(iters)
        w, b = t.dqn.get_params()
        st = t.dqn.qmax_stats()
        c = t.counters()
        loss = t.dqn.last_loss()
        t.close()
        return w, b, st, c, loss
    w_s, b_s, st, c, loss_s = run(_capi.QMAX_SCREENED)
    w_s2, b_s2, st2, c2, loss_s2 = run(_capi.QMAX_SCREENED)
    assert np.array_equal(w_s, w_s2) and np.array_equal(b_s, b_s2) and st == st2 and c == c2 and loss_s == loss_s2
    assert st[0] == 6 and st[1] == 6 * 8192 and 1.0 <= st[2] / st[1] < 16.0
    w_f, b_f, st_f, c_f, loss_f = run(_capi.QMAX_FULL)
    assert st_f[0] == 0 and c_f == c
    assert np.abs(w_s - w_f).max() < 1e-7 and np.abs(b_s - b_f).max() < 1e-7 and abs(loss_s - loss_f) < 1e-4 * abs(loss_f)


@pytest.mark.parametrize("sizes,mode", [(CFG2_NET, 0), ((1260, 512, 512, 512, 8100), 0), ((1260, 128, 256, 8100), 1)])   # (the as-written
def test_select_head_riding_on_the_last_hidden_product(xq, sizes, mode):                                                 # backprop has no 128 -> 256)
    """From 2048 games on, the trainer's select chain (dqn_q90_boards) takes Q[0..95] out of the last hidden product itself (EPI_HEAD:
    one k-slab of the head per 64-column tile, q_head_finish_kernel adds them) — the stand-alone head (xq_dqn_forward_boards_dev,
    k-slabs of the same 64 columns) must give the same bits: the two loops then play the same moves and learn the same weights.
    The stand-alone head against the oracle on a few boards."""
    n, iters, seed = 2048, 3, 77
    cfg = xq.TrainerConfig(n_games=n, layer_sizes=sizes, learning_rate=0.01, gamma=0.99, epsilon=0.1, replay_capacity=4 * n,
                           minibatch=n, td_net=0, backprop_mode=mode, target_sync_interval=0, mean_gradient=1, seed=seed, first_game_id=0)
    t = xq.Trainer(cfg)
    w0, b0 = t.dqn.get_params()
    t.random_plies(20)
    tb0, tm0 = t.env.get_state()
    t.step(iters)
    tw, tb = t.dqn.get_params()
    tboards, tmeta = t.env.get_state()

    env = xq.VecEnv(n, seed=seed, first_game_id=0)
    for _ in range(20):
        env.selfplay_step_dev(0, 96, 0.1)
    b_, m_ = env.get_state()
    assert np.array_equal(b_, tb0) and np.array_equal(m_, tm0)
    d = xq.DQN(sizes, 0.01, 0.99, seed=1)
    d.set_params(w0, b0); d.updateTargetNetwork()
    rp = xq.ReplayBuffer(4 * n, seed=seed + 0x1234567)
    import torch
    for it in range(iters):
        q = d.q_boards(env, 96)
        if it == 0:
            qh = q.cpu().numpy()
            for i in range(0, n, 257):
                want = xo.nn_forward(sizes, w0, b0, xo.state_repr(xo.board_from(b_[i])))[:96]
                assert np.abs(qh[i] - want).max() < 2e-5
        env.selfplay_step_dev(q.data_ptr(), 96, 0.1, replay=rp)
        torch.cuda.synchronize()
        rp.sample(n)
        d.td_grads_replay(rp, n, td_net=0, mode=mode)
        d.apply_grads(0.01, 1.0 / n)
    w, b = d.get_params()
    boards, meta = env.get_state()
    assert np.array_equal(boards, tboards) and np.array_equal(meta, tmeta)
    assert np.array_equal(w, tw) and np.array_equal(b, tb)
    t.close(); env.close(); d.close(); rp.close()


def test_bench_composition_against_the_oracle_directly(xq):
    """the 8192-wide loop EXACTLY as bench.py configures it — 1 M-slot ring filled to capacity, collect overlapped on its
    own stream, exact screening of max_a' Q(s',a'), layer 0 of s' derived, TD target inside the refine blocks, fused tail launches, slab
    sums inside the SGD kernel — checked against the fp64 oracle itself, not through another HIP path: Q(s,a) and y of 64 transitions
    of the last minibatch, and the update of one output row + bias (every sample of that action, ~90 backward passes in fp64).
    The minibatch is re-derived on the host from the trainer's documented Philox stream and sampling window."""
    from cn_chess_ai_amd import _capi
    from test_dqn_gpu import QTOL, PTOL
    n, cap, lr, seed = 8192, 1 << 20, 0.001, 0x5EED
    cfg = xq.TrainerConfig(n_games=n, layer_sizes=CFG2_NET, learning_rate=lr, gamma=0.99, epsilon=0.1, replay_capacity=cap, minibatch=n,
                           td_net=_capi.TD_ONLINE_NET, backprop_mode=_capi.BACKPROP_REFERENCE, target_sync_interval=10, mean_gradient=1,
                           seed=seed, first_game_id=0, overlap_collect=1, collects_per_update=1)
    t = xq.Trainer(cfg)
    t.dqn.set_qmax_mode(_capi.QMAX_SCREENED)
    t.dqn.set_l0_derive(True)
    t.dqn.set_fused_apply(True)
    t.random_plies(300)
    for _ in range(cap // n):
        t.collect()                                   # ring filled to capacity, as bench.py does before it times anything
    for _ in range(2):
        t.learn_grads(); t.collect(); t.learn_apply(1)
    w2, b2 = t.dqn.get_params()                        # theta before the third update
    size, _, total = t.replay.stats()
    assert size == cap
    wpos = total % cap
    t.learn_grads(); t.collect(); t.learn_apply(1)
    w3, b3 = t.dqn.get_params()
    st = t.dqn.qmax_stats()
    assert st[0] == 3 and st[1] == 3 * n               # all three steps took the screened route
    qsa, y = t.dqn.last_td_values(n)
    # the third minibatch: sample call #2 of the ring's stream, drawn from the ring minus the n slots that iteration's collect writes
    rseed = seed + 0x1234567
    key = (rseed & 0xFFFFFFFF, rseed >> 32)
    start, count = (wpos + n) % cap, cap - n
    slots = [(start + xo.philox((i, 0, 2, 1), key)[0] % count) % cap for i in range(n)]
    assert not any(wpos <= s < wpos + n for s in slots)
    rng = np.random.default_rng(4)
    pick = rng.choice(n, 64, replace=False)
    trans = {}

    def transition(i):
        if i not in trans:
            trans[i] = t.replay.get(slots[i])          # (board, action.to, reward, done, next board): reference-scale rewards
        return trans[i]

    worst_q = worst_y = 0.0
    for i in pick:
        s, a, r, dn, s2 = transition(int(i))
        assert a >= 0
        x, x2 = xo.state_repr(xo.board_from(s)), xo.state_repr(xo.board_from(s2))
        tq = xo.td_target(CFG2_NET, w2, b2, x, x2, int(a), float(r), int(dn), 0.99)
        q = xo.nn_forward(CFG2_NET, w2, b2, x)[int(a)]
        worst_q = max(worst_q, abs(float(qsa[i]) - q))
        # y = r + gamma max Q(s'): |r| is in the thousands (chessai.cpp:311-345), so the budget on y is relative to its size in fp32
        worst_y = max(worst_y, abs(float(y[i]) - tq[int(a)]) / max(1.0, abs(tq[int(a)])))
    assert worst_q < QTOL and worst_y < 1e-6, (worst_q, worst_y)
    # one output row: every sample of the minibatch whose action.to is a* contributes delta * a_last to W_out[a*] and delta to b_out[a*]
    acts = np.array([transition(i)[1] for i in range(n)])
    counts = np.bincount(acts[acts >= 0], minlength=90)
    a_star = int(np.argsort(counts)[45])               # an action of middling popularity (~90 samples)
    gw, gb = np.zeros_like(w2), np.zeros_like(b2)
    for i in np.nonzero(acts == a_star)[0]:
        s, a, r, dn, s2 = transition(int(i))
        x, x2 = xo.state_repr(xo.board_from(s)), xo.state_repr(xo.board_from(s2))
        tq = xo.td_target(CFG2_NET, w2, b2, x, x2, int(a), float(r), int(dn), 0.99)
        xo.nn_accum_grad(CFG2_NET, w2, b2, x, tq, 0, gw, gb)
    wo = 1260 * 256 + 256 * 256                        # W_out starts here in the reference flat layout
    row = slice(wo + a_star * 256, wo + (a_star + 1) * 256)
    want_w = w2[row] - lr / n * gw[row]
    want_b = b2[512 + a_star] - lr / n * gb[512 + a_star]
    assert np.abs(gw[row]).max() > 0
    assert np.abs(w3[row] - want_w).max() < PTOL and abs(b3[512 + a_star] - want_b) < PTOL
    # ... and the step was not a no-op on that row
    assert np.abs(w3[row] - w2[row]).max() > 0
    t.close()


def test_timing_knobs_do_not_change_a_bit(xq):
    """INTEGRATION.md lists the environment variables the library reads as A/B knobs for TIMING (event flags, the fork as a stop event, the SGD
    kernel's load width, the screening pass's block -> XCD map, the weight-gradient product's launch): six updates of the bench's schedule at 2048
    games give the same weights, boards, Q(s,a) and targets, bit for bit, under each of them.  One child process per setting (read once per process)."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "knob_probe.py")
    got = {}
    for knob in ("", "XQ_EVENT_SYSFENCE=1", "XQ_FORK_STOP_EVENT=0", "XQ_SGD_SCALAR=1", "XQ_SCREEN_XCD=0", "XQ_SCREEN_XCD=1", "XQ_TAIL_GRAD_EARLY=1", "XQ_REFINE_WHOLE=1"):
        env = dict(os.environ)
        if knob:
            k, v = knob.split("=")
            env[k] = v
        out = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("KNOB_PROBE")]
        assert out.returncode == 0 and line, (knob, out.stdout[-400:], out.stderr[-800:])
        _, digest, screened = line[0].split()
        assert int(screened) == 6, (knob, screened)            # every update took the screened route
        got[knob] = digest
    assert len(set(got.values())) == 1, got

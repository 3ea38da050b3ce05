"""GPU parity tests of the Q-network path (through the C ABI) — `pytest -m gpu`.

HIP fp32/MFMA path vs the fp64 CPU oracle (oracle/xq_oracle.c restating reference dqn.cu / dqn.cpp / chessai.cpp) on
identical inputs.  Tolerance, from BASELINE.json north_star: Q-values within 1e-4 (fp32 vs the fp64 reference).
"""
import ctypes as C
import os

import numpy as np
import pytest

import xqoracle as xo

pytestmark = pytest.mark.gpu

QTOL = 1e-4          # north_star tolerance on Q-values
PTOL = 2e-5          # parameters after one update (|w| <= 0.05 + update; fp32 storage)

REF_NET = [1260, 128, 8100]
CFG2_NET = [1260, 256, 256, 8100]


@pytest.fixture(scope="module")
def xq():
    import cn_chess_ai_amd as m
    assert m._capi.device_count() > 0
    return m


@pytest.fixture(scope="module")
def trace(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_trace.npz"))


def one_hot(boards):
    boards = np.asarray(boards).reshape(-1, 90)
    x = np.zeros((len(boards), 1260))
    for i, b in enumerate(boards):
        for s in np.nonzero(b)[0]:
            x[i, s * 14 + b[s] - 1] = 1.0
    return x


def make_net(xq, sizes, seed=1, lr=0.001, gamma=0.99):
    w, b = xo.init_weights(sizes, seed)
    b = np.random.default_rng(seed + 100).uniform(-0.05, 0.05, size=len(b))   # non-zero biases exercise the bias path
    d = xq.DQN(sizes, lr, gamma, seed=seed)
    d.set_params(w, b)
    d.updateTargetNetwork()
    return d, w.copy(), b.copy()


def transitions(trace, idx):
    """(s, action.to, reward, done, s') from consecutive records of the reference trace (valid moves only)."""
    L = xo.lib()
    S, A, R, D, S2 = [], [], [], [], []
    for i in idx:
        b = xo.board_from(trace["board"][i], trace["moveCount"][i], trace["player"][i], trace["redScore"][i],
                          trace["blackScore"][i])
        mover = b.currentPlayer
        fr, fc, tr, tc = (int(x) for x in trace["move"][i])
        S.append(b.squares())
        L.xqo_move_piece(C.byref(b), fr, fc, tr, tc)
        A.append(tr * 9 + tc)
        R.append(float(L.xqo_evaluate_board(C.byref(b), mover, b.moveCount)))
        D.append(int(L.xqo_check_game_over(C.byref(b)) or b.moveCount + 1 >= 200))
        S2.append(b.squares())
    return (np.array(S, np.uint8), np.array(A, np.int32), np.array(R, np.float32), np.array(D, np.uint8),
            np.array(S2, np.uint8))


def valid_indices(trace, n, seed=0):
    ok = np.nonzero((trace["valid"] == 1) & (trace["over"] == 0))[0]
    rng = np.random.default_rng(seed)
    pick = rng.choice(ok, size=n, replace=False)
    # make sure terminal / near-cap transitions are present
    late = ok[trace["moveCount"][ok] >= 198][:4]
    caps = ok[np.isin(trace["captured"][ok], (1, 8))][:2]
    special = np.concatenate([late, caps])[:max(n // 4, 1)]
    pick[:len(special)] = special
    return pick


# ------------------------------------------------------------------------------------------------ forward
@pytest.mark.parametrize("sizes", [REF_NET, CFG2_NET, [1260, 64, 96, 200], [40, 24, 24, 56]])
def test_forward_matches_oracle(xq, trace, sizes):
    d, w, b = make_net(xq, sizes, seed=3)
    rng = np.random.default_rng(0)
    n = 37
    if sizes[0] == 1260:
        x = one_hot(trace["board"][rng.choice(len(trace["board"]), n, replace=False)])
        x[-5:] = rng.uniform(-1, 1, size=(5, 1260))            # dense (non one-hot) inputs: getQValues takes any vector
    else:
        x = rng.uniform(-1, 1, size=(n, sizes[0]))
    q = d.getQValues(x)
    want = np.stack([xo.nn_forward(sizes, w, b, xi) for xi in x])
    assert q.shape == want.shape
    assert np.abs(q - want).max() < QTOL
    assert np.abs(q - want).max() < 5e-6                          # in practice ~1e-7
    q1 = d.getQValues(x[0])                                      # batch-1 call, like upstream
    assert np.abs(q1 - want[0]).max() < 5e-6
    d.close()


def test_board_input_equals_dense_input(xq, trace):
    d, w, b = make_net(xq, CFG2_NET, seed=4)
    n = 200
    boards = trace["board"][:n]
    env = xq.VecEnv(n)
    env.set_state(boards)
    q96 = d.q_boards(env, 96).cpu().numpy()
    qfull = d.q_boards(env, 8100).cpu().numpy()
    dense = d.getQValues(one_hot(boards))
    assert np.abs(qfull - dense).max() < 2e-6
    assert np.abs(q96 - dense[:, :96]).max() < 2e-6
    want = np.stack([xo.nn_forward(CFG2_NET, w, b, xo.state_repr(xo.board_from(bd))) for bd in boards[:20]])
    assert np.abs(qfull[:20] - want).max() < 5e-6
    env.close(); d.close()


# ------------------------------------------------------------------------------------------------ backprop
@pytest.mark.parametrize("sizes,mode", [(REF_NET, 0), (REF_NET, 1), (CFG2_NET, 0), (CFG2_NET, 1),
                                        ([1260, 512, 512, 512, 8100], 0)])
def test_backpropagate_single_sample_matches_oracle(xq, trace, sizes, mode):
    """DQN::backpropagate(state, target, lr) with batch 1 — exactly the upstream call (dqn.cu:323-467)."""
    d, w, b = make_net(xq, sizes, seed=5)
    x = one_hot(trace["board"][100])[0]
    target = xo.nn_forward(sizes, w, b, x)
    target[33] = -0.35                                   # the TD-modified entry (chessai.cpp:124/127)
    target[7000] += 0.01                                 # and an arbitrary far entry: the API takes any target
    lr = 0.05
    assert xo.nn_backprop(sizes, w, b, x, target, lr, mode) == 0
    d.backpropagate(x, target, lr, 1.0, mode)
    gw, gb = d.get_params()
    assert np.abs(gw - w).max() < PTOL and np.abs(gb - b).max() < PTOL
    q = d.getQValues(x)
    assert np.abs(q - xo.nn_forward(sizes, w, b, x)).max() < QTOL
    d.close()


def test_backpropagate_modes_differ_and_batch_rule(xq, trace):
    """reference vs textbook deltas really differ; the minibatch rule = sum of per-sample gradients at fixed weights."""
    sizes = REF_NET
    n = 6
    xs = one_hot(trace["board"][200:200 + n])
    d0, w, b = make_net(xq, sizes, seed=6)
    d1, _, _ = make_net(xq, sizes, seed=6)
    targets = np.stack([xo.nn_forward(sizes, w, b, x) for x in xs])
    targets[np.arange(n), np.arange(n) * 7] = np.linspace(-0.9, 0.9, n)
    lr, scale = 0.02, 1.0 / n
    res = {}
    for mode, d in ((0, d0), (1, d1)):
        gw, gb = np.zeros_like(w), np.zeros_like(b)
        for x, t in zip(xs, targets):
            assert xo.nn_accum_grad(sizes, w, b, x, t, mode, gw, gb) == 0
        want_w, want_b = w - lr * scale * gw, b - lr * scale * gb
        d.backpropagate(xs, targets, lr, scale, mode)
        got_w, got_b = d.get_params()
        assert np.abs(got_w - want_w).max() < PTOL and np.abs(got_b - want_b).max() < PTOL
        res[mode] = got_w
    assert np.abs(res[0] - res[1]).max() > 1e-5          # the shifted-stride hidden delta is a different update
    d0.close(); d1.close()


def test_reference_mode_rejects_undefined_topology(xq):
    d = xq.DQN([1260, 32, 64, 128], seed=1)
    x, t = np.zeros(1260), np.zeros(128)
    with pytest.raises(xq.XqError) as e:
        d.backpropagate(x, t, 0.01, 1.0, 0)
    assert e.value.code == 5
    d.backpropagate(x, t, 0.01, 1.0, 1)                  # textbook mode is defined everywhere
    d.close()


# ------------------------------------------------------------------------------------------------ TD step
def oracle_td_update(sizes, w, b, wt, bt, S, A, R, D, S2, gamma, lr, scale, mode):
    gw, gb = np.zeros_like(w), np.zeros_like(b)
    qsa, ys = [], []
    for s, a, r, dn, s2 in zip(S, A, R, D, S2):
        x = xo.state_repr(xo.board_from(s))
        tq = xo.nn_forward(sizes, w, b, x)                                   # chessai.cpp:122
        qsa.append(tq[a])
        if dn:
            y = float(r)
        else:
            y = float(r) + gamma * xo.nn_forward(sizes, wt, bt, xo.state_repr(xo.board_from(s2))).max()   # :126-127
        tq[a] = y
        ys.append(y)
        assert xo.nn_accum_grad(sizes, w, b, x, tq, mode, gw, gb) == 0      # :131
    return w - lr * scale * gw, b - lr * scale * gb, np.array(qsa), np.array(ys)


@pytest.mark.parametrize("sizes,td_net,mode", [(REF_NET, 0, 0), (CFG2_NET, 0, 0), (CFG2_NET, 1, 0), (CFG2_NET, 1, 1)])
def test_td_update_matches_oracle(xq, trace, sizes, td_net, mode):
    n = 48
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, n, seed=1))
    assert D.sum() >= 2 and (D == 0).sum() > 10
    d, w, b = make_net(xq, sizes, seed=7)
    wt, bt = w, b
    if td_net == 1:                                       # make the target net different from the online net
        wt, bt = xo.init_weights(sizes, 99)
        d.set_params(wt, bt, net=1)
    R = R / 1000.0                                        # keep |target| O(1) so tanh' is not ~0 everywhere
    lr, scale = 0.05, 1.0 / n
    want_w, want_b, want_q, want_y = oracle_td_update(sizes, w, b, wt, bt, S, A, R, D, S2, 0.99, lr, scale, mode)
    qsa, y = d.td_update(S, S2, A, R, D, td_net=td_net, mode=mode, learning_rate=lr, grad_scale=scale)
    assert np.abs(qsa - want_q).max() < QTOL and np.abs(y - want_y).max() < QTOL
    got_w, got_b = d.get_params()
    assert np.abs(got_w - want_w).max() < PTOL and np.abs(got_b - want_b).max() < PTOL
    assert np.abs(got_w - w).max() > 1e-4                # something was learnt
    # untouched output rows (>= 96) stay bit-identical: action.to < 90
    hl = sizes[-2]
    wo_out = sum(sizes[i] * sizes[i + 1] for i in range(len(sizes) - 2))
    assert np.array_equal(got_w[wo_out + 96 * hl:], w.astype(np.float32).astype(np.float64)[wo_out + 96 * hl:])
    x = xo.state_repr(xo.board_from(S[0]))
    assert np.abs(d.getQValues(x) - xo.nn_forward(sizes, want_w, want_b, x)).max() < QTOL
    d.close()


def test_td_update_raw_rewards_and_sequence(xq, trace):
    """Upstream-scale rewards (+-thousands against tanh outputs) through 3 consecutive updates, reference net."""
    sizes = REF_NET
    d, w, b = make_net(xq, sizes, seed=8)
    lr = 0.001
    for k in range(3):
        S, A, R, D, S2 = transitions(trace, valid_indices(trace, 16, seed=10 + k))
        w, b, _, _ = oracle_td_update(sizes, w, b, w, b, S, A, R, D, S2, 0.99, lr, 1.0 / 16, 0)
        d.td_update(S, S2, A, R, D, td_net=0, mode=0, learning_rate=lr, grad_scale=1.0 / 16)
    got_w, got_b = d.get_params()
    assert np.abs(got_w - w).max() < 1e-4 and np.abs(got_b - b).max() < 1e-4
    xs = one_hot(trace["board"][300:310])
    want = np.stack([xo.nn_forward(sizes, w, b, x) for x in xs])
    assert np.abs(d.getQValues(xs) - want).max() < QTOL
    d.close()


def test_td_update_equals_dense_backpropagate(xq, trace):
    """The sparse TD path (delta in columns 0..95) must equal DQN::backpropagate on the dense target vector."""
    sizes = CFG2_NET
    n = 24
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, n, seed=2))
    R = R / 500.0
    da, w, b = make_net(xq, sizes, seed=9)
    db, _, _ = make_net(xq, sizes, seed=9)
    xs = one_hot(S)
    q = db.getQValues(xs)
    nq = db.getQValues(one_hot(S2))
    tq = q.copy()
    for i in range(n):
        tq[i, A[i]] = R[i] if D[i] else R[i] + 0.99 * nq[i].max()
    db.backpropagate(xs, tq, 0.03, 1.0 / n, 0)
    da.td_update(S, S2, A, R, D, td_net=0, mode=0, learning_rate=0.03, grad_scale=1.0 / n)
    wa, ba = da.get_params()
    wb, bb = db.get_params()
    assert np.abs(wa - wb).max() < 2e-6 and np.abs(ba - bb).max() < 2e-6
    da.close(); db.close()


# ------------------------------------------------------------------------------------------------ DQN facade
def test_select_action_matches_oracle(xq, trace):
    sizes = REF_NET
    d, w, b = make_net(xq, sizes, seed=11)
    L = xo.lib()
    for i in (0, 50, 400, 1234):
        bd = xo.board_from(trace["board"][i])
        codes, n = xo.all_valid_actions(bd, int(trace["player"][i]))
        if n == 0:
            continue
        x = xo.state_repr(bd)
        q = xo.nn_forward(sizes, w, b, x)
        acts = [(int(c) // 90, int(c) % 90) for c in codes]
        rm = 2147483647                                      # RAND_MAX: randValue = rand() / RAND_MAX (dqn.cpp:30)
        for k1, r2 in ((rm // 2, 0), (rm // 20, 12345), (214748364, 7), (214748365, 3), (rm, 1)):
            want = L.xqo_select_action(q.ctypes.data_as(C.POINTER(C.c_double)), len(q),
                                       codes.ctypes.data_as(C.POINTER(C.c_uint16)), n, k1, r2, rm, 0.1)
            assert d.selectAction(x, 0.1, acts, rand1=k1 / rm, rand2=r2) == acts[want]
    with pytest.raises(RuntimeError):
        d.selectAction(np.zeros(1260), 0.1, [])
    d.close()


def test_target_network_and_dqn_train(xq, trace):
    sizes = REF_NET
    d, w, b = make_net(xq, sizes, seed=12)
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, 4, seed=3))
    d.td_update(S, S2, A, R / 1000, D, learning_rate=0.05)
    w1, b1 = d.get_params(0)
    wt, bt = d.get_params(1)
    assert np.abs(wt - w).max() < 1e-8 and np.abs(w1 - wt).max() > 1e-5      # target still the old weights
    d.updateTargetNetwork()
    wt, bt = d.get_params(1)
    assert np.array_equal(wt, w1) and np.array_equal(bt, b1)                # now the TRAINED weights (not upstream's stale copy)
    # DQN::train (dqn.cpp:157-172): single transition, target net
    x, x2 = xo.state_repr(xo.board_from(S[0])), xo.state_repr(xo.board_from(S2[0]))
    q = xo.nn_forward(sizes, w1, b1, x)
    q[A[0]] = 0.25 + 0.99 * xo.nn_forward(sizes, wt, bt, x2).max()
    xo.nn_backprop(sizes, w1, b1, x, q, 0.001, 0)
    d.train(x, int(A[0]), 0.25, x2, False)
    w2, b2 = d.get_params(0)
    assert np.abs(w2 - w1).max() < PTOL
    d.close()


def test_model_file_format(xq, tmp_path):
    d, w, b = make_net(xq, REF_NET, seed=13)
    p = tmp_path / "model.bin"
    d.saveModel(p)
    raw = p.read_bytes()
    assert len(raw) == 9650484                                              # verified upstream size
    nw, nb = 1198080, 8228
    fw = np.frombuffer(raw[:nw * 8], dtype="<f8")
    fb = np.frombuffer(raw[nw * 8:(nw + nb) * 8], dtype="<f8")
    assert np.array_equal(fw, w.astype(np.float32).astype(np.float64)) and np.array_equal(fb, b.astype(np.float32).astype(np.float64))
    tail = raw[(nw + nb) * 8:]
    assert tail == (3).to_bytes(8, "big") + b"".join(int(s).to_bytes(4, "big") for s in REF_NET)
    d2 = xq.DQN(REF_NET, seed=77)
    d2.loadModel(p)
    w2, b2 = d2.get_params()
    assert np.array_equal(w2, fw) and np.array_equal(b2, fb)
    d3 = xq.DQN(CFG2_NET, seed=1)
    with pytest.raises(xq.XqError) as e:
        d3.loadModel(p)
    assert e.value.code == 4
    with pytest.raises(xq.XqError):
        d3.loadModel(tmp_path / "missing.bin")
    d.close(); d2.close(); d3.close()


def test_invalid_arguments(xq):
    with pytest.raises(xq.XqError) as e:
        xq.DQN([1260])
    assert e.value.code == 1
    d = xq.DQN([1260, 16, 32], seed=1)                  # < 96 outputs: dense API only
    with pytest.raises(xq.XqError):
        d.td_update(np.zeros((1, 90)), np.zeros((1, 90)), [0], [0.0], [0])
    d.close()


def test_td_update_config4_topology(xq, trace):
    """BASELINE configs[3] network (512,512,512): reference-compatible TD step vs oracle (layer-0 gradient kernel with two
    accumulator sets, 4 hidden-delta GEMMs with the as-written cross-layer view)."""
    sizes = [1260, 512, 512, 512, 8100]
    n = 12
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, n, seed=5))
    d, w, b = make_net(xq, sizes, seed=15)
    R = R / 1000.0
    lr, scale = 0.05, 1.0 / n
    want_w, want_b, want_q, want_y = oracle_td_update(sizes, w, b, w, b, S, A, R, D, S2, 0.99, lr, scale, 0)
    qsa, y = d.td_update(S, S2, A, R, D, td_net=0, mode=0, learning_rate=lr, grad_scale=scale)
    assert np.abs(qsa - want_q).max() < QTOL and np.abs(y - want_y).max() < QTOL
    got_w, got_b = d.get_params()
    assert np.abs(got_w - want_w).max() < PTOL and np.abs(got_b - want_b).max() < PTOL
    d.close()


def test_td_update_full_batch_properties(xq):
    """BASELINE-size minibatch (8192): size-independent properties — linearity of the update in grad_scale, determinism
    (bitwise equal reruns), rows >= 96 of the output layer untouched, empty slots (action -1) contribute nothing."""
    sizes = CFG2_NET
    n = 8192
    env = xq.VecEnv(n, seed=77)
    for _ in range(30):
        env.selfplay_step(None)
    S, _ = env.get_state()
    res = env.selfplay_step(None)
    S2, _ = env.get_state()
    A = (res["action"] % 90).astype(np.int32)
    R = (res["reward"] / 100.0).astype(np.float32)
    D = res["done"]
    runs = []
    for scale in (64.0 / n, 64.0 / n, 128.0 / n):
        d, w, b = make_net(xq, sizes, seed=21)
        d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=1.0, grad_scale=scale)
        runs.append(d.get_params())
        d.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])      # deterministic
    w32 = w.astype(np.float32).astype(np.float64)
    dw1, dw2 = runs[0][0] - w32, runs[2][0] - w32
    m = np.abs(dw1) > 2e-3
    assert m.sum() > 1000 and np.abs(dw2[m] / dw1[m] - 2.0).max() < 1e-2                          # linear in grad_scale
    wo_out = 1260 * 256 + 256 * 256
    assert np.array_equal(runs[0][0][wo_out + 96 * 256:], w32[wo_out + 96 * 256:])
    # a batch whose second half is "empty slots" equals the first half alone (same scale)
    d1, _, _ = make_net(xq, sizes, seed=21)
    d2, _, _ = make_net(xq, sizes, seed=21)
    A2 = A.copy(); A2[n // 2:] = -1
    d1.td_update(S, S2, A2, R, D, td_net=1, mode=0, learning_rate=1.0, grad_scale=64.0 / n)
    d2.td_update(S[:n // 2], S2[:n // 2], A[:n // 2], R[:n // 2], D[:n // 2], td_net=1, mode=0, learning_rate=1.0,
                 grad_scale=64.0 / n)
    w1, b1 = d1.get_params(); w2, b2 = d2.get_params()
    assert np.abs(w1 - w2).max() < 1e-5 and np.abs(b1 - b2).max() < 1e-5
    env.close(); d1.close(); d2.close()


def test_td_targets_on_the_persistent_gemm_match_oracle(xq):
    """The column-max GEMM's persistent form (>= 512 output tiles; here 64 x 9 for a ragged batch of 1100) against the oracle:
    y = r + gamma * max_k Q_target(s')[k] per sample, target net with large random biases on every layer (the bias vector is
    staged through LDS there, rows >= 8100 padded with -inf), Q(s, a) from the online net."""
    sizes = CFG2_NET
    n = 1100
    env = xq.VecEnv(n, seed=4321)
    for _ in range(25):
        env.selfplay_step(None)
    S, _ = env.get_state()
    res = env.selfplay_step(None)
    S2, _ = env.get_state()
    A = (res["action"] % 90).astype(np.int32)
    R = (res["reward"] / 100.0).astype(np.float32)
    D = res["done"].copy()
    D[::7] = 1                                           # plenty of terminal samples in the mix
    d, w, b = make_net(xq, sizes, seed=31)
    wt, _ = xo.init_weights(sizes, 77)
    wt = wt * 3.0                                         # spread the pre-activations so that the arg-max is not a bias artefact
    bt = np.random.default_rng(5).uniform(-0.4, 0.4, size=len(b))
    d.set_params(wt, bt, net=1)
    qsa, y = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    want_y = np.empty(n)
    want_q = np.empty(n)
    arg = set()
    for i in range(n):
        want_q[i] = xo.nn_forward(sizes, w, b, xo.state_repr(xo.board_from(S[i])))[A[i]]
        if D[i]:
            want_y[i] = float(R[i])
        else:
            q2 = xo.nn_forward(sizes, wt, bt, xo.state_repr(xo.board_from(S2[i])))
            arg.add(int(q2.argmax()))
            want_y[i] = float(R[i]) + 0.99 * q2.max()
    assert len(arg) > 20 and max(arg) >= 96              # the max really ranges over all 8100 outputs
    assert np.abs(y - want_y).max() < QTOL and np.abs(qsa - want_q).max() < QTOL
    env.close(); d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,scale", [(CFG2_NET, 3.0), (CFG2_NET, 0.2), ((1260, 128, 8100), 1.0), ((1260, 512, 512, 512, 8100), 1.0)])
def test_screened_qmax_equals_full_product(xq, sizes, scale):
    """xq_dqn_set_qmax_mode(XQ_QMAX_SCREENED): the TD targets are the maxima of fp32-evaluated outputs — equal to the full fp32
    column-max GEMM up to summation order (MFMA k-order vs one fp32 fma chain per candidate) and to the oracle within QTOL.
    Weight scales 0.2 .. 3 move the pre-activations from the linear range deep into tanh saturation; every layer has biases."""
    from cn_chess_ai_amd import _capi
    n = 1100
    env = xq.VecEnv(n, seed=99)
    for _ in range(31):
        env.selfplay_step(None)
    S, _ = env.get_state()
    res = env.selfplay_step(None)
    S2, _ = env.get_state()
    A = (res["action"] % 90).astype(np.int32)
    R = (res["reward"] / 100.0).astype(np.float32)
    D = res["done"].copy()
    D[::5] = 1
    d, w, b = make_net(xq, sizes, seed=8)
    wt, _ = xo.init_weights(sizes, 78)
    wt = wt * scale
    bt = np.random.default_rng(6).uniform(-0.3, 0.3, size=len(b))
    d.set_params(wt, bt, net=1)
    ys = {}
    for mode in (_capi.QMAX_FULL, _capi.QMAX_SCREENED):
        d.set_qmax_mode(mode)
        qsa, y = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
        ys[mode] = (qsa.copy(), y.copy())
    steps, samples, pairs, whole = d.qmax_stats()
    assert steps == 1 and samples == n
    assert pairs >= n                                         # every sample has at least its own maximum as a candidate
    assert pairs < 64 * n                                     # ... and the screen really prunes: far fewer than the 254 groups
    assert np.array_equal(ys[0][0], ys[1][0])                 # Q(s, a) does not depend on the mode
    assert np.abs(ys[0][1] - ys[1][1]).max() < 2e-6
    want_y = np.empty(n)
    for i in range(0, n, 3):
        if D[i]:
            want_y[i] = float(R[i])
        else:
            want_y[i] = float(R[i]) + 0.99 * xo.nn_forward(sizes, wt, bt, xo.state_repr(xo.board_from(S2[i]))).max()
    assert np.abs(ys[1][1][::3] - want_y[::3]).max() < QTOL
    # online-net rule as well (the screen reads the online weights then), and a second step reuses the buffers
    d.set_qmax_mode(_capi.QMAX_FULL)
    _, y_full = d.td_update(S, S2, A, R, D, td_net=0, mode=0, learning_rate=0.0, grad_scale=1.0)
    d.set_qmax_mode(_capi.QMAX_SCREENED)
    _, y_scr = d.td_update(S, S2, A, R, D, td_net=0, mode=0, learning_rate=0.0, grad_scale=1.0)
    assert np.abs(y_full - y_scr).max() < 2e-6
    env.close(); d.close()


@pytest.mark.gpu
def test_screened_qmax_degenerate_outputs(xq):
    """Worst case for the screen: every output row identical (all 8100 outputs tie) — every group is a candidate, most of them
    as whole groups; the result must still be the fp32 maximum.  Then a net whose maximum sits in the last, partly padded group."""
    from cn_chess_ai_amd import _capi
    sizes = CFG2_NET
    n = 1100
    env = xq.VecEnv(n, seed=5)
    for _ in range(9):
        env.selfplay_step(None)
    S, _ = env.get_state()
    res = env.selfplay_step(None)
    S2, _ = env.get_state()
    A = (res["action"] % 90).astype(np.int32)
    R = np.zeros(n, np.float32)
    D = np.zeros(n, np.uint8)
    d, w, b = make_net(xq, sizes, seed=8)
    wt, bt0 = xo.init_weights(sizes, 79)
    nw_out = 8100 * 256
    wt = wt.copy(); bt = np.zeros(len(b))
    wt[-nw_out:] = np.tile(wt[-nw_out:-nw_out + 256], 8100)          # all output rows equal
    d.set_params(wt, bt, net=1)
    d.set_qmax_mode(_capi.QMAX_FULL)
    _, y_full = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    d.set_qmax_mode(_capi.QMAX_SCREENED)
    _, y_scr = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    _, _, pairs, whole = d.qmax_stats()
    assert np.abs(y_full - y_scr).max() < 2e-6
    assert pairs >= 250 * n and whole >= 200 * n
    d.set_refine_stage(1)                                               # every group popular: 85 sweeps of the staged pass per block
    _, y_staged = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    d.set_refine_stage(-1)
    assert np.array_equal(y_staged, y_scr)
    # guard: a net like this one is detected from the counters of the first 32 screened steps and the steps after the next check
    # boundary run the full product — same results, and the screened-step counter stops
    # (the counters queued behind screened step 32 are evaluated at screened step 64: the fallback step depends on the step count only)
    for _ in range(75):
        _, y_again = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    steps_screened = d.qmax_stats()[0]
    assert steps_screened == 64, steps_screened
    assert np.abs(y_again - y_full).max() < 2e-6
    d.set_qmax_mode(_capi.QMAX_SCREENED)                                # an explicit request switches the screen back on
    # the winner is output 8099 (last row of the last, padded tile)
    bt[-1] = 5.0
    d.set_params(wt, bt, net=1)
    _, y_scr = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    d.set_qmax_mode(_capi.QMAX_FULL)
    _, y_full = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    assert np.abs(y_full - y_scr).max() < 2e-6 and y_scr.min() > 0.98
    # near-ties: every output row = one base row + a perturbation of the size of a bf16 rounding step, so the bf16 pass cannot tell
    # the rows apart (its own arg-max is mostly NOT the true one) and the result stands or falls with the error bound
    rng = np.random.default_rng(11)
    base = wt[-nw_out:-nw_out + 256].copy()
    rows = base[None, :] * (1.0 + rng.uniform(-2.0 ** -9, 2.0 ** -9, size=(8100, 256)))
    wt[-nw_out:] = rows.reshape(-1)
    bt[-1] = 0.0
    d.set_params(wt, bt, net=1)
    d.set_qmax_mode(_capi.QMAX_SCREENED)
    before = d.qmax_stats()
    _, y_scr = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    after = d.qmax_stats()
    assert after[2] - before[2] > 100 * n                              # hundreds of groups per sample stay candidates
    for i in range(0, n, 11):
        q2 = xo.nn_forward(sizes, wt, bt, xo.state_repr(xo.board_from(S2[i])))
        assert abs(y_scr[i] - 0.99 * q2.max()) < 2e-6, (i, y_scr[i], 0.99 * q2.max())
    env.close(); d.close()


@pytest.mark.gpu
def test_online_td_next_state_chain_derived_and_direct(xq):
    """With the online TD rule the layer-0 sums of s' are derived from those of s (minus the rows of the squares that changed, plus
    the rows of what stands there now) when the two boards are a move apart, and gathered in full otherwise.  Both routes against
    the oracle: real transitions (one move apart), and pairs of unrelated positions (dozens of squares differ)."""
    sizes = CFG2_NET
    n = 300
    env = xq.VecEnv(n, seed=321)
    for _ in range(17):
        env.selfplay_step(None)
    S, _ = env.get_state()
    res = env.selfplay_step(None)
    S2, _ = env.get_state()
    A = (res["action"] % 90).astype(np.int32)
    R = (res["reward"] / 100.0).astype(np.float32)
    D = np.zeros(n, np.uint8)
    d, w, b = make_net(xq, sizes, seed=12)
    ys = {}
    for derive in (False, True):                     # library default: the direct gather (the reference's summation order)
        d.set_l0_derive(derive)
        for name, nxt in (("one move apart", S2), ("unrelated", np.roll(S2, 7, axis=0))):
            assert ((S != nxt).sum(axis=1) > 8).any() == (name == "unrelated")
            qsa, y = d.td_update(S, nxt, A, R, D, td_net=0, mode=0, learning_rate=0.0, grad_scale=1.0)
            ys[(derive, name)] = y.copy()
            for i in range(0, n, 7):
                want = float(R[i]) + 0.99 * xo.nn_forward(sizes, w, b, xo.state_repr(xo.board_from(nxt[i]))).max()
                assert abs(y[i] - want) < 2e-6, (derive, name, i, y[i], want)
    # the two routes differ only in summation order: ~1e-7, not bit-identical on real transitions, identical where nothing is derived
    assert np.abs(ys[(True, "one move apart")] - ys[(False, "one move apart")]).max() < 2e-6
    assert np.array_equal(ys[(True, "unrelated")], ys[(False, "unrelated")]) or np.abs(ys[(True, "unrelated")] - ys[(False, "unrelated")]).max() < 2e-6
    env.close(); d.close()


def test_screened_qmax_when_one_group_holds_the_two_largest_outputs(xq):
    """The regime about a fifth of the runs from time-seeded weights train into (tools/whole_seed_scan.py): the trained rows dominate and the two
    largest outputs of nearly every sample sit in ONE 32-row group within the bf16 bound of each other, so the screen asks for that WHOLE group
    for nearly every sample.  qmax_refine2_kernel then stages the group's rows of W and the block's activation rows in LDS (one block per CU,
    K = 256).  Rows 0..31 = one base row + perturbations of a bf16 rounding step, lifted above the rest by their bias; 8192 samples."""
    from cn_chess_ai_amd import _capi
    sizes = CFG2_NET
    n = 8192
    env = xq.VecEnv(n, seed=6)
    for _ in range(7):
        env.selfplay_step(None)
    S, _ = env.get_state()
    res = env.selfplay_step(None)
    S2, _ = env.get_state()
    A = (res["action"] % 90).astype(np.int32)
    R = np.zeros(n, np.float32)
    D = np.zeros(n, np.uint8)
    d, w, b = make_net(xq, sizes, seed=21)
    rng = np.random.default_rng(3)
    wt = w.copy(); bt = b.copy()
    nw_out = 8100 * 256
    base = wt[-nw_out:-nw_out + 256].copy()
    for r in range(32):
        wt[len(wt) - nw_out + r * 256: len(wt) - nw_out + (r + 1) * 256] = base * (1.0 + 2.0 ** -9 * rng.standard_normal(256))
    bt[len(bt) - 8100: len(bt) - 8100 + 32] += 0.75
    d.set_params(wt, bt, net=1)
    d.set_qmax_mode(_capi.QMAX_FULL)
    _, y_full = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    d.set_qmax_mode(_capi.QMAX_SCREENED)
    ys = {}
    for stage in (0, 1):                                         # whole groups from global memory / the popular ones through LDS
        d.set_refine_stage(stage)
        _, ys[stage] = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    steps, samples, pairs, whole = d.qmax_stats()
    assert steps == 2 and samples == 2 * n
    assert whole >= 0.9 * 2 * n, (pairs, whole)                  # the regime the test is named after
    assert np.array_equal(ys[0], ys[1])                          # same dots, same order inside a dot: same bits
    assert np.abs(y_full - ys[1]).max() < 2e-6
    # a block whose samples ask for MORE than three popular groups (two sweeps of the staged pass) and for unpopular ones beside them: rows
    # 32..127 join the tie, so groups 0..3 are whole for every sample; then only every fourth sample's action row differs — nothing to do with
    # the maximum, the targets must not move
    for r in range(32, 128):
        wt[len(wt) - nw_out + r * 256: len(wt) - nw_out + (r + 1) * 256] = base * (1.0 + 2.0 ** -9 * rng.standard_normal(256))
    bt[len(bt) - 8100 + 32: len(bt) - 8100 + 128] += 0.75
    d.set_params(wt, bt, net=1)
    d.set_qmax_mode(_capi.QMAX_FULL)
    _, y_full = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    d.set_qmax_mode(_capi.QMAX_SCREENED)
    for stage in (0, 1):
        d.set_refine_stage(stage)
        _, ys[stage] = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
    assert np.array_equal(ys[0], ys[1])
    assert np.abs(y_full - ys[1]).max() < 2e-6



@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [CFG2_NET, (1260, 512, 512, 512, 8100)])
def test_screen_shadow_follows_every_parameter_change(xq, sizes):
    """The screening pass keeps the bf16 copy (and the row-norm / bias maxima) of output rows >= 96 from step to step — the TD rule
    only ever writes rows 0..95 — and converts everything again after set_params / updateTargetNetwork / a dense backpropagate.
    Each of those routes changes a row >= 96 so that it becomes the maximum; a stale shadow would miss it."""
    from cn_chess_ai_amd import _capi
    n = 1100
    env = xq.VecEnv(n, seed=17)
    for _ in range(12):
        env.selfplay_step(None)
    S, _ = env.get_state()
    res = env.selfplay_step(None)
    S2, _ = env.get_state()
    A = (res["action"] % 90).astype(np.int32)
    R = np.zeros(n, np.float32)
    D = np.zeros(n, np.uint8)
    d, w, b = make_net(xq, sizes, seed=21)
    H = sizes[-2]
    nb_out0 = len(b) - 8100

    def both(td_net):
        d.set_qmax_mode(_capi.QMAX_FULL)
        _, yf = d.td_update(S, S2, A, R, D, td_net=td_net, mode=0, learning_rate=0.0, grad_scale=1.0)
        d.set_qmax_mode(_capi.QMAX_SCREENED)
        _, ys = d.td_update(S, S2, A, R, D, td_net=td_net, mode=0, learning_rate=0.0, grad_scale=1.0)
        assert np.abs(yf - ys).max() < 2e-6
        return ys

    for td_net in (0, 1):
        y0 = both(td_net)
        both(td_net)                                            # a second screened step runs on the kept shadow
        # (a) set_params: bias of row 5000 up => that output wins everywhere
        b2 = b.copy(); b2[nb_out0 + 5000] = 3.0
        d.set_params(w, b2, net=td_net)
        y1 = both(td_net)
        assert y1.min() > 0.99 * np.tanh(2.0) and np.abs(y1 - y0).max() > 0.1
        # (b) weights of row 7001 scaled up (the row NORM maximum changes, and with it the bound)
        w2 = w.copy(); w2[-8100 * H + 7001 * H:-8100 * H + 7002 * H] *= 40.0
        d.set_params(w2, b, net=td_net)
        y2 = both(td_net)
        for i in range(0, n, 97):
            q2 = xo.nn_forward(sizes, w2, b, xo.state_repr(xo.board_from(S2[i])))
            assert abs(y2[i] - 0.99 * q2.max()) < QTOL
        d.set_params(w, b, net=td_net)
    # (c) updateTargetNetwork copies a changed online net into the target net the screen had cached
    both(1)
    b3 = b.copy(); b3[nb_out0 + 6500] = 2.5
    d.set_params(w, b3, net=0)
    d.updateTargetNetwork()
    y3 = both(1)
    assert y3.min() > 0.99 * np.tanh(1.5)
    # (d) a TD step with a learning rate really leaves rows >= 96 alone (what the kept shadow relies on)
    d.set_params(w, b, net=0)
    both(0)
    d.set_qmax_mode(_capi.QMAX_SCREENED)
    d.td_update(S, S2, A, R, D, td_net=0, mode=0, learning_rate=0.05, grad_scale=1.0 / n)
    w4, b4 = d.get_params()
    assert np.array_equal(w4[-8100 * H + 96 * H:], w[-8100 * H + 96 * H:].astype(np.float32).astype(np.float64))
    assert not np.array_equal(w4[-8100 * H:-8100 * H + 96 * H], w[-8100 * H:-8100 * H + 96 * H].astype(np.float32).astype(np.float64))
    both(0)
    env.close(); d.close()


@pytest.mark.gpu
def test_screened_qmax_with_non_finite_weights(xq):
    """a diverged net (an inf / NaN weight) must not make the screened maximum differ silently from the full product's:
    fmaxf semantics — a NaN output never wins, the largest finite (or +inf) output does."""
    from cn_chess_ai_amd import _capi
    sizes = CFG2_NET
    n = 1100
    env = xq.VecEnv(n, seed=23)
    for _ in range(15):
        env.selfplay_step(None)
    S, _ = env.get_state()
    res = env.selfplay_step(None)
    S2, _ = env.get_state()
    A = (res["action"] % 90).astype(np.int32)
    R = np.zeros(n, np.float32)
    D = np.zeros(n, np.uint8)
    d, w, b = make_net(xq, sizes, seed=4)
    for bad in (np.nan, np.inf):
        w2 = w.copy()
        w2[-8100 * 256 + 4321 * 256 + 17] = bad              # one weight of output row 4321
        d.set_params(w2, b, net=1)
        d.set_qmax_mode(_capi.QMAX_FULL)
        _, y_full = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
        d.set_qmax_mode(_capi.QMAX_SCREENED)
        _, y_scr = d.td_update(S, S2, A, R, D, td_net=1, mode=0, learning_rate=0.0, grad_scale=1.0)
        both_nan = np.isnan(y_full) & np.isnan(y_scr)
        assert (both_nan | (np.abs(y_full - y_scr) < 2e-6)).all(), bad
        assert np.isfinite(y_scr).mean() > 0.3
    env.close(); d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,mode,n,screened", [(REF_NET, 0, 700, False), (CFG2_NET, 0, 1100, False), (CFG2_NET, 1, 8192, False),
                                                   ((1260, 512, 512, 512, 8100), 0, 1500, False), ((1260, 64, 96, 200), 1, 333, False),
                                                   (CFG2_NET, 0, 1100, True), (CFG2_NET, 1, 8192, True), ((1260, 512, 512, 512, 8100), 0, 1500, True)])
def test_td_tail_launches_equal_the_two_stream_path_bitwise(xq, sizes, mode, n, screened):
    """xq_dqn_set_td_tail: the gradient half of the TD step as fused launches on one stream (blocks of the delta product, the weight
    gradients, the output-layer / layer-0 sums and the bias column sums in shared grids) against the same kernels launched one by
    one on two streams — every block runs the same body on the same operands, so two updates leave bit-identical parameters.
    Nets with one, two and three hidden layers, partial tiles (n not a multiple of 64), both backprop modes.  With exact screening
    the same switch also moves the TD target / delta arithmetic into the refine kernel's blocks (256-wide last hidden layer)."""
    from cn_chess_ai_amd import _capi
    env = xq.VecEnv(n, seed=5)
    for _ in range(23):
        env.selfplay_step(None)
    S, _ = env.get_state()
    res = env.selfplay_step(None)
    S2, _ = env.get_state()
    A = (res["action"] % 90).astype(np.int32)
    R = (res["reward"] / 100.0).astype(np.float32)
    D = res["done"].copy()
    D[::9] = 1
    out = {}
    # fused_apply: the partial sums of the step stay pending until the SGD kernel adds them; without it ONE launch behind the fused
    # grids reduces all of them into the gradient buffer (what a reader of the buffer, or an all-reduce, needs)
    for tail, fused in ((True, True), (True, False), (False, True), (False, False)):
        d, w, b = make_net(xq, sizes, seed=31)
        d.set_fused_apply(fused)
        d.set_td_tail(tail)
        d.set_qmax_mode(_capi.QMAX_SCREENED if screened else _capi.QMAX_FULL)
        for _ in range(2):
            qsa, y = d.td_update(S, S2, A, R, D, td_net=0, mode=mode, learning_rate=0.05, grad_scale=1.0 / n)
        out[(tail, fused)] = (d.get_params(), qsa.copy(), y.copy())
        d.close()
    (w1, b1), q1, y1 = out[(True, True)]
    for key in ((True, False), (False, True), (False, False)):
        (w0, b0), q0, y0 = out[key]
        assert np.array_equal(q1, q0) and np.array_equal(y1, y0), key
        assert np.array_equal(w1, w0) and np.array_equal(b1, b0), key
    wi, bi = xo.init_weights(sizes, 31)
    assert np.abs(w1 - wi).max() > 0                # ... and the updates did move the weights
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,n", [(CFG2_NET, 2048), ((1260, 512, 512, 512, 8100), 2048), (REF_NET, 300)])
def test_select_chain_layer0_sums_kept_between_plies(xq, sizes, n):
    """xq_dqn_set_l0_derive also covers the select chain (xq_dqn_select_q_dev = what xq_trainer_collect runs): its layer-0 sums are kept
    per game, and while the online parameters do not change the next call derives them from the kept ones (two squares per move; a
    game that ended shows the start position and is summed in full).  First call and every call after a parameter update: the
    full sum in the direct gather's order => the same bits as xq_dqn_forward_boards_dev.  Derived calls: a different summation
    order, ~1e-7.  Several plies in a row (the rounding of successive derivations adds up), games that end in between, a parameter
    update in between."""
    env = xq.VecEnv(n, seed=17)
    for _ in range(140):                            # mid-game: some games end (and reset) during the plies below
        env.selfplay_step(None)
    d, w, b = make_net(xq, sizes, seed=5)
    d.set_l0_derive(True)
    q_full = d.q_boards(env, 96).cpu().numpy()
    q_sel = d.select_q(env).cpu().numpy()
    assert np.array_equal(q_sel, q_full)            # nothing kept yet: full sums
    worst = 0.0
    for ply in range(6):
        env.selfplay_step(None)
        q_full = d.q_boards(env, 96).cpu().numpy()
        q_sel = d.select_q(env).cpu().numpy()       # derived from the sums kept by the previous call
        worst = max(worst, float(np.abs(q_sel - q_full).max()))
        assert np.abs(q_sel - q_full).max() < 2e-6, ply
    assert worst > 0 or sizes is REF_NET            # ... and it really was the other summation order somewhere
    S, _ = env.get_state()
    for i in range(0, n, max(1, n // 9)):
        want = xo.nn_forward(sizes, w, b, xo.state_repr(xo.board_from(S[i])))[:96]
        assert np.abs(q_sel[i] - want).max() < 2e-5
    # a parameter update drops the kept sums: the next call is a full sum again, bit-identical to the direct gather
    res = env.selfplay_step(None)
    S2, _ = env.get_state()
    d.set_fused_apply(True)
    d.td_update(S, S2, (res["action"] % 90).astype(np.int32), (res["reward"] / 100.0).astype(np.float32), res["done"], td_net=0, mode=0,
                learning_rate=0.05, grad_scale=1.0 / n)
    assert np.array_equal(d.select_q(env).cpu().numpy(), d.q_boards(env, 96).cpu().numpy())
    env.close(); d.close()


@pytest.mark.gpu
def test_replay_draws_are_ordered_against_their_consumer_across_streams(xq):
    """A ring with a stream of its own draws its slot list there; xq_dqn_td_grads_replay reads the list on the Q-net's stream.  Drawing
    again while the previous TD step is still queued used to overwrite the list under it (found by the overlapped-trainer
    composition once the step enqueued faster).  The library orders the two itself: a loop that never synchronises between an
    update and the next draw must give the bits of the same loop with a device synchronisation after every update."""
    import torch
    sizes, n, cap, iters = CFG2_NET, 4096, 1 << 15, 8
    env = xq.VecEnv(n, seed=3)
    rp0 = xq.ReplayBuffer(cap, seed=11)
    for _ in range(8):                                   # fill the ring with random plies
        env.selfplay_step_dev(0, 96, 0.1, replay=rp0)
    torch.cuda.synchronize()
    rp0.close()
    out = []
    for sync in (True, False):
        env2 = xq.VecEnv(n, seed=3)
        rp = xq.ReplayBuffer(cap, seed=11)
        for _ in range(8):
            env2.selfplay_step_dev(0, 96, 0.1, replay=rp)
        torch.cuda.synchronize()
        d, w, b = make_net(xq, sizes, seed=2)
        for it in range(iters):
            rp.sample(n)                                 # own stream of the ring; the list is consumed on the Q-net's stream
            d.td_grads_replay(rp, n, td_net=0, mode=0)
            d.apply_grads(0.05, 1.0 / n)
            if sync:
                torch.cuda.synchronize()
        out.append(d.get_params())
        d.close(); rp.close(); env2.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    env.close()


@pytest.mark.gpu
def test_hidden_tanh_accuracy(xq):
    """The forward epilogue of the hidden products computes tanh as 1 - 2 / (e^{2|x|} + 1) on the hardware exp2 / reciprocal, with the
    odd Taylor polynomial below |x| = 0.05 (xq_gemm.hip.h::tanh_hidden) where NeuralNetwork::forward (dqn.cu:184-195) calls libm in
    fp64.  Measured THROUGH the GEMM: identity weights make a hidden unit tanh(x_j), the output layer (libm tanhf) returns
    tanh(tanh(x_j)), whose derivative w.r.t. the hidden value is >= 0.42 — an error e of the hidden tanh shows as >= 0.42 e.
    Budget: 3e-7 absolute, 1e-5 relative (north_star holds Q to 1e-4)."""
    H = 128
    d = xq.DQN([H, H, H], 0.001, 0.99, seed=1)
    w = np.concatenate([np.eye(H).ravel(), np.eye(H).ravel()])
    d.set_params(w, np.zeros(2 * H))
    mags = np.concatenate([np.logspace(-6, np.log10(0.2), 3000), np.linspace(0.04, 0.06, 800), np.linspace(0.2, 20.0, 4392)])
    x = (mags * np.where(np.arange(len(mags)) & 1, -1.0, 1.0)).astype(np.float32).astype(np.float64).reshape(64, H)
    q = d.getQValues(x)
    inner = np.tanh(x)
    want = np.tanh(inner)
    err = np.abs(q - want)
    assert err.max() < 3e-7, err.max()
    rel = err / np.abs(want)
    assert rel.max() < 1e-5, rel.max()
    assert np.all(np.sign(q) == np.sign(x))
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,n", [(CFG2_NET, 1100), ([1260, 64, 8100], 300), (CFG2_NET, 2048), ([1260, 512, 512, 8100], 1024), ([1260, 128, 128, 8100], 3072)])
def test_layer0_gradient_on_the_matrix_pipe_matches_the_segmented_sums(xq, trace, sizes, n):
    """xq_dqn_set_l0_grad_mode(1): gW0 = one-hot^T x delta_0 as a bf16 MFMA product with delta_0 split exactly into three bf16 values
    (xq_l0grad.hip.h).  Same TD step, both modes: every parameter outside layer 0 bit-identical, layer 0 equal up to the summation
    order (<= two fp32 ulps of the stored weight), and the matrix-pipe update itself within PTOL of the fp64 oracle.
    Shapes: 1100 samples = a partial chunk (zero-padded planes, split kernel of its own); 2048 / 3072 = whole chunks (planes written by the
    delta product's epilogue, selector words riding in fused launch 1); widths 64 / 128 / 256 / 512 = 2 / 4 / 8 / 16 column blocks."""
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, n, seed=6))
    R = R / 1000.0
    lr, scale = 0.05, 1.0 / n
    outs = []
    for mode in (0, 1):
        d, w, b = make_net(xq, sizes, seed=9)
        d.set_l0_grad_mode(mode)
        d.td_update(S, S2, A, R, D, td_net=0, mode=0, learning_rate=lr, grad_scale=scale)
        outs.append(d.get_params())
        d.close()
    (w0, b0), (w1, b1) = outs
    n0 = sizes[0] * sizes[1]
    assert np.array_equal(w0[n0:], w1[n0:]) and np.array_equal(b0, b1)
    upd = np.abs(w0[:n0] - w[:n0]).max()
    # the parameters are stored in fp32: |w| <= 0.06 => one ulp is 3.7e-9; the two summation orders may round the stored value apart by it
    assert upd > 0 and np.abs(w0[:n0] - w1[:n0]).max() <= 8e-9 and not np.array_equal(w0[:n0], w[:n0])
    if n <= 300:
        want_w, want_b, _, _ = oracle_td_update(sizes, w, b, w, b, S, A, R, D, S2, 0.99, lr, scale, 0)
        assert np.abs(w1 - want_w).max() < PTOL and np.abs(b1 - want_b).max() < PTOL

"""The HIP Q-network against the REAL reference NN runtime directly (no oracle in between).

tests/golden/ref_nn.npz = outputs of /root/reference/src/dqn.cu's own kernels run on an MI355X (see
tests/test_ref_nn_golden.py for how the fixture is made and what it can pin).  Here the product path — the C ABI's
xq_dqn_forward / xq_dqn_backpropagate in its as-written mode, fp32 on the matrix pipe — is fed the same parameters and
inputs and compared with what the reference left: Q-values within north_star's 1e-4 (in practice 1e-6), parameters after
one update within PTOL of the reference's fp64 ones.
"""
import numpy as np
import pytest

import refnn

pytestmark = pytest.mark.gpu

QTOL = 1e-4          # north_star tolerance on Q-values
PTOL = 2e-5          # parameters after one update (fp32 storage of values up to ~0.2)
G = refnn.Golden()
# xq_dqn_create takes 1..7 hidden layers (include/xq_capi.h); the fixture's {20, 30} net has none and stays an oracle-only case
TOPOS = [t for t in G.topologies if len(t[1]) >= 3]


@pytest.fixture(scope="module")
def xq():
    import cn_chess_ai_amd as m
    assert m._capi.device_count() > 0
    return m


def ids(t):
    return "-".join(str(s) for s in t[1])


@pytest.mark.parametrize("topo", TOPOS, ids=ids)
def test_forward_matches_the_reference_run(xq, topo):
    seed, sizes = topo
    w, b = refnn.params(seed, sizes)
    d = xq.DQN(sizes, 0.001, 0.99, seed=1)
    d.set_params(w, b)
    qpos = G.get(sizes, "q_positions")
    xs = np.stack([G.input(sizes, f"fwd{k}", seed) for k in range(6)])
    q = d.getQValues(xs)
    worst = 0.0
    for k in range(6):
        want = G.get(sizes, f"fwd{k}_q")
        worst = max(worst, np.abs(q[k][qpos] - want).max())
        assert abs(q[k].max() - G.get(sizes, f"fwd{k}_max_sum")[0]) < QTOL
        am = int(G.get(sizes, f"fwd{k}_argmax")[0])
        assert q[k][am] >= q[k].max() - 2e-6                      # the reference's arg-max is a maximum here too (fp32 ties aside)
    assert worst < QTOL and worst < 5e-6
    d.close()


@pytest.mark.parametrize("topo", TOPOS, ids=ids)
def test_backpropagate_matches_the_reference_run(xq, topo):
    seed, sizes = topo
    w, b = refnn.params(seed, sizes)
    nhid = sum(sizes[1:-1])
    n0 = sizes[0] * sizes[1]
    d = xq.DQN(sizes, 0.001, 0.99, seed=1)
    for u in range(4):
        tag = f"bp{u}"
        d.set_params(w, b)
        x = G.input(sizes, tag, seed)
        a = int(G.get(sizes, tag + "_action")[0])
        y, lr = G.get(sizes, tag + "_y_lr")
        t = d.getQValues(x).astype(np.float64)                  # the caller's own Q, as chessai.cpp:121-133 builds the target
        t[a] = y
        d.backpropagate(x, t, lr, 1.0, 0)                       # mode 0 = the hidden delta as written upstream
        gw, gb = d.get_params()
        if nhid:
            assert np.abs(gb[:nhid] - G.get(sizes, tag + "_hidden_biases")).max() < PTOL
        ob = G.get(sizes, tag + "_out_biases")
        assert np.abs(gb[nhid:nhid + len(ob)] - ob).max() < PTOL
        rows, cols = G.get(sizes, tag + "_w0_rows"), G.get(sizes, tag + "_w0_cols")
        got0 = gw[:n0].reshape(sizes[1], sizes[0])[np.ix_(rows, cols)]
        ref0 = G.get(sizes, tag + "_w0").reshape(len(rows), len(cols))
        assert np.abs(got0 - ref0).max() < PTOL
        # the update itself, not just the parameters: relative to the step the reference took
        step = np.abs(ref0 - w[:n0].reshape(sizes[1], sizes[0])[np.ix_(rows, cols)]).max()
        if step > 1e-4:
            assert np.abs(got0 - ref0).max() < 2e-3 * step + 1e-7
    d.close()


@pytest.mark.refnn_live
def test_live_reference_run_on_a_fresh_seed(xq, tmp_path):
    """The reference NN runtime itself (oracle/_ref/xqref_nn travels with the snapshot), run NOW on this GPU with a seed the
    fixture does not hold: its Q-values and its one-step update against the oracle (fp64, 1e-14) and the HIP path.
    OPT-IN (XQ_RUN_REFERENCE_NN=1): upstream's backpropagate reads device memory it has released (dqn.cu:371 / :441) and indexes
    zs[l] past its end (:420); the committed fixtures (ref_nn.npz, ref_nn_seq.npz) pin the same values, so the default `-m gpu`
    run never executes that binary on a shared GPU box."""
    import os
    import subprocess
    if os.environ.get("XQ_RUN_REFERENCE_NN") != "1":
        pytest.skip("opt-in: set XQ_RUN_REFERENCE_NN=1 to run upstream's NN runtime (undefined behaviour included) on this GPU")

    import gen_golden_nn as gg
    import xqoracle as xo
    if not os.path.exists(gg.BIN):
        pytest.skip("oracle/_ref/xqref_nn not built (needs /root/reference in the authoring container)")
    probe = subprocess.run([gg.BIN, "probe"], capture_output=True, text=True)
    if probe.returncode != 0:
        pytest.skip("allocator probe: small blocks are not sub-allocated here; not running the reference's backpropagate")
    seed, sizes = 4242, [1260, 256, 256, 8100]
    out = str(tmp_path / "live.bin")
    subprocess.run(["timeout", "-k", "10", "120", gg.BIN, "nn", out, str(seed)] + [str(s) for s in sizes], check=True)
    rec = gg.parse(out)
    w, b = refnn.params(seed, sizes)
    d = xq.DQN(sizes, 0.001, 0.99, seed=1)
    d.set_params(w, b)
    qpos = rec["q_positions"]

    def inp(tag):
        x = np.zeros(sizes[0])
        if tag + "_onehot" in rec:
            x[rec[tag + "_onehot"]] = 1.0
        else:
            x[:] = rec[tag + "_dense"]
        return x
    for k in range(6):
        x = inp(f"fwd{k}")
        assert np.abs(xo.nn_forward(sizes, w, b, x)[qpos] - rec[f"fwd{k}_q"]).max() <= 1e-14
        assert np.abs(d.getQValues(x)[qpos] - rec[f"fwd{k}_q"]).max() < 5e-6
    nhid, n0 = sum(sizes[1:-1]), sizes[0] * sizes[1]
    for u in range(4):
        tag = f"bp{u}"
        x = inp(tag)
        a, (y, lr) = int(rec[tag + "_action"][0]), rec[tag + "_y_lr"]
        t = xo.nn_forward(sizes, w, b, x)
        t[a] = y
        w2, b2 = w.copy(), b.copy()
        assert xo.nn_backprop(sizes, w2, b2, x, t, lr, 0) == 0
        rows, cols = rec[tag + "_w0_rows"], rec[tag + "_w0_cols"]
        ref0 = rec[tag + "_w0"].reshape(len(rows), len(cols))
        assert np.abs(b2[:nhid] - rec[tag + "_hidden_biases"]).max() <= 1e-14
        assert np.abs(w2[:n0].reshape(sizes[1], sizes[0])[np.ix_(rows, cols)] - ref0).max() <= 1e-14
        d.set_params(w, b)
        d.backpropagate(x, t, lr, 1.0, 0)
        gw, gb = d.get_params()
        assert np.abs(gb[:nhid] - rec[tag + "_hidden_biases"]).max() < PTOL
        assert np.abs(gw[:n0].reshape(sizes[1], sizes[0])[np.ix_(rows, cols)] - ref0).max() < PTOL
    d.close()


@pytest.mark.parametrize("topo", [t for t in TOPOS if t[1][0] == 1260], ids=ids)
def test_td_hot_path_matches_the_reference_run(xq, topo):
    """The TD step itself — xq_dqn_td_grads + apply from packed BOARDS (layer-0 gather, sparse output delta, hidden deltas as
    written, layer-0 segmented sums, bias sums, SGD) — against single NeuralNetwork::backpropagate calls of the reference run.
    The fixture's one-hot inputs are boards (distinct squares, one of 14 planes each); a terminal transition (done = 1) with reward
    y gives exactly the reference's target: the net's own Q with entry `action` replaced by y (chessai.cpp:121-133)."""
    seed, sizes = topo
    w, b = refnn.params(seed, sizes)
    nhid, n0 = sum(sizes[1:-1]), sizes[0] * sizes[1]
    d = xq.DQN(sizes, 0.001, 0.99, seed=1)
    for u in range(3):
        tag = f"bp{u}"
        idx = G.get(sizes, tag + "_onehot")
        board = np.zeros((1, 90), dtype=np.uint8)
        board[0, idx // 14] = idx % 14 + 1
        a = int(G.get(sizes, tag + "_action")[0])
        y, lr = G.get(sizes, tag + "_y_lr")
        d.set_params(w, b)
        qsa, yy = d.td_update(board, board, np.array([a], np.int32), np.array([y], np.float32), np.array([1], np.uint8),
                              td_net=0, mode=0, learning_rate=float(lr), grad_scale=1.0)
        assert abs(yy[0] - y) < 1e-6
        gw, gb = d.get_params()
        assert np.abs(gb[:nhid] - G.get(sizes, tag + "_hidden_biases")).max() < PTOL
        ob = G.get(sizes, tag + "_out_biases")
        assert np.abs(gb[nhid:nhid + len(ob)] - ob).max() < PTOL
        rows, cols = G.get(sizes, tag + "_w0_rows"), G.get(sizes, tag + "_w0_cols")
        got0 = gw[:n0].reshape(sizes[1], sizes[0])[np.ix_(rows, cols)]
        ref0 = G.get(sizes, tag + "_w0").reshape(len(rows), len(cols))
        step = np.abs(ref0 - w[:n0].reshape(sizes[1], sizes[0])[np.ix_(rows, cols)]).max()
        assert np.abs(got0 - ref0).max() < PTOL and np.abs(got0 - ref0).max() < 2e-3 * step + 1e-7
        # the one-hidden-layer reference net: upstream's output-layer update came out of released memory intact in this run
        # (tests/test_ref_nn_golden.py reports it), so the whole parameter set is comparable there
        if len(sizes) == 3:
            pos = refnn.sample_positions(seed, 1, sizes)[:64]
            assert np.abs(gw[n0 + pos] - G.get(sizes, f"{tag}_ub_w1")).max() < PTOL
    d.close()

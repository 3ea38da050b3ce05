"""The oracle's NN half against the REAL reference NN runtime, executed on an MI355X.

tests/golden/ref_nn.npz holds outputs of /root/reference/src/dqn.cu itself — its five kernels and the host code of
NeuralNetwork::forward / ::backpropagate, compiled for gfx950 through the image's hipify-perl (API identifiers only;
oracle/Makefile target `refnn`, oracle/ref/ref_nn_driver.cpp, oracle/gen_golden_nn.py) and run on the GPU box.  This
pins by EXECUTION what rounds 1-3 could only restate: rows N3/N4 (forward, dqn.cu:184-260), N5 (backpropagate with the
hidden delta exactly as written, :275-467), N6 (copyWeightsAndBiasesFrom copies stale host vectors, :507-515) and N8
(constructor: counts, offsets, init range, :14-146) of SURVEY.md §8.

Asserted: everything free of undefined behaviour upstream — Q-values, every bias, all of layer 0's weights (they carry
the as-written hidden delta through every layer above).  The updated weights of layers >= 1 depend on a read of device
memory the reference has already released (freed :371, read :441): whatever the allocator left there.  They are in the
fixture as `ub_*`; where HIP's allocator kept the block intact they equal the oracle's "stale bytes intact" model, and
this file reports, but does not require, that.
"""
import numpy as np
import pytest

import refnn
import xqoracle as xo

G = refnn.Golden()
TOPOS = G.topologies
# fp64 on both sides; differences are FMA contraction (-ffp-contract=off in both builds) and tanh ulps
ATOL = 1e-14


def ids(t):
    return "-".join(str(s) for s in t[1])


def test_fixture_comes_from_a_run_whose_stale_reads_stayed_in_mapped_memory():
    assert G.probe["small_blocks_share_one_2mb_block"] is True
    assert [t for _, t in TOPOS][:3] == [[1260, 128, 8100], [1260, 256, 256, 8100], [1260, 512, 512, 512, 8100]]   # BASELINE's topologies


@pytest.mark.parametrize("topo", TOPOS, ids=ids)
def test_constructor_counts_offsets_and_init_range(topo):
    """N8 (dqn.cu:14-57, 96-146): reference flat layout, U(-0.05, 0.05) weights, zero biases, device copy uploaded."""
    seed, sizes = topo
    nw, nb = xo.nn_counts(sizes)
    assert tuple(G.get(sizes, "counts")) == (nw, nb) == refnn.counts(sizes)
    wo = np.cumsum([0] + [a * b for a, b in zip(sizes[:-1], sizes[1:])])[:-1]
    bo = np.cumsum([0] + list(sizes[1:]))[:-1]
    assert (G.get(sizes, "weight_offsets") == wo).all() and (G.get(sizes, "bias_offsets") == bo).all()
    lo, hi, mean = G.get(sizes, "init_w_min_max_mean")
    assert -0.05 <= lo < hi <= 0.05 and abs(mean) < 0.01
    if nw > 100000:
        assert lo < -0.0499 and hi > 0.0499          # the whole range is used
    assert G.get(sizes, "init_b_absmax")[0] == 0.0
    assert G.get(sizes, "init_forward_of_zero_absmax")[0] == 0.0
    if sizes == [1260, 128, 8100]:
        assert nw == 1198080 and nb == 8228


@pytest.mark.parametrize("topo", TOPOS, ids=ids)
def test_forward_matches_the_reference_kernels(topo):
    """N3 / N4: xqo_nn_forward against forwardKernel (6 arguments, bias added last) run by NeuralNetwork::forward."""
    seed, sizes = topo
    w, b = refnn.params(seed, sizes)
    qpos = G.get(sizes, "q_positions")
    for k in range(6):
        x = G.input(sizes, f"fwd{k}", seed)
        q = xo.nn_forward(sizes, w, b, x)
        assert np.abs(q[qpos] - G.get(sizes, f"fwd{k}_q")).max() <= ATOL
        mx, sm = G.get(sizes, f"fwd{k}_max_sum")
        assert abs(q.max() - mx) <= ATOL and abs(q.sum() - sm) <= ATOL * len(q)
        assert int(np.argmax(q)) == int(G.get(sizes, f"fwd{k}_argmax")[0])


@pytest.mark.parametrize("topo", TOPOS, ids=ids)
def test_backpropagate_matches_the_reference_kernels(topo):
    """N5: one NeuralNetwork::backpropagate from known parameters — xqo_nn_backprop in its as-written mode (0) against the
    biases and layer-0 weights the reference left on the device."""
    seed, sizes = topo
    w, b = refnn.params(seed, sizes)
    nL = len(sizes) - 1
    nhid = sum(sizes[1:-1])
    n0 = sizes[0] * sizes[1]
    moved = 0.0
    for u in range(4):
        tag = f"bp{u}"
        x = G.input(sizes, tag, seed)
        a = int(G.get(sizes, tag + "_action")[0])
        y, lr = G.get(sizes, tag + "_y_lr")
        t = xo.nn_forward(sizes, w, b, x)
        t[a] = y                                              # chessai.cpp:121-133: the net's own Q with one entry replaced
        w2, b2 = w.copy(), b.copy()
        assert xo.nn_backprop(sizes, w2, b2, x, t, lr, 0) == 0
        if nhid:
            assert np.abs(b2[:nhid] - G.get(sizes, tag + "_hidden_biases")).max() <= ATOL
        ob = G.get(sizes, tag + "_out_biases")
        assert np.abs(b2[nhid:nhid + len(ob)] - ob).max() <= ATOL
        assert G.get(sizes, tag + "_out_biases_rest_maxdiff")[0] < 1e-15       # a - target = ulps of the two forward kernels
        if nhid + len(ob) < len(b):
            assert np.abs(b2[nhid + len(ob):] - b[nhid + len(ob):]).max() < 1e-15
        rows, cols = G.get(sizes, tag + "_w0_rows"), G.get(sizes, tag + "_w0_cols")
        W0 = w2[:n0].reshape(sizes[1], sizes[0])
        ref0 = G.get(sizes, tag + "_w0").reshape(len(rows), len(cols))
        assert np.abs(W0[np.ix_(rows, cols)] - ref0).max() <= ATOL
        moved = max(moved, np.abs(ref0 - w[:n0].reshape(sizes[1], sizes[0])[np.ix_(rows, cols)]).max())
        assert int(G.get(sizes, tag + "_w0_changed_elsewhere")[0]) == 0           # one-hot input: only its columns move
        if G.has(sizes, tag + "_onehot"):
            mask = np.ones(sizes[0], dtype=bool)
            mask[cols] = False
            assert (W0[:, mask] == w[:n0].reshape(sizes[1], sizes[0])[:, mask]).all()
    assert moved > 1e-6, "the fixture's updates must be visible, or the comparison above pins nothing"


def test_as_written_hidden_delta_is_what_the_reference_computes():
    """The quirk of dqn.cu:406-423 (sizes shifted by one layer, truncated sum, cross-layer stride), pinned by execution: on
    every net with two or more hidden layers the textbook rule gives different layer-0 weights than the reference left."""
    for seed, sizes in TOPOS:
        if len(sizes) < 4:
            continue
        w, b = refnn.params(seed, sizes)
        n0 = sizes[0] * sizes[1]
        x = G.input(sizes, "bp2", seed)
        a = int(G.get(sizes, "bp2_action")[0])
        y, lr = G.get(sizes, "bp2_y_lr")
        t = xo.nn_forward(sizes, w, b, x)
        t[a] = y
        rows, cols = G.get(sizes, "bp2_w0_rows"), G.get(sizes, "bp2_w0_cols")
        ref0 = G.get(sizes, "bp2_w0").reshape(len(rows), len(cols))
        w_ref, b_ref = w.copy(), b.copy()
        w_txt, b_txt = w.copy(), b.copy()
        xo.nn_backprop(sizes, w_ref, b_ref, x, t, lr, 0)
        xo.nn_backprop(sizes, w_txt, b_txt, x, t, lr, 1)
        d_ref = np.abs(w_ref[:n0].reshape(sizes[1], sizes[0])[np.ix_(rows, cols)] - ref0).max()
        d_txt = np.abs(w_txt[:n0].reshape(sizes[1], sizes[0])[np.ix_(rows, cols)] - ref0).max()
        assert d_ref <= ATOL < 1e-9 < d_txt, (sizes, d_ref, d_txt)


@pytest.mark.parametrize("topo", TOPOS, ids=ids)
def test_copy_of_a_trained_net_is_the_untrained_net_upstream(topo):
    """N6 (dqn.cu:507-515): the copy answers exactly like the net before the update; the update itself moved the outputs."""
    _, sizes = topo
    d_pre, d_post, d_move = G.get(sizes, "copy_vs_pre_vs_post_moved")
    assert d_pre == 0.0 and d_move > 1e-3 and d_post == d_move


def test_report_weights_that_depend_on_released_memory(capsys):
    """Not a requirement: which layers' weight updates equal the oracle's "released block still holds the activation" model."""
    lines = []
    for seed, sizes in TOPOS:
        w, b = refnn.params(seed, sizes)
        nL = len(sizes) - 1
        for u in range(4):
            tag = f"bp{u}"
            x = G.input(sizes, tag, seed)
            a = int(G.get(sizes, tag + "_action")[0])
            y, lr = G.get(sizes, tag + "_y_lr")
            t = xo.nn_forward(sizes, w, b, x)
            t[a] = y
            w2, b2 = w.copy(), b.copy()
            xo.nn_backprop(sizes, w2, b2, x, t, lr, 0)
            off = sizes[0] * sizes[1]
            for l in range(1, nL):
                pos = refnn.sample_positions(seed, l, sizes)[:64]
                d = np.abs(w2[off + pos] - G.get(sizes, f"{tag}_ub_w{l}")).max()
                lines.append(f"{'-'.join(map(str, sizes))} {tag} layer {l}: {'intact' if d <= ATOL else 'reused'} ({d:.1e})")
                off += sizes[l] * sizes[l + 1]
    with capsys.disabled():
        n_ok = sum("intact" in s for s in lines)
        print(f"\n[ref_nn] weights of layers >= 1 after the read of released memory: {n_ok} of {len(lines)} equal the stale-bytes-intact model")

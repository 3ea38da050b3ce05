"""CPU checks of the BUILD-DEFINED part of the oracle (oracle/xq_oracle_ext.c: Double DQN, prioritized replay, bf16 Q-net).

There is no upstream code to pin these to; what CAN be pinned is that the generalised TD step reduces to the restated reference
step (chessai.cpp:122-131 / dqn.cpp:157-172) where the two overlap, and that the sampler really draws proportionally.
"""
import ctypes as C
import os

import numpy as np

import xqoracle as xo
from test_dqn_gpu import oracle_td_update, transitions, valid_indices
from test_config5_gpu import oracle_update


def test_ext_oracle_reduces_to_the_reference_restatement(golden_dir):
    """td_rule 0 / 1 of the generalised step in fp64 ARE chessai.cpp:122-131 / dqn.cpp:157-172: same gradients as the base oracle."""
    trace = np.load(os.path.join(golden_dir, "ref_trace.npz"))
    sizes = [1260, 48, 48, 8100]
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, 12, seed=3))
    R = R / 1000.0
    w, b = xo.init_weights(sizes, 5)
    wt, bt = xo.init_weights(sizes, 6)
    for td_rule, (tw, tb) in ((0, (w, b)), (1, (wt, bt))):
        for mode in (0, 1):
            a = oracle_td_update(sizes, w, b, tw, tb, S, A, R, D, S2, 0.99, 0.05, 1 / 12, mode)
            e = oracle_update(sizes, w, b, wt, bt, S, A, R, D, S2, 0.99, 0.05, 1 / 12, mode, td_rule)
            assert np.abs(a[0] - e[0]).max() < 1e-12 and np.abs(a[1] - e[1]).max() < 1e-12
            assert np.abs(a[2] - e[2]).max() < 1e-12 and np.abs(a[3] - e[3]).max() < 1e-12
    # Double DQN picks the online arg-max and reads the target net there
    x2 = xo.state_repr(xo.board_from(S2[0]))
    gw, gb = np.zeros_like(w), np.zeros_like(b)
    q, y, star = xo.ext_td_accum(sizes, w, b, wt, bt, xo.state_repr(xo.board_from(S[0])), x2, int(A[0]), float(R[0]), 0, 0.99, 2, 1, False,
                                 1.0, gw, gb)
    zo, zt = xo.ext_forward(sizes, w, b, x2)[1], xo.ext_forward(sizes, wt, bt, x2)[1]
    assert star == int(np.argmax(zo)) and abs(y - (R[0] + 0.99 * np.tanh(zt[star]))) < 1e-15


def test_bf16_rounding_is_round_to_nearest_even():
    f = xo.lib().xqo_bf16_round
    # 1 + 2^-8 is a tie between 1.0 and 1 + 2^-7: even mantissa (1.0) wins; 1 + 3*2^-8 ties to 1 + 2^-6 (even) not 1 + 2^-7
    assert f(1.0) == 1.0 and f(1.00390625) == 1.0 and f(1.01171875) == 1.015625 and f(1.005) == 1.0078125 and f(-1.005) == -1.0078125
    x = np.random.default_rng(0).normal(size=1000).astype(np.float32)
    r = np.array([f(float(v)) for v in x], dtype=np.float32)
    assert (r.view(np.uint32) & 0xFFFF == 0).all() and np.abs(r - x).max() <= np.abs(x).max() * 2.0 ** -8


def test_sum_tree_sampler_is_proportional():
    cap = 3000
    rng = np.random.default_rng(1)
    prio = rng.uniform(0, 1, cap).astype(np.float32)
    prio[rng.choice(cap, 500, replace=False)] = 0
    tree = xo.per_build(prio)
    total = xo.lib().xqo_per_total(tree.ctypes.data_as(C.POINTER(C.c_float)), cap)
    assert abs(total - prio.astype(np.float64).sum()) < 1e-3 * total
    # every mass u lands on the leaf whose cumulative interval contains it
    cum = np.cumsum(prio.astype(np.float64))
    for u in np.linspace(0, total * 0.9999, 400):
        leaf = xo.lib().xqo_per_descend(tree.ctypes.data_as(C.POINTER(C.c_float)), cap, C.c_float(u))
        assert prio[leaf] > 0 and abs(np.searchsorted(cum, u, side="right") - leaf) <= 1
    counts = np.zeros(cap)
    for call in range(40):
        slots, w, wmax = xo.per_sample(tree, cap, 2048, 7, call, int((prio > 0).sum()), 0.4)
        np.add.at(counts, slots, 1)
        assert (prio[slots] > 0).all() and wmax == w.max()
    big = prio > 0.5
    assert abs(counts[big].sum() / counts.sum() - prio[big].sum() / prio.sum()) < 0.01


def test_bf16_backward_definition_stays_close_to_the_fp32_backward(golden_dir):
    """XQ_PRECISION_BF16_FULL (oracle bf16 = 2): bf16 operands in the lower hidden deltas and the hidden weight gradients.  Same forward
    as bf16 = 1 (bit-identical Q and y); the update differs from the fp32-backward one by a few bf16 roundings per product — stated
    tolerance: 2 % of the update's largest entry per parameter block — and it is not the same function (the rounding is applied)."""
    trace = np.load(os.path.join(golden_dir, "ref_trace.npz"))
    sizes = [1260, 64, 64, 64, 8100]
    n = 24
    S, A, R, D, S2 = transitions(trace, valid_indices(trace, n, seed=5))
    R = R / 1000.0
    w, b = xo.init_weights(sizes, 5)
    b = np.random.default_rng(2).uniform(-0.05, 0.05, size=len(b))
    wt, bt = xo.init_weights(sizes, 6)
    for mode in (0, 1):
        u1 = oracle_update(sizes, w, b, wt, bt, S, A, R, D, S2, 0.99, 1.0, 1.0 / n, mode, 2, bf16=1)
        u2 = oracle_update(sizes, w, b, wt, bt, S, A, R, D, S2, 0.99, 1.0, 1.0 / n, mode, 2, bf16=2)
        assert np.array_equal(u1[2], u2[2]) and np.array_equal(u1[3], u2[3]) and np.array_equal(u1[4], u2[4])
        dw1, dw2 = u1[0] - w, u2[0] - w
        off = 0
        for l in range(len(sizes) - 1):
            cnt = sizes[l] * sizes[l + 1]
            a, c = dw1[off:off + cnt], dw2[off:off + cnt]
            assert np.abs(a).max() > 0
            assert np.abs(a - c).max() <= 0.02 * np.abs(a).max(), (mode, l)
            if l == len(sizes) - 2:
                assert np.array_equal(a, c)                 # output layer: untouched by the mode
            elif l >= 1:
                assert np.abs(a - c).max() > 0              # hidden layers: really rounded
            off += cnt
        assert np.abs((u1[1] - b) - (u2[1] - b)).max() <= 0.02 * np.abs(u1[1] - b).max()

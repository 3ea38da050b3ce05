"""A K-ply TRAJECTORY of the reference's own NN runtime on its own net {1260, 128, 8100} (tests/golden/ref_nn_seq.npz; made by
oracle/gen_golden_nn_seq.py from `xqref_nn seq`: /root/reference/src/dqn.cu through hipify-perl, run on an MI355X).

tests/golden/ref_nn.npz pins single forward / backpropagate calls from fixed parameters.  Here the parameters are carried from ply to
ply by the reference itself: per ply the NN half of ChessAI::train's loop body (chessai.cpp:121-133) — targetQ = getQValues(state),
targetQ[action.to] = done ? r : r + gamma * max(getQValues(nextState)), backpropagate(state, targetQ, lr) — on 17 consecutive positions
of a real-rules-engine game with evaluateBoard's integer rewards.  This pins by EXECUTION row N1's NN half and, on this net, the
output-layer update of dqn.cu:438-446 as well: backpropagate reads the hidden activation after releasing it (freed :371, read :441), and the
fixture's `intact` flags say at which plies the released block still held it (all of them in the recorded run — then the whole
trajectory, output layer included, is comparable end to end).

CPU: the oracle (fp64) follows the trajectory to <= 1e-12.  GPU: the HIP TD path (fp32, batch 1, as-written backprop) follows it to
north_star's 1e-4 on Q-values after every one of the updates.
"""
import os

import numpy as np
import pytest

import refnn
import xqoracle as xo

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_nn_seq.npz")
Z = np.load(PATH)
SIZES = [int(s) for s in Z["sizes"]]
SEED, LR, GAMMA = int(Z["seed_lr_gamma"][0]), float(Z["seed_lr_gamma"][1]), float(Z["seed_lr_gamma"][2])
K = len(Z["action_to"])
GOOD = int(Z["first_bad_ply"][0])          # plies [0, GOOD) ran with the released block intact


def onehot(codes):
    """chessai.cpp:268-289: index sq * 14 + (piece code - 1)"""
    x = np.zeros(SIZES[0])
    x[[sq * 14 + int(c) - 1 for sq, c in enumerate(codes) if c]] = 1.0
    return x


def test_the_recorded_run_is_usable():
    assert SIZES == [1260, 128, 8100] and K == 16 and (LR, GAMMA) == (0.001, 0.99)
    assert GOOD >= 8, "the allocator handed the released block on too early in the recorded run: regenerate the fixture"
    assert Z["intact"][:GOOD].all()
    # reference-scale rewards (evaluateBoard's integers, chessai.cpp:311-345) and at least one capture inside the run
    assert np.abs(Z["reward"]).max() >= 5 and len(set(Z["reward"].tolist())) >= 3


def test_oracle_follows_the_reference_trajectory():
    w, b = refnn.params(SEED, SIZES)
    nhid = SIZES[1]
    off1 = SIZES[0] * SIZES[1]
    pos = refnn.sample_positions(SEED, 1, SIZES)[:64]
    worst = dict(q=0.0, y=0.0, bias=0.0, w0=0.0, w1=0.0)
    for t in range(GOOD):
        x = xo.state_repr(xo.board_from(Z["states"][t]))
        x2 = xo.state_repr(xo.board_from(Z["next_states"][t]))
        q = xo.nn_forward(SIZES, w, b, x)
        worst["q"] = max(worst["q"], np.abs(q[:96] - Z[f"ply{t}_q"]).max())
        m = xo.nn_forward(SIZES, w, b, x2).max()
        y = Z["reward"][t] if Z["done"][t] else Z["reward"][t] + GAMMA * m
        worst["y"] = max(worst["y"], abs(m - Z[f"ply{t}_maxq2_y"][0]), abs(y - Z[f"ply{t}_maxq2_y"][1]))
        tq = xo.td_target(SIZES, w, b, x, x2, int(Z["action_to"][t]), float(Z["reward"][t]), int(Z["done"][t]), GAMMA)
        assert xo.nn_backprop(SIZES, w, b, x, tq, LR, 0) == 0
        worst["bias"] = max(worst["bias"], np.abs(b[:nhid] - Z[f"ply{t}_hidden_biases"]).max(), np.abs(b[nhid:nhid + 96] - Z[f"ply{t}_out_biases"]).max())
        cols = Z[f"ply{t}_w0_cols"]
        got0 = w[:off1].reshape(SIZES[1], SIZES[0])[:16][:, cols]
        worst["w0"] = max(worst["w0"], np.abs(got0 - Z[f"ply{t}_w0"].reshape(16, len(cols))).max())
        worst["w1"] = max(worst["w1"], np.abs(w[off1 + pos] - Z[f"ply{t}_ub_w1"]).max())      # the output-layer update, comparable here
    assert all(v <= 1e-12 for v in worst.values()), worst
    if GOOD == K:
        x = xo.state_repr(xo.board_from(Z["states"][0]))
        assert np.abs(xo.nn_forward(SIZES, w, b, x)[:96] - Z["final_q_of_first_state"]).max() <= 1e-12


@pytest.mark.gpu
def test_hip_td_path_follows_the_reference_trajectory():
    """xq_dqn_td_update (packed boards, sparse output delta, as-written hidden delta, fp32 on the matrix pipe), one transition per call:
    Q(state)[0..95] before every update and the TD target of every ply against what the reference's runtime computed."""
    import cn_chess_ai_amd as xq
    assert xq._capi.device_count() > 0
    w, b = refnn.params(SEED, SIZES)
    d = xq.DQN(SIZES, LR, GAMMA, seed=1)
    d.set_params(w, b)
    nhid = SIZES[1]
    worst_q = worst_y = 0.0
    for t in range(GOOD):
        x = onehot(Z["states"][t])
        q = d.getQValues(x)
        worst_q = max(worst_q, np.abs(q[:96] - Z[f"ply{t}_q"]).max())
        qsa, y = d.td_update(Z["states"][t][None], Z["next_states"][t][None], Z["action_to"][t:t + 1], Z["reward"][t:t + 1], Z["done"][t:t + 1],
                             td_net=0, mode=0, learning_rate=LR, grad_scale=1.0)
        worst_y = max(worst_y, abs(float(y[0]) - Z[f"ply{t}_maxq2_y"][1]))
        gw, gb = d.get_params()
        assert np.abs(gb[:nhid] - Z[f"ply{t}_hidden_biases"]).max() < 2e-5 and np.abs(gb[nhid:nhid + 96] - Z[f"ply{t}_out_biases"]).max() < 2e-5
    assert worst_q < 1e-4 and worst_y < 1e-4 * max(1.0, np.abs(Z["reward"]).max()), (worst_q, worst_y)
    if GOOD == K:
        x = onehot(Z["states"][0])
        assert np.abs(d.getQValues(x)[:96] - Z["final_q_of_first_state"]).max() < 1e-4          # after 16 updates
    d.close()

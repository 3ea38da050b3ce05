"""Pins oracle/xq_oracle.c to the REAL reference rules engine (not gpu).

tests/golden/ref_trace.npz and ref_validmat.npz hold OUTPUTS of /root/reference/src/chessboard.cpp compiled
unmodified (oracle/ref/ref_driver.cpp + oracle/gen_golden.py).  Known answers quoted from SURVEY.md pin the
chessai.cpp / dqn.cu restatements that cannot be executed here.
"""
import ctypes as C
import os

import numpy as np
import pytest

import xqoracle as xo

START_RED_MOVES = ("0->9 0->18 1->20 1->18 2->22 2->18 3->13 4->13 5->13 6->26 6->22 7->26 7->24 8->17 8->26 "
                   "19->20 19->21 19->22 19->23 19->24 19->18 19->28 19->37 19->46 19->55 19->82 19->10 25->26 25->24 "
                   "25->23 25->22 25->21 25->20 25->34 25->43 25->52 25->61 25->88 25->16 27->36 29->38 31->40 33->42 "
                   "35->44")  # SURVEY.md Appendix B (reference output)


@pytest.fixture(scope="module")
def trace(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_trace.npz"))


def _board(t, i):
    return xo.board_from(t["board"][i], t["moveCount"][i], t["player"][i], t["redScore"][i], t["blackScore"][i])


def test_trace_covers_enough(trace):
    n = len(trace["moveCount"])
    assert n >= 2000                                    # >= 2k positions, both sides
    assert int(trace["over"].sum()) > 0                 # post-game-over attempts present (E14 "still works after game over")
    assert int((trace["valid"] == 0).sum()) > 50        # invalid attempts present (no state change)
    assert int((trace["captured"] > 0).sum()) > 100     # captures present
    caps = set(trace["captured"].tolist())
    assert 1 in caps or 8 in caps                       # a general was captured at least once


def test_ordered_move_lists_match_reference(trace):
    L = xo.lib()
    n = len(trace["moveCount"])
    for i in range(n):
        b = _board(trace, i)
        for colour, key in ((0, "red"), (1, "black")):
            want = trace[key][trace[key + "_off"][i]:trace[key + "_off"][i + 1]]
            got, cnt = xo.all_valid_actions(b, colour)
            assert cnt == len(want), (i, key)
            assert np.array_equal(got, want), (i, key)
        assert L.xqo_check_game_over(C.byref(b)) == trace["over"][i]
        assert L.xqo_get_winner(C.byref(b)) == trace["winner"][i]


def test_move_piece_matches_reference(trace):
    L = xo.lib()
    n = len(trace["moveCount"])
    start = xo.new_board().squares()
    checked = 0
    for i in range(n):
        b = _board(trace, i)
        fr, fc, tr, tc = (int(x) for x in trace["move"][i])
        assert L.xqo_is_valid_move(C.byref(b), fr, fc, tr, tc) == trace["valid"][i], i
        before = (b.squares(), b.moveCount, b.currentPlayer, b.redScore, b.blackScore)
        cap = L.xqo_move_piece(C.byref(b), fr, fc, tr, tc)
        assert cap == trace["captured"][i], i
        if not trace["valid"][i]:                       # invalid -> Empty piece, NO state change
            assert cap == 0
            assert np.array_equal(b.squares(), before[0])
            assert (b.moveCount, b.currentPlayer, b.redScore, b.blackScore) == before[1:]
        if i + 1 < n:
            new_game = trace["moveCount"][i + 1] == 0 and np.array_equal(trace["board"][i + 1], start) \
                and trace["over"][i]
            if not new_game:
                assert np.array_equal(b.squares(), trace["board"][i + 1]), i
                assert b.moveCount == trace["moveCount"][i + 1]
                assert b.currentPlayer == trace["player"][i + 1]
                assert b.redScore == trace["redScore"][i + 1]
                assert b.blackScore == trace["blackScore"][i + 1]
                checked += 1
    assert checked > 2000


def test_is_valid_move_matrix_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_validmat.npz"))
    for i in range(len(g["board"])):
        b = xo.board_from(g["board"][i])
        want = np.unpackbits(g["valid_bits"][i])[:8100]
        got = xo.valid_matrix(b)
        assert np.array_equal(got, want), i
        # generator set == validator set, no duplicates
        for colour in (0, 1):
            codes, cnt = xo.all_valid_actions(b, colour)
            own = np.array([(c > 0) and ((c > 7) == (colour == 1)) for c in g["board"][i]])
            want_codes = sorted(int(k) for k in np.nonzero(want)[0] if own[k // 90])
            assert sorted(int(c) for c in codes) == want_codes
            assert len(set(codes.tolist())) == cnt


def test_positions_with_more_than_64_moves_match_reference(golden_dir):
    """ref_bigmoves.npz: 160 real-reference positions in which a side has 65..77 moves (second half of the 128-entry lists)."""
    t = np.load(os.path.join(golden_dir, "ref_bigmoves.npz"))
    L = xo.lib()
    n = len(t["moveCount"])
    counts = np.maximum(np.diff(t["red_off"]), np.diff(t["black_off"]))
    assert n >= 100 and counts.min() > 64 and counts.max() >= 75
    for i in range(n):
        b = _board(t, i)
        for colour, key in ((0, "red"), (1, "black")):
            want = t[key][t[key + "_off"][i]:t[key + "_off"][i + 1]]
            got, cnt = xo.all_valid_actions(b, colour)
            assert cnt == len(want) and np.array_equal(got, want), (i, key)
        fr, fc, tr, tc = (int(x) for x in t["move"][i])
        assert L.xqo_is_valid_move(C.byref(b), fr, fc, tr, tc) == t["valid"][i]
        assert L.xqo_move_piece(C.byref(b), fr, fc, tr, tc) == t["captured"][i]


def test_public_piece_validators_match_reference(golden_dir):
    """ref_rulemat.npz: isValid{General..Soldier}Move (chessboard.h:50-56) of the real reference on 24 positions — all
    in-board (from, to) pairs per piece type plus 64 queries each with coordinates outside the board."""
    g = np.load(os.path.join(golden_dir, "ref_rulemat.npz"))
    L = xo.lib()
    for i in range(len(g["board"])):
        b = xo.board_from(g["board"][i])
        want = np.unpackbits(g["rule_bits"][i])[:7 * 8100].reshape(7, 8100)
        for k in range(0, 7 * 8100, 7):                  # every 7th entry of the table: 8100 checks per position
            t, ft = divmod(k, 8100)
            f, to = divmod(ft, 90)
            r = L.xqo_piece_rule(C.byref(b), t + 1, f // 9, f % 9, to // 9, to % 9)
            assert max(r, 0) == want[t, ft], (i, t, f, to)
        for q, res in zip(g["query"][i], g["query_result"][i]):
            assert L.xqo_piece_rule(C.byref(b), *(int(x) for x in q)) == res, (i, q)


def test_start_position_known_answers():
    b = xo.new_board()
    red, n = xo.all_valid_actions(b, 0)
    want = [int(a) * 90 + int(c) for a, c in (m.split("->") for m in START_RED_MOVES.split())]
    assert n == 44 and red.tolist() == want
    black, nb = xo.all_valid_actions(b, 1)
    assert nb == 44
    L = xo.lib()
    # start position, moveCount 9 -> 0, 10 -> -1, 199 -> -19 (int truncation toward zero)
    assert L.xqo_evaluate_board(C.byref(b), 0, 9) == 0
    assert L.xqo_evaluate_board(C.byref(b), 0, 10) == -1
    assert L.xqo_evaluate_board(C.byref(b), 1, 199) == -19
    assert L.xqo_check_game_over(C.byref(b)) == 0
    assert L.xqo_get_winner(C.byref(b)) == 0            # first general in index order is Red's (E16)
    idx = xo.state_indices(b)
    assert len(idx) == 32 and idx[0] == 0 * 14 + 4 and idx[4] == 4 * 14 + 0   # chariot at sq0, general at sq4
    s = xo.state_repr(b)
    assert s.sum() == 32 and s[89 * 14 + 7 + 4] == 1.0                         # black chariot at sq 89


def test_material_reward_truncation():
    L = xo.lib()
    b = xo.new_board()
    b.sq[0] = 0                                          # red loses a chariot: -90 for red, +90 for black
    assert L.xqo_evaluate_board(C.byref(b), 0, 0) == -90
    assert L.xqo_evaluate_board(C.byref(b), 0, 5) == -90   # -90.5 truncates toward zero
    assert L.xqo_evaluate_board(C.byref(b), 0, 10) == -91
    assert L.xqo_evaluate_board(C.byref(b), 1, 5) == 89    # 89.5 -> 89


def test_move_cap_and_winner():
    L = xo.lib()
    b = xo.new_board()
    b.moveCount = 200
    assert L.xqo_check_game_over(C.byref(b)) == 1
    assert L.xqo_get_winner(C.byref(b)) == 0            # capped game reports Red (E16, verified upstream)
    assert L.xqo_move_piece(C.byref(b), 3, 0, 4, 0) == 0 and b.moveCount == 201   # still moves after game over


def test_nn_structural_known_answers():
    nw, nb = xo.nn_counts([1260, 128, 8100])
    assert (nw, nb) == (1198080, 8228)
    assert nw * 8 + nb * 8 + 8 + 3 * 4 == 9650484        # model file size, SURVEY §5 checkpoint row
    assert xo.nn_counts([1260, 256, 256, 8100])[0] == 2461696
    assert xo.nn_counts([1260, 512, 512, 512, 8100])[0] == 5316608


def test_select_action_semantics():
    L = xo.lib()
    q = np.zeros(8100)
    q[[9, 18, 20]] = [0.5, 0.7, 0.7]
    codes = np.array([0 * 90 + 9, 0 * 90 + 18, 1 * 90 + 20, 1 * 90 + 18], dtype=np.uint16)
    pq = q.ctypes.data_as(C.POINTER(C.c_double))
    pc = codes.ctypes.data_as(C.POINTER(C.c_uint16))
    RM = 2147483647
    assert L.xqo_select_action(pq, 8100, pc, 4, RM, 0, RM, 0.1) == 1       # first strict max (0.7 at index 1)
    assert L.xqo_select_action(pq, 8100, pc, 4, 0, 7, RM, 0.1) == 7 % 4    # explore: rand() % n
    assert L.xqo_select_action(pq, 8100, pc, 0, 0, 0, RM, 0.1) == -1       # empty list (upstream throws)
    q[:] = -2.0                                                            # below every tanh output, still picks first
    assert L.xqo_select_action(pq, 8100, pc, 4, RM, 0, RM, 0.1) == 0


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert xo.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert xo.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert xo.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.skipif(not os.path.exists(xo.REF_BIN), reason="real-reference binary only exists in the authoring container")
def test_live_cross_check_against_reference_binary(tmp_path):
    import subprocess
    out = tmp_path / "t.bin"
    subprocess.check_call([xo.REF_BIN, "trace", "12345", "6", str(out)])
    rec = np.fromfile(str(out), dtype=xo.REF_RECORD)
    L = xo.lib()
    for r in rec:
        b = xo.board_from(r["board"], r["moveCount"], r["player"], r["redScore"], r["blackScore"])
        for colour, key, nkey in ((0, "red", "nRed"), (1, "black", "nBlack")):
            got, cnt = xo.all_valid_actions(b, colour)
            assert cnt == r[nkey] and np.array_equal(got, r[key][:cnt])
        assert L.xqo_move_piece(C.byref(b), int(r["fr"]), int(r["fc"]), int(r["tr"]), int(r["tc"])) == r["captured"]

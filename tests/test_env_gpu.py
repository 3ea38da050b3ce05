"""GPU parity tests of the batched rules engine (through the C ABI) — `pytest -m gpu`.

HIP path vs (a) the committed outputs of the REAL reference rules engine (tests/golden/ref_*.npz) and
(b) the CPU oracle on the same seeded inputs.  Bar: bit-exact.
"""
import ctypes as C
import os

import numpy as np
import pytest

import xqoracle as xo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def xq():
    import cn_chess_ai_amd as m
    assert m._capi.device_count() > 0, "no HIP device visible"
    return m


@pytest.fixture(scope="module")
def trace(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_trace.npz"))


def _meta(t):
    return np.stack([t["moveCount"], t["player"], t["redScore"], t["blackScore"]], axis=1).astype(np.int32)


def _ragged(t, key, i):
    return t[key][t[key + "_off"][i]:t[key + "_off"][i + 1]]


def test_start_position(xq):
    env = xq.VecEnv(5)
    boards, meta = env.get_state()
    assert np.array_equal(boards, np.tile(xq.START_BOARD, (5, 1)))
    assert np.array_equal(boards[0], xo.new_board().squares())
    assert not meta.any()
    codes, counts = env.legal_moves(-1)
    want, n = xo.all_valid_actions(xo.new_board(), 0)
    assert (counts == 44).all() and np.array_equal(codes[3, :44], want)
    assert not codes[:, 44:].any()
    env.close()


def test_legal_moves_match_reference_trace(xq, trace):
    n = len(trace["moveCount"])
    env = xq.VecEnv(n)
    env.set_state(trace["board"], _meta(trace))
    for colour, key in ((0, "red"), (1, "black")):
        codes, counts = env.legal_moves(colour)
        want_counts = np.diff(trace[key + "_off"])
        assert np.array_equal(counts, want_counts)
        for i in range(n):
            assert np.array_equal(codes[i, :counts[i]], _ragged(trace, key, i)), (key, i)
    codes, counts = env.legal_moves(-1)     # side to move per game
    for i in range(n):
        key = "red" if trace["player"][i] == 0 else "black"
        assert np.array_equal(codes[i, :counts[i]], _ragged(trace, key, i)), i
    assert np.array_equal(env.get_winner(), trace["winner"])          # getWinner() of the real reference, every record
    env.close()


def test_valid_matrix_matches_reference(xq, golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_validmat.npz"))
    n = len(g["board"])
    env = xq.VecEnv(n)
    env.set_state(g["board"])
    for i in range(n):
        want = np.unpackbits(g["valid_bits"][i])[:8100]
        assert np.array_equal(env.valid_matrix(i), want), i
    env.close()


def test_positions_with_more_than_64_moves(xq, golden_dir):
    """The `lane + 64 < n_moves` half of move generation, compaction and select on real-reference positions with 65..77 moves."""
    t = np.load(os.path.join(golden_dir, "ref_bigmoves.npz"))
    n = len(t["moveCount"])
    env = xq.VecEnv(n)
    env.set_state(t["board"], _meta(t))
    for colour, key in ((0, "red"), (1, "black")):
        codes, counts = env.legal_moves(colour)
        assert np.array_equal(counts, np.diff(t[key + "_off"]))
        for i in range(n):
            assert np.array_equal(codes[i, :counts[i]], _ragged(t, key, i)), (key, i)
    # greedy select over the long lists: Q rows that put the maximum on a move in the second half of the list (index >= 64)
    codes, counts = env.legal_moves(-1)
    q = np.full((n, 90), -0.5, dtype=np.float32)
    want = np.zeros(n, dtype=np.int64)
    for i in range(n):
        k = int(counts[i]) - 1 - (i % 5) if counts[i] > 64 else int(counts[i]) - 1
        to = int(codes[i, k]) % 90
        q[i, to] = 0.25 + 0.001 * i
        want[i] = next(int(c) for c in codes[i, :counts[i]] if int(c) % 90 == to)     # first move onto that square wins
    res = env.selfplay_step(q, 0.0)
    assert np.array_equal(res["action"], want) and np.array_equal(res["n_moves"], counts)
    mv = t["move"].astype(np.int64)
    env.set_state(t["board"], _meta(t))
    res = env.step(((mv[:, 0] * 9 + mv[:, 1]) * 90 + mv[:, 2] * 9 + mv[:, 3]).astype(np.int32), auto_reset=False)
    assert np.array_equal(res["valid"], t["valid"]) and np.array_equal(res["captured"], t["captured"])
    env.close()


def test_public_piece_validators_match_reference(xq, golden_dir):
    """xq_env_rule_matrix / xq_env_rule_query == the real reference's isValid{General..Soldier}Move (chessboard.h:50-56)."""
    g = np.load(os.path.join(golden_dir, "ref_rulemat.npz"))
    n = len(g["board"])
    env = xq.VecEnv(n)
    env.set_state(g["board"])
    for i in range(n):
        want = np.unpackbits(g["rule_bits"][i])[:7 * 8100].reshape(7, 8100)
        assert np.array_equal(env.rule_matrix(i), want), i
        for q, res in zip(g["query"][i][:16], g["query_result"][i][:16]):
            assert env.rule_query(i, *(int(x) for x in q)) == bool(res), (i, q)
    with pytest.raises(xq._capi.XqError) as e:
        env.rule_query(0, 5, 3, 3, 3, 3)                 # from == to on a chariot: loop overflow upstream
    assert e.value.code == 5
    env.close()


def test_probing_a_finished_game_fabricates_nothing(xq):
    """A rejected move on an already finished board reports terminated/winner but adds no episode, no win, no reset."""
    env = xq.VecEnv(2)
    boards = np.zeros((2, 90), dtype=np.uint8)
    boards[:, 4] = 1                                       # lone red general: checkGameOver() is true
    env.set_state(boards, np.array([[30, 1, 5, 7], [30, 1, 5, 7]], dtype=np.int32))
    for _ in range(4):
        res = env.step(np.array([-1, 8100], dtype=np.int32), auto_reset=True)
        assert (res["terminated"] == 1).all() and (res["winner"] == 0).all() and (res["valid"] == 0).all()
    b2, m2 = env.get_state()
    assert np.array_equal(b2, boards) and (m2 == [30, 1, 5, 7]).all()
    c = env.counters()
    assert c["episodes"] == 0 and c["red_wins"] == 0 and len(env.drain_episodes()[0]) == 0
    res = env.step(np.array([4 * 90 + 13, -1], dtype=np.int32), auto_reset=True)    # a VALID move on a finished board still plays
    assert res["valid"][0] == 1 and res["terminated"][0] == 1
    c = env.counters()
    assert c["episodes"] == 1 and c["red_wins"] == 1 and len(env.drain_episodes()[0]) == 1
    b3, m3 = env.get_state()
    assert np.array_equal(b3[0], xq.START_BOARD) and not m3[0].any() and np.array_equal(b3[1], boards[1])
    env.close()


def test_step_matches_reference_trace(xq, trace):
    """movePiece with the reference's recorded attempts (valid, invalid, out-of-turn, out-of-board, after game over)."""
    n = len(trace["moveCount"])
    L = xo.lib()
    env = xq.VecEnv(n)
    env.set_state(trace["board"], _meta(trace))
    mv = trace["move"].astype(np.int64)
    inb = ((mv >= 0).all(axis=1)) & (mv[:, 0] < 10) & (mv[:, 2] < 10) & (mv[:, 1] < 9) & (mv[:, 3] < 9)
    actions = np.where(inb, (mv[:, 0] * 9 + mv[:, 1]) * 90 + mv[:, 2] * 9 + mv[:, 3], -1).astype(np.int32)
    res = env.step(actions, auto_reset=False)
    boards, meta = env.get_state()
    assert np.array_equal(res["valid"], trace["valid"])
    assert np.array_equal(res["captured"], trace["captured"])
    for i in range(n):
        b = xo.board_from(trace["board"][i], *(_meta(trace)[i]))
        mover = b.currentPlayer
        L.xqo_move_piece(C.byref(b), *(int(x) for x in trace["move"][i]))
        assert np.array_equal(boards[i], b.squares()), i
        assert tuple(meta[i]) == (b.moveCount, b.currentPlayer, b.redScore, b.blackScore), i
        over = L.xqo_check_game_over(C.byref(b))
        assert res["reward"][i] == L.xqo_evaluate_board(C.byref(b), mover, b.moveCount), i
        assert res["terminated"][i] == over
        assert res["done"][i] == int(over or b.moveCount + 1 >= 200)
        assert res["winner"][i] == (L.xqo_get_winner(C.byref(b)) if over else 2)
        assert (res["move_count"][i], res["red_score"][i], res["black_score"][i]) == (b.moveCount, b.redScore, b.blackScore)
    # consecutive records of the reference trace chain: state after move i == snapshot i+1
    start = xo.new_board().squares()
    for i in range(n - 1):
        new_game = trace["moveCount"][i + 1] == 0 and np.array_equal(trace["board"][i + 1], start) and trace["over"][i]
        if not new_game:
            assert np.array_equal(boards[i], trace["board"][i + 1]), i
    env.close()


def test_step_rejects_out_of_range_actions_and_auto_resets(xq):
    env = xq.VecEnv(4)
    res = env.step(np.array([-5, 8100, 99999, 27 * 90 + 36], dtype=np.int32))
    assert res["valid"].tolist() == [0, 0, 0, 1]
    boards, meta = env.get_state()
    assert np.array_equal(boards[:3], np.tile(xq.START_BOARD, (3, 1))) and not meta[:3].any()
    assert meta[3].tolist() == [1, 1, 0, 0]
    # a game at the move cap is over -> auto reset on the next step call
    env.set_state(np.tile(xq.START_BOARD, (4, 1)), np.array([[199, 1, 30, 40]] * 4, dtype=np.int32))
    res = env.step(np.array([-1, 62 * 90 + 53, 62 * 90 + 53, 62 * 90 + 53], dtype=np.int32))
    assert res["terminated"].tolist() == [0, 1, 1, 1] and res["done"].tolist() == [1, 1, 1, 1]
    assert res["winner"].tolist() == [2, 0, 0, 0]                  # capped game reports Red (E16)
    boards, meta = env.get_state()
    assert meta[0].tolist() == [199, 1, 30, 40] and not meta[1:].any()
    assert np.array_equal(boards[1], xq.START_BOARD)
    env.close()


def test_reward_truncation_all_move_counts(xq):
    """evaluateBoard's `score -= moveCount*0.1` on an int (chessai.cpp:343) for every moveCount and many materials."""
    L = xo.lib()
    rng = np.random.default_rng(7)
    n = 200 * 6
    boards = np.tile(xo.new_board().squares(), (n, 1))
    meta = np.zeros((n, 4), dtype=np.int32)
    for i in range(n):
        mc = i % 200
        kill = rng.choice(90, size=rng.integers(0, 12), replace=False)     # random material imbalance
        boards[i, kill] = np.where(np.isin(boards[i, kill], (1, 8)), boards[i, kill], 0)   # keep generals
        meta[i] = (mc, rng.integers(0, 2), 0, 0)
    env = xq.VecEnv(n)
    env.set_state(boards, meta)
    res = env.step(np.full(n, -1, dtype=np.int32), auto_reset=False)   # invalid action: state untouched, reward still evaluated
    for i in range(n):
        b = xo.board_from(boards[i], *meta[i])
        assert res["reward"][i] == L.xqo_evaluate_board(C.byref(b), int(meta[i, 1]), int(meta[i, 0])), i
    env.close()


@pytest.mark.parametrize("policy", ["random", "q"])
def test_selfplay_matches_oracle(xq, policy):
    """Whole plies on device vs the oracle's restatement of the loop body, same Philox streams, 400 plies x 96 games."""
    n, steps, seed, first = 96, 400, 0x5EED, 1000
    env = xq.VecEnv(n, seed=seed, first_game_id=first)
    boards = [xo.new_board() for _ in range(n)]
    plies = np.zeros(n, dtype=np.int64)
    rng = np.random.default_rng(3)
    eps = 0.1
    n_term = 0
    for t in range(steps):
        q = None
        if policy == "q":
            q = np.tanh(rng.normal(size=(n, 90))).astype(np.float32)
            if t % 7 == 0:
                q[:, :] = q[:, :1]                      # all equal: first-max tie-break must pick validActions[0]
        res = env.selfplay_step(q, eps)
        for g in range(n):
            o = xo.selfplay_step(boards[g], None if q is None else q[g], seed, first + g, int(plies[g]),
                                 xo.eps_to_u32(eps))
            r = res[g]
            assert r["action"] == o.action_code, (t, g)
            assert r["n_moves"] == o.n_moves and r["reward"] == o.reward
            assert (r["done"], r["terminated"], r["winner"]) == (o.done, o.terminated, o.winner)
            assert (r["move_count"], r["red_score"], r["black_score"]) == (o.moveCount, o.redScore, o.blackScore)
            if policy == "q":
                assert r["explored"] == o.explored
            if o.action_code >= 0:
                plies[g] += 1
            n_term += int(o.terminated)
    got, meta = env.get_state()
    for g in range(n):
        b = boards[g]
        assert np.array_equal(got[g], b.squares())
        assert tuple(meta[g]) == (b.moveCount, b.currentPlayer, b.redScore, b.blackScore)
    rec, total = env.drain_episodes()
    assert total == n_term == len(rec) and n_term > 50
    c = env.counters()
    assert c["plies"] == plies.sum() and c["episodes"] == n_term
    assert c["red_wins"] + c["black_wins"] == n_term
    env.close()


def test_selfplay_full_size_invariants(xq):
    """BASELINE config size (8192 games): size-independent properties over 450 plies of random play."""
    n, steps = 8192, 450
    env = xq.VecEnv(n, seed=1234)
    tot_moves, tot_plies, term = 0, 0, 0
    for t in range(steps):
        res = env.selfplay_step(None)
        assert (res["action"] >= 0).all() or (res["n_moves"][res["action"] < 0] == 0).all()
        tot_moves += int(res["n_moves"].sum()); tot_plies += n
        term += int(res["terminated"].sum())
        assert (res["done"] >= res["terminated"]).all()
        assert (res["move_count"] <= 200).all()
    boards, meta = env.get_state()
    assert (boards <= 14).all()
    for code, cap in ((1, 1), (8, 1), (2, 2), (9, 2), (3, 2), (10, 2), (4, 2), (11, 2), (5, 2), (12, 2), (6, 2), (13, 2),
                      (7, 5), (14, 5)):
        assert ((boards == code).sum(axis=1) <= cap).all()          # material never grows
    assert ((boards == 1).sum(axis=1) == 1).all() and ((boards == 8).sum(axis=1) == 1).all()   # live games keep both generals
    c = env.counters()
    assert c["plies"] == tot_plies and c["episodes"] == term
    rec, total = env.drain_episodes(1 << 20)
    assert total == term
    mean_moves = tot_moves / tot_plies
    assert 30 < mean_moves < 46            # reference workload: 38.3 legal moves/ply
    mean_len = tot_plies / max(term, 1)
    assert 120 < mean_len < 190            # reference: 153 plies/game
    capped = float((rec["move_count"] >= 200).mean())
    assert 0.3 < capped < 0.6              # reference: 45 % of games hit the 200-ply cap
    env.close()


def test_replay_ring_receives_transitions(xq):
    n = 64
    env = xq.VecEnv(n, seed=9)
    rp = xq.ReplayBuffer(4 * n, seed=5)
    prev, _ = env.get_state()
    env.selfplay_step_dev(replay=rp)
    nxt, _ = env.get_state()
    assert rp.stats() == (n, 4 * n, n)
    for slot in (0, 17, 63):
        b, a, r, d, nb = rp.get(slot)
        assert np.array_equal(b, prev[slot]) and np.array_equal(nb, nxt[slot]) and d == 0 and 0 <= a < 90
        assert (b != nb).sum() in (1, 2)
    for _ in range(5):
        env.selfplay_step_dev(replay=rp)
    size, cap, tot = rp.stats()
    assert size == cap == 4 * n and tot == 6 * n
    slots = rp.sample(1000)
    assert slots.min() >= 0 and slots.max() < cap and len(np.unique(slots)) > 100
    # the sampler is the documented Philox stream
    want = [xo.philox([i, 0, 0, 1], [5, 0])[0] % cap for i in range(10)]
    assert slots[:10].tolist() == want
    rp.push(prev[:3], [5, 6, 7], [1.5, -2.0, 0.0], [0, 1, 0], nxt[:3])
    env.close(); rp.close()


def test_side_without_any_action_ends_the_episode(xq):
    """chessai.cpp:100-103: an empty action list breaks the loop.  Black (to move): general hemmed in by its own horses,
    every horse leg blocked by a red piece -> no pseudo-legal move at all."""
    b = np.zeros(90, dtype=np.uint8)
    sq = lambda r, c: r * 9 + c
    b[sq(0, 4)] = 1                                   # red general
    b[sq(9, 4)] = 8                                   # black general
    for r, c in ((9, 3), (9, 5), (8, 4)):
        b[sq(r, c)] = 11                              # black horses
    for r, c in ((9, 2), (8, 3), (9, 6), (8, 5), (7, 4)):
        b[sq(r, c)] = 7                               # red soldiers on every leg square
    ob = xo.board_from(b, 17, 1, 30, 40)
    codes, n = xo.all_valid_actions(ob, 1)
    assert n == 0
    env = xq.VecEnv(2, seed=3)
    env.set_state(np.stack([b, b]), np.array([[17, 1, 30, 40], [17, 0, 30, 40]], dtype=np.int32))
    codes, counts = env.legal_moves(-1)
    assert counts[0] == 0 and counts[1] > 0
    res = env.selfplay_step(None)
    o = xo.selfplay_step(ob, None, 3, 0, 0, 0)
    assert (res[0]["action"], res[0]["n_moves"], res[0]["terminated"], res[0]["done"], res[0]["winner"]) == (-1, 0, 1, 1, 0)
    assert (o.action_code, o.n_moves, o.terminated, o.done, o.winner) == (-1, 0, 1, 1, 0)
    boards, meta = env.get_state()
    assert np.array_equal(boards[0], xq.START_BOARD) and not meta[0].any()       # auto-reset
    assert res[1]["action"] >= 0 and res[1]["terminated"] == 0
    rec, total = env.drain_episodes()
    assert total == 1 and rec[0]["reserved"] == 1 and rec[0]["move_count"] == 17
    env.close()


def test_replay_ring_wraps_and_skips_empty_transitions(xq):
    n, cap = 48, 100                                   # capacity not a multiple of n: the write position wraps mid-batch
    env = xq.VecEnv(n, seed=21)
    rp = xq.ReplayBuffer(cap, seed=1)
    snaps = []
    for t in range(5):
        prev, _ = env.get_state()
        env.selfplay_step_dev(replay=rp)
        nxt, _ = env.get_state()
        snaps.append((prev, nxt))
    size, capacity, total = rp.stats()
    assert (size, capacity, total) == (cap, cap, 5 * n)
    # transition g of step t sits in slot (t*n + g) % cap unless a later step overwrote it
    for t, g in ((4, 0), (4, 47), (3, 10), (2, 40)):
        slot = (t * n + g) % cap
        later = [(tt, gg) for tt in range(t + 1, 5) for gg in range(n) if (tt * n + gg) % cap == slot]
        if later:
            continue
        bd, a, r, d, nb = rp.get(slot)
        assert np.array_equal(bd, snaps[t][0][g]) and 0 <= a < 90
        if not d:
            assert np.array_equal(nb, snaps[t][1][g])
    env.close(); rp.close()


def test_game_id_sharding_at_config5_size(xq):
    """131 072 games in one env (BASELINE configs[4] size) vs small envs created with first_game_id offsets: a game's
    trajectory depends only on (seed, game id, ply), so any shard of the big batch must equal the same ids stepped alone —
    the property the multi-GPU sharding (cn_chess_ai_amd/dist.py::shard_games) relies on."""
    n_big, plies = 131072, 60
    big = xq.VecEnv(n_big, seed=0xC0FFEE)
    for _ in range(plies):
        big.selfplay_step_dev()
    bb, bm = big.get_state()
    cb = big.counters()
    assert cb["plies"] == n_big * plies
    for first in (0, 8192 * 7 + 5, n_big - 64):
        small = xq.VecEnv(64, seed=0xC0FFEE, first_game_id=first)
        for _ in range(plies):
            small.selfplay_step_dev()
        sb, sm = small.get_state()
        assert np.array_equal(sb, bb[first:first + 64]) and np.array_equal(sm, bm[first:first + 64])
        small.close()
    big.close()

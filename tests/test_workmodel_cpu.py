"""bench.py's per-kernel prices (cn_chess_ai_amd/workmodel.py) against the figures DESIGN.md section 5 states (not gpu: plain arithmetic)."""
import pytest

from cn_chess_ai_amd import workmodel as wm


def test_headline_configuration_figures():
    w = wm.step_work((1260, 256, 256, 8100), 8192, 8192, plies=1, screened=True, derive=True)
    # the screening pass: 2 x 8100 x 8192 x 256 = 33.97 GFLOP on the bf16 pipe, ~25 MB of operands + partial arrays
    assert abs(w["gemm_qmax_screen"]["flops"] - 2 * 8100 * 8192 * 256) < 1 and w["gemm_qmax_screen"]["peak"] == 2500.0
    assert 24e6 < w["gemm_qmax_screen"]["hbm_bytes"] < 27e6
    # the grouped hidden product of the two forward chains: 2 x 2 x 8192 x 256 x 256 FLOP on the fp32 pipe
    assert w["gemm_hidden_fwd"]["flops"] == 2.0 * 2 * 8192 * 256 * 256 and w["gemm_hidden_fwd"]["peak"] == 157.3
    # env kernel: 593 B per game and ply (SURVEY section 8d + the Q row)
    assert wm.ENV_BYTES_PER_GAME == 593 and w["env_selfplay_step"]["hbm_bytes"] == 593.0 * 8192
    # fused launch 2: delta_0 8 MB + boards + 8 chunks of 1260 x 256 partial sums + the 256 x 256 x 8192 weight-gradient product
    assert 40e6 < w["td_tail_l0"]["hbm_bytes"] < 50e6 and w["td_tail_l0"]["bound"] == "hbm"
    assert "td_target_delta" not in w                      # the TD target rides in the refine kernel at a 256-wide last hidden layer
    for k, v in w.items():
        assert v["hbm_bytes"] >= 0 and v["flops"] >= 0 and v["bound"] in ("hbm", "mfma"), k
    ach, frac = wm.price(w["td_tail_l0"], 44.7)
    assert abs(frac - ach / 8000.0) < 1e-12 and 0.1 < frac < 0.15


def test_other_configurations_have_every_kernel_of_their_step():
    w4 = wm.step_work((1260, 512, 512, 512, 8100), 8192, 8192, plies=4, screened=True)
    assert {"td_tail_deltas", "td_tail_l0", "td_target_delta", "gemm_hidden_fwd@select", "env_selfplay_step"} <= set(w4)
    assert w4["gemm_hidden_fwd"]["flops"] == 2.0 * 2 * 8192 * 512 * 512          # per launch (two launches per step)
    w5 = wm.step_work((1260, 512, 512, 512, 8100), 16384, 16384, bf16=True, bf16_bwd=True, td="double", screened=False, prioritized=True)
    assert w5["gemm_qmax_rowmax"]["peak"] == 2500.0 and "gemm_qmax_screen" not in w5
    assert wm.pick_splits(256, 256, 8192) == 32 and wm.pick_splits(512, 512, 8192) == 8

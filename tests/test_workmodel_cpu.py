"""bench.py's per-kernel prices (cn_chess_ai_amd/workmodel.py) against the figures DESIGN.md section 5 states (not gpu: plain arithmetic)."""
import pytest

from cn_chess_ai_amd import workmodel as wm


def test_headline_configuration_figures():
    w = wm.step_work((1260, 256, 256, 8100), 8192, 8192, plies=1, screened=True, derive=True, l0_mfma=False)
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


def test_matrix_pipe_layer0_gradient_figures():
    """the library default since round 5: launch 2 reads the three bf16 planes of delta_0 and the selector words instead of delta rows"""
    w = wm.step_work((1260, 256, 256, 8100), 8192, 8192)
    seg = wm.step_work((1260, 256, 256, 8100), 8192, 8192, l0_mfma=False)
    assert w["td_tail_l0"]["mfma_bf16_flops"] == 2.0 * 80 * 16 * 256 * 8192 * 3            # 16.1 GFLOP on the bf16 pipe
    assert 55e6 < w["td_tail_l0"]["hbm_bytes"] < 65e6 and w["td_tail_deltas"]["hbm_bytes"] > seg["td_tail_deltas"]["hbm_bytes"]
    assert wm.rocprof_kernel("td_tail_l0", (1260, 256, 256, 8100), 8192) == "td_tail_kernel<30u, true>("
    assert wm.rocprof_kernel("td_tail_l0", (1260, 256, 256, 8100), 8192, l0_mfma=False) == "td_tail_kernel<30u, false>("
    assert wm.rocprof_kernel("td_tail_deltas", (1260, 256, 256, 8100), 8192) == "td_tail_kernel<39u, false>("
    assert wm.match_kernel("void xq::td_tail_kernel<30u, true>(xq::TailArgs)", "td_tail_kernel<30u, true>(")
    assert not wm.match_kernel("void xq::td_tail_kernel<30u, false>(xq::TailArgs)", "td_tail_kernel<30u, true>(")


CONFIGS = {2: dict(layers=(1260, 256, 256, 8100), minibatch=8192, n_games=8192, plies=1),
           4: dict(layers=(1260, 512, 512, 512, 8100), minibatch=8192, n_games=8192, plies=4),
           5: dict(layers=(1260, 512, 512, 512, 8100), minibatch=16384, n_games=16384, plies=1, bf16=True, bf16_bwd=True, td="double",
                   screened=False, prioritized=True)}


@pytest.mark.parametrize("config", sorted(CONFIGS))
def test_newest_profile_set_names_every_kernel_of_the_step(config):
    """VERDICT r4 #4 / Next #5: every offline figure of a bench line comes from ONE profile set — the newest rNN_x tag under profiles/
    with both a kernel-stats CSV and a PMC summary for the configuration — and that set must hold exactly one row for every kernel
    instance the work model names (a renamed kernel or a changed template argument goes red here, not silently stale in the line)."""
    import csv
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    best = wm.newest_profile_set(root, config)
    assert best, "no complete profile set for config %d under profiles/" % config
    tag, stats_csv, pmc_json = best
    if tag < "r05":
        pytest.skip("newest complete set for config %d is %s: made before the kernel names of round 5" % (config, tag))
    c = dict(CONFIGS[config])
    layers, mb, ng, plies = c.pop("layers"), c.pop("minibatch"), c.pop("n_games"), c.pop("plies")
    work = wm.step_work(layers, mb, ng, plies, **c)
    stats_names = [r["Name"] for r in csv.DictReader(open(stats_csv))]
    pmc_names = list(json.load(open(pmc_json)))
    rare = {"target_sync_copy"}                          # a runtime copy kernel, not one of ours: present, but shared with other copies
    for bracket in work:
        want = wm.rocprof_kernel(bracket, layers, mb, td=c.get("td", "online"), bf16=c.get("bf16", False))
        if want is None or bracket in rare:
            continue
        for inst in (want if isinstance(want, list) else [want]):
            for where, names in (("kernel stats", stats_names), ("PMC summary", pmc_names)):
                hits = [x for x in names if wm.match_kernel(x, inst)]
                assert len(hits) == 1, "%s of %s: %d rows match %r (bracket %s)" % (where, tag, len(hits), inst, bracket)

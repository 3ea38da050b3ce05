#!/usr/bin/env python3
"""How many delta rows does the layer-0 gradient have to touch?  (Next #1.)

gW0^T[(sq, piece)] = sum of delta_0 over the samples with `piece` on `sq` (dqn.cu:310-319 on the one-hot of chessai.cpp:268-289).
The segmented-sum kernel reads one row per (sample, occupied square): R_now = sum_sq (N - empty(sq)).  Because the 15 classes of a
square (14 piece planes + empty) partition the samples, the plane of the MOST COMMON class of a square could be had as
`column sum - the other planes - the rows of the samples where the square is empty`: R_min = sum_sq (N - max_class(sq)).
This tool measures both on the live boards of the bench's training loop as it runs (the ring holds the last 128 plies of these games).
usage: python tools/l0_class_hist.py [--updates 20000]"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cn_chess_ai_amd as xq
from cn_chess_ai_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--updates", type=int, default=20000)
ap.add_argument("--out", default="")
a = ap.parse_args()
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
cfg = xq.TrainerConfig(n_games=8192, layer_sizes=(1260, 256, 256, 8100), replay_capacity=1 << 20, minibatch=8192,
                       td_net=_capi.TD_ONLINE_NET, overlap_collect=1, mean_gradient=1, target_sync_interval=10)
t = xq.Trainer(cfg, stream=C.c_void_p(ts.cuda_stream))
t.dqn.set_qmax_mode(_capi.QMAX_SCREENED)
t.dqn.set_fused_apply(True)
t.synchronize()
START = t.env.get_state()[0][0].copy()               # the env starts from chessboard.cpp:8-29's position
t.random_plies(300)
for _ in range(128):
    t.collect()


def census(tag, updates):
    t.synchronize()
    boards, _ = t.env.get_state()                     # [n][90] piece codes 0..14
    n = boards.shape[0]
    cls = np.stack([(boards == c).sum(0) for c in range(15)])        # [15][90]
    occupied = n - cls[0]
    best = cls.max(0)
    r_now, r_min = int(occupied.sum()), int((n - best).sum())
    home_keep = float(np.mean([cls[1:, s].max() / n for s in range(90) if cls[1:, s].max() > 0.3 * n])) if (cls[1:].max(0) > 0.3 * n).any() else 0.0
    rec = {"at": tag, "updates": updates, "boards": n, "pieces_per_board": r_now / n, "rows_segmented": r_now, "rows_majority_complement": r_min,
           "ratio": r_min / max(r_now, 1), "squares_where_a_piece_is_the_majority": int((cls[1:].max(0) > cls[0]).sum()),
           "mean_share_of_the_commonest_piece_on_squares_above_30pct": home_keep}
    # rows that are the START position's piece on its own square (what a gather kernel could keep in LDS: 32 rows)
    if START is not None:
        hot = int(((boards == START[None, :]) & (boards != 0)).sum())
        rec["rows_of_the_start_position"] = hot
        rec["share_of_start_rows"] = hot / max(r_now, 1)
    occ = np.sort(occupied.astype(np.float64) / n)[::-1]
    rec["occupancy_sorted_top"] = [round(float(x), 3) for x in occ[:56]]
    rec["rows_outside_the_top_k_squares"] = {str(k): int(round(float(occ[k:].sum() * n))) for k in (24, 32, 40, 48, 56, 64, 72)}
    print(json.dumps(rec), flush=True)
    return rec


recs = [census("ring filled by random play + 128 plies of the fresh net", 0)]
done = 0
for upto in (100, 1000, 5000, a.updates):
    if upto > a.updates:
        break
    for _ in range(upto - done):
        t.learn_grads(); t.collect(); t.learn_apply(1)
    done = upto
    recs.append(census("training loop", done))
if a.out:
    with open(a.out, "w") as f:
        json.dump(recs, f, indent=1)
t.close()

"""Report half of tools/detector_census.py: the oracle's contour calls (oracle/ is test infrastructure, hence this file is under
tests/) on the poses a stamped launch of the rollout-step kernel looked at — per workgroup the candidates, rows, contour points,
pieces, vertex counts — held against that workgroup's barrier P -> barrier R time.  python tests/detector_census_report.py IN.npz"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def report(path):
    from oracle import oracle
    z = np.load(path)
    s = z["stamps"].astype(np.float64) / 100.0
    pose, done = z["pose"], z["done"].astype(bool)
    n = pose.shape[0]
    pr = s[:, 12] - s[:, 10]                               # P passed -> R passed, wave 0
    w0 = s[:, 11] - s[:, 10]                               # wave 0's own groups
    w4 = s[:, 15] - s[:, 10]                               # wave 4's
    wg_total = s[:, 3] - s[:, 0]
    rows = np.zeros(n, int); pts = np.zeros(n, int); pieces = np.zeros(n, int); approx = np.zeros(n, int); vis = np.zeros(n, int)
    for i in range(n):
        if done[i]:
            continue
        v, c, area, ratio, spans, verts, length, apx, more = oracle.camera_pixel_debug(pose[i, 0:3], pose[i, 3:7], full=True)
        if not spans:
            continue
        js = sorted(spans)
        # the kernels' border pre-check drops blobs that touch the border before the pool sees them
        if js[0] == 0 or js[-1] == 127 or min(a for a, b in spans.values()) == 0 or max(b for a, b in spans.values()) == 127:
            continue
        rows[i] = js[-1] - js[0] + 1
        pts[i] = more["points"]
        pieces[i] = more["contours"]
        approx[i] = more["approx"]
        vis[i] = v
    cand = rows > 0
    G = n // 256
    per = lambda a: a.reshape(G, 256)
    ncand = per(cand).sum(1)
    out = {"drones": n, "candidates_frac": round(float(cand.mean()), 4), "done_frac": round(float(done.mean()), 4),
           "rows_pct_10_50_90_max": [int(np.percentile(rows[cand], p)) for p in (10, 50, 90, 100)],
           "points_pct_10_50_90_max": [int(np.percentile(pts[cand], p)) for p in (10, 50, 90, 100)],
           "several_pieces_frac_of_candidates": round(float((pieces[cand] > 1).mean()), 4),
           "approx_counts": {int(k): int((approx[cand] == k).sum()) for k in np.unique(approx[cand])},
           "visible_frac_of_candidates": round(float(vis[cand].mean()), 4),
           "P_to_R_us_pct_10_50_90_max": [round(float(np.percentile(pr, p)), 2) for p in (10, 50, 90, 100)],
           "workgroup_us_pct_50_90_max": [round(float(np.percentile(wg_total, p)), 2) for p in (50, 90, 100)],
           "candidates_per_workgroup_pct_10_50_90_max": [int(np.percentile(ncand, p)) for p in (10, 50, 90, 100)]}
    # what explains a slow workgroup: linear fit of P -> R on a few per-workgroup counts
    feats = {"candidates": ncand, "rows_sum": per(rows).sum(1), "points_sum": per(pts).sum(1), "points_max": per(pts).max(1),
             "rows_max": per(rows).max(1), "several_pieces": per(pieces > 1).sum(1), "not_four": per(cand & (approx != 4)).sum(1),
             "octets_over_64": np.maximum(ncand + per(rows > 16).sum(1) - 64, 0)}
    out["corr_with_P_to_R"] = {k: round(float(np.corrcoef(v, pr)[0, 1]), 3) if v.std() > 0 else None for k, v in feats.items()}
    A = np.stack([np.ones(G)] + [feats[k].astype(float) for k in ("candidates", "points_max", "several_pieces", "not_four")], 1)
    coef, *_ = np.linalg.lstsq(A, pr, rcond=None)
    out["fit_us"] = dict(zip(["const", "per_candidate", "per_point_of_max", "per_several_pieces", "per_not_four"], [round(float(c), 3) for c in coef]))
    out["fit_residual_us"] = round(float(np.std(pr - A @ coef)), 3)
    if s.shape[1] == 64:          # the probe build: per-wave times of the contour phase, first and second (warm) pass
        past_p, sched1, end1 = s[:, 40:48], s[:, 48:56], s[:, 16:24]
        start2, sched2, end2 = s[:, 24:32], s[:, 56:64], s[:, 32:40]
        pct = lambda a: [round(float(np.percentile(a, p)), 2) for p in (10, 50, 90, 100)]
        out["probe"] = {
            "per_wave_first_pass_us_median": [round(float(np.median(end1[:, w] - past_p[:, w])), 2) for w in range(8)],
            "per_wave_second_pass_us_median": [round(float(np.median(end2[:, w] - start2[:, w])), 2) for w in range(8)],
            "schedule_first_pass_us_pct": pct(sched1 - past_p), "schedule_second_pass_us_pct": pct(sched2 - start2),
            "slowest_wave_first_pass_us_pct": pct((end1 - past_p).max(1)), "slowest_wave_second_pass_us_pct": pct((end2 - start2).max(1)),
            "which_wave_is_slowest_first_pass": np.bincount((end1 - past_p).argmax(1), minlength=8).tolist(),
            "barrier_P_skew_us_pct": pct(past_p.max(1) - past_p.min(1)),
        }
        # per wave: its octets' blobs (the schedule of fpv_pixel_tile.h replayed) against its time
        wave_pts, wave_rows, wave_n = np.zeros((G, 8)), np.zeros((G, 8)), np.zeros((G, 8))
        for g in range(G):
            r = rows[g * 256:(g + 1) * 256]; p_ = pts[g * 256:(g + 1) * 256]
            nc = int((r > 0).sum()); room = 64 - nc
            big = next((b for b in (16, 20, 24, 28) if int((r > b).sum()) <= room), 36)
            order = [i for i in range(256) if r[i] > big]
            octs = [i for i in order for _ in (0, 1)] + [i for i in range(256) if 0 < r[i] <= big]
            for o, i in enumerate(octs):
                w = (o // 8) % 8
                wave_pts[g, w] = max(wave_pts[g, w], p_[i] / (2 if r[i] > big else 1)); wave_rows[g, w] = max(wave_rows[g, w], r[i] / (2 if r[i] > big else 1))
                wave_n[g, w] += 1
        t1 = (end1 - sched1).ravel(); t2 = (end2 - sched2).ravel()
        busy = wave_n.ravel() > 0
        A2 = np.stack([np.ones(busy.sum()), wave_pts.ravel()[busy], wave_rows.ravel()[busy], wave_n.ravel()[busy]], 1)
        for nm, tt in (("first", t1), ("second", t2)):
            c2, *_ = np.linalg.lstsq(A2, tt[busy], rcond=None)
            out["probe"][f"wave_fit_{nm}_pass_us"] = dict(zip(["const", "per_point_per_lane_of_its_largest", "per_row_per_lane_of_its_largest", "per_octet"],
                                                             [round(float(c), 3) for c in c2]))
            out["probe"][f"wave_fit_{nm}_residual_us"] = round(float(np.std(tt[busy] - A2 @ c2)), 3)
        out["probe"]["idle_waves_frac"] = round(float((~busy).mean()), 3)
    slow = np.argsort(-pr)[:8]
    out["slowest_workgroups"] = [{"P_to_R": round(float(pr[g]), 2), "wave0": round(float(w0[g]), 2), "wave4": round(float(w4[g]), 2),
                                  **{k: int(v[g]) for k, v in feats.items()}} for g in slow]
    print(json.dumps(out, indent=1))



if __name__ == "__main__":
    report(sys.argv[1])

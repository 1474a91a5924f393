"""Comparison helpers shared by the parity tests (fp32 HIP path versus the fp64 oracle).

The default detector is the reference's pixel rules (docs/SPEC.md §6b, hover.py:157-222): the camera features are integer
counts on a 128 x 128 grid.  The fp32 path carries ~1e-5 px of rounding in the projected corners (and in the fp32 state
behind them), so an edge that passes that close to a pixel centre rasterises ONE pixel differently from the fp64 oracle: a
one-pixel change of the bounding box, a few px^2 of contour, one pixel of one vertex, or the visibility decision itself when
a gate sits on the boundary.  Such rows are rare — measured 8 of 65 536 drones in one step, 0.012 %, none of them a visibility
flip; 4 of 16 384 on poses built to stress the contour calls; over trajectories of 40 - 45 steps at most 0.1 % of the rows of a step
outside the trajectory tolerances, 0.013 - 0.026 % on average (profiles/r05_flip_rates.json, tools/flip_rates.py: fp32 engine
against fp64 engine; the budgets below are 4 - 8 x those) — and cannot be avoided short of computing in fp64; `pixel_flips` finds them, bounds how
many there are and how large the change is, and the callers hold every other row — and every word of the flipped rows that
does not derive from the camera features — to the 1e-5 bar.
"""
import numpy as np

FEATURE_OBS = [7, 8, 11, 13, 14]            # centre(2), area, visible, ratio (hover.py:253-267)
FEATURE_STATE = [26, 27, 28, 29, 35]        # feature memory prev_center / prev_area / prev_ratio + ep_return (takes the reward)


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))


def pixel_flips(obs, obs_ref, max_frac=0.001, tol=1e-5):
    """Rows whose camera features differ between the fp32 path and the oracle by more than `tol`.  Asserts that they are at most
    `max_frac` of the rows (and at most 2 rows in small batches) and that each is a one-pixel effect: a pixel more or less in the
    contour, the bounding box or — since the vertices are cv2.approxPolyDP's, which may pick another contour point when a span end
    moves — a vertex that moves along its edge by a few pixels (the centre is the mean of four vertices: 1 / 256 per pixel)."""
    obs, obs_ref = np.asarray(obs, np.float64), np.asarray(obs_ref, np.float64)
    bad = (np.abs(obs[:, FEATURE_OBS] - obs_ref[:, FEATURE_OBS]) > tol).any(axis=1)
    n = obs.shape[0]
    assert bad.sum() <= max(2, int(max_frac * n)), f"{bad.sum()} of {n} rows differ in the camera features"
    same_vis = bad & (obs[:, 13] == obs_ref[:, 13])
    if same_vis.any():
        assert np.abs(obs[same_vis, 11] - obs_ref[same_vis, 11]).max() <= 6 / 16384          # a few px^2 of contour
        assert np.abs(obs[same_vis, 14] - obs_ref[same_vis, 14]).max() <= 0.2                # one pixel of w or h
        assert np.abs(obs[same_vis, 7:9] - obs_ref[same_vis, 7:9]).max() <= 8 / 256 + 1e-6   # a vertex a few pixels along its edge
    return bad


def state_close_except_flips(st, st_ref, flips, tol=1e-5):
    """Every state word within tol, except the camera-feature memory and ep_return of the flipped rows."""
    ok = close(st, st_ref, tol)
    skip = np.zeros_like(ok)
    skip[np.ix_(np.where(flips)[0], FEATURE_STATE)] = True
    return ok | skip


def obs_close_except_flips(obs, obs_ref, flips, tol, scale=None):
    scale = np.ones(obs.shape[1]) if scale is None else scale
    ok = close(np.asarray(obs) * scale, np.asarray(obs_ref) * scale, tol)
    skip = np.zeros_like(ok)
    skip[np.ix_(np.where(flips)[0], FEATURE_OBS)] = True
    return ok | skip


TRAJECTORY_BAD_FRAC = 0.004      # 4 x the worst step of profiles/r05_flip_rates.json (trajectory_*: 0.00073 and 0.00098 of the rows)


def rows_close(a, b, atol=0.0, rtol=0.0, max_bad_frac=TRAJECTORY_BAD_FRAC, what=""):
    """Trajectory-level comparisons (an fp32 device trajectory against the fp64 oracle's over tens of steps): every row within
    the tolerance except a small fraction — the rows a one-pixel flip of the detector touched at this step or, through the
    feature memory (prev_center / prev_area / prev_ratio, hover.py:270-272), at the previous one.  The budget is derived from
    the measured rate (TRAJECTORY_BAD_FRAC; two rows in batches too small for a fraction)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ok = np.abs(a - b) <= atol + rtol * np.abs(b)
    ok = ok.reshape(ok.shape[0], -1).all(axis=1) if ok.ndim > 1 else ok
    bad = int((~ok).sum())
    assert bad <= max(2, int(max_bad_frac * ok.shape[0])), f"{what}: {bad} of {ok.shape[0]} rows outside the tolerance"
    return ok


def adversarial_states(oracle, n, seed):
    """Canonical [n, 36] states whose POSES stress the detector's contour calls (hover.py:187-203): a third of the drones close to
    the panel's plane (edge-on slivers, blobs in several pieces), a third anywhere in front of it (blobs of up to ~120 rows), a third
    far away (blobs of a few pixels), all aimed roughly at the panel, at rest.  Most of them are outside the flight dome: their
    step terminates, and the terminal observation — with the camera features of that pose — is what the done ledger keeps."""
    import math
    rng = np.random.Generator(np.random.Philox(seed))
    st = oracle.synthetic_states(n, seed=seed + 1)
    st[:, 7:13] = 0.0                                       # at rest: the pose after the step is the pose set here, nearly
    for i in range(n):
        mode = i % 3
        if mode == 0:
            pos = np.array([rng.uniform(4.0, 4.97), rng.uniform(-3, 3), rng.uniform(3, 9)])
        elif mode == 1:
            pos = np.array([rng.uniform(-3, 4.9), rng.uniform(-4, 4), rng.uniform(0.5, 9)])
        else:
            pos = np.array([rng.uniform(-40, 4.0), rng.uniform(-10, 10), rng.uniform(0.5, 20)])
        tgt = np.array([4.98, rng.uniform(-1.5, 1.5), rng.uniform(4.5, 7.5)])
        d = tgt - pos
        eul = np.array([rng.normal(0, 0.6), -math.atan2(d[2], math.hypot(d[0], d[1])) + math.radians(25) + rng.normal(0, 0.3),
                        math.atan2(d[1], d[0]) + rng.normal(0, 0.3)])
        st[i, 0:3] = pos
        st[i, 3:7] = oracle.quat_from_euler(eul)
        st[i, 23:26] = eul
    return st


def final_observations(obs, term, trunc, led_idx, led_obs):
    """[n, O] observations OF THE STEP: the returned row for a drone that goes on, the ledger's terminal observation for one whose
    episode ended (the returned row is then the reset observation, SB3 VecEnv semantics)."""
    out = np.array(obs, np.float64, copy=True)
    done = (np.asarray(term) | np.asarray(trunc)).astype(bool)
    assert done.sum() == len(led_idx) and set(np.where(done)[0]) == set(int(i) for i in led_idx)
    out[np.asarray(led_idx, np.int64)] = led_obs
    return out

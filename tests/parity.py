"""Comparison helpers shared by the parity tests (fp32 HIP path versus the fp64 oracle).

The default detector is the reference's pixel rules (docs/SPEC.md §6b, hover.py:157-222): the camera features are integer
counts on a 128 x 128 grid.  The fp32 path carries ~1e-5 px of rounding in the projected corners (and in the fp32 state
behind them), so an edge that passes that close to a pixel centre rasterises ONE pixel differently from the fp64 oracle: a
one-pixel change of the bounding box, a few px^2 of contour, one pixel of one vertex, or the visibility decision itself when
a gate sits on the boundary.  Such rows are rare (measured ~0.3 % of the drones that see the panel per step,
profiles/r03_pixel_flip_rate.json) and cannot be avoided short of computing in fp64; `pixel_flips` finds them, bounds how
many there are and how large the change is, and the callers hold every other row — and every word of the flipped rows that
does not derive from the camera features — to the 1e-5 bar.
"""
import numpy as np

FEATURE_OBS = [7, 8, 11, 13, 14]            # centre(2), area, visible, ratio (hover.py:253-267)
FEATURE_STATE = [26, 27, 28, 29, 35]        # feature memory prev_center / prev_area / prev_ratio + ep_return (takes the reward)


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))


def pixel_flips(obs, obs_ref, max_frac=0.01, tol=1e-5):
    """Rows whose camera features differ between the fp32 path and the oracle by more than `tol`.  Asserts that they are at most
    `max_frac` of the rows (and at most 2 rows in small batches) and that each is a one-pixel effect."""
    obs, obs_ref = np.asarray(obs, np.float64), np.asarray(obs_ref, np.float64)
    bad = (np.abs(obs[:, FEATURE_OBS] - obs_ref[:, FEATURE_OBS]) > tol).any(axis=1)
    n = obs.shape[0]
    assert bad.sum() <= max(2, int(max_frac * n)), f"{bad.sum()} of {n} rows differ in the camera features"
    same_vis = bad & (obs[:, 13] == obs_ref[:, 13])
    if same_vis.any():
        assert np.abs(obs[same_vis, 11] - obs_ref[same_vis, 11]).max() <= 6 / 16384          # a few px^2 of contour
        assert np.abs(obs[same_vis, 14] - obs_ref[same_vis, 14]).max() <= 0.2                # one pixel of w or h
        assert np.abs(obs[same_vis, 7:9] - obs_ref[same_vis, 7:9]).max() <= 2 / 128 + 1e-6   # one pixel of one vertex
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


def rows_close(a, b, atol=0.0, rtol=0.0, max_bad_frac=0.02, what=""):
    """Trajectory-level comparisons (an fp32 device trajectory against the fp64 oracle's over tens of steps): every row within
    the tolerance except a small fraction — the rows a one-pixel flip of the detector touched at this step or, through the
    feature memory (prev_center / prev_area / prev_ratio, hover.py:270-272), at the previous one."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ok = np.abs(a - b) <= atol + rtol * np.abs(b)
    ok = ok.reshape(ok.shape[0], -1).all(axis=1) if ok.ndim > 1 else ok
    bad = int((~ok).sum())
    assert bad <= max(2, int(max_bad_frac * ok.shape[0])), f"{what}: {bad} of {ok.shape[0]} rows outside the tolerance"
    return ok

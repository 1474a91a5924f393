"""Pixel detector (docs/SPEC.md §6b; simulation/hover.py:157-222) — CPU checks of the oracle's restatement.

What pins it: the reference's own constants.  hover.py:50-51 sets target_area = 0.013 and target_ratio = 1.53; from the
reset pose (hover.py:78-79: origin, level, resting on the floor) the pixel rules give a 20 x 13 bounding box
(20 / 13 = 1.538...) and a contour of 207 px^2 (207 / 16384 = 0.01263), i.e. exactly those constants at the precision
they are written with — for pixel centres at (i + 0.5, j + 0.5) and the camera tilted UP by 25 deg at the body origin
(SPEC §1's reading of hover.py:83-90).  The continuous projection of SPEC §6 gives 0.01435 / 1.4995 there.
OpenCV's contour rules are restated from their documentation (cv2 is not installed): parity unpinned; the product's
cheap vertex rule is compared with a full Douglas-Peucker restatement (tests/cv_restatement.py).
"""
import math

import numpy as np
import pytest

from tests import cv_restatement as cv

Q_LEVEL = np.array([0.0, 0.0, 0.0, 1.0])


def _quat(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = (math.cos(roll / 2), math.sin(roll / 2), math.cos(pitch / 2), math.sin(pitch / 2),
                              math.cos(yaw / 2), math.sin(yaw / 2))
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                     cr * cp * cy + sr * sp * sy])


def _numpy_spans(pos, tilt_deg=25.0, centre=0.5):
    """Independent rasteriser of the level-drone view: row spans of the panel for a camera tilted up by tilt_deg,
    sampling pixel (i, j) at (i + centre, j + centre)."""
    c, s = math.cos(math.radians(tilt_deg)), math.sin(math.radians(tilt_deg))
    uv = []
    for p in [(4.98, 1, 5), (4.98, -1, 5), (4.98, -1, 7), (4.98, 1, 7)]:        # hover.py:116-155
        d = np.array(p) - pos
        depth, up, right = c * d[0] + s * d[2], -s * d[0] + c * d[2], -d[1]
        uv.append((64 + 64 * right / depth, 64 - 64 * up / depth))
    rows = []
    for j in range(128):
        y, xs = j + centre, []
        for k in range(4):
            (xa, ya), (xb, yb) = uv[k], uv[(k + 1) % 4]
            if (ya <= y) != (yb <= y):
                xs.append(xa + (y - ya) * (xb - xa) / (yb - ya))
        if len(xs) == 2:
            a, b = math.ceil(min(xs) - centre), math.floor(max(xs) - centre)
            if a <= b:
                rows.append((j, a, b))
    return rows


def test_reset_pose_reproduces_the_references_target_constants(oracle):
    pos = np.array([0.0, 0.0, 0.01])                      # hover.py:78 start pose on the 0.02 m thick body (cf2x.urdf:29)
    vis, c, area, ratio = oracle.camera(pos, Q_LEVEL, "pixel")
    assert vis == 1
    assert ratio == 20 / 13 and math.floor(ratio * 100) / 100 == 1.53          # hover.py:51 target_ratio = 1.53
    assert area == 207 / 16384 and round(area, 3) == 0.013                     # hover.py:50 target_area = 0.013
    assert c[0] == -0.5 / 64 and abs(c[1] + 0.46875) < 1e-12                   # integer pixel coordinates: half-pixel bias (:197-203)
    # the continuous projection does not land on them
    va, ca, aa, ra = oracle.camera(pos, Q_LEVEL, "analytic")
    assert va == 1 and abs(aa - 0.014346) < 1e-5 and abs(ra - 1.4995) < 1e-4
    # the reset observation carries them (hover.py:112-113)
    st, obs = oracle.reset_states(1, "pixel")
    assert obs[0, 11] == area and obs[0, 14] == ratio and obs[0, 13] == 1.0


def test_camera_reading_is_pinned_by_those_constants():
    """Which camera readings reproduce (20 x 13, 0.013) from the start pose: the tilt within [23, 26] deg UP and pixel
    centres at half-integers.  Integer sample positions give 19 / 13 = 1.46, other tilts another box."""
    pos = np.array([0.0, 0.0, 0.01])

    def box(rows):
        return (max(r[2] for r in rows) - min(r[1] for r in rows) + 1, len(rows)) if rows else None

    assert box(_numpy_spans(pos, 25.0, 0.5)) == (20, 13)
    assert box(_numpy_spans(pos, 25.0, 0.0)) == (19, 13)
    def area3(rows):
        return round(cv.contour_area(cv.contour_from_spans(rows)) / 16384, 3)

    good = [t for t in range(0, 46)
            if box(_numpy_spans(pos, float(t), 0.5)) == (20, 13) and area3(_numpy_spans(pos, float(t), 0.5)) == 0.013]
    assert 25 in good and set(good) <= set(range(21, 27)), good      # only a narrow band of upward tilts fits both constants
    assert box(_numpy_spans(pos, -25.0, 0.5)) is None      # tilted down the panel is out of frame
    assert box(_numpy_spans(pos, 0.0, 0.5)) is None        # level camera: out of frame as well


def test_where_in_the_dome_the_targets_are_attained(oracle):
    """Level drone, panel straight ahead (yaw 0): the poses inside the 3 m dome (hover.py:37) whose pixel features equal the
    targets exactly lie in a thin band through the start pose (|x| <= 0.4 m, z <= 0.6 m on a 0.1 m grid) — the targets
    are the view from where the drone spawns."""
    hits = []
    for x in np.arange(-2.9, 2.91, 0.1):
        for z in np.arange(0.01, 2.9, 0.1):
            if x * x + z * z > 9.0:
                continue
            vis, c, area, ratio = oracle.camera(np.array([x, 0.0, z]), Q_LEVEL, "pixel")
            if vis and ratio == 20 / 13 and round(area, 3) == 0.013:
                hits.append((round(x, 2), round(z, 2)))
    assert (0.0, 0.01) in hits
    assert 10 <= len(hits) <= 40 and all(abs(x) <= 0.41 and z <= 0.62 for x, z in hits), hits


def test_spans_match_an_independent_rasteriser(oracle):
    rng = np.random.default_rng(4)
    checked = 0
    for _ in range(300):
        pos = np.array([rng.uniform(-2.5, 1.5), 0.0, rng.uniform(0.01, 2.0)])
        vis, c, area, ratio, spans, verts, length, apx = oracle.camera_pixel_debug(pos, Q_LEVEL)
        rows = _numpy_spans(pos)
        got = [(j, spans[j][0], spans[j][1]) for j in sorted(spans)]
        if rows and all(0 < r[0] < 127 for r in rows):
            assert got == rows
            checked += 1
    assert checked > 100


def test_contour_formulas_and_vertex_rule_against_full_douglas_peucker(oracle):
    """On random poses: the closed-form contour area / arcLength equal those of the explicit pixel contour; the visibility
    decision equals the full approxPolyDP restatement's; the four vertices coincide in most cases and the centre differs
    by at most two pixels' worth of one vertex (2 / 128) — the measured gap of the product's cheap vertex rule."""
    st = oracle.synthetic_states(12000, seed=5)
    blobs = vis_agree = same = 0
    cdiff = []
    for i in range(st.shape[0]):
        vis, c, area, ratio, spans, verts, length, apx = oracle.camera_pixel_debug(st[i, 0:3], st[i, 3:7])
        if not spans or length < 0:
            continue
        rows = [(j, spans[j][0], spans[j][1]) for j in sorted(spans)]
        pts = cv.contour_from_spans(rows)
        blobs += 1
        assert abs(cv.contour_area(pts) - apx) < 1e-9 and abs(cv.arc_length(pts) - length) < 1e-9
        v2, cx, cy, a2, r2, approx = cv.detect(rows)
        vis_agree += bool(vis) == v2
        if vis and v2:
            assert a2 == area and r2 == ratio
            assert abs(ratio * len(rows) - round(ratio * len(rows))) < 1e-9   # obs[14]: a ratio of small integers (hover.py:209-213)
            same += set(map(tuple, verts)) == set(approx)
            cdiff.append(max(abs(cx - c[0]), abs(cy - c[1])))
    assert blobs > 1500
    assert vis_agree == blobs
    assert same >= 0.65 * len(cdiff)
    assert max(cdiff) <= 2 / 128 + 1e-12 and np.mean(np.array(cdiff) <= 1 / 128 + 1e-12) >= 0.99


def test_border_rule_and_invisibility(oracle):
    # panel partly out of frame: rolled / yawed away -> invisible, features zero (hover.py:245-248)
    vis, c, area, ratio = oracle.camera(np.array([0.0, 0.0, 1.0]), _quat(0.0, 0.0, 1.2), "pixel")
    assert vis == 0 and area == 0.0 and ratio == 0.0 and c[0] == 0.0 and c[1] == 0.0
    # facing away
    assert oracle.camera(np.array([0.0, 0.0, 1.0]), _quat(0.0, 0.0, math.pi), "pixel")[0] == 0
    # pitching the nose moves the panel through the top / bottom image rows: as soon as a red pixel sits in a border row the
    # target is invisible (hover.py:180-185) although most of the panel is still in frame
    found = 0
    for pitch in np.arange(-0.9, 0.9, 0.005):
        vis, c, area, ratio, spans, *_ = oracle.camera_pixel_debug(np.array([0.0, 0.0, 1.0]), _quat(0.0, pitch, 0.0))
        if spans and (min(spans) == 0 or max(spans) == 127) and len(spans) >= 5:
            assert vis == 0
            found += 1
    assert found >= 3


def test_step_uses_the_selected_detector(oracle):
    st = oracle.synthetic_states(256, seed=9)
    act = np.zeros((256, 4))
    a = st.copy(); b = st.copy()
    obs_a, rew_a, *_ = oracle.step_batch(oracle.make_cfg(noise_ratio=0.0, detector="analytic"), a, act, 0)
    obs_b, rew_b, *_ = oracle.step_batch(oracle.make_cfg(noise_ratio=0.0, detector="pixel"), b, act, 0)
    both = (obs_a[:, 13] == 1) & (obs_b[:, 13] == 1)
    assert both.sum() > 10
    assert (obs_a[both, 14] != obs_b[both, 14]).any()
    assert np.allclose(obs_a[:, 0:7], obs_b[:, 0:7]) and np.allclose(a[:, 0:26], b[:, 0:26])
    # the reward's target term follows the detector's features (hover.py:296-320)
    i = np.where(both)[0][0]
    t_a = -np.hypot(*obs_a[i, 7:9]) - abs(obs_a[i, 11] - 0.013) - abs(obs_a[i, 14] - 1.53)
    t_b = -np.hypot(*obs_b[i, 7:9]) - abs(obs_b[i, 11] - 0.013) - abs(obs_b[i, 14] - 1.53)
    assert abs((rew_b[i] - rew_a[i]) - (t_b - t_a)) < 1e-12


def _run(rows):
    """PixelRun of fpv_dynamics.h in integers: what a walk over `rows` [(j, a, b), ...] leaves."""
    r = dict(n=0, j0=0, jn=-1, a0=0, b0=0, pa=0, pb=0, W=0, S=0, H=0, viol=-1000, amin=128, bmax=-1)
    for j, a, b in rows:
        if r["n"] == 0:
            r.update(j0=j, a0=a, b0=b, pa=a, pb=b)
        da, db = abs(a - r["pa"]), abs(b - r["pb"])
        r["viol"] = max(r["viol"], a - r["pb"], r["pa"] - b)
        r["S"] += da + db
        r["H"] += min(da, 1) + min(db, 1)
        r["W"] += b - a
        r["n"] += 1
        r["amin"], r["bmax"] = min(r["amin"], a), max(r["bmax"], b)
        r.update(pa=a, pb=b, jn=j)
    return r


def _merge(p, h):
    if h["n"] == 0:
        return dict(p)
    if p["n"] == 0:
        return dict(h)
    da, db = abs(h["a0"] - p["pa"]), abs(h["b0"] - p["pb"])
    return dict(n=p["n"] + h["n"], j0=p["j0"], a0=p["a0"], b0=p["b0"], jn=h["jn"], pa=h["pa"], pb=h["pb"], W=p["W"] + h["W"],
                S=p["S"] + h["S"] + da + db, H=p["H"] + h["H"] + min(da, 1) + min(db, 1),
                viol=max(p["viol"], h["viol"], h["a0"] - p["pb"], p["pa"] - h["b0"]), amin=min(p["amin"], h["amin"]),
                bmax=max(p["bmax"], h["bmax"]))


def test_run_closed_forms_equal_the_per_pair_sums_and_concatenate(oracle):
    """The HIP kernels reduce the blob's rows to a PixelRun (fpv_dynamics.h) and take contour area / arcLength from closed
    forms; two waves may each walk a part of the rows and concatenate.  Checked here in integers against the oracle's
    per-pair sums (SPEC §6b item 5) on random poses, for every split point of every blob."""
    st = oracle.synthetic_states(6000, seed=8)
    blobs = 0
    for i in range(st.shape[0]):
        vis, c, area, ratio, spans, verts, length, apx = oracle.camera_pixel_debug(st[i, 0:3], st[i, 3:7])
        if not spans or length < 0:
            continue
        rows = [(j, spans[j][0], spans[j][1]) for j in sorted(spans)]
        r = _run(rows)
        gap = r["jn"] - r["j0"] + 1 != r["n"]
        if gap or r["viol"] > 1:
            assert vis == 0
            continue
        blobs += 1
        w0, wn = r["b0"] - r["a0"], r["pb"] - r["pa"]
        assert r["W"] - 0.5 * (w0 + wn) - 0.5 * r["S"] + 0.5 * r["H"] == apx
        straight = w0 + wn + r["S"] + 2 * (r["n"] - 1) - 2 * r["H"]
        assert straight + 1.4142135623730951 * r["H"] == length
        if vis:
            assert ratio == (r["bmax"] - r["amin"] + 1) / (r["jn"] - r["j0"] + 1)
        for cut in range(len(rows) + 1):
            m = _merge(_run(rows[:cut]), _run(rows[cut:]))
            # (viol carries each run's first row against itself, -(b - a) <= 0: only its comparison with 1 is used)
            assert {k: v for k, v in m.items() if k != "viol"} == {k: v for k, v in r.items() if k != "viol"}
            assert (m["viol"] > 1) == (r["viol"] > 1)
    assert blobs > 800

"""Pixel detector (docs/SPEC.md §6b; simulation/hover.py:157-222) — CPU checks of the oracle's restatement.

What pins it: the reference's own constants.  hover.py:50-51 sets target_area = 0.013 and target_ratio = 1.53; from the
reset pose (hover.py:78-79: origin, level, resting on the floor) the pixel rules give a 20 x 13 bounding box
(20 / 13 = 1.538...) and a contour of 207 px^2 (207 / 16384 = 0.01263), i.e. exactly those constants at the precision
they are written with — for pixel centres at (i + 0.5, j + 0.5) and the camera tilted UP by 25 deg at the body origin
(SPEC §1's reading of hover.py:83-90).  The continuous projection of SPEC §6 gives 0.01435 / 1.4995 there.
OpenCV's calls are restated from its published algorithms (cv2 is not installed: parity unpinned) twice and independently:
oracle/cv_restate.c on the mask (Suzuki-Abe border following, approx.cpp's Douglas-Peucker), tests/cv_restatement.py in closed
form from the row spans; the two are held against each other here.
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


def _adversarial_pose(rng):
    """Poses that stress the contour calls: inside the dome, close to the panel's plane (edge-on slivers, blobs in several pieces)
    and far away (blobs of a few pixels), aimed roughly at the panel."""
    mode = rng.integers(0, 3)
    if mode == 0:
        pos = np.array([rng.uniform(4.0, 4.97), rng.uniform(-3, 3), rng.uniform(3, 9)])
    elif mode == 1:
        pos = np.array([rng.uniform(-3, 4.9), rng.uniform(-4, 4), rng.uniform(0, 9)])
    else:
        pos = np.array([rng.uniform(-40, 4.0), rng.uniform(-10, 10), rng.uniform(0, 20)])
    tgt = np.array([4.98, rng.uniform(-1.5, 1.5), rng.uniform(4.5, 7.5)])
    d = tgt - pos
    yaw = math.atan2(d[1], d[0]) + rng.normal(0, 0.3)
    pitch = -math.atan2(d[2], math.hypot(d[0], d[1])) + math.radians(25) + rng.normal(0, 0.3)
    return pos, _quat(rng.normal(0, 0.6), pitch, yaw)


def _run_integers(rows):
    """SPEC §6b item 5's integers of one 8-connected component."""
    r = _run(rows)
    return r["W"], r["S"], r["H"], r["b0"] - r["a0"], r["pb"] - r["pa"], r["n"]


def test_oracle_equals_the_closed_form_restatement_on_every_blob(oracle):
    """oracle/cv_restate.c follows the border of the MASK (Suzuki-Abe as cv2.findContours runs it) and runs approx.cpp's
    Douglas-Peucker on that contour; tests/cv_restatement.py states the contour of a row-span blob in closed form and has its own
    Douglas-Peucker.  Pose by pose they must agree on everything: the chosen contour point by point, len(contours), len(approx),
    the visibility decision, the four vertices, arcLength, and every feature bit for bit.  Poses: the synthetic states of the
    benchmark plus adversarial ones (slivers in several pieces, tiny and huge blobs)."""
    rng = np.random.default_rng(11)
    st = oracle.synthetic_states(4000, seed=5)
    poses = [(st[i, 0:3], st[i, 3:7]) for i in range(st.shape[0])] + [_adversarial_pose(rng) for _ in range(9000)]
    blobs = seen = pieces = not_four = 0
    for pos, quat in poses:
        vis, c, area, ratio, spans, verts, length, apx, more = oracle.camera_pixel_debug(pos, quat, full=True)
        if not spans:
            assert vis == 0
            continue
        rows = [(j, spans[j][0], spans[j][1]) for j in sorted(spans)]
        v2, cx, cy, a2, r2, approx, contour = cv.detect(rows)
        blobs += 1
        assert bool(vis) == v2
        if any(j == 0 or j == 127 or a == 0 or b == 127 for j, a, b in rows):
            assert vis == 0                                                    # hover.py:180-185
            continue
        comps = cv.components(rows)
        pieces += len(comps) > 1
        assert more["contours"] == len(comps) and more["contour"] == contour and more["approx"] == len(approx)
        assert cv.arc_length(contour) == length and cv.contour_area(contour) == apx
        not_four += len(approx) != 4
        # the closed forms the kernels use (SPEC §6b item 5) equal the explicit contour's area and, up to the float32 rounding of
        # its diagonal runs, its length
        chosen = [cmp for cmp in comps if cv.contour_from_spans(cmp) == contour][-1]
        W, S, H, w0, wn, n = _run_integers(chosen)
        assert W - 0.5 * (w0 + wn) - 0.5 * S + 0.5 * H == apx
        assert abs(w0 + wn + S + 2 * (n - 1) - 2 * H + math.sqrt(2.0) * H - length) < 1e-5 * max(length, 1.0)
        if vis:
            seen += 1
            assert (cx, cy, a2, r2) == (c[0], c[1], area, ratio)
            assert sorted(map(tuple, verts)) == sorted(approx)
            assert abs(ratio * len(chosen) - round(ratio * len(chosen))) < 1e-9   # obs[14]: a ratio of small integers (:209-213)
    assert blobs > 8000 and seen > 4000 and pieces > 200 and not_four > 1000, (blobs, seen, pieces, not_four)


def test_known_contours_of_small_masks(oracle):
    """cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) on masks whose answer is common knowledge: a filled rectangle comes
    back as its four corners from the top-left one downwards (counter-clockwise on the screen), a single pixel as itself, a
    one-pixel line as its two ends; the list starts with the contour found last; a blob inside a ring's hole is not external."""
    m = np.zeros((16, 16), np.uint8)
    m[3:7, 2:9] = 1
    assert oracle.cv_find_contours(m) == [[(2, 3), (2, 6), (8, 6), (8, 3)]]
    m[10, 12] = 1
    m[12, 1:5] = 1
    assert oracle.cv_find_contours(m) == [[(1, 12), (4, 12)], [(12, 10)], [(2, 3), (2, 6), (8, 6), (8, 3)]]
    ring = np.zeros((16, 16), np.uint8)
    ring[2:13, 2:13] = 1
    ring[4:11, 4:11] = 0
    ring[7, 7] = 1
    assert oracle.cv_find_contours(ring) == [[(2, 2), (2, 12), (12, 12), (12, 2)]]
    # a diamond: every border pixel is a direction change of the chain except along its straight diagonals
    d = np.zeros((9, 9), np.uint8)
    for y in range(9):
        w = 4 - abs(y - 4)
        d[y, 4 - w:4 + w + 1] = 1
    assert oracle.cv_find_contours(d) == [[(4, 0), (0, 4), (4, 8), (8, 4)]]
    c = [(4, 0), (0, 4), (4, 8), (8, 4)]
    assert oracle.cv_contour_area(c) == 32.0 and oracle.cv_bounding_rect(c) == (0, 0, 9, 9)
    assert oracle.cv_arc_length(c) == 4.0 * float(np.sqrt(np.float32(32.0)))            # float32 segment lengths
    assert oracle.cv_approx_poly_dp(c, 0.04 * oracle.cv_arc_length(c)) == c      # the three sweeps end on vertex 0 again
    # a U: the cavity is open, so a blob inside it IS external; cv2 lists it first (found last)
    u = np.zeros((12, 12), np.uint8)
    u[2:10, 2] = u[2:10, 8] = 1
    u[9, 2:9] = 1
    u[5, 5] = 1
    got = oracle.cv_find_contours(u)
    assert len(got) == 2 and got[0] == [(5, 5)]


def test_float32_centre_is_exact_and_diagonal_runs_round_as_sqrt():
    """hover.py:203 stores the centre as float32: for four integer vertices below 128 the value sum / 256 - 1 is a multiple of
    1 / 256 in [-1, 1), which float32 holds exactly.  And the kernels' float32(k * sqrt 2 in double) is float32 sqrt(2 k^2),
    cv2.arcLength's length of a run of k diagonal steps."""
    for s in range(0, 4 * 127 + 1):
        assert float(np.float32(s / 4.0 / 64.0 - 1.0)) == s / 256.0 - 1.0
    for k in range(1, 200):
        assert np.float32(np.float64(k) * 1.4142135623730951) == np.sqrt(np.float32(2 * k * k), dtype=np.float32)


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
            continue                                       # several pieces: cv2 takes the largest (the test above)
        blobs += 1
        w0, wn = r["b0"] - r["a0"], r["pb"] - r["pa"]
        assert r["W"] - 0.5 * (w0 + wn) - 0.5 * r["S"] + 0.5 * r["H"] == apx
        straight = w0 + wn + r["S"] + 2 * (r["n"] - 1) - 2 * r["H"]
        assert abs(straight + 1.4142135623730951 * r["H"] - length) < 1e-5 * length       # (float32 rounding of the diagonal runs)
        if vis:
            assert ratio == (r["bmax"] - r["amin"] + 1) / (r["jn"] - r["j0"] + 1)
        for cut in range(len(rows) + 1):
            m = _merge(_run(rows[:cut]), _run(rows[cut:]))
            # (viol carries each run's first row against itself, -(b - a) <= 0: only its comparison with 1 is used)
            assert {k: v for k, v in m.items() if k != "viol"} == {k: v for k, v in r.items() if k != "viol"}
            assert (m["viol"] > 1) == (r["viol"] > 1)
    assert blobs > 800

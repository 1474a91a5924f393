"""Second, independent restatement of the OpenCV calls of the reference's detector (simulation/hover.py:187-213), in pure
Python and specialised to what the detector's masks are: one span of red pixels per image row.

oracle/cv_restate.c restates cv2.findContours / arcLength / contourArea / boundingRect / approxPolyDP for ANY mask (Suzuki-Abe
border following on the pixel grid).  This file states the same results in closed form from the row spans — the form the HIP
kernels use (docs/SPEC.md §6b) — so that tests/test_detector_cpu.py can hold the two against each other pose by pose:

* the outer border of an 8-connected run of rows, in cv2's order: it starts at the first pixel of the top row and runs
  counter-clockwise as seen on the screen — down the left ends, along the bottom row, up the right ends, back along the top
  row.  Between the left ends of two rows (a above, a' below): a' < a: one step SW, then W along the lower row; a' == a: one
  step S; a' > a: E along the upper row to a' - 1, then one step SE.  Up the right ends (b below, b' above): b' > b: one step
  NE, then E along the upper row; b' == b: N; b' < b: W along the lower row to b' + 1, then NW;
* CHAIN_APPROX_SIMPLE keeps the points where the direction changes (the start point always);
* cv2.arcLength adds segment lengths taken in float32 into a double; contourArea is the shoelace sum; boundingRect the
  integer extent;
* approxPolyDP on a closed curve as approx.cpp runs it (three farthest-point sweeps, the slice stack, the clean-up pass).

OpenCV is not installed here and ships no source in /root/reference: PARITY UNPINNED for both restatements; this one exists to
catch mistakes in either.  Pure Python, small inputs only.
"""
from __future__ import annotations

import math

import numpy as np

# direction codes of contours.cpp: 0 = E, 1 = NE, 2 = N, 3 = NW, 4 = W, 5 = SW, 6 = S, 7 = SE (y grows downwards)
_STEP = {(1, 0): 0, (1, -1): 1, (0, -1): 2, (-1, -1): 3, (-1, 0): 4, (-1, 1): 5, (0, 1): 6, (1, 1): 7}


def components(rows):
    """rows: [(j, a, b), ...] sorted by j, one span per row.  Splits them into 8-connected components (consecutive rows whose
    spans touch or overlap diagonally), top to bottom."""
    out, cur = [], []
    for r in rows:
        if cur:
            j, a, b = cur[-1]
            if r[0] != j + 1 or r[1] > b + 1 or r[2] < a - 1:
                out.append(cur)
                cur = []
        cur.append(r)
    if cur:
        out.append(cur)
    return out


def border_path(rows):
    """Every pixel of the outer border of ONE 8-connected component (CHAIN_APPROX_NONE), in cv2's order, as a closed list:
    the last point is followed by the first again."""
    j0, a0, b0 = rows[0]
    if len(rows) == 1:
        return [(x, j0) for x in range(a0, b0 + 1)] + [(x, j0) for x in range(b0 - 1, a0, -1)]
    pts = [(a0, j0)]
    for (j, a, b), (jn, an, bn) in zip(rows, rows[1:]):                     # down the left ends
        if an < a:
            pts += [(x, jn) for x in range(a - 1, an - 1, -1)]
        elif an == a:
            pts.append((an, jn))
        else:
            pts += [(x, j) for x in range(a + 1, an)]
            pts.append((an, jn))
    jl, al, bl = rows[-1]
    pts += [(x, jl) for x in range(al + 1, bl + 1)]                         # along the bottom row
    for (j, a, b), (jp, ap, bp) in zip(rows[::-1], rows[-2::-1]):           # up the right ends
        if bp > b:
            pts += [(x, jp) for x in range(b + 1, bp + 1)]
        elif bp == b:
            pts.append((bp, jp))
        else:
            pts += [(x, j) for x in range(b - 1, bp, -1)]
            pts.append((bp, jp))
    pts += [(x, j0) for x in range(b0 - 1, a0, -1)]                         # back along the top row, short of the start
    if len(pts) > 1 and pts[-1] == pts[0]:                                  # (a one-pixel top row: the way up ends on the start)
        pts.pop()
    return pts


def compress(path):
    """CHAIN_APPROX_SIMPLE of a closed pixel path: the points whose outgoing step differs from their incoming one."""
    n = len(path)
    if n == 1:
        return list(path)
    out = []
    for i, p in enumerate(path):
        q, r = path[i - 1], path[(i + 1) % n]
        if _STEP[(p[0] - q[0], p[1] - q[1])] != _STEP[(r[0] - p[0], r[1] - p[1])] or i == 0:
            out.append(p)
    return out


def contour_from_spans(rows):
    """cv2.findContours(..., CHAIN_APPROX_SIMPLE)'s contour of one 8-connected component given as row spans."""
    return compress(border_path(rows))


def arc_length(pts):
    """cv2.arcLength(pts, True): float32 segment lengths added into a double."""
    total = 0.0
    for i in range(len(pts)):
        dx = np.float32(pts[i][0]) - np.float32(pts[i - 1][0])
        dy = np.float32(pts[i][1]) - np.float32(pts[i - 1][1])
        total += float(np.sqrt(dx * dx + dy * dy, dtype=np.float32))
    return total if len(pts) > 1 else 0.0


def contour_area(pts):
    s = 0.0
    for i in range(len(pts)):
        x0, y0 = pts[i - 1]
        x1, y1 = pts[i]
        s += x0 * y1 - x1 * y0
    return abs(s) * 0.5


def bounding_rect(pts):
    xs = [p[0] for p in pts]
    ys = [p[1] for p in pts]
    return min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1


def approx_poly_dp_closed(src, eps):
    count = len(src)
    if count == 0:
        return []
    eps2 = eps * eps
    dst = []
    stack = []
    pos = 0
    right_start = 0
    le_eps = False
    start_pt = None
    for _ in range(3):
        max_dist = 0.0
        pos = (pos + right_start) % count
        start_pt = src[pos]
        pos = (pos + 1) % count
        for j in range(1, count):
            pt = src[pos]
            pos = (pos + 1) % count
            d = (pt[0] - start_pt[0]) ** 2 + (pt[1] - start_pt[1]) ** 2
            if d > max_dist:
                max_dist = d
                right_start = j
        le_eps = max_dist <= eps2
    if not le_eps:
        slice_start = pos % count
        right = (right_start + slice_start) % count
        stack.append((right, slice_start))        # right_slice: start = right, end = slice.start
        stack.append((slice_start, right))        # slice
    else:
        dst.append(start_pt)
    while stack:
        s_start, s_end = stack.pop()
        end_pt = src[s_end]
        pos = s_start
        start_pt = src[pos]
        pos = (pos + 1) % count
        if pos != s_end:
            dx, dy = end_pt[0] - start_pt[0], end_pt[1] - start_pt[1]
            max_dist = 0.0
            split = s_start
            while pos != s_end:
                pt = src[pos]
                pos = (pos + 1) % count
                d = abs((pt[1] - start_pt[1]) * dx - (pt[0] - start_pt[0]) * dy)
                if d > max_dist:
                    max_dist = d
                    split = (pos + count - 1) % count
            le = max_dist * max_dist <= eps2 * (dx * dx + dy * dy)
        else:
            le = True
            start_pt = src[s_start]
        if le:
            dst.append(start_pt)
        else:
            stack.append((split, s_end))
            stack.append((s_start, split))
    # clean-up pass: drop the middle point of almost straight triples (in place, as approx.cpp does)
    count = new_count = len(dst)
    out = list(dst)
    pos = count - 1
    start_pt = out[pos]
    pos = (pos + 1) % count
    wpos = pos
    pt = out[pos]
    pos = (pos + 1) % count
    i = 0
    while i < count and new_count > 2:
        end_pt = out[pos]
        pos = (pos + 1) % count
        dx, dy = end_pt[0] - start_pt[0], end_pt[1] - start_pt[1]
        dist = abs((pt[0] - start_pt[0]) * dy - (pt[1] - start_pt[1]) * dx)
        sip = (pt[0] - start_pt[0]) * (end_pt[0] - pt[0]) + (pt[1] - start_pt[1]) * (end_pt[1] - pt[1])
        if dist * dist <= 0.5 * eps2 * (dx * dx + dy * dy) and dx != 0 and dy != 0 and sip >= 0:
            new_count -= 1
            out[wpos] = start_pt = end_pt
            wpos = (wpos + 1) % count
            pt = out[pos]
            pos = (pos + 1) % count
            i += 2
            continue
        out[wpos] = start_pt = pt
        wpos = (wpos + 1) % count
        pt = end_pt
        i += 1
    return out[:new_count]


def detect(rows):
    """The reference's feature extraction (hover.py:180-213) from the mask's row spans [(j, a, b), ...].  Returns
    (visible, center_x, center_y, area, ratio, approx, contour)."""
    none = (False, 0.0, 0.0, 0.0, 0.0, [], [])
    if not rows:
        return none
    comps = components(rows)
    contours = [contour_from_spans(c) for c in comps][::-1]                 # cv2's list: the contour found last comes first
    if any(j == 0 or j == 127 or a == 0 or b == 127 for j, a, b in rows):   # :180-185, on the whole mask
        return none
    contour = max(contours, key=contour_area)                               # :189 (the first maximum of the list)
    eps = 0.04 * arc_length(contour)                                        # :190
    approx = approx_poly_dp_closed(contour, eps)                            # :191
    if len(approx) != 4:                                                    # :193
        return (False, 0.0, 0.0, 0.0, 0.0, approx, contour)
    cx = float(np.float32(sum(p[0] for p in approx) / 4.0 / 64.0 - 1.0))    # :197-203
    cy = float(np.float32(sum(p[1] for p in approx) / 4.0 / 64.0 - 1.0))
    _, _, w, h = bounding_rect(contour)
    return True, cx, cy, contour_area(contour) / 16384.0, w / h, approx, contour

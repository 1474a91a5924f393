"""Test-side restatement of the OpenCV calls of the reference's detector (simulation/hover.py:187-213) on a binary
mask given as row spans: the outer contour of an 8-connected, row-convex blob, cv2.arcLength, cv2.contourArea
(Green's formula over the contour), cv2.boundingRect and cv2.approxPolyDP (Douglas-Peucker as OpenCV's approx.cpp
runs it on a closed curve: three farthest-point iterations to pick the two start vertices, recursive splitting on
the perpendicular distance, then the clean-up pass over almost straight triples).

OpenCV is not installed here and ships no source in /root/reference, so this is written from the documented algorithm
and the author's recollection of approx.cpp: PARITY UNPINNED.  Its use: quantify how often the product's cheap vertex
rule (docs/SPEC.md §6b: covered pixel nearest to each projected corner + the epsilon gate on the diagonals) gives what
the full procedure gives — tests/test_detector_cpu.py.  Pure Python, small inputs only.
"""
from __future__ import annotations

import math


def contour_from_spans(rows):
    """rows: list of (j, a, b) for consecutive image rows j (top to bottom), a <= b pixel columns.
    Returns the outer contour as a closed list of pixel points (x, y), clockwise as seen in the image, starting at the
    top-left pixel — every boundary pixel listed (CHAIN_APPROX_NONE)."""
    pts = []
    j0, a0, b0 = rows[0]
    pts += [(x, j0) for x in range(a0, b0 + 1)]                     # top row, left to right
    for (j, a, b), (jn, an, bn) in zip(rows, rows[1:]):              # down the right side
        if bn > b:
            pts += [(x, jn) for x in range(b + 1, bn + 1)]           # diagonal step, then along the lower row
        elif bn == b:
            pts.append((bn, jn))
        else:
            pts += [(x, j) for x in range(b - 1, bn, -1)]            # back along row j to bn + 1, then the diagonal step
            pts.append((bn, jn))
    jl, al, bl = rows[-1]
    pts += [(x, jl) for x in range(bl - 1, al - 1, -1)]             # bottom row, right to left
    for (j, a, b), (jp, ap, bp) in zip(rows[::-1], rows[-2::-1]):    # up the left side
        if ap < a:
            pts += [(x, jp) for x in range(a - 1, ap - 1, -1)]
        elif ap == a:
            pts.append((ap, jp))
        else:
            pts += [(x, j) for x in range(a + 1, ap)]
            pts.append((ap, jp))
    # drop consecutive duplicates and the closing duplicate of the start
    out = []
    for p in pts:
        if not out or out[-1] != p:
            out.append(p)
    while len(out) > 1 and out[-1] == out[0]:
        out.pop()
    return out


def arc_length(pts):
    return sum(math.hypot(pts[i][0] - pts[i - 1][0], pts[i][1] - pts[i - 1][1]) for i in range(len(pts)))


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
    # clean-up pass: drop the middle point of almost straight triples
    count = new_count = len(dst)
    if count < 3:
        return dst
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
    return out[:new_count] if new_count < count else out


def detect(rows):
    """The reference's feature extraction (hover.py:187-213) from the mask's row spans.  Returns
    (visible, center_x, center_y, area, ratio) with 128 x 128 normalisation, or (False, ...) when the approximation is
    not a quadrilateral."""
    pts = contour_from_spans(rows)
    eps = 0.04 * arc_length(pts)
    approx = approx_poly_dp_closed(pts, eps)
    if len(approx) != 4:
        return False, 0.0, 0.0, 0.0, 0.0, approx
    cx = sum(p[0] for p in approx) / 4.0 / 64.0 - 1.0
    cy = sum(p[1] for p in approx) / 4.0 / 64.0 - 1.0
    _, _, w, h = bounding_rect(pts)
    return True, cx, cy, contour_area(pts) / 16384.0, w / h, approx

// fpv_polydp.h — cv2.arcLength + cv2.approxPolyDP of the reference's detector (simulation/hover.py:189-203) on a blob given as
// one span of red pixels per image row, worked by a GROUP of lanes (docs/SPEC.md §6b items 5-6).
//
// What the reference runs: contours = cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE); contour = max(contours,
// key=cv2.contourArea); approx = cv2.approxPolyDP(contour, 0.04 * cv2.arcLength(contour, True), True); four vertices or the
// target is invisible; centre = their mean.  OpenCV (opencv-python 4.11.0.86, simulation/uv.lock:655) is in neither
// /root/reference nor this image: its published algorithms are restated (parity unpinned), here in the form a GPU lane group
// can run without ever listing the contour:
//
//  * The outer border of an 8-connected run of rows c0..c1 with spans [a_j, b_j] starts at (a_c0, c0) and runs down the left
//    ends, along the bottom row, up the right ends and back along the top row.  Between the left ends of two rows (a above,
//    a' below): a' < a: one step SW, then W along the lower row; a' = a: S; a' > a: E along the upper row to a' - 1, then SE.
//    Up the right ends (b below, b' above): b' > b: NE, then E along the upper row; b' = b: N; b' < b: W along the lower row to
//    b' + 1, then NW.
//  * SLOTS.  Every row contributes three points per side, in path order: where the path enters the row, the span end, where it
//    leaves the row — left side of row j: (max(a_j, a_{j-1} - 1), j), (a_j, j), (max(a_j, a_{j+1} - 1), j); right side:
//    (min(b_j, b_{j+1} + 1), j), (b_j, j), (min(b_j, b_{j-1} + 1), j); a missing neighbour row counts as the row itself.  The 6 n
//    slots, left side downwards then right side upwards, are a sub-sequence of the pixel path that contains every point at
//    which the path changes direction, i.e. every CHAIN_APPROX_SIMPLE vertex, in order; the others lie on the straight runs
//    between two vertices (or repeat one).  Douglas-Peucker's searches maximise a convex function of the point — squared
//    distance to a point, |distance to a line| — with strict ">" (the first maximum in path order wins): along a straight run
//    such a function is below the larger of its two ends, or constant with the earlier end winning the tie, so a point that is
//    no vertex is never selected and the search over the slots returns the very point the search over cv2's contour returns.
//  * cv2.arcLength takes every segment's length in float32 and adds them into a double.  Axis-parallel segments are integers;
//    a run of k equal diagonal steps is float32(sqrt(2 k^2)) = float32(k * sqrt 2 in double) for every k < 200
//    (tests/test_detector_cpu.py).  All those floats are multiples of 2^-23 below 256, so their sum is exact in any order: it is
//    accumulated as two integers.  Two successive diagonal steps belong to one run iff no axis-parallel step lies between them:
//    going down the left ends with d = a' - a: (d_prev = -1 and d < 0) or (d_prev > 0 and d = 1); going up the right ends
//    with e = b' - b: (e_prev = 1 and e > 0) or (e_prev < 0 and e = -1).
//  * approx.cpp's closed-curve procedure as it stands: three farthest-point sweeps, the slice stack (kept as the stack of pending
//    slice ends: the slices are worked left to right), acceptance by max_dist^2 <= eps^2 |chord|^2 in double, and the clean-up
//    pass, in place.  The clean-up pass drops at most every other vertex, so ten or more vertices can never become four: the
//    recursion is abandoned (invisible) as soon as emitted + pending vertices exceed nine.
//
// Everything is integer arithmetic on the spans except eps (double), so fp32 and fp64 handles take the same decisions from the
// same spans, and the CPU oracle's (oracle/cv_restate.c, written independently on the explicit contour) to the bit.
//
// The code is plain C++ over two policies — Grp (lane(), size(), max_u32(), add_i32(): all-reduce over the group) and Span
// (operator()(row, a, b): span of a blob row, false when the row is empty) — so that tests/host/polydp_host.cpp can compile
// this very file with g++ (a group of one lane) and tests/test_polydp_cpu.py can hold it against the oracle without a GPU.
#pragma once
#include <stdint.h>

namespace fpv {

struct PolyBlob {      // one 8-connected component: rows c0 .. c1 and the integers of SPEC §6b item 5 over them
    int c0, c1;        // first / last row
    int W, S, H;       // sum of widths b - a; sum over consecutive rows of |a' - a| + |b' - b|; number of span ends that moved
    int amin, bmax;    // bounding columns
};

struct PolyResult {
    int vis;           // approxPolyDP returned four vertices
    int sx, sy;        // sums of their coordinates (centre = sum / 256 - 1, hover.py:197-203)
    int area2;         // twice cv2.contourArea of the contour
    int w, h;          // cv2.boundingRect's size
};

namespace polydp {

__host__ __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ __forceinline__ int iabs(int a) { return a < 0 ? -a : a; }

// slot i of the blob (see above): left side downwards, then right side upwards
template <class Span>
__host__ __device__ __forceinline__ void slot(const Span& span, int c0, int c1, int n, int i, int& x, int& y) {
    const int side = i >= 3 * n ? 1 : 0;
    const int ii = side ? i - 3 * n : i;
    const int r = (ii * 0xAAAB) >> 17;                     // ii / 3 (exact below 2^15; ii < 768)
    const int k = ii - 3 * r;
    const int j = side ? c1 - r : c0 + r;
    const int nb = imin(imax(j + (side ? 1 - k : k - 1), c0), c1);     // the row the path comes from (k = 0) / goes to (k = 2)
    int a, b, na, nbb;
    span(j, a, b);
    span(nb, na, nbb);
    x = side ? imin(b, nbb + 1) : imax(a, na - 1);
    y = j;
}

// First maximum, in path order, over the slots base + 1 .. base + count (mod P) of |p - s|^2 (kPoint) or of
// |(p - s) x (dx, dy)|, packed as value << 10 | (1023 - q), q = 1 .. count; a value of 0 never wins (strict ">" from 0).
template <bool kPoint, class Grp, class Span>
__host__ __device__ __forceinline__ uint32_t sweep(const Grp& g, const Span& span, int c0, int c1, int n, int P, int base, int count,
                                                   int sx, int sy, int dx, int dy) {
    uint32_t best = 0u;
    for (int q = 1 + g.lane(); q <= count; q += g.size()) {
        int i = base + q;
        if (i >= P) i -= P;
        int x, y;
        slot(span, c0, c1, n, i, x, y);
        const int ex = x - sx, ey = y - sy;
        const int d = kPoint ? ex * ex + ey * ey : iabs(ey * dx - ex * dy);
        const uint32_t key = ((uint32_t)d << 10) | (uint32_t)(1023 - q);
        best = key > best ? key : best;
    }
    return g.max_u32(best);
}

// d of the transition s along a side: left (side 0) from row c0 + s to c0 + s + 1: a' - a; right (side 1) from row c1 - s up to
// c1 - s - 1: b' - b
template <class Span>
__host__ __device__ __forceinline__ int delta(const Span& span, int c0, int c1, int side, int s) {
    int a, b, a2, b2;
    if (side) { span(c1 - s, a, b); span(c1 - s - 1, a2, b2); return b2 - b; }
    span(c0 + s, a, b); span(c0 + s + 1, a2, b2);
    return a2 - a;
}
__host__ __device__ __forceinline__ bool same_run(int prev, int cur, int side) {
    return side ? ((prev == 1 && cur > 0) || (prev < 0 && cur == -1)) : ((prev == -1 && cur < 0) || (prev > 0 && cur == 1));
}

struct Verts {          // up to nine vertices, x | y << 8, packed
    uint64_t v0, v1;
    uint32_t v2;
    __host__ __device__ __forceinline__ uint32_t get(int t) const {
        const uint64_t w = t < 4 ? v0 : v1;
        const uint32_t p = (uint32_t)(w >> ((t & 3) << 4)) & 0xFFFFu;
        return t == 8 ? v2 : p;
    }
    __host__ __device__ __forceinline__ void set(int t, uint32_t p) {
        const int sh = (t & 3) << 4;
        const uint64_t m = ~((uint64_t)0xFFFFu << sh), q = (uint64_t)p << sh;
        if (t < 4) v0 = (v0 & m) | q;
        else if (t < 8) v1 = (v1 & m) | q;
        else v2 = p;
    }
};

}  // namespace polydp

// The largest component by cv2.contourArea among the non-empty rows j0 .. jn of a drone's span table (ties: the lowest one,
// which cv2.findContours lists first — it lists the contour found last first — and max() keeps).  Serial: every lane of the
// group walks the rows (blobs that are not one 8-connected run of rows are slivers seen edge-on; rare).
template <class Span>
__host__ __device__ __forceinline__ PolyBlob poly_largest_component(const Span& span, int j0, int jn) {
    using namespace polydp;
    PolyBlob best = {0, -1, 0, 0, 0, 128, -1};
    int best_area2 = -1;
    PolyBlob cur = {0, -1, 0, 0, 0, 128, -1};
    int pa = 0, pb = 0, w0 = 0;
    bool open = false;
    for (int j = j0; j <= jn + 1; ++j) {
        int a = 0, b = 0;
        const bool have = j <= jn && span(j, a, b);
        const bool joins = open && have && !(a > pb + 1 || b < pa - 1);
        if (open && !joins) {                                  // the component ends with row j - 1
            const int area2 = 2 * cur.W - (w0 + (pb - pa)) - cur.S + cur.H;
            if (area2 >= best_area2) { best_area2 = area2; best = cur; best.c1 = j - 1; }
            open = false;
        }
        if (have) {
            if (!open) { cur = PolyBlob{j, j, 0, 0, 0, 128, -1}; w0 = b - a; open = true; }
            else {
                const int da = iabs(a - pa), db = iabs(b - pb);
                cur.S += da + db;
                cur.H += (da > 0) + (db > 0);
            }
            cur.W += b - a;
            cur.amin = imin(cur.amin, a); cur.bmax = imax(cur.bmax, b);
            pa = a; pb = b;
        }
    }
    return best;
}

// hover.py:189-213 for the component `bl` (an 8-connected run of non-empty rows).  Every lane of the group calls it with the
// same arguments and gets the same result.
template <class Grp, class Span>
__host__ __device__ __forceinline__ PolyResult poly_detect(const Grp& g, const Span& span, const PolyBlob& bl) {
    using namespace polydp;
    const int c0 = bl.c0, c1 = bl.c1, n = c1 - c0 + 1, P = 6 * n;
    int a0, b0, an, bn;
    span(c0, a0, b0);
    span(c1, an, bn);
    const int w0 = b0 - a0, wn = bn - an;
    PolyResult res = {0, 0, 0, 2 * bl.W - (w0 + wn) - bl.S + bl.H, bl.bmax - bl.amin + 1, n};
    // ---- cv2.arcLength: axis-parallel steps as one integer, the runs of diagonal steps in float32 each (exact sums)
    const int straight = w0 + wn + bl.S + 2 * (n - 1) - 2 * bl.H;
    int gi = 0, gf = 0;
    for (int t = g.lane(); t < 2 * (n - 1); t += g.size()) {
        const int side = t >= n - 1 ? 1 : 0, s = side ? t - (n - 1) : t;
        const int d = delta(span, c0, c1, side, s);
        if (d == 0) continue;
        if (s < n - 2 && same_run(d, delta(span, c0, c1, side, s + 1), side)) continue;     // not the last step of its run
        int k = 1, cur = d;
        for (int q = s - 1; q >= 0; --q) {
            const int dp = delta(span, c0, c1, side, q);
            if (!same_run(dp, cur, side)) break;
            ++k; cur = dp;
        }
        const double gv = (double)(float)((double)k * 1.4142135623730951);
        const int whole = (int)gv;
        gi += whole;
        gf += (int)((gv - (double)whole) * 8388608.0);
    }
    gi = g.add_i32(gi);
    gf = g.add_i32(gf);
    const double len = (double)(straight + gi) + (double)gf * (1.0 / 8388608.0);
    const double eps = 0.04 * len;                             // hover.py:190
    const double eps2 = eps * eps;
    // ---- approxPolyDP, 1: three farthest-point sweeps
    int sx, sy, s2 = 0;
    uint32_t key = 0u;
    for (int it = 0; it < 3; ++it) {
        slot(span, c0, c1, n, s2, sx, sy);
        key = sweep<true>(g, span, c0, c1, n, P, s2, P - 1, sx, sy, 0, 0);
        if (it < 2) {
            s2 += (key >> 10) ? 1023 - (int)(key & 1023u) : 0;
            if (s2 >= P) s2 -= P;
        }
    }
    if ((double)(key >> 10) <= eps2) return res;               // one vertex
    int e = s2 + 1023 - (int)(key & 1023u);
    if (e >= P) e -= P;
    // ---- 2, 3: the slices, left to right.  pend: ends of the slices not yet accepted, the nearest on top (10 bits each)
    uint64_t pend_lo = ((uint64_t)s2 << 10) | (uint64_t)e, pend_hi = 0ull;
    int depth = 2, cnt = 0, s = s2;
    Verts vs = {0ull, 0ull, 0u};
    while (depth > 0) {
        const int en = (int)(pend_lo & 1023u);
        int between = en - s - 1;
        if (between < 0) between += P;
        int ex, ey;
        slot(span, c0, c1, n, s, sx, sy);
        slot(span, c0, c1, n, en, ex, ey);
        const int dx = ex - sx, dy = ey - sy;
        bool accept = true;
        if (between > 0) {
            key = sweep<false>(g, span, c0, c1, n, P, s, between, sx, sy, dx, dy);
            const double md = (double)(key >> 10);
            accept = md * md <= eps2 * (double)(dx * dx + dy * dy);
        }
        if (accept) {
            if (cnt >= 9) return res;
            vs.set(cnt++, (uint32_t)sx | (uint32_t)sy << 8);
            s = en;
            pend_lo = (pend_lo >> 10) | (pend_hi << 54);
            pend_hi >>= 10;
            --depth;
        } else {
            int m = s + 1023 - (int)(key & 1023u);
            if (m >= P) m -= P;
            pend_hi = (pend_hi << 10) | (pend_lo >> 54);
            pend_lo = (pend_lo << 10) | (uint64_t)m;
            ++depth;
            if (cnt + depth > 9) return res;                   // ten or more vertices: never four after the clean-up pass
        }
    }
    // ---- 4: the clean-up pass (closed curve, in place)
    const int count = cnt;
    int new_count = cnt, pos = count - 1, wpos;
    uint32_t start_pt = vs.get(pos);
    if (++pos >= count) pos = 0;
    wpos = pos;
    uint32_t pt = vs.get(pos);
    if (++pos >= count) pos = 0;
    for (int i = 0; i < count && new_count > 2; ++i) {
        const uint32_t end_pt = vs.get(pos);
        if (++pos >= count) pos = 0;
        const int stx = (int)(start_pt & 255u), sty = (int)(start_pt >> 8), px = (int)(pt & 255u), py = (int)(pt >> 8);
        const int enx = (int)(end_pt & 255u), eny = (int)(end_pt >> 8);
        const int dx = enx - stx, dy = eny - sty;
        const double dist = (double)iabs((px - stx) * dy - (py - sty) * dx);
        const int sip = (px - stx) * (enx - px) + (py - sty) * (eny - py);
        if (dist * dist <= 0.5 * eps2 * (double)(dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
            --new_count;
            start_pt = end_pt;
            vs.set(wpos, start_pt);
            if (++wpos >= count) wpos = 0;
            pt = vs.get(pos);
            if (++pos >= count) pos = 0;
            ++i;
            continue;
        }
        start_pt = pt;
        vs.set(wpos, start_pt);
        if (++wpos >= count) wpos = 0;
        pt = end_pt;
    }
    if (new_count != 4) return res;                            // hover.py:193
    for (int t = 0; t < 4; ++t) {
        const uint32_t p = vs.get(t);
        res.sx += (int)(p & 255u);
        res.sy += (int)(p >> 8);
    }
    res.vis = 1;
    return res;
}

}  // namespace fpv

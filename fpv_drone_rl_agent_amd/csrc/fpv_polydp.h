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
// The code is plain C++ over two policies — Grp (lane(), size(), all-reduces max_u32 / max_i32 / add_i32, the inclusive scan
// scan_add and fence(), which orders the group's writes to its scratch before its reads) and Span (operator()(row, a, b): span of
// a blob row, false when the row is empty) — so that tests/host/polydp_host.cpp can compile this very file with g++ — a group of
// one lane, and groups of eight lanes played by eight threads in lockstep — and tests/test_polydp_cpu.py can hold it against the
// oracle without a GPU.
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

namespace polydp {

struct Stack {            // pending slice ends: position | x << 10 | y << 17, the nearest on top
    uint32_t e[10];
    __host__ __device__ __forceinline__ void push(uint32_t v) {
#pragma unroll
        for (int i = 9; i > 0; --i) e[i] = e[i - 1];
        e[0] = v;
    }
    __host__ __device__ __forceinline__ void pop() {
#pragma unroll
        for (int i = 0; i < 9; ++i) e[i] = e[i + 1];
    }
};

// the clean-up pass of approx.cpp on the packed vertex list (closed curve, in place); returns the new count
__host__ __device__ __forceinline__ int cleanup(Verts& vs, int count, double eps2) {
    int new_count = count, pos = count - 1, wpos;
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
    return new_count;
}

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

// hover.py:189-213 for the component `bl` (an 8-connected run of non-empty rows), from the span table alone (no contour list:
// the slots are evaluated where they are needed).  Every lane of the group calls it with the same arguments and gets the same
// result.  This is the general routine — any blob, any size — and the slow one: poly_blob below takes the blobs that are one
// piece and whose contour fits its registers, and falls back to this.
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
    const int new_count = cleanup(vs, cnt, eps2);
    if (new_count != 4) return res;                            // hover.py:193
    for (int t = 0; t < 4; ++t) {
        const uint32_t p = vs.get(t);
        res.sx += (int)(p & 255u);
        res.sy += (int)(p >> 8);
    }
    res.vis = 1;
    return res;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The fast path: one blob, one lane group, the contour in registers.
//
// tab[0 .. R): the span table of the blob's rows jlo .. jlo + R - 1 (a | b << 8, 0x00FF for a row without red pixels), with
// tab[-1] = tab[R] = 0x00FF around it, private to the group; list[0 .. W x TRC): scratch, private to the group.  One pass over the rows (row r on lane r mod W) gives the border
// rule (hover.py:180-185), the number of pieces, the integers of SPEC §6b item 5 and cv2's CHAIN_APPROX_SIMPLE contour itself:
// per row and side the point where the path enters the row (kept when it is not the span end: a turn from a diagonal into the
// row), the span end (kept unless the path runs straight through it: S,S / N,N or two diagonal steps of one run) and the point
// where it leaves the row (kept when it is not the span end).  A group scan places the left side's points upwards from list[0]
// and the right side's downwards from list[S - 1]; the P points then sit in registers, point p on lane p mod W.  cv2.arcLength
// = the closed-form count of axis-parallel steps + float32 lengths of the diagonal segments of that contour; approxPolyDP's
// sweeps run over the registers (positions are renumbered from the first vertex, so a slice is a plain range and the
// first-maximum rule is a constant low part of the key), its slice stack lives in registers with the points beside the indices.
// A blob in several pieces, or with more than S / W x TRC contour points, goes through poly_detect above.

struct TabSpan {          // Span policy over a group's private table
    const uint16_t* tab;
    int jlo;
    __host__ __device__ __forceinline__ bool operator()(int row, int& a, int& b) const {
        const uint32_t ab = tab[row - jlo];
        a = (int)(ab & 255u); b = (int)(ab >> 8);
        return a <= b;
    }
};

template <int TRC, class Grp>
__host__ __device__ __forceinline__ PolyResult poly_blob(const Grp& g, const uint16_t* tab, uint16_t* list, const uint32_t* gtab, int jlo, int R) {
    using namespace polydp;
    const int W = g.size(), S = W * TRC - 1;                       // lanes of the group; contour points it can hold (list[S]: a spare entry)
    const int l = g.lane();
    PolyResult res = {0, 0, 0, 0, 0, 0};
    // ---- one pass over the rows: statistics, contour points.  tab[-1] and tab[R] read as "no red pixel".
    int sumA = 0, sumB = 0, sumC = 0, amin = 128, bmax = -1;       // W | S << 14; H | n << 8 | firsts << 16; j0 | jn << 7 | w0 << 14 | wn << 21
    int baseL = 0, baseR = 0;
    for (int t0 = 0; t0 < R; t0 += W) {
        const int r = imin(t0 + l, R);                             // (a lane beyond the last row looks at the empty row R)
        const uint32_t own = tab[r], up = tab[r - 1], dn = tab[r + 1];
        const int a = (int)(own & 255u), b = (int)(own >> 8), pa = (int)(up & 255u), pb = (int)(up >> 8), na = (int)(dn & 255u), nb = (int)(dn >> 8);
        const bool ne = a <= b;
        // a row above / below that does not touch this one (8-connectivity) belongs to another piece
        const bool first = ne & ((pa > pb) | (a > pb + 1) | (b < pa - 1)), last = ne & ((na > nb) | (na > b + 1) | (nb < a - 1));
        const int row = jlo + r;
        const bool below = ne & !last, above = ne & !first;            // the piece goes on below / above this row
        const int da = below ? na - a : 0, db = below ? nb - b : 0;    // to the row below
        const int ua = above ? a - pa : 0, ub = above ? pb - b : 0;    // left end: from the row above; right end: to the row above
        const int one = ne ? 1 : 0, fl = first ? 1 : 0, la = last ? 1 : 0;
        sumA += ne ? (b - a) | (iabs(da) + iabs(db)) << 14 : 0;
        sumB += ((da != 0) + (db != 0)) | one << 8 | fl << 16;
        sumC += (first ? row | (b - a) << 14 : 0) + (last ? row << 7 | (b - a) << 21 : 0);
        amin = ne ? imin(amin, a) : amin;
        bmax = ne ? imax(bmax, b) : bmax;
        // left side, going down: ua = a - a_above, da = a_below - a.  Kept: the point where a diagonal step enters the row short of
        // its end; the end, unless the path runs straight through it; the point where the path leaves the row for a diagonal step
        const int kin = (ua < -1) & one, kout = (da > 1) & one;
        const int kend = one & (fl | la | ((((ua == 0) & (da == 0)) | ((ua == -1) & (da < 0)) | ((ua > 0) & (da == 1))) ^ 1));
        // right side, going up: -db = b - b_below (arriving), ub = b_above - b (leaving)
        const int rin = (db < -1) & one, rout = (ub < -1) & one;
        const int rend = one & (fl | la | ((((db == 0) & (ub == 0)) | ((db == -1) & (ub > 0)) | ((db > 0) & (ub == -1))) ^ 1));
        const int cL = kin + kend + kout, cR = rin + rend + rout;
        const int incl = g.scan_add(cL | cR << 8), tot = g.max_i32(incl);
        const int exL = baseL + (incl & 255) - cL;                 // the left side's points go upwards from list[0]
        const int exR = S - 1 - (baseR + (incl >> 8) - cR);        // the right side's, in the reverse of the path's order, downwards from list[S - 1]
        const uint32_t y = (uint32_t)row << 8;
        // (a point that is not kept, or that no longer fits, goes to the spare entry list[S]: no branch around the stores)
        const int iL0 = exL, iL1 = exL + kin, iL2 = exL + kin + kend, iR0 = exR, iR1 = exR - rout, iR2 = exR - rout - rend;
        list[(kin && iL0 < S) ? iL0 : S] = (uint16_t)((uint32_t)(pa - 1) | y);
        list[(kend && iL1 < S) ? iL1 : S] = (uint16_t)((uint32_t)a | y);
        list[(kout && iL2 < S) ? iL2 : S] = (uint16_t)((uint32_t)(na - 1) | y);
        list[(rout && iR0 >= 0) ? iR0 : S] = (uint16_t)((uint32_t)(pb + 1) | y);
        list[(rend && iR1 >= 0) ? iR1 : S] = (uint16_t)((uint32_t)b | y);
        list[(rin && iR2 >= 0) ? iR2 : S] = (uint16_t)((uint32_t)(nb + 1) | y);
        baseL += tot & 255; baseR += tot >> 8;
    }
    sumA = g.add_i32(sumA); sumB = g.add_i32(sumB); sumC = g.add_i32(sumC);
    amin = -g.max_i32(-amin); bmax = g.max_i32(bmax);
    const int n = (sumB >> 8) & 255, pieces = ((sumB >> 16) & 127);
    if (n == 0) return res;
    // hover.py:180-185 — the whole mask: any red pixel in the first / last row or column
    {
        int jmin = 128, jmax = -1;
        // (first / last rows of the pieces: with one piece sumC holds them; with several the span table's ends decide)
        if (pieces == 1) { jmin = sumC & 127; jmax = (sumC >> 7) & 127; }
        else {
            for (int r = 0; r < R; ++r) if ((tab[r] & 255u) <= (uint32_t)(tab[r] >> 8)) { jmin = jlo + r; break; }
            for (int r = R - 1; r >= 0; --r) if ((tab[r] & 255u) <= (uint32_t)(tab[r] >> 8)) { jmax = jlo + r; break; }
        }
        if (amin == 0 || bmax == 127 || jmin == 0 || jmax == 127) return res;
    }
    const int P = baseL + baseR;
    if (pieces != 1 || P > S) {                                    // several pieces (cv2 takes the largest), or a contour beyond the registers
        const TabSpan span{tab, jlo};
        PolyBlob bl;
        if (pieces == 1) bl = PolyBlob{sumC & 127, (sumC >> 7) & 127, sumA & 16383, sumA >> 14, sumB & 255, amin, bmax};
        else {
            int j0 = jlo, jn = jlo + R - 1;
            bl = poly_largest_component(span, j0, jn);
        }
        return poly_detect(g, span, bl);
    }
    g.fence();                                                     // the list is complete
    const int j0 = sumC & 127, jn = (sumC >> 7) & 127, w0 = (sumC >> 14) & 127, wn = (sumC >> 21) & 127;
    const int Wsum = sumA & 16383, Ssum = sumA >> 14, H = sumB & 255;
    res.area2 = 2 * Wsum - (w0 + wn) - Ssum + H;
    res.w = bmax - amin + 1;
    res.h = jn - j0 + 1;
    // ---- the contour into registers (point p on lane p mod W), and cv2.arcLength from its diagonal segments
    const int shift = S - P;                                       // list position of point p >= baseL: p + shift
    uint32_t pt[TRC];
    int gi = 0, gf = 0;
    // every read of a half of the registers is issued before the first one is used: three LDS round trips per half, not three
    // per block of W points; the second half only when some blob of the wave has that many points
#pragma unroll
    for (int t = 0; t < TRC; ++t) pt[t] = 0u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        constexpr int kHalf = (TRC + 1) / 2;
        if (h == 0 || kHalf * W < P) {
            uint32_t nx[kHalf], gv[kHalf];
#pragma unroll
            for (int t = 0; t < kHalf; ++t) {
                if (h * kHalf + t < TRC) {
                    const int p = (h * kHalf + t) * W + l;
                    int q = p + 1;
                    if (q >= P) q = 0;
                    const bool in = p < P;
                    pt[h * kHalf + t] = list[in ? (p < baseL ? p : p + shift) : 0];
                    nx[t] = list[in ? (q < baseL ? q : q + shift) : 0];
                }
            }
#pragma unroll
            for (int t = 0; t < kHalf; ++t) {
                if (h * kHalf + t < TRC) {
                    const uint32_t me = pt[h * kHalf + t];
                    const int adx = iabs((int)(nx[t] & 255u) - (int)(me & 255u)), ady = iabs((int)(nx[t] >> 8) - (int)(me >> 8));
                    gv[t] = gtab[ady & (0 - (int)((adx == ady) & (adx != 0) & (int)((h * kHalf + t) * W + l < P)))];   // a run of |dy| diagonal steps
                }
            }
#pragma unroll
            for (int t = 0; t < kHalf; ++t) {
                if (h * kHalf + t < TRC) {
                    gi += (int)(gv[t] >> 23);
                    gf += (int)(gv[t] & 0x7FFFFFu);
                }
            }
        }
    }
    gi = g.add_i32(gi);
    gf = g.add_i32(gf);
    const int straight = w0 + wn + Ssum + 2 * (n - 1) - 2 * H;
    const double len = (double)(straight + gi) + (double)gf * (1.0 / 8388608.0);
    const double eps = 0.04 * len;                                 // hover.py:190
    const double eps2 = eps * eps;
    // ---- approxPolyDP, 1: three farthest-point sweeps; positions are cyclic from the sweep's start
    int s2 = 0;
    uint32_t spt = list[0], key = 0u;
    for (int it = 0; it < 3; ++it) {
        const int sx = (int)(spt & 255u), sy = (int)(spt >> 8);
        uint32_t best = 0u;
#pragma unroll
        for (int t = 0; t < TRC; ++t) {
            if (t * W < P) {
                const int p = t * W + l;
                int q = p - s2;
                if (q < 0) q += P;
                const int ex = (int)(pt[t] & 255u) - sx, ey = (int)(pt[t] >> 8) - sy;
                const uint32_t k = ((uint32_t)(ex * ex + ey * ey) << 10) | (uint32_t)(1023 - q);
                best = (p < P && k > best) ? k : best;
            }
        }
        key = g.max_u32(best);
        if (it < 2) {
            s2 += (key >> 10) ? 1023 - (int)(key & 1023u) : 0;
            if (s2 >= P) s2 -= P;
            spt = list[s2 < baseL ? s2 : s2 + shift];
        }
    }
    if ((double)(key >> 10) <= eps2) return res;                   // one vertex
    // From here on positions count from the first vertex s2 (position 0, and P when the curve closes), so no slice wraps.
    const int e = 1023 - (int)(key & 1023u);                       // the second vertex
    uint32_t ept;
    {
        int ap = s2 + e;
        if (ap >= P) ap -= P;
        ept = list[ap < baseL ? ap : ap + shift];
    }
    Verts vs = {0ull, 0ull, 0u};
    int cnt = 0;
    // ---- 2, 3 on the usual course: both slices of the first level in ONE sweep (they share the chord, so a point's distance is
    // the same number for either), then the up to four slices of the second level in one sweep (a slice the first level accepted
    // is looked at again, with the same outcome).  When every second-level slice is accepted the vertices are final: at most
    // four.  Otherwise the slice stack below does it all again, to any depth.
    bool regular;
    {
        const int sx = (int)(spt & 255u), sy = (int)(spt >> 8), ex = (int)(ept & 255u), ey = (int)(ept >> 8);
        const int dxA = ex - sx, dyA = ey - sy, cA = sy * dxA - sx * dyA;
        uint32_t kA = 0u, kB = 0u;
#pragma unroll
        for (int t = 0; t < TRC; ++t) {
            if (t * W < P) {
                const int p = t * W + l;
                int q = p - s2;
                if (q < 0) q += P;
                const int d = iabs((int)(pt[t] >> 8) * dxA - (int)(pt[t] & 255u) * dyA - cA);
                const uint32_t k = p < P ? ((uint32_t)d << 10) | (uint32_t)(1023 - q) : 0u;
                kA = (q < e && k > kA) ? k : kA;
                kB = (q > e && k > kB) ? k : kB;
            }
        }
        kA = g.max_u32(kA);
        kB = g.max_u32(kB);
        const double limA = eps2 * (double)(dxA * dxA + dyA * dyA), mdA = (double)(kA >> 10), mdB = (double)(kB >> 10);
        const bool splitA = mdA * mdA > limA, splitB = mdB * mdB > limA;
        const int mA = splitA ? 1023 - (int)(kA & 1023u) : 0, mB = splitB ? 1023 - (int)(kB & 1023u) : e;
        int apA = s2 + mA, apB = s2 + mB;
        if (apA >= P) apA -= P;
        if (apB >= P) apB -= P;
        const uint32_t ptA = list[apA < baseL ? apA : apA + shift], ptB = list[apB < baseL ? apB : apB + shift];
        // chords 0: s2 -> mA, 1: mA -> e, 2: e -> mB, 3: mB -> s2 (a level-one slice that was accepted: an empty chord and itself)
        const int ax = (int)(ptA & 255u), ay = (int)(ptA >> 8), bx = (int)(ptB & 255u), by = (int)(ptB >> 8);
        const int dx0 = ax - sx, dy0 = ay - sy, dx1 = ex - ax, dy1 = ey - ay, dx2 = bx - ex, dy2 = by - ey, dx3 = sx - bx, dy3 = sy - by;
        const int c0 = sy * dx0 - sx * dy0, c1 = ay * dx1 - ax * dy1, c2 = ey * dx2 - ex * dy2, c3 = by * dx3 - bx * dy3;
        uint32_t k0 = 0u, k1 = 0u, k2 = 0u, k3 = 0u;
#pragma unroll
        for (int t = 0; t < TRC; ++t) {
            if (t * W < P) {
                const int p = t * W + l;
                int q = p - s2;
                if (q < 0) q += P;
                const bool hi = q > e, in1 = q > mA, in3 = q > mB;      // slice: hi ? (in3 ? 3 : 2) : (in1 ? 1 : 0)
                const int dx = hi ? (in3 ? dx3 : dx2) : (in1 ? dx1 : dx0), dy = hi ? (in3 ? dy3 : dy2) : (in1 ? dy1 : dy0);
                const int cc = hi ? (in3 ? c3 : c2) : (in1 ? c1 : c0);
                const int d = iabs((int)(pt[t] >> 8) * dx - (int)(pt[t] & 255u) * dy - cc);
                const uint32_t k = p < P ? ((uint32_t)d << 10) | (uint32_t)(1023 - q) : 0u;
                const uint32_t klo = hi ? 0u : k, khi = hi ? k : 0u;
                k0 = (!in1 && klo > k0) ? klo : k0;
                k1 = (in1 && klo > k1) ? klo : k1;
                k2 = (!in3 && khi > k2) ? khi : k2;
                k3 = (in3 && khi > k3) ? khi : k3;
            }
        }
        k0 = g.max_u32(k0); k1 = g.max_u32(k1); k2 = g.max_u32(k2); k3 = g.max_u32(k3);
        const double m0 = (double)(k0 >> 10), m1 = (double)(k1 >> 10), m2 = (double)(k2 >> 10), m3 = (double)(k3 >> 10);
        const bool a0 = m0 * m0 <= eps2 * (double)(dx0 * dx0 + dy0 * dy0), a1 = m1 * m1 <= eps2 * (double)(dx1 * dx1 + dy1 * dy1);
        const bool a2 = m2 * m2 <= eps2 * (double)(dx2 * dx2 + dy2 * dy2), a3 = m3 * m3 <= eps2 * (double)(dx3 * dx3 + dy3 * dy3);
        regular = a0 && a1 && a2 && a3;
        if (regular) {
            cnt = 2 + (splitA ? 1 : 0) + (splitB ? 1 : 0);
            if (cnt != 4) return res;                              // two or three vertices stay two or three (hover.py:193)
            // The clean-up pass removes a vertex as soon as one of the four triples of successive vertices qualifies — and then
            // there are three left.  The triples are tested side by side, one per lane.
            const int i = l & 3;
            const uint32_t vp = i == 0 ? ptB : (i == 1 ? spt : (i == 2 ? ptA : ept));          // vertex i - 1
            const uint32_t vc = i == 0 ? spt : (i == 1 ? ptA : (i == 2 ? ept : ptB));          // vertex i
            const uint32_t vn = i == 0 ? ptA : (i == 1 ? ept : (i == 2 ? ptB : spt));          // vertex i + 1
            const int stx = (int)(vp & 255u), sty = (int)(vp >> 8), px = (int)(vc & 255u), py = (int)(vc >> 8);
            const int enx = (int)(vn & 255u), eny = (int)(vn >> 8), ddx = enx - stx, ddy = eny - sty;
            const double dist = (double)iabs((px - stx) * ddy - (py - sty) * ddx);
            const int sip = (px - stx) * (enx - px) + (py - sty) * (eny - py);
            const bool drop = dist * dist <= 0.5 * eps2 * (double)(ddx * ddx + ddy * ddy) && ddx != 0 && ddy != 0 && sip >= 0;
            if (g.max_i32(drop ? 1 : 0) != 0) return res;
            res.sx = sx + ax + ex + bx;
            res.sy = sy + ay + ey + by;
            res.vis = 1;
            return res;
        }
        // ---- 2, 3 in general: what the two levels found goes onto a stack of pending slice ends (the nearest on top) — each with
        // its point, and marked when the slice that ends there is already known to be accepted; a rejected slice gets its split
        // point on top of its end.  Then the slices are worked left to right, to any depth.
        int apS[4] = {s2 + 1023 - (int)(k0 & 1023u), s2 + 1023 - (int)(k1 & 1023u), s2 + 1023 - (int)(k2 & 1023u), s2 + 1023 - (int)(k3 & 1023u)};
        uint32_t ptS[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (apS[i] >= P) apS[i] -= P;
            if (apS[i] >= P) apS[i] = 0;                           // (an untouched key: the entry is not used)
            ptS[i] = list[apS[i] < baseL ? apS[i] : apS[i] + shift];
        }
        Stack st;
#pragma unroll
        for (int i = 0; i < 10; ++i) st.e[i] = 0u;
        int depth = 0;
#define POLY_ENTRY(pos_, pt_, known_) ((uint32_t)(pos_) | ((pt_) & 255u) << 10 | ((pt_) >> 8) << 17 | ((known_) ? 1u << 24 : 0u))
        st.push(POLY_ENTRY(P, spt, splitB ? a3 : true)); ++depth;                          // position P = the first vertex again
        if (splitB) {
            if (!a3) { st.push(POLY_ENTRY(1023 - (int)(k3 & 1023u), ptS[3], false)); ++depth; }
            st.push(POLY_ENTRY(mB, ptB, a2)); ++depth;
            if (!a2) { st.push(POLY_ENTRY(1023 - (int)(k2 & 1023u), ptS[2], false)); ++depth; }
        }
        st.push(POLY_ENTRY(e, ept, splitA ? a1 : true)); ++depth;
        if (splitA) {
            if (!a1) { st.push(POLY_ENTRY(1023 - (int)(k1 & 1023u), ptS[1], false)); ++depth; }
            st.push(POLY_ENTRY(mA, ptA, a0)); ++depth;
            if (!a0) { st.push(POLY_ENTRY(1023 - (int)(k0 & 1023u), ptS[0], false)); ++depth; }
        }
#undef POLY_ENTRY
        int s = 0;
        uint32_t cur = spt;                                                            // the slice's start point
        while (depth > 0) {
            const uint32_t top = st.e[0];
            const int en = (int)(top & 1023u), ex = (int)((top >> 10) & 127u), ey = (int)((top >> 17) & 127u);
            const int sx2 = (int)(cur & 255u), sy2 = (int)(cur >> 8), dx = ex - sx2, dy = ey - sy2;
            const uint32_t between = (uint32_t)(en - s - 1);
            bool accept = true;
            uint32_t kb = 0u;
            if (between > 0u && (top >> 24) == 0u) {
                const int c0 = sy2 * dx - sx2 * dy;                                    // (y - sy) dx - (x - sx) dy = y dx - x dy - c0
#pragma unroll
                for (int t = 0; t < TRC; ++t) {
                    if (t * W < P) {
                        int q = t * W + l - s2;                                        // position from the first vertex
                        if (q < 0) q += P;
                        const int d = iabs((int)(pt[t] >> 8) * dx - (int)(pt[t] & 255u) * dy - c0);
                        const uint32_t k = ((uint32_t)d << 10) | (uint32_t)(1023 - q);
                        const bool inside = (uint32_t)(q - s - 1) < between && t * W + l < P;
                        kb = (inside && k > kb) ? k : kb;
                    }
                }
                kb = g.max_u32(kb);
                const double md = (double)(kb >> 10);
                accept = md * md <= eps2 * (double)(dx * dx + dy * dy);
            }
            if (accept) {
                if (cnt >= 9) return res;
                vs.set(cnt++, cur);
                s = en;
                cur = (uint32_t)ex | (uint32_t)ey << 8;
                st.pop();
                --depth;
            } else {
                const int m = 1023 - (int)(kb & 1023u);                                // position of the split point
                int ap = m + s2;
                if (ap >= P) ap -= P;
                const uint32_t mp = list[ap < baseL ? ap : ap + shift];
                st.push((uint32_t)m | (mp & 255u) << 10 | (mp >> 8) << 17);
                ++depth;
                if (cnt + depth > 9) return res;               // ten or more vertices: never four after the clean-up pass
            }
        }
    }
    // ---- 4: the clean-up pass
    const int new_count = cleanup(vs, cnt, eps2);
    if (new_count != 4) return res;                                // hover.py:193
    for (int t = 0; t < 4; ++t) {
        const uint32_t p4 = vs.get(t);
        res.sx += (int)(p4 & 255u);
        res.sy += (int)(p4 >> 8);
    }
    res.vis = 1;
    return res;
}

// cv2.arcLength's float32 length of a run of k equal diagonal steps in units of 2^-23 (exact: a float below 256): the table
// poly_blob takes as gtab[0 .. 127]
__host__ __device__ __forceinline__ uint32_t poly_run_length_q23(int k) {
    return (uint32_t)((double)(float)((double)k * 1.4142135623730951) * 8388608.0);
}

}  // namespace fpv

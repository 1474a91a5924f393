/*
 * cv_restate.c — see cv_restate.h.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (opencv-python 4.11.0.86 of
 * simulation/uv.lock:655-656 is neither in /root/reference nor in this image): the published algorithms restated,
 * call sites simulation/hover.py:187-213.
 */
#include "cv_restate.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ cv2.findContours (hover.py:187)
 * Suzuki & Abe 1985, "Topological structural analysis of digitized binary images by border following", as OpenCV's
 * contours.cpp runs it: the image gets a one-pixel frame of zeros, the raster scan starts an OUTER border where a pixel
 * of value 1 follows a 0 in its row, RETR_EXTERNAL keeps it only when the last border pixel passed in this row is not a
 * left-hand border of an already traced contour (mark > 0), the border is followed through the 8-neighbourhood with
 * direction codes 0 = E, 1 = NE, 2 = N, ... 7 = SE (y down), traced pixels are marked 2, or 2 | -128 where the east
 * neighbour was found empty, and CHAIN_APPROX_SIMPLE writes a point wherever the direction code changes (the start
 * point always: it is entered with code s ^ 4 of its last neighbour, never the code it is left with). */
static const int CODE_DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int CODE_DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

static int fetch_contour(int8_t* i0, int step, int ox, int oy, cvr_point* pts, int cap) {
    const int8_t nbd = 2;
    int deltas[16], n = 0;
    for (int i = 0; i < 8; ++i) deltas[i] = deltas[i + 8] = CODE_DX[i] + CODE_DY[i] * step;
    cvr_point pt = {ox, oy};
    int8_t *i1, *i3, *i4 = 0;
    int s = 4, s_end = 4;                                  /* an outer border: the search starts at W */
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);
    if (s == s_end) {                                      /* a single pixel */
        *i0 = (int8_t)(nbd | -128);
        if (n >= cap) return -1;
        pts[n++] = pt;
        return n;
    }
    i3 = i0;
    int prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        while (s < 15) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (int8_t)(nbd | -128);      /* the east neighbour was examined and empty */
        else if (*i3 == 1) *i3 = nbd;
        if (s != prev_s) {                                 /* CHAIN_APPROX_SIMPLE: the direction changes here */
            if (n >= cap) return -1;
            pts[n++] = pt;
            prev_s = s;
        }
        pt.x += CODE_DX[s];
        pt.y += CODE_DY[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
    return n;
}

int cvr_find_contours_external(const uint8_t* mask, int w, int h, cvr_point* pts, int cap_pts, int32_t* start,
                               int cap_contours) {
    const int step = w + 2;
    int8_t* img = (int8_t*)calloc((size_t)(h + 2) * (size_t)step, 1);
    if (!img) return -1;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) img[(y + 1) * step + x + 1] = mask[y * w + x] ? 1 : 0;
    int nc = 0, np = 0, fail = 0;
    for (int y = 1; y <= h && !fail; ++y) {
        int prev = 0, lnbd_x = 0;
        for (int x = 1; x <= w; ++x) {
            int p = img[y * step + x];
            if (p == prev) continue;
            int traced = 0;
            if (prev == 0 && p == 1) {                                      /* candidate start of an outer border */
                if (!(img[y * step + lnbd_x] > 0)) {                        /* RETR_EXTERNAL: not inside a traced contour */
                    if (nc >= cap_contours) { fail = 1; break; }
                    int n = fetch_contour(img + y * step + x, step, x - 1, y - 1, pts + np, cap_pts - np);
                    if (n < 0) { fail = 1; break; }
                    start[nc++] = np;
                    np += n;
                    lnbd_x = x;
                    traced = 1;
                }
            }
            /* (a hole border — p == 0 behind a pixel >= 1 — is never kept by RETR_EXTERNAL) */
            prev = traced ? img[y * step + x] : p;
            if (prev & -2) lnbd_x = x;
        }
    }
    free(img);
    if (fail) return -1;
    start[nc] = np;
    /* cv2 hands the list back with the contour found last in front */
    cvr_point* tmp = (cvr_point*)malloc(sizeof(cvr_point) * (size_t)(np > 0 ? np : 1));
    int32_t* ts = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nc + 1));
    if (!tmp || !ts) { free(tmp); free(ts); return -1; }
    int q = 0;
    for (int c = nc - 1; c >= 0; --c) {
        ts[nc - 1 - c] = q;
        memcpy(tmp + q, pts + start[c], sizeof(cvr_point) * (size_t)(start[c + 1] - start[c]));
        q += start[c + 1] - start[c];
    }
    ts[nc] = q;
    memcpy(pts, tmp, sizeof(cvr_point) * (size_t)np);
    memcpy(start, ts, sizeof(int32_t) * (size_t)(nc + 1));
    free(tmp); free(ts);
    return nc;
}

/* ------------------------------------------------------------------ cv2.arcLength(contour, True) (hover.py:190)
 * shapedescr.cpp: every segment's length is taken in SINGLE precision (Point2f differences, sqrt of a float) and added
 * into a double. */
double cvr_arc_length_closed(const cvr_point* pts, int n) {
    if (n <= 1) return 0.0;
    double perimeter = 0.0;
    float px = (float)pts[n - 1].x, py = (float)pts[n - 1].y;
    for (int i = 0; i < n; ++i) {
        float x = (float)pts[i].x, y = (float)pts[i].y;
        float dx = x - px, dy = y - py;
        perimeter += sqrtf(dx * dx + dy * dy);
        px = x; py = y;
    }
    return perimeter;
}

/* ------------------------------------------------------------------ cv2.contourArea(contour) (hover.py:189, :206) */
double cvr_contour_area(const cvr_point* pts, int n) {
    if (n == 0) return 0.0;
    double a00 = 0.0;
    float px = (float)pts[n - 1].x, py = (float)pts[n - 1].y;
    for (int i = 0; i < n; ++i) {
        float x = (float)pts[i].x, y = (float)pts[i].y;
        a00 += (double)px * y - (double)py * x;
        px = x; py = y;
    }
    return fabs(a00 * 0.5);
}

/* ------------------------------------------------------------------ cv2.boundingRect(contour) (hover.py:209) */
void cvr_bounding_rect(const cvr_point* pts, int n, int32_t xywh[4]) {
    if (n == 0) { xywh[0] = xywh[1] = xywh[2] = xywh[3] = 0; return; }
    int xmin = pts[0].x, xmax = pts[0].x, ymin = pts[0].y, ymax = pts[0].y;
    for (int i = 1; i < n; ++i) {
        if (pts[i].x < xmin) xmin = pts[i].x;
        if (pts[i].x > xmax) xmax = pts[i].x;
        if (pts[i].y < ymin) ymin = pts[i].y;
        if (pts[i].y > ymax) ymax = pts[i].y;
    }
    xywh[0] = xmin; xywh[1] = ymin; xywh[2] = xmax - xmin + 1; xywh[3] = ymax - ymin + 1;
}

/* ------------------------------------------------------------------ cv2.approxPolyDP(contour, eps, True) (hover.py:191)
 * approx.cpp, approxPolyDP_<int> on a closed curve: (1) three farthest-point sweeps pick the two start vertices (each sweep
 * starts at the point the previous one found farthest; strict ">" keeps the first maximum in contour order), (2) the two
 * slices between them go on a stack, (3) a slice is accepted when no point between its ends lies farther than eps from
 * the chord — compared as max_dist^2 <= eps^2 * |chord|^2 — and is split at its farthest point otherwise; an accepted
 * slice writes its start point, (4) the clean-up pass drops the middle one of three successive output points when it
 * lies within eps / sqrt(2) of the line through its neighbours, that line is neither horizontal nor vertical and the
 * turn is not a reversal — in place, so the wrap-around at the end reads what the pass itself wrote. */
typedef struct { int start, end; } cvr_range;

int cvr_approx_poly_dp_closed(const cvr_point* src, int count, double eps, cvr_point* dst) {
    if (count == 0) return 0;
    cvr_range* stack = (cvr_range*)malloc(sizeof(cvr_range) * (size_t)(2 * count + 8));
    if (!stack) return -1;
    int top = 0, new_count = 0, pos = 0;
    cvr_range slice = {0, 0}, right_slice = {0, 0};
    cvr_point start_pt = {-1000000, -1000000}, end_pt = {0, 0}, pt = {0, 0};
    int le_eps = 0;
    eps *= eps;
#define READ_PT(p_, pos_) do { (p_) = src[pos_]; if (++(pos_) >= count) (pos_) = 0; } while (0)
    for (int it = 0; it < 3; ++it) {
        double max_dist = 0.0;
        pos = (pos + right_slice.start) % count;
        READ_PT(start_pt, pos);
        for (int j = 1; j < count; ++j) {
            READ_PT(pt, pos);
            double dx = pt.x - start_pt.x, dy = pt.y - start_pt.y;
            double dist = dx * dx + dy * dy;
            if (dist > max_dist) { max_dist = dist; right_slice.start = j; }
        }
        le_eps = max_dist <= eps;
    }
    if (!le_eps) {
        right_slice.end = slice.start = pos % count;
        slice.end = right_slice.start = (right_slice.start + slice.start) % count;
        stack[top++] = right_slice;
        stack[top++] = slice;
    } else {
        dst[new_count++] = start_pt;
    }
    while (top > 0) {
        slice = stack[--top];
        end_pt = src[slice.end];
        pos = slice.start;
        READ_PT(start_pt, pos);
        if (pos != slice.end) {
            double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y, max_dist = 0.0;
            while (pos != slice.end) {
                READ_PT(pt, pos);
                double dist = fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
                if (dist > max_dist) { max_dist = dist; right_slice.start = (pos + count - 1) % count; }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = 1;
            start_pt = src[slice.start];
        }
        if (le_eps) {
            dst[new_count++] = start_pt;
        } else {
            right_slice.end = slice.end;
            slice.end = right_slice.start;
            stack[top++] = right_slice;
            stack[top++] = slice;
        }
    }
#undef READ_PT
    free(stack);
    /* the clean-up pass, closed curve */
    count = new_count;
    pos = count - 1;
#define READ_DST_PT(p_, pos_) do { (p_) = dst[pos_]; if (++(pos_) >= count) (pos_) = 0; } while (0)
    READ_DST_PT(start_pt, pos);
    int wpos = pos;
    READ_DST_PT(pt, pos);
    for (int i = 0; i < count && new_count > 2; ++i) {
        READ_DST_PT(end_pt, pos);
        double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
        double dist = fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
        double sip = (double)(pt.x - start_pt.x) * (end_pt.x - pt.x) + (double)(pt.y - start_pt.y) * (end_pt.y - pt.y);
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
            new_count--;
            dst[wpos] = start_pt = end_pt;
            if (++wpos >= count) wpos = 0;
            READ_DST_PT(pt, pos);
            i++;
            continue;
        }
        dst[wpos] = start_pt = pt;
        if (++wpos >= count) wpos = 0;
        pt = end_pt;
    }
#undef READ_DST_PT
    return new_count;
}

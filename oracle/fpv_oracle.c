/*
 * fpv_oracle.c — CPU restatement of simulation/hover.py's step()/reset() path plus the
 * docs/SPEC.md physics.  TEST INFRASTRUCTURE ONLY (see fpv_oracle.h header comment).
 * PARITY UNPINNED versus PyFlyt/pybullet; hover.py-level rules are pinned by the KATs of
 * SURVEY.md §8c (tests/test_oracle_kat.py).
 */
#include "fpv_oracle.h"
#include "cv_restate.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ constants (SPEC §1) */
static const double MASS = 0.1;                       /* cf2x.urdf:10 */
static const double INERTIA[3] = {3e-5, 3e-5, 5e-5};  /* cf2x.urdf:12 */
static const double PROP_X[4] = {0.028, -0.028, 0.028, -0.028}; /* cf2x.urdf:33-76 */
static const double PROP_Y[4] = {-0.028, 0.028, 0.028, -0.028};
static const double SPIN[4] = {-1.0, -1.0, 1.0, 1.0};
static const double F_MAX = 1.0;                      /* total_thrust/4, cf2x.yaml:2 */
static const double KQ_OVER_KF = 7.94e-12 / 3.16e-10; /* cf2x.yaml:3-4 */
static const double TAU = 0.01;                       /* cf2x.yaml:6 */
static const double K_LIN = 0.5 * 1.225 * 1.5 * 3.0e-4; /* cf2x.yaml:9-10 */
static const double K_PQR = 1.0e-4;                   /* cf2x.yaml:11 */
static const double PID_KP[3] = {2.0e-2, 2.0e-2, 4.0e-2}; /* cf2x.yaml:16-19 */
static const double PID_KI[3] = {2.5e-7, 2.5e-7, 1.35e-4};
static const double PID_KD[3] = {5.0e-5, 5.0e-5, 0.0};
static const double PID_LIM[3] = {1.0, 1.0, 1.0};
static const double GRAV = 9.81;
static const double DT = 1.0 / 240.0;                 /* hover.py:23 */
static const double T_CTRL = 1.0 / 120.0;
static const int SUBSTEPS = 12;                       /* hover.py:24 x 2 */
static const double AGENT_DT = 1.0 / 40.0;            /* hover.py:25 */
static const double MAX_STEPS = 400.0;                /* hover.py:35 */
static const double DOME = 3.0;                       /* hover.py:37 */
static const double FLOOR = 0.1;                      /* hover.py:38 */
static const double TARGET_AREA = 0.013;              /* hover.py:50 */
static const double TARGET_RATIO = 1.53;              /* hover.py:51 */
static const double Z_REST = 0.01;                    /* half of cf2x.urdf:29 box height */
static const double PWM_IDLE = 0.05;
static const double V_CLAMP = 50.0, W_CLAMP = 100.0;
static const double PANEL_X = 4.98, PANEL_ZC = 6.0;   /* hover.py:116-155 */
static const double CAM_TILT_DEG = 25.0;              /* hover.py:85 (sign: SPEC §1) */

static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* ------------------------------------------------------------------ Philox (SPEC §9) */
void fpv_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void fpv_oracle_normal4(uint64_t seed, uint64_t drone_id, uint64_t tick, double g[4]) {
    /* SPEC §9: one Philox call per sub-step pair, eight 16-bit uniforms; tick = agent_tick*12 + k */
    uint64_t pair = tick >> 1;
    uint32_t ctr[4] = {(uint32_t)drone_id, (uint32_t)pair, (uint32_t)(pair >> 32), 0x46505631u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t x[4];
    fpv_oracle_philox4x32_10(ctr, key, x);
    uint32_t h[8];
    for (int j = 0; j < 4; ++j) { h[2 * j] = x[j] & 0xFFFFu; h[2 * j + 1] = x[j] >> 16; }
    const uint32_t* hh = h + 4 * (tick & 1);
    double u[4];
    for (int j = 0; j < 4; ++j) u[j] = ((double)hh[j] + 0.5) * (1.0 / 65536.0);
    double ra = sqrt(-2.0 * log(u[0])), rb = sqrt(-2.0 * log(u[2]));
    g[0] = ra * cos(2.0 * M_PI * u[1]);
    g[1] = ra * sin(2.0 * M_PI * u[1]);
    g[2] = rb * cos(2.0 * M_PI * u[3]);
    g[3] = rb * sin(2.0 * M_PI * u[3]);
}

/* ------------------------------------------------------------------ attitude (SPEC §5) */
double fpv_oracle_wrap(double d) {
    /* hover.py:229  (d + pi) % (2 pi) - pi, Python modulo */
    double m = d + M_PI;
    if (m < 0.0) m += 2.0 * M_PI;
    else if (m >= 2.0 * M_PI) m -= 2.0 * M_PI;
    return m - M_PI;
}

void fpv_oracle_quat_from_euler(const double e[3], double q[4]) {
    /* hover.py:233 pybullet.getQuaternionFromEuler — ZYX, xyzw */
    double sr = sin(0.5 * e[0]), cr = cos(0.5 * e[0]);
    double sp = sin(0.5 * e[1]), cp = cos(0.5 * e[1]);
    double sy = sin(0.5 * e[2]), cy = cos(0.5 * e[2]);
    q[0] = sr * cp * cy - cr * sp * sy;
    q[1] = cr * sp * cy + sr * cp * sy;
    q[2] = cr * cp * sy - sr * sp * cy;
    q[3] = cr * cp * cy + sr * sp * sy;
}

void fpv_oracle_euler_from_quat(const double q[4], double e[3]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double s = -2.0 * (x * z - w * y);
    if (s <= -0.99999) {
        e[0] = 0.0; e[1] = -0.5 * M_PI; e[2] = 2.0 * atan2(x, -y);
    } else if (s >= 0.99999) {
        e[0] = 0.0; e[1] = 0.5 * M_PI; e[2] = 2.0 * atan2(-x, y);
    } else {
        e[0] = atan2(2.0 * (y * z + w * x), w * w - x * x - y * y + z * z);
        e[1] = asin(s);
        e[2] = atan2(2.0 * (x * y + w * z), w * w + x * x - y * y - z * z);
    }
}

void fpv_oracle_scale_action(const double a[4], double sp[4]) {
    /* hover.py:337-341 */
    sp[0] = a[0] * 30.0;
    sp[1] = a[1] * 30.0;
    sp[2] = a[2] * -30.0;
    sp[3] = (a[3] + 1.0) / 2.0;
}

static void rotmat(const double q[4], double R[3][3]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0][0] = 1.0 - 2.0 * (y * y + z * z); R[0][1] = 2.0 * (x * y - w * z); R[0][2] = 2.0 * (x * z + w * y);
    R[1][0] = 2.0 * (x * y + w * z); R[1][1] = 1.0 - 2.0 * (x * x + z * z); R[1][2] = 2.0 * (y * z - w * x);
    R[2][0] = 2.0 * (x * z - w * y); R[2][1] = 2.0 * (y * z + w * x); R[2][2] = 1.0 - 2.0 * (x * x + y * y);
}

/* ------------------------------------------------------------------ camera (SPEC §6) */
int fpv_oracle_camera(const double pos[3], const double quat[4], double center[2], double* area,
                      double* ratio) {
    static const double CY[4] = {1.0, -1.0, -1.0, 1.0};
    static const double CZ[4] = {-1.0, -1.0, 1.0, 1.0};
    double R[3][3];
    rotmat(quat, R);
    double c = cos(CAM_TILT_DEG * M_PI / 180.0), s = sin(CAM_TILT_DEG * M_PI / 180.0);
    double u[4], v[4];
    int ok = 1;
    for (int j = 0; j < 4; ++j) {
        double dw[3] = {PANEL_X - pos[0], CY[j] - pos[1], PANEL_ZC + CZ[j] - pos[2]};
        double db[3];
        for (int i = 0; i < 3; ++i) db[i] = R[0][i] * dw[0] + R[1][i] * dw[1] + R[2][i] * dw[2];
        double depth = c * db[0] + s * db[2];
        double up = -s * db[0] + c * db[2];
        double right = -db[1];
        if (!(depth > 0.1)) { ok = 0; depth = 1.0; }
        u[j] = 64.0 + 64.0 * right / depth;
        v[j] = 64.0 - 64.0 * up / depth;
        if (!(u[j] >= 1.0 && u[j] <= 127.0 && v[j] >= 1.0 && v[j] <= 127.0)) ok = 0;
    }
    double sh = 0.0, umin = u[0], umax = u[0], vmin = v[0], vmax = v[0];
    for (int j = 0; j < 4; ++j) {
        int n = (j + 1) & 3;
        sh += u[j] * v[n] - u[n] * v[j];
        if (u[j] < umin) umin = u[j];
        if (u[j] > umax) umax = u[j];
        if (v[j] < vmin) vmin = v[j];
        if (v[j] > vmax) vmax = v[j];
    }
    double apx = 0.5 * fabs(sh);
    if (!(apx >= 1.0)) ok = 0;
    if (!ok) {
        center[0] = center[1] = 0.0; *area = 0.0; *ratio = 0.0;
        return 0;
    }
    center[0] = 0.25 * (u[0] + u[1] + u[2] + u[3]) / 64.0 - 1.0;
    center[1] = 0.25 * (v[0] + v[1] + v[2] + v[3]) / 64.0 - 1.0;
    *area = apx / 16384.0;
    double h = vmax - vmin;
    *ratio = h > 0.0 ? (umax - umin) / h : 0.0;
    return 1;
}

/* ------------------------------------------------------------------ pixel detector (SPEC §6b)
 * hover.py:157-222 on a 128 x 128 rendering of the panel.  The renderer (pybullet) is not on disk, so the mask is stated
 * directly: a pixel is red when its centre (i + 0.5, j + 0.5) lies in the projected front face.  From the mask on this is
 * the reference's own sequence of calls — the border test (:180-185), cv2.findContours(RETR_EXTERNAL,
 * CHAIN_APPROX_SIMPLE) (:187), max by cv2.contourArea (:189), epsilon = 0.04 * cv2.arcLength (:190), cv2.approxPolyDP
 * (:191), "four vertices or invisible" (:193), the mean of the four vertices (:197-203), contourArea / 16384 (:206),
 * cv2.boundingRect's w / h (:209-213) — through the restatement of those calls in cv_restate.c (opencv-python is absent:
 * PARITY UNPINNED for cv2 and for the renderer).  The reference's own constants pin the conventions: from the reset pose
 * this gives w / h = 20 / 13 = 1.538 and area = 207 / 16384 = 0.01263, i.e. hover.py:50-51's target_ratio = 1.53 and
 * target_area = 0.013. */
static int row_span(const double u[4], const double v[4], const double slope[4], int j, int* a, int* b) {
    double y = j + 0.5, xl = 0.0, xr = 0.0;
    int n = 0;
    for (int k = 0; k < 4; ++k) {
        int m = (k + 1) & 3;
        if ((v[k] <= y) != (v[m] <= y)) {                 /* the edge crosses this row of pixel centres */
            double x = u[k] + (y - v[k]) * slope[k];
            if (n == 0) { xl = xr = x; }
            else { if (x < xl) xl = x; if (x > xr) xr = x; }
            ++n;
        }
    }
    if (n < 2) return 0;
    double fa = ceil(xl - 0.5), fb = floor(xr - 0.5);
    if (fa < 0.0) fa = 0.0;
    if (fb > 127.0) fb = 127.0;
    if (fa > fb) return 0;
    *a = (int)fa; *b = (int)fb;
    return 1;
}

/* the red mask of the view from (pos, quat), drawn into an all-zero mask[j * 128 + i]; returns 0 when a corner lies behind
 * the camera (nothing is drawn then).  y0 .. y1: the rows that received red pixels (y0 > y1: none).
 * spans (may be NULL): [128][2] = (a, b) per row, -1 for rows without a red pixel */
static int panel_mask(const double pos[3], const double quat[4], uint8_t* mask, double* spans, int* y0, int* y1) {
    static const double CY[4] = {1.0, -1.0, -1.0, 1.0};
    static const double CZ[4] = {-1.0, -1.0, 1.0, 1.0};
    double R[3][3];
    rotmat(quat, R);
    double c = cos(CAM_TILT_DEG * M_PI / 180.0), s = sin(CAM_TILT_DEG * M_PI / 180.0);
    double u[4], v[4];
    int ok = 1;
    *y0 = 128; *y1 = -1;
    if (spans) for (int i = 0; i < 256; ++i) spans[i] = -1.0;
    for (int j = 0; j < 4; ++j) {
        double dw[3] = {PANEL_X - pos[0], CY[j] - pos[1], PANEL_ZC + CZ[j] - pos[2]};
        double db[3];
        for (int i = 0; i < 3; ++i) db[i] = R[0][i] * dw[0] + R[1][i] * dw[1] + R[2][i] * dw[2];
        double depth = c * db[0] + s * db[2];
        double up = -s * db[0] + c * db[2];
        double right = -db[1];
        if (!(depth > 0.1)) { ok = 0; depth = 1.0; }
        double k = 64.0 / depth;
        u[j] = 64.0 + k * right;
        v[j] = 64.0 - k * up;
    }
    if (!ok) return 0;
    double slope[4];                                      /* du / dv of the four edges (unused when an edge is level) */
    for (int j = 0; j < 4; ++j) {
        int m = (j + 1) & 3;
        slope[j] = v[m] != v[j] ? (u[m] - u[j]) / (v[m] - v[j]) : 0.0;
    }
    double vmin = v[0], vmax = v[0];
    for (int j = 1; j < 4; ++j) { if (v[j] < vmin) vmin = v[j]; if (v[j] > vmax) vmax = v[j]; }
    double flo = ceil(vmin - 0.5), fhi = floor(vmax - 0.5);
    if (flo < 0.0) flo = 0.0;
    if (fhi > 127.0) fhi = 127.0;
    if (!(flo <= fhi)) return 1;
    for (int j = (int)flo; j <= (int)fhi; ++j) {
        int a, b;
        if (!row_span(u, v, slope, j, &a, &b)) continue;               /* (a sliver between two pixel centres) */
        if (spans) { spans[2 * j] = a; spans[2 * j + 1] = b; }
        memset(mask + 128 * j + a, 1, (size_t)(b - a + 1));
        if (j < *y0) *y0 = j;
        *y1 = j;
    }
    return 1;
}

/* hover.py:157-222 from the mask on, for a mask whose red pixels all lie in rows y0 .. y1 (the scan of findContours is
 * confined to those rows: the rows above and below hold nothing to find).  dbg (may be NULL, FPV_ORACLE_PIXEL_DBG doubles):
 * [256..263] the four vertices, [264] arcLength, [265] contourArea, [266] len(approx), [267] len(contours), [268] points of the
 * chosen contour, then the contour's points (x, y) from [272] (at most 512 of them) */
static int detect_rows(const uint8_t* mask, int y0, int y1, double center[2], double* area, double* ratio, double* dbg) {
    enum { CAP_PTS = 4 * 128 * 128, CAP_C = 128 * 128 / 2 + 1 };
    static __thread cvr_point pts[CAP_PTS];
    static __thread int32_t start[CAP_C + 1];
    static __thread cvr_point approx[CAP_PTS];
    center[0] = center[1] = 0.0; *area = 0.0; *ratio = 0.0;
    if (y0 > y1) { if (dbg) dbg[267] = 0; return 0; }                              /* no red pixel at all */
    int red_at_edges = 0;                                                          /* hover.py:180-185 */
    for (int i = 0; i < 128; ++i)
        if (mask[i] || mask[127 * 128 + i]) red_at_edges = 1;
    for (int j = y0; j <= y1; ++j)
        if (mask[128 * j] || mask[128 * j + 127]) red_at_edges = 1;
    int nc = cvr_find_contours_external(mask + 128 * y0, 128, y1 - y0 + 1, pts, CAP_PTS, start, CAP_C);      /* :187 */
    if (dbg) dbg[267] = nc;
    if (nc <= 0 || red_at_edges) return 0;                                                      /* :188 */
    for (int i = 0; i < start[nc]; ++i) pts[i].y += y0;
    int best = 0;                                                                               /* :189 max(): first maximum */
    double best_area = cvr_contour_area(pts + start[0], start[1] - start[0]);
    for (int c = 1; c < nc; ++c) {
        double a = cvr_contour_area(pts + start[c], start[c + 1] - start[c]);
        if (a > best_area) { best_area = a; best = c; }
    }
    const cvr_point* contour = pts + start[best];
    int n = start[best + 1] - start[best];
    double len = cvr_arc_length_closed(contour, n);
    double epsilon = 0.04 * len;                                                                /* :190 */
    int na = cvr_approx_poly_dp_closed(contour, n, epsilon, approx);                            /* :191 */
    if (dbg) {
        dbg[264] = len; dbg[265] = best_area; dbg[266] = na; dbg[268] = n;
        for (int i = 0; i < n && i < 512; ++i) { dbg[272 + 2 * i] = contour[i].x; dbg[273 + 2 * i] = contour[i].y; }
    }
    if (na != 4) return 0;                                                                      /* :193 */
    if (dbg) for (int k = 0; k < 4; ++k) { dbg[256 + 2 * k] = approx[k].x; dbg[257 + 2 * k] = approx[k].y; }
    double mx = (double)(approx[0].x + approx[1].x + approx[2].x + approx[3].x) / 4.0;          /* :197-198 np.mean */
    double my = (double)(approx[0].y + approx[1].y + approx[2].y + approx[3].y) / 4.0;
    /* :200-203: / (128 / 2) - 1, then np.array(..., dtype=np.float32).  The four vertices are integers below 128, so the
     * value is a multiple of 1 / 256 in [-1, 1): the single-precision rounding is exact (kept as the reference has it) */
    center[0] = (double)(float)(mx / 64.0 - 1.0);
    center[1] = (double)(float)(my / 64.0 - 1.0);
    *area = best_area / (128.0 * 128.0);                                                        /* :206 */
    int32_t r[4];
    cvr_bounding_rect(contour, n, r);                                                           /* :209 */
    *ratio = r[3] > 0 ? (double)r[2] / (double)r[3] : 0.0;                                      /* :210-213 */
    return 1;
}

int fpv_oracle_detect_mask(const uint8_t* mask, double center[2], double* area, double* ratio, double* dbg) {
    int y0 = 128, y1 = -1;
    for (int j = 0; j < 128; ++j)
        for (int i = 0; i < 128; ++i)
            if (mask[128 * j + i]) { if (j < y0) y0 = j; y1 = j; break; }
    return detect_rows(mask, y0, y1, center, area, ratio, dbg);
}

static int camera_pixel(const double pos[3], const double quat[4], double center[2], double* area, double* ratio,
                        double* dbg) {
    static __thread uint8_t mask[128 * 128];
    static __thread int dirty0 = 0, dirty1 = -1;               /* rows of `mask` that the previous call left red */
    if (dbg) for (int i = 0; i < FPV_ORACLE_PIXEL_DBG; ++i) dbg[i] = -1.0;
    center[0] = center[1] = 0.0; *area = 0.0; *ratio = 0.0;
    if (dirty0 <= dirty1) memset(mask + 128 * dirty0, 0, (size_t)(128 * (dirty1 - dirty0 + 1)));
    dirty0 = 0; dirty1 = -1;
    int y0, y1;
    if (!panel_mask(pos, quat, mask, dbg, &y0, &y1)) return 0;
    dirty0 = y0; dirty1 = y1;
    return detect_rows(mask, y0, y1, center, area, ratio, dbg);
}

int fpv_oracle_camera_pixel(const double pos[3], const double quat[4], double center[2], double* area,
                            double* ratio) {
    return camera_pixel(pos, quat, center, area, ratio, NULL);
}

/* test hook: the same, also returning the row spans, the four vertices, arcLength, the contour area in pixels and the
 * chosen contour itself (layout above) */
int fpv_oracle_camera_pixel_debug(const double pos[3], const double quat[4], double center[2], double* area,
                                  double* ratio, double dbg[FPV_ORACLE_PIXEL_DBG]) {
    return camera_pixel(pos, quat, center, area, ratio, dbg);
}

/* ------------------------------------------------------------------ reward (SPEC §7) */
double fpv_oracle_reward(const double pos[3], const double euler[3], double omega_z,
                         const double center[2], double area, double ratio, int visible,
                         const double action[4], const double prev_action[4], double step_count,
                         int floor_check, uint32_t* info) {
    double reward = -0.1;                                              /* hover.py:343 */
    uint32_t bits = 0;
    if (step_count > MAX_STEPS) bits |= FPV_ORACLE_TRUNC;              /* :275 */
    if (sqrt(pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2]) > DOME) { /* :278-281 */
        reward = -100.0;
        bits |= FPV_ORACLE_TERM | FPV_ORACLE_OOB;
    }
    if (step_count > 30.0 && floor_check) {                            /* :283-290 */
        if (pos[2] < FLOOR) {
            reward = -100.0;
            bits |= FPV_ORACLE_TERM | FPV_ORACLE_FLOOR;
        }
    }
    double target;
    if (visible) {                                                     /* :296-317 */
        target = -sqrt(center[0] * center[0] + center[1] * center[1]) - fabs(area - TARGET_AREA) -
                 fabs(ratio - TARGET_RATIO);
    } else {
        target = -2.0;                                                 /* :320 */
    }
    reward -= 0.01 * omega_z * omega_z;                                /* :322-324 */
    reward += target - sqrt(euler[0] * euler[0] + euler[1] * euler[1]); /* :326-327 */
    double d2 = 0.0;
    for (int i = 0; i < 4; ++i) d2 += (action[i] - prev_action[i]) * (action[i] - prev_action[i]);
    reward -= 0.2 * sqrt(d2);                                          /* :329-331 */
    reward += 1.0;                                                     /* :332 */
    *info = bits;
    return reward;
}

/* ------------------------------------------------------------------ sub-step (SPEC §4) */
static void substep_p(const fpv_oracle_cfg* cfg, fpv_oracle_state* s, const double sp[4], double pwm[4], int k,
                      uint64_t drone_id, uint64_t tick, double mass, double k_lin, double k_pqr);

void fpv_oracle_substep(const fpv_oracle_cfg* cfg, fpv_oracle_state* s, const double sp[4],
                        double pwm[4], int k, uint64_t drone_id, uint64_t tick) {
    substep_p(cfg, s, sp, pwm, k, drone_id, tick, MASS, K_LIN, K_PQR);
}

static void substep_p(const fpv_oracle_cfg* cfg, fpv_oracle_state* s, const double sp[4], double pwm[4], int k,
                      uint64_t drone_id, uint64_t tick, double mass, double k_lin, double k_pqr) {
    /* 1. control */
    if ((k & 1) == 0) {
        double u[3];
        for (int i = 0; i < 3; ++i) {
            double e = sp[i] - s->omega[i];
            s->pid_int[i] = clampd(s->pid_int[i] + PID_KI[i] * e * T_CTRL, -PID_LIM[i], PID_LIM[i]);
            double d = PID_KD[i] * (e - s->pid_prev[i]) / T_CTRL;
            s->pid_prev[i] = e;
            u[i] = clampd(PID_KP[i] * e + s->pid_int[i] + d, -PID_LIM[i], PID_LIM[i]);
        }
        double T = clampd(sp[3], 0.0, 1.0);
        pwm[0] = -u[0] - u[1] - u[2] + T;
        pwm[1] = +u[0] + u[1] - u[2] + T;
        pwm[2] = +u[0] - u[1] + u[2] + T;
        pwm[3] = -u[0] + u[1] + u[2] + T;
        double hi = fmax(fmax(pwm[0], pwm[1]), fmax(pwm[2], pwm[3]));
        if (hi > 1.0)
            for (int i = 0; i < 4; ++i) pwm[i] /= hi;
        double lo = fmin(fmin(pwm[0], pwm[1]), fmin(pwm[2], pwm[3]));
        if (lo < PWM_IDLE)
            for (int i = 0; i < 4; ++i) pwm[i] += (1.0 - pwm[i]) / (1.0 - lo) * (PWM_IDLE - lo);
    }
    /* 2. motors */
    double f[4];
    double g[4] = {0, 0, 0, 0};
    if (cfg->noise_ratio != 0.0) fpv_oracle_normal4(cfg->seed, drone_id, tick, g);
    for (int i = 0; i < 4; ++i) {
        s->thr[i] += (DT / TAU) * (pwm[i] - s->thr[i]);
        if (cfg->noise_ratio != 0.0) s->thr[i] += g[i] * s->thr[i] * cfg->noise_ratio;
        f[i] = F_MAX * s->thr[i] * s->thr[i];
    }
    /* 3. wrench in the body frame */
    double R[3][3];
    rotmat(s->quat, R);
    double vb[3];
    for (int i = 0; i < 3; ++i) vb[i] = R[0][i] * s->vel[0] + R[1][i] * s->vel[1] + R[2][i] * s->vel[2];
    double Fb[3], tb[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) Fb[i] = -k_lin * fabs(vb[i]) * vb[i];
    for (int i = 0; i < 4; ++i) {
        Fb[2] += f[i];
        tb[0] += PROP_Y[i] * f[i];
        tb[1] -= PROP_X[i] * f[i];
        tb[2] += KQ_OVER_KF * SPIN[i] * f[i];
    }
    for (int i = 0; i < 3; ++i) tb[i] -= k_pqr * fabs(s->omega[i]) * s->omega[i];
    /* 4. accelerations */
    double acc[3];
    for (int i = 0; i < 3; ++i) acc[i] = (R[i][0] * Fb[0] + R[i][1] * Fb[1] + R[i][2] * Fb[2]) / mass;
    acc[2] -= GRAV;
    const double* w = s->omega;
    double Iw[3] = {INERTIA[0] * w[0], INERTIA[1] * w[1], INERTIA[2] * w[2]};
    double gyro[3] = {w[1] * Iw[2] - w[2] * Iw[1], w[2] * Iw[0] - w[0] * Iw[2], w[0] * Iw[1] - w[1] * Iw[0]};
    /* 5. semi-implicit Euler */
    for (int i = 0; i < 3; ++i) {
        s->vel[i] = clampd(s->vel[i] + acc[i] * DT, -V_CLAMP, V_CLAMP);
        s->omega[i] = clampd(s->omega[i] + (tb[i] - gyro[i]) / INERTIA[i] * DT, -W_CLAMP, W_CLAMP);
        s->pos[i] += s->vel[i] * DT;
    }
    double wn = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double theta = wn * DT;
    if (theta > 0.25 * M_PI) theta = 0.25 * M_PI;
    double scale = (wn < 1e-3) ? (0.5 * DT - DT * DT * DT / 48.0 * wn * wn) : sin(0.5 * theta) / wn;
    double dq[4] = {w[0] * scale, w[1] * scale, w[2] * scale, cos(0.5 * theta)};
    const double* q = s->quat;
    double nq[4] = {
        q[3] * dq[0] + q[0] * dq[3] + q[1] * dq[2] - q[2] * dq[1],
        q[3] * dq[1] - q[0] * dq[2] + q[1] * dq[3] + q[2] * dq[0],
        q[3] * dq[2] + q[0] * dq[1] - q[1] * dq[0] + q[2] * dq[3],
        q[3] * dq[3] - q[0] * dq[0] - q[1] * dq[1] - q[2] * dq[2],
    };
    double inv = 1.0 / sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
    for (int i = 0; i < 4; ++i) s->quat[i] = nq[i] * inv;
    /* 6. ground */
    if (s->pos[2] < Z_REST) {
        s->pos[2] = Z_REST;
        s->vel[0] = 0.0; s->vel[1] = 0.0;
        if (s->vel[2] < 0.0) s->vel[2] = 0.0;
        s->omega[0] = 0.0; s->omega[1] = 0.0;
    }
}

/* ------------------------------------------------------------------ observation (SPEC §6) */
static void observe(fpv_oracle_state* s, const double euler[3], const double action[4],
                    double obs[FPV_ORACLE_OBS_WORDS], double center[2], double* area, double* ratio,
                    int* visible, int detector) {
    for (int i = 0; i < 3; ++i) obs[i] = fpv_oracle_wrap(euler[i] - s->prev_euler[i]) / AGENT_DT; /* :228-230 */
    fpv_oracle_quat_from_euler(euler, obs + 3);                                                 /* :233 */
    *visible = detector ? fpv_oracle_camera_pixel(s->pos, s->quat, center, area, ratio)
                        : fpv_oracle_camera(s->pos, s->quat, center, area, ratio);
    obs[7] = center[0]; obs[8] = center[1];           /* hover.py:253-267 */
    obs[9] = s->prev_center[0]; obs[10] = s->prev_center[1];
    obs[11] = *area; obs[12] = s->prev_area;
    obs[13] = *visible ? 1.0 : 0.0;
    obs[14] = *ratio; obs[15] = s->prev_ratio;
    for (int i = 0; i < 4; ++i) obs[16 + i] = action[i];
    s->prev_center[0] = center[0]; s->prev_center[1] = center[1]; /* :270-272 */
    s->prev_area = *area;
    s->prev_ratio = *ratio;
}

void fpv_oracle_reset_state_d(fpv_oracle_state* s, double obs[FPV_ORACLE_OBS_WORDS], int detector) {
    /* hover.py:72-114, SPEC §8; prev_action is preserved (hover.py:101 vs :357) */
    double keep[4];
    memcpy(keep, s->prev_action, sizeof keep);
    memset(s, 0, sizeof *s);
    memcpy(s->prev_action, keep, sizeof keep);
    s->pos[2] = Z_REST;
    s->quat[3] = 1.0;
    double thr = 0.0;
    for (int i = 0; i < 20; ++i) thr += (DT / TAU) * (PWM_IDLE - thr); /* hover.py:109-110, noise-free */
    for (int i = 0; i < 4; ++i) s->thr[i] = thr;
    double euler[3] = {0, 0, 0}, zero[4] = {0, 0, 0, 0}, c[2], a, r;
    int vis;
    double tmp[FPV_ORACLE_OBS_WORDS];
    observe(s, euler, zero, obs ? obs : tmp, c, &a, &r, &vis, detector); /* hover.py:112-113 */
}

void fpv_oracle_reset_state(fpv_oracle_state* s, double obs[FPV_ORACLE_OBS_WORDS]) { fpv_oracle_reset_state_d(s, obs, 0); }

uint32_t fpv_oracle_step_one(const fpv_oracle_cfg* cfg, fpv_oracle_state* s, const double action[4],
                             uint64_t drone_id, uint64_t agent_tick, double obs[FPV_ORACLE_OBS_WORDS],
                             double* reward) {
    double sp[4], pwm[4] = {0, 0, 0, 0};
    fpv_oracle_scale_action(action, sp);
    for (int k = 0; k < SUBSTEPS; ++k)                                /* hover.py:346-349 */
        fpv_oracle_substep(cfg, s, sp, pwm, k, drone_id, agent_tick * (uint64_t)SUBSTEPS + (uint64_t)k);
    double euler[3], center[2], area, ratio;
    int visible;
    fpv_oracle_euler_from_quat(s->quat, euler);
    observe(s, euler, action, obs, center, &area, &ratio, &visible, cfg->detector); /* hover.py:351 */
    uint32_t info;
    *reward = fpv_oracle_reward(s->pos, euler, s->omega[2], center, area, ratio, visible, action,
                                s->prev_action, s->step_count, cfg->floor_check, &info); /* :352 */
    memcpy(s->prev_euler, euler, sizeof euler);                       /* :354 */
    s->step_count += 1.0;                                             /* :356 */
    memcpy(s->prev_action, action, 4 * sizeof(double));               /* :357 */
    s->ep_return += *reward;
    return info;
}

/* ------------------------------------------------------------------ batch */
void fpv_oracle_reset_batch_d(int64_t n, double* state, double* obs, const uint8_t* mask, int detector) {
    fpv_oracle_state* st = (fpv_oracle_state*)state;
    for (int64_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        fpv_oracle_reset_state_d(&st[i], obs ? obs + i * FPV_ORACLE_OBS_WORDS : NULL, detector);
    }
}

void fpv_oracle_reset_batch(int64_t n, double* state, double* obs, const uint8_t* mask) {
    fpv_oracle_reset_batch_d(n, state, obs, mask, 0);
}

void fpv_oracle_step_batch(const fpv_oracle_cfg* cfg, int64_t n, double* state, const double* actions,
                           uint64_t agent_tick, double* obs, double* reward, uint8_t* terminated,
                           uint8_t* truncated, int64_t* led_count, int32_t* led_idx,
                           uint32_t* led_info, double* led_ret, double* led_obs, int nthreads) {
    fpv_oracle_state* st = (fpv_oracle_state*)state;
    uint32_t* bits = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    double* tobs = (double*)malloc((size_t)n * FPV_ORACLE_OBS_WORDS * sizeof(double));
    double* tret = (double*)malloc((size_t)n * sizeof(double));
    int nt = nthreads > 0 ? nthreads : 1;
    (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double* o = obs + i * FPV_ORACLE_OBS_WORDS;
        uint32_t b = fpv_oracle_step_one(cfg, &st[i], actions + 4 * i, cfg->id_offset + (uint64_t)i,
                                         agent_tick, o, &reward[i]);
        terminated[i] = (b & FPV_ORACLE_TERM) ? 1 : 0;
        truncated[i] = (b & FPV_ORACLE_TRUNC) ? 1 : 0;
        bits[i] = b;
        if (b & (FPV_ORACLE_TERM | FPV_ORACLE_TRUNC)) {
            memcpy(tobs + i * FPV_ORACLE_OBS_WORDS, o, FPV_ORACLE_OBS_WORDS * sizeof(double));
            tret[i] = st[i].ep_return;
            bits[i] |= ((uint32_t)st[i].step_count) << 8;
            fpv_oracle_reset_state_d(&st[i], o, cfg->detector);       /* SPEC §8 auto-reset */
        }
    }
    if (led_count) {
        int64_t c = *led_count;
        for (int64_t i = 0; i < n; ++i) {
            if (!(bits[i] & (FPV_ORACLE_TERM | FPV_ORACLE_TRUNC))) continue;
            led_idx[c] = (int32_t)i;
            led_info[c] = bits[i];
            led_ret[c] = tret[i];
            memcpy(led_obs + c * FPV_ORACLE_OBS_WORDS, tobs + i * FPV_ORACLE_OBS_WORDS,
                   FPV_ORACLE_OBS_WORDS * sizeof(double));
            ++c;
        }
        *led_count = c;
    }
    free(bits); free(tobs); free(tret);
}

/* ------------------------------------------------------------------ GAE */
void fpv_oracle_gae(int64_t T, int64_t n, const double* rewards, const double* values,
                    const uint8_t* episode_starts, const double* last_values,
                    const uint8_t* last_dones, double gamma, double lam, double* adv,
                    double* returns) {
    for (int64_t i = 0; i < n; ++i) {
        double last = 0.0;
        for (int64_t t = T - 1; t >= 0; --t) {
            double nnt, nv;
            if (t == T - 1) { nnt = 1.0 - (double)last_dones[i]; nv = last_values[i]; }
            else { nnt = 1.0 - (double)episode_starts[(t + 1) * n + i]; nv = values[(t + 1) * n + i]; }
            double delta = rewards[t * n + i] + gamma * nv * nnt - values[t * n + i];
            last = delta + gamma * lam * nnt * last;
            adv[t * n + i] = last;
            returns[t * n + i] = last + values[t * n + i];
        }
    }
}

/* ------------------------------------------------------------------ yaw task (SPEC §10, yaw.py) */
static void yaw_uniforms(uint64_t seed, uint64_t drone_id, uint64_t tick, double u[4]) {
    uint32_t ctr[4] = {(uint32_t)drone_id, (uint32_t)tick, (uint32_t)(tick >> 32), 0x59415731u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t x[4];
    fpv_oracle_philox4x32_10(ctr, key, x);
    for (int j = 0; j < 4; ++j) u[j] = ((double)(x[j] >> 8) + 0.5) * (1.0 / 16777216.0);
}

static int yaw_observe(const double* y, const double euler[3], const double angvel[3], double obs[FPV_ORACLE_YAW_OBS]) {
    const fpv_oracle_state* s = (const fpv_oracle_state*)y;
    double c[2], area, ratio;
    int vis = fpv_oracle_camera(s->pos, s->quat, c, &area, &ratio);
    for (int i = 0; i < 3; ++i) obs[i] = euler[i] / M_PI;             /* yaw.py:59 */
    obs[3] = c[0]; obs[4] = c[1];                                     /* yaw.py:62-63 ([BUILD] panel centre) */
    for (int i = 0; i < 3; ++i) obs[5 + i] = angvel[i];               /* yaw.py:66 */
    for (int i = 0; i < 4; ++i) obs[8 + i] = y[36 + i];               /* yaw.py:69 */
    return vis;
}

void fpv_oracle_yaw_reset_state(const fpv_oracle_cfg* cfg, double* y, uint64_t drone_id, uint64_t tick,
                                double obs[FPV_ORACLE_YAW_OBS]) {
    fpv_oracle_state* s = (fpv_oracle_state*)y;
    double u[4];
    yaw_uniforms(cfg->seed, drone_id, tick, u);
    memset(y, 0, FPV_ORACLE_YAW_STATE * sizeof(double));
    double psi = M_PI * (2.0 * u[0] - 1.0);                            /* yaw.py:79 */
    s->pos[2] = Z_REST;
    s->quat[2] = sin(0.5 * psi); s->quat[3] = cos(0.5 * psi);
    double thr = 0.0;
    for (int i = 0; i < 20; ++i) thr += (DT / TAU) * (PWM_IDLE - thr);
    for (int i = 0; i < 4; ++i) s->thr[i] = thr;
    s->prev_euler[2] = psi;
    y[40] = cfg->domain_rand ? 0.08 + 0.04 * u[1] : MASS;
    y[41] = cfg->domain_rand ? 1.0 + u[2] : 1.5;
    y[42] = cfg->domain_rand ? (0.5 + 1.5 * u[3]) * 1.0e-4 : K_PQR;
    double euler[3] = {0.0, 0.0, psi}, zero[3] = {0, 0, 0}, tmp[FPV_ORACLE_YAW_OBS];
    yaw_observe(y, euler, zero, obs ? obs : tmp);
}

uint32_t fpv_oracle_yaw_step_one(const fpv_oracle_cfg* cfg, double* y, double action, uint64_t drone_id,
                                 uint64_t agent_tick, double obs[FPV_ORACLE_YAW_OBS], double* reward) {
    fpv_oracle_state* s = (fpv_oracle_state*)y;
    double prev_a = y[39];
    y[36] = y[37]; y[37] = y[38]; y[38] = y[39]; y[39] = action;      /* yaw.py:107 */
    double full[4] = {0.0, 0.0, action, -1.0}, sp[4], pwm[4] = {0, 0, 0, 0};  /* yaw.py:110-115 */
    fpv_oracle_scale_action(full, sp);                                /* yaw.py:118-121 */
    double k_lin = 0.5 * 1.225 * y[41] * 3.0e-4;
    for (int k = 0; k < 2; ++k)                                       /* yaw.py:126-127: one aviary step */
        substep_p(cfg, s, sp, pwm, k, drone_id, agent_tick * 2u + (uint64_t)k, y[40], k_lin, y[42]);
    double euler[3], angvel[3];
    fpv_oracle_euler_from_quat(s->quat, euler);
    uint32_t bits = 0;
    if (s->step_count >= MAX_STEPS) bits |= FPV_ORACLE_TRUNC;         /* yaw.py:138 */
    if (sqrt(s->pos[0] * s->pos[0] + s->pos[1] * s->pos[1] + s->pos[2] * s->pos[2]) > DOME)
        bits |= FPV_ORACLE_TERM | FPV_ORACLE_OOB;                     /* yaw.py:141-143 */
    s->step_count += 1.0;                                             /* yaw.py:145 */
    for (int i = 0; i < 3; ++i) angvel[i] = clampd(fpv_oracle_wrap(euler[i] - s->prev_euler[i]) * 120.0 / 30.0, -1.0, 1.0);
    int vis = yaw_observe(y, euler, angvel, obs);
    double target = vis ? -sqrt(obs[3] * obs[3] + obs[4] * obs[4]) : -2.0;
    *reward = 1.0 + target - 0.01 * s->omega[2] * s->omega[2] - 0.2 * fabs(action - prev_a);
    memcpy(s->prev_euler, euler, sizeof euler);
    s->ep_return += *reward;
    return bits;
}

void fpv_oracle_yaw_reset_batch(const fpv_oracle_cfg* cfg, int64_t n, double* state, double* obs,
                                const uint8_t* mask, uint64_t tick) {
    for (int64_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        fpv_oracle_yaw_reset_state(cfg, state + i * FPV_ORACLE_YAW_STATE, cfg->id_offset + (uint64_t)i, tick,
                                   obs ? obs + i * FPV_ORACLE_YAW_OBS : NULL);
    }
}

void fpv_oracle_yaw_step_batch(const fpv_oracle_cfg* cfg, int64_t n, double* state, const double* actions,
                               uint64_t agent_tick, double* obs, double* reward, uint8_t* terminated,
                               uint8_t* truncated, int64_t* led_count, int32_t* led_idx, uint32_t* led_info,
                               double* led_ret, double* led_obs) {
    for (int64_t i = 0; i < n; ++i) {
        double* y = state + i * FPV_ORACLE_YAW_STATE;
        double* o = obs + i * FPV_ORACLE_YAW_OBS;
        uint32_t b = fpv_oracle_yaw_step_one(cfg, y, actions[i], cfg->id_offset + (uint64_t)i, agent_tick, o, &reward[i]);
        terminated[i] = (b & FPV_ORACLE_TERM) ? 1 : 0;
        truncated[i] = (b & FPV_ORACLE_TRUNC) ? 1 : 0;
        if (b & (FPV_ORACLE_TERM | FPV_ORACLE_TRUNC)) {
            if (led_count) {
                int64_t c = (*led_count)++;
                led_idx[c] = (int32_t)i;
                led_info[c] = b | (((uint32_t)((fpv_oracle_state*)y)->step_count) << 8);
                led_ret[c] = ((fpv_oracle_state*)y)->ep_return;
                memcpy(led_obs + c * FPV_ORACLE_YAW_OBS, o, FPV_ORACLE_YAW_OBS * sizeof(double));
            }
            fpv_oracle_yaw_reset_state(cfg, y, cfg->id_offset + (uint64_t)i, agent_tick, o);
        }
    }
}

int fpv_oracle_version(void) { return 2; }

/*
 * cv_restate.h — CPU restatement of the four OpenCV calls of the reference's detector.  TEST INFRASTRUCTURE ONLY
 * (same rule as fpv_oracle.h: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).
 *
 * simulation/hover.py:187-213 runs, on the thresholded 128 x 128 mask,
 *     contours, _ = cv2.findContours(red_mask, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)        (:187)
 *     contour = max(contours, key=cv2.contourArea)                                                (:189)
 *     epsilon = 0.04 * cv2.arcLength(contour, True)                                               (:190)
 *     approx = cv2.approxPolyDP(contour, epsilon, True)                                           (:191)
 *     ... cv2.contourArea(contour) (:206), cv2.boundingRect(contour) (:209)
 * The algorithm lives in a third-party dependency that is absent from /root/reference and from this image:
 * opencv-python 4.11.0.86 (simulation/uv.lock:655-656).  What follows restates its published algorithm — Suzuki-Abe
 * border following as modules/imgproc/src/contours.cpp runs it for RETR_EXTERNAL + CHAIN_APPROX_SIMPLE, and
 * approx.cpp's closed-curve Douglas-Peucker — from the authors' knowledge of that source: PARITY UNPINNED (no cv2 output is
 * available to check it against; the reference holds no fixture for it).  It is anchored on the reference's own constants
 * instead: from the reset pose these calls give w / h = 20 / 13 and contourArea = 207, i.e. hover.py:50-51's
 * target_ratio = 1.53 and target_area = 0.013 (tests/test_detector_cpu.py).
 */
#ifndef CV_RESTATE_H
#define CV_RESTATE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int32_t x, y; } cvr_point;

/* cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) on a w x h mask of 0 / non-zero bytes (w, h <= 256).
 * Contours come back in cv2's list order (the contour found LAST by the raster scan is first in the list): contour c is
 * pts[start[c] .. start[c + 1]).  Returns the number of contours, or -1 when cap_pts / cap_contours is too small. */
int cvr_find_contours_external(const uint8_t* mask, int w, int h, cvr_point* pts, int cap_pts, int32_t* start,
                               int cap_contours);

double cvr_arc_length_closed(const cvr_point* pts, int n);    /* cv2.arcLength(contour, True) for an integer contour */
double cvr_contour_area(const cvr_point* pts, int n);         /* cv2.contourArea(contour) */
void cvr_bounding_rect(const cvr_point* pts, int n, int32_t xywh[4]);                  /* cv2.boundingRect(contour) */

/* cv2.approxPolyDP(contour, eps, True) for an integer contour; dst needs room for n points.  Returns the vertex count. */
int cvr_approx_poly_dp_closed(const cvr_point* src, int n, double eps, cvr_point* dst);

#ifdef __cplusplus
}
#endif
#endif

"""The kernels' contour code (fpv_drone_rl_agent_amd/csrc/fpv_polydp.h) held against the oracle WITHOUT a GPU.

The header is plain C++ over a lane-group policy; tests/host/polydp_host.cpp compiles that very file with g++.  Two routines:
poly_blob — what the kernels run on a blob that is one piece: cv2's CHAIN_APPROX_SIMPLE contour built from the span table in one
pass, arcLength from it, Douglas-Peucker with the two first levels of slices swept together and a slice stack for anything deeper
— played by ONE lane, and by EIGHT and SIXTEEN lanes in lockstep (threads that meet at every cross-lane operation, which is how
the scan, the lane-strided positions and the all-reduces of the kernels' DPP groups are exercised here); and poly_detect — the
general routine over six "slot" points per image row, which poly_blob falls back to for blobs in several pieces (with
poly_largest_component) and for contours beyond its registers.  The oracle (oracle/cv_restate.c) follows the border on the mask
and runs approx.cpp's procedure on that contour.  Same spans in, so every output must be equal — visibility, the sums of the four
vertices, 2 x contourArea, boundingRect — on the benchmark's synthetic states and on adversarial poses (slivers in several pieces,
blobs of a few pixels, blobs of a hundred rows, approximations with 1 .. 9 vertices).  The -m gpu tests then check the kernels
against the same oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.test_detector_cpu import _adversarial_pose

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("polydp") / "libpolydp_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", "-Wno-unknown-pragmas", "-pthread", "-o", out,
                           os.path.join(ROOT, "tests", "host", "polydp_host.cpp")])
    lib = C.CDLL(out)
    lib.polydp_detect_spans.restype = C.c_int
    return lib


ONE_LANE, EIGHT_LANES, SLOTS_ONLY, SIXTEEN_LANES = 0, 1, 2, 3


def _detect(lib, spans, mode=ONE_LANE):
    tab = np.full((128, 2), -1, np.int16)
    for j, (a, b) in spans.items():
        tab[j] = (a, b)
    out = (C.c_int * 6)()
    vis = lib.polydp_detect_spans(tab.ctypes.data_as(C.POINTER(C.c_int16)), mode, out)
    return vis, list(out)


def test_kernel_side_douglas_peucker_equals_the_oracle(oracle, host_lib):
    rng = np.random.default_rng(21)
    st = oracle.synthetic_states(6000, seed=12)
    poses = [(st[i, 0:3], st[i, 3:7]) for i in range(st.shape[0])] + [_adversarial_pose(rng) for _ in range(14000)]
    blobs = seen = pieces = 0
    counts = set()
    for pos, quat in poses:
        vis, c, area, ratio, spans, verts, length, apx, more = oracle.camera_pixel_debug(pos, quat, full=True)
        if not spans:
            continue
        blobs += 1
        # the lockstep groups are slow on a CPU (threads that meet dozens of times per blob): every 12th blob goes through them
        modes = (ONE_LANE, SLOTS_ONLY) + ((EIGHT_LANES, SIXTEEN_LANES) if blobs % 12 == 0 else ())
        for mode in modes:
            got, out = _detect(host_lib, spans, mode)
            assert got == vis, mode                                # (-1: the lanes of a group disagreed)
            if vis:
                assert out[1] == verts[:, 0].sum() and out[2] == verts[:, 1].sum()
                assert out[1] / 256.0 - 1.0 == c[0] and out[2] / 256.0 - 1.0 == c[1]
                assert out[3] == 2 * apx and out[3] / 2.0 / 16384.0 == area and out[4] / out[5] == ratio
        pieces += more["contours"] > 1
        counts.add(more["approx"])
        seen += vis
    assert blobs > 12000 and seen > 6000 and pieces > 300 and counts >= {1, 2, 3, 4, 5, 6, 7, 8}, (blobs, seen, pieces, counts)


def test_hand_made_span_tables(oracle, host_lib):
    """Blobs no pose produces easily: one pixel, one row, a one-pixel column, a thin diagonal (every border pixel visited twice),
    a 100-row rectangle, two pieces of equal area (cv2 lists the lower one first and max() keeps it)."""
    def both(spans, mode=ONE_LANE):
        mask = np.zeros((128, 128), np.uint8)
        for j, (a, b) in spans.items():
            mask[j, a:b + 1] = 1
        return oracle.detect_mask(mask), _detect(host_lib, spans, mode)

    for spans in ({60: (60, 60)}, {60: (40, 90)}, {j: (64, 64) for j in range(20, 90)}, {j: (j, j + 1) for j in range(10, 100)},
                  {j: (20, 110) for j in range(10, 110)}, {j: (30 + j // 3, 90 - j // 2) for j in range(5, 100)},
                  {**{j: (20, 40) for j in range(10, 20)}, **{j: (60, 80) for j in range(40, 50)}},
                  {**{j: (20, 40) for j in range(10, 20)}, **{j: (60, 80) for j in range(21, 31)}},
                  {j: (64 - (j - 5), 64 + (j - 5)) for j in range(5, 60)}):
        for mode in (ONE_LANE, EIGHT_LANES, SLOTS_ONLY, SIXTEEN_LANES):
            (vis, c, area, ratio), (got, out) = both(spans, mode)
            assert got == vis, (spans, mode)
            if vis:
                assert (out[1] / 256.0 - 1.0, out[2] / 256.0 - 1.0, out[3] / 2.0 / 16384.0, out[4] / out[5]) == (c[0], c[1], area, ratio)
    # the two equal pieces: the lower one is taken
    spans = {**{j: (20, 40) for j in range(10, 20)}, **{j: (60, 80) for j in range(40, 50)}}
    (vis, c, area, ratio), _ = both(spans)
    assert vis == 1 and c[0] == 70 / 64.0 - 1.0 and c[1] == 44.5 / 64.0 - 1.0

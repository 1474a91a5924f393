"""ctypes loader for the CPU oracle + numpy restatements of the SB3-side arithmetic.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never from fpv_drone_rl_agent_amd/ (the product).

PARITY UNPINNED versus PyFlyt/pybullet (see fpv_oracle.h).  The numpy pieces restate
SB3 2.7.0's VecNormalize / RunningMeanStd (call sites simulation/train_hover.py:43,
simulation/test_hover.py:8-10; algorithm as published) — SB3 is not
installed either, so these are pinned only by their own closed-form tests.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libfpv_oracle.so")

STATE_WORDS = 36
OBS_WORDS = 20
YAW_STATE_WORDS = 44
YAW_OBS_WORDS = 12
TERM, TRUNC, OOB, FLOOR = 1, 2, 4, 8


def build(force: bool = False) -> str:
    """Compile oracle/fpv_oracle.c with gcc (make).  Returns the .so path."""
    srcs = [os.path.join(_HERE, f) for f in ("fpv_oracle.c", "fpv_oracle.h", "fpv_cpu.c", "cv_restate.c", "cv_restate.h",
                                                os.path.join("..", "include", "fpv_env.h"))]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfpv_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB


class _Cfg(C.Structure):
    _fields_ = [("floor_check", C.c_int32), ("domain_rand", C.c_int32), ("noise_ratio", C.c_double),
                ("seed", C.c_uint64), ("id_offset", C.c_uint64), ("detector", C.c_int32), ("reserved", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        dp = C.POINTER(C.c_double)
        _lib.fpv_oracle_wrap.restype = C.c_double
        _lib.fpv_oracle_wrap.argtypes = [C.c_double]
        _lib.fpv_oracle_reward.restype = C.c_double
        _lib.fpv_oracle_reward.argtypes = [dp, dp, C.c_double, dp, C.c_double, C.c_double, C.c_int,
                                           dp, dp, C.c_double, C.c_int, C.POINTER(C.c_uint32)]
        _lib.fpv_oracle_camera.restype = C.c_int
        _lib.fpv_oracle_camera_pixel.restype = C.c_int
        _lib.fpv_oracle_step_one.restype = C.c_uint32
        _lib.fpv_oracle_version.restype = C.c_int
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


DETECTORS = {"analytic": 0, "pixel": 1}


def make_cfg(floor_check=True, noise_ratio=0.02, seed=0, id_offset=0, domain_rand=False, detector="pixel"):
    """detector: "pixel" (the 128 x 128 rules of hover.py:157-222, SPEC §6b; the default, as in the product) or "analytic"
    (the continuous projection of SPEC §6)."""
    return _Cfg(int(bool(floor_check)), int(bool(domain_rand)), float(noise_ratio), int(seed), int(id_offset),
                DETECTORS[detector] if isinstance(detector, str) else int(detector), 0)


# ---------------------------------------------------------------- small pieces
def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().fpv_oracle_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def normal4(seed, drone_id, tick):
    g = np.zeros(4)
    lib().fpv_oracle_normal4(C.c_uint64(seed), C.c_uint64(drone_id), C.c_uint64(tick), _dp(g))
    return g


def wrap(d):
    return float(lib().fpv_oracle_wrap(float(d)))


def quat_from_euler(e):
    e = _f64(e)
    q = np.zeros(4)
    lib().fpv_oracle_quat_from_euler(_dp(e), _dp(q))
    return q


def euler_from_quat(q):
    q = _f64(q)
    e = np.zeros(3)
    lib().fpv_oracle_euler_from_quat(_dp(q), _dp(e))
    return e


def scale_action(a):
    a = _f64(a)
    sp = np.zeros(4)
    lib().fpv_oracle_scale_action(_dp(a), _dp(sp))
    return sp


def camera(pos, quat, detector="pixel"):
    pos, quat = _f64(pos), _f64(quat)
    c = np.zeros(2)
    area, ratio = C.c_double(), C.c_double()
    fn = lib().fpv_oracle_camera_pixel if DETECTORS.get(detector, detector) else lib().fpv_oracle_camera
    vis = fn(_dp(pos), _dp(quat), _dp(c), C.byref(area), C.byref(ratio))
    return int(vis), c, area.value, ratio.value


PIXEL_DBG = 272 + 1024


def camera_pixel_debug(pos, quat, full=False):
    """(visible, center, area, ratio, spans {row: (a, b)}, vertices [4][2], arcLength, contour area in pixels); with full=True
    also a dict {approx: len(approx), contours: len(contours), contour: [(x, y), ...] of the chosen contour}."""
    pos, quat = _f64(pos), _f64(quat)
    c = np.zeros(2)
    dbg = np.full(PIXEL_DBG, -1.0)
    area, ratio = C.c_double(), C.c_double()
    lib().fpv_oracle_camera_pixel_debug.restype = C.c_int
    vis = lib().fpv_oracle_camera_pixel_debug(_dp(pos), _dp(quat), _dp(c), C.byref(area), C.byref(ratio), _dp(dbg))
    spans = {j: (int(dbg[2 * j]), int(dbg[2 * j + 1])) for j in range(128) if dbg[2 * j] >= 0}
    out = (int(vis), c, area.value, ratio.value, spans, dbg[256:264].reshape(4, 2).astype(int), dbg[264], dbg[265])
    if not full:
        return out
    n = int(dbg[268])
    more = dict(approx=int(dbg[266]), contours=int(dbg[267]),
                contour=[(int(dbg[272 + 2 * i]), int(dbg[273 + 2 * i])) for i in range(max(min(n, 512), 0))], points=n)
    return out + (more,)


class _Pt(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32)]


def detect_mask(mask):
    """hover.py:180-213 on a 128 x 128 mask (nonzero = red).  (visible, center, area, ratio)."""
    m = np.ascontiguousarray((np.asarray(mask) != 0).astype(np.uint8))
    assert m.shape == (128, 128)
    c = np.zeros(2)
    area, ratio = C.c_double(), C.c_double()
    lib().fpv_oracle_detect_mask.restype = C.c_int
    vis = lib().fpv_oracle_detect_mask(m.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(c), C.byref(area), C.byref(ratio), None)
    return int(vis), c, area.value, ratio.value


def cv_find_contours(mask):
    """cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) restated (oracle/cv_restate.c): list of [(x, y), ...]."""
    m = np.ascontiguousarray((np.asarray(mask) != 0).astype(np.uint8))
    h, w = m.shape
    cap = 4 * h * w + 16
    pts = (_Pt * cap)()
    start = (C.c_int32 * (h * w // 2 + 2))()
    lib().cvr_find_contours_external.restype = C.c_int
    nc = lib().cvr_find_contours_external(m.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(w), C.c_int(h), pts, C.c_int(cap), start,
                                          C.c_int(h * w // 2 + 1))
    assert nc >= 0
    return [[(pts[i].x, pts[i].y) for i in range(start[c], start[c + 1])] for c in range(nc)]


def _pts(points):
    arr = (_Pt * max(len(points), 1))()
    for i, (x, y) in enumerate(points):
        arr[i].x, arr[i].y = int(x), int(y)
    return arr


def cv_arc_length(points):
    lib().cvr_arc_length_closed.restype = C.c_double
    return float(lib().cvr_arc_length_closed(_pts(points), C.c_int(len(points))))


def cv_contour_area(points):
    lib().cvr_contour_area.restype = C.c_double
    return float(lib().cvr_contour_area(_pts(points), C.c_int(len(points))))


def cv_bounding_rect(points):
    r = (C.c_int32 * 4)()
    lib().cvr_bounding_rect(_pts(points), C.c_int(len(points)), r)
    return tuple(int(v) for v in r)


def cv_approx_poly_dp(points, eps):
    dst = (_Pt * max(len(points), 1))()
    lib().cvr_approx_poly_dp_closed.restype = C.c_int
    n = lib().cvr_approx_poly_dp_closed(_pts(points), C.c_int(len(points)), C.c_double(eps), dst)
    return [(dst[i].x, dst[i].y) for i in range(n)]


def reward(pos, euler, omega_z, center, area, ratio, visible, action, prev_action, step_count,
           floor_check=True):
    info = C.c_uint32()
    args = [_f64(x) for x in (pos, euler, center, action, prev_action)]
    r = lib().fpv_oracle_reward(_dp(args[0]), _dp(args[1]), float(omega_z), _dp(args[2]), float(area),
                                float(ratio), int(visible), _dp(args[3]), _dp(args[4]),
                                float(step_count), int(floor_check), C.byref(info))
    return float(r), int(info.value)


# ---------------------------------------------------------------- state / batch
def reset_states(n, detector="pixel"):
    """n freshly reset drones.  Returns (state [n,36], obs [n,20])."""
    st = np.zeros((n, STATE_WORDS))
    obs = np.zeros((n, OBS_WORDS))
    lib().fpv_oracle_reset_batch_d(C.c_int64(n), _dp(st), _dp(obs), None, C.c_int(DETECTORS.get(detector, detector)))
    return st, obs


def reset_masked(state, obs, mask, detector="pixel"):
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    lib().fpv_oracle_reset_batch_d(C.c_int64(state.shape[0]), _dp(state), _dp(obs),
                                   m.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(DETECTORS.get(detector, detector)))


def step_one(cfg, state_row, action, drone_id, agent_tick):
    """One drone, no auto-reset.  state_row (36,) is updated in place."""
    assert state_row.dtype == np.float64 and state_row.flags.c_contiguous
    a = _f64(action)
    obs = np.zeros(OBS_WORDS)
    r = C.c_double()
    bits = lib().fpv_oracle_step_one(C.byref(cfg), _dp(state_row), _dp(a), C.c_uint64(drone_id),
                                     C.c_uint64(agent_tick), _dp(obs), C.byref(r))
    return obs, r.value, int(bits)


class Ledger:
    def __init__(self, cap, obs_words=OBS_WORDS):
        self.count = C.c_int64(0)
        self.idx = np.zeros(cap, np.int32)
        self.info = np.zeros(cap, np.uint32)
        self.ret = np.zeros(cap)
        self.obs = np.zeros((cap, obs_words))

    def rows(self):
        c = self.count.value
        return self.idx[:c], self.info[:c], self.ret[:c], self.obs[:c]


def step_batch(cfg, state, actions, agent_tick, ledger=None, nthreads=1):
    """VecEnv-semantics step (auto-reset).  state [n,36] float64 updated in place."""
    n = state.shape[0]
    assert state.dtype == np.float64 and state.flags.c_contiguous
    a = _f64(actions)
    obs = np.zeros((n, OBS_WORDS))
    rew = np.zeros(n)
    term = np.zeros(n, np.uint8)
    trunc = np.zeros(n, np.uint8)
    u8 = C.POINTER(C.c_uint8)
    if ledger is None:
        ledger = Ledger(n)
        ledger_tmp = True
    else:
        ledger_tmp = False
    lib().fpv_oracle_step_batch(
        C.byref(cfg), C.c_int64(n), _dp(state), _dp(a), C.c_uint64(agent_tick), _dp(obs), _dp(rew),
        term.ctypes.data_as(u8), trunc.ctypes.data_as(u8), C.byref(ledger.count),
        ledger.idx.ctypes.data_as(C.POINTER(C.c_int32)),
        ledger.info.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(ledger.ret), _dp(ledger.obs),
        C.c_int(nthreads))
    return obs, rew, term, trunc, (ledger if not ledger_tmp else ledger)


# ---------------------------------------------------------------- yaw task (SPEC §10)
def yaw_reset_states(cfg, n, tick=0):
    st = np.zeros((n, YAW_STATE_WORDS))
    obs = np.zeros((n, YAW_OBS_WORDS))
    lib().fpv_oracle_yaw_reset_batch(C.byref(cfg), C.c_int64(n), _dp(st), _dp(obs), None, C.c_uint64(tick))
    return st, obs


def yaw_step_batch(cfg, state, actions, agent_tick, ledger=None):
    n = state.shape[0]
    assert state.dtype == np.float64 and state.flags.c_contiguous and state.shape[1] == YAW_STATE_WORDS
    a = _f64(actions).reshape(n)
    obs = np.zeros((n, YAW_OBS_WORDS))
    rew = np.zeros(n)
    term = np.zeros(n, np.uint8)
    trunc = np.zeros(n, np.uint8)
    if ledger is None:
        ledger = Ledger(n, obs_words=YAW_OBS_WORDS)
    u8 = C.POINTER(C.c_uint8)
    lib().fpv_oracle_yaw_step_batch(
        C.byref(cfg), C.c_int64(n), _dp(state), _dp(a), C.c_uint64(agent_tick), _dp(obs), _dp(rew),
        term.ctypes.data_as(u8), trunc.ctypes.data_as(u8), C.byref(ledger.count),
        ledger.idx.ctypes.data_as(C.POINTER(C.c_int32)), ledger.info.ctypes.data_as(C.POINTER(C.c_uint32)),
        _dp(ledger.ret), _dp(ledger.obs))
    return obs, rew, term, trunc, ledger


def gae(rewards, values, episode_starts, last_values, last_dones, gamma=0.99, lam=0.95):
    r, v, lv = _f64(rewards), _f64(values), _f64(last_values)
    es = np.ascontiguousarray(episode_starts, np.uint8)
    ld = np.ascontiguousarray(last_dones, np.uint8)
    T, n = r.shape
    adv = np.zeros((T, n))
    ret = np.zeros((T, n))
    u8 = C.POINTER(C.c_uint8)
    lib().fpv_oracle_gae(C.c_int64(T), C.c_int64(n), _dp(r), _dp(v), es.ctypes.data_as(u8), _dp(lv),
                         ld.ctypes.data_as(u8), C.c_double(gamma), C.c_double(lam), _dp(adv), _dp(ret))
    return adv, ret


# ---------------------------------------------------------------- synthetic states
def synthetic_states(n, seed=0):
    """Seeded airborne random-init states of BASELINE.md §3 / SURVEY.md §8(d), canonical [n,36]."""
    rng = np.random.Generator(np.random.Philox(seed))
    st, _ = reset_states(n)
    # position uniform in the ball |p| < 2.5 with z in [0.3, 2.5] (rejection, vectorised)
    pos = np.zeros((n, 3))
    todo = np.arange(n)
    while todo.size:
        c = rng.uniform([-2.5, -2.5, 0.3], [2.5, 2.5, 2.5], size=(todo.size, 3))
        ok = np.linalg.norm(c, axis=1) < 2.5
        pos[todo[ok]] = c[ok]
        todo = todo[~ok]
    st[:, 0:3] = pos
    eul = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-np.pi, np.pi, n)], 1)
    sr, cr = np.sin(eul[:, 0] / 2), np.cos(eul[:, 0] / 2)
    sp, cp = np.sin(eul[:, 1] / 2), np.cos(eul[:, 1] / 2)
    sy, cy = np.sin(eul[:, 2] / 2), np.cos(eul[:, 2] / 2)
    st[:, 3] = sr * cp * cy - cr * sp * sy
    st[:, 4] = cr * sp * cy + sr * cp * sy
    st[:, 5] = cr * cp * sy - sr * sp * cy
    st[:, 6] = cr * cp * cy + sr * sp * sy
    # body lin vel U(-1,1)^3 -> world frame
    vb = rng.uniform(-1, 1, (n, 3))
    x, y, z, w = st[:, 3], st[:, 4], st[:, 5], st[:, 6]
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    st[:, 7:10] = np.einsum("nij,nj->ni", R, vb)
    st[:, 10:13] = rng.uniform(-2, 2, (n, 3))
    st[:, 13:17] = rng.uniform(0.3, 0.7, (n, 4))
    st[:, 17:23] = 0.0
    st[:, 23:26] = eul
    st[:, 26:30] = 0.0
    st[:, 30:34] = 0.0
    st[:, 34] = rng.integers(0, 401, n).astype(np.float64)
    st[:, 35] = 0.0
    return st


# ---------------------------------------------------------------- VecNormalize (numpy restatement)
class RunningMeanStd:
    """SB3 RunningMeanStd: parallel-variance merge, epsilon-count start."""

    def __init__(self, shape=(), epsilon=1e-4):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = float(epsilon)

    def update(self, arr):
        arr = np.asarray(arr, np.float64)
        bm, bv, bc = arr.mean(axis=0), arr.var(axis=0), arr.shape[0]
        self.update_from_moments(bm, bv, bc)

    def update_from_moments(self, bm, bv, bc):
        delta = bm - self.mean
        tot = self.count + bc
        new_mean = self.mean + delta * bc / tot
        m2 = self.var * self.count + bv * bc + np.square(delta) * self.count * bc / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot


class VecNormalizeRef:
    """Arithmetic of SB3 VecNormalize(norm_obs=True, norm_reward=True) — train_hover.py:43."""

    def __init__(self, n, obs_dim=OBS_WORDS, gamma=0.99, clip_obs=10.0, clip_reward=10.0, eps=1e-8):
        self.obs_rms = RunningMeanStd((obs_dim,))
        self.ret_rms = RunningMeanStd(())
        self.returns = np.zeros(n)
        self.gamma, self.clip_obs, self.clip_reward, self.eps = gamma, clip_obs, clip_reward, eps
        self.training, self.norm_obs, self.norm_reward = True, True, True

    def normalize_obs(self, obs):
        if not self.norm_obs:
            return obs
        return np.clip((obs - self.obs_rms.mean) / np.sqrt(self.obs_rms.var + self.eps),
                       -self.clip_obs, self.clip_obs)

    def normalize_reward(self, r):
        if not self.norm_reward:
            return r
        return np.clip(r / np.sqrt(self.ret_rms.var + self.eps), -self.clip_reward, self.clip_reward)

    def reset(self, obs):
        self.returns[:] = 0.0
        if self.training and self.norm_obs:
            self.obs_rms.update(obs)
        return self.normalize_obs(obs)

    def step(self, obs, rewards, dones):
        if self.training and self.norm_obs:
            self.obs_rms.update(obs)
        o = self.normalize_obs(obs)
        if self.training:
            self.returns = self.returns * self.gamma + rewards
            self.ret_rms.update(self.returns)
        r = self.normalize_reward(rewards)
        self.returns[np.asarray(dones, bool)] = 0.0
        return o, r

"""Known-answer tests pinning the oracle to what simulation/hover.py states itself.

Values are the hand-derived KATs of SURVEY.md §8(c) (the reference ships no tests or
golden vectors).  Physics inside Aviary.step() is NOT pinned: parity with
PyFlyt/pybullet is unpinned (docs/SPEC.md header).
"""
import math

import numpy as np
import pytest


def test_philox_known_answer(oracle):
    # Random123 kat_vectors: philox4x32-10, zero counter/key and all-ones counter/key
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = 0xFFFFFFFF
    assert oracle.philox([f, f, f, f], [f, f]) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_normal4_moments(oracle):
    g = np.array([oracle.normal4(7, i, t) for i in range(200) for t in range(50)]).ravel()
    assert abs(g.mean()) < 0.02 and abs(g.std() - 1.0) < 0.02
    assert np.allclose(oracle.normal4(7, 3, 9), oracle.normal4(7, 3, 9))
    assert not np.allclose(oracle.normal4(7, 3, 9), oracle.normal4(7, 4, 9))


@pytest.mark.parametrize("e,q", [
    ((0.1, -0.2, 0.3), (0.06407135, -0.09115755, 0.1534393, 0.98185617)),
    ((1.0, 0.5, -2.0), (0.43367995, -0.27357221, -0.77958954, 0.35961103)),
    ((0.0, 0.0, math.pi), (0.0, 0.0, 1.0, 6.1232e-17)),
])
def test_quat_from_euler_kat(oracle, e, q):
    # hover.py:233 (pybullet.getQuaternionFromEuler)
    assert np.allclose(oracle.quat_from_euler(e), q, atol=1e-8)


def test_quat_from_euler_matches_scipy(oracle):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    for e in rng.uniform(-3, 3, (50, 3)):
        e[1] = np.clip(e[1], -1.5, 1.5)
        assert np.allclose(oracle.quat_from_euler(e), Rotation.from_euler("xyz", e).as_quat(), atol=1e-12)
        back = oracle.euler_from_quat(oracle.quat_from_euler(e))
        assert np.allclose(back, e, atol=1e-9)


def test_euler_gimbal_branch(oracle):
    q = oracle.quat_from_euler((0.0, math.pi / 2, 0.4))
    e = oracle.euler_from_quat(q)
    assert e[0] == 0.0 and abs(e[1] - math.pi / 2) < 1e-12


@pytest.mark.parametrize("d,w", [
    (3.5, -2.78318531), (-3.5, 2.78318531), (math.pi, -math.pi), (-math.pi, -math.pi),
    (6.0, -0.28318531), (0.25, 0.25),
])
def test_wrap_kat(oracle, d, w):
    # hover.py:229 — Python modulo, +pi -> -pi
    assert abs(oracle.wrap(d) - w) < 1e-8
    assert abs(oracle.wrap(d) - ((d + math.pi) % (2 * math.pi) - math.pi)) < 1e-12


def test_action_scaling_kat(oracle):
    # hover.py:337-341
    assert np.allclose(oracle.scale_action((1, -1, 0.5, 0)), (30, -30, -15, 0.5))


def test_reward_kat_visible(oracle):
    # -0.1 - 0.04 + (-0.5 - 0.007 - 0.13) - 0 - 0.1 + 1.0 = 0.123
    a, pa = np.array([0.5, 0, 0, 0.0]), np.zeros(4)
    r, info = oracle.reward((0, 0, 1), (0, 0, 0.7), 2.0, (0.3, -0.4), 0.02, 1.4, 1, a, pa, 5)
    assert abs(r - 0.123) < 1e-12 and info == 0


def test_reward_kat_invisible(oracle):
    # -0.1 - 2.0 - 0.5 + 1.0 = -1.6
    r, info = oracle.reward((0, 0, 1), (0.3, 0.4, 0), 0.0, (0, 0), 0, 0, 0, np.zeros(4), np.zeros(4), 5)
    assert abs(r + 1.6) < 1e-12 and info == 0


def test_reward_out_of_bounds_is_assignment(oracle):
    # hover.py:279: reward restarts from -100, shaping terms and +1 still apply
    r, info = oracle.reward((3.1, 0, 1), (0.3, 0.4, 0), 0.0, (0, 0), 0, 0, 0, np.zeros(4), np.zeros(4), 5)
    assert abs(r - (-100.0 - 2.0 - 0.5 + 1.0)) < 1e-12
    assert info == (oracle.TERM | oracle.OOB)


def test_floor_check_window_and_render(oracle):
    z = np.zeros(4)
    # step_count > 30 needed: call with step_count 30 -> no check; 31 -> terminates (hover.py:283)
    _, i30 = oracle.reward((0, 0, 0.05), (0, 0, 0), 0, (0, 0), 0, 0, 0, z, z, 30)
    _, i31 = oracle.reward((0, 0, 0.05), (0, 0, 0), 0, (0, 0), 0, 0, 0, z, z, 31)
    _, i31r = oracle.reward((0, 0, 0.05), (0, 0, 0), 0, (0, 0), 0, 0, 0, z, z, 31, floor_check=False)
    assert i30 == 0 and i31 == (oracle.TERM | oracle.FLOOR) and i31r == 0
    # truncation uses > (hover.py:275)
    _, t400 = oracle.reward((0, 0, 1), (0, 0, 0), 0, (0, 0), 0, 0, 0, z, z, 400)
    _, t401 = oracle.reward((0, 0, 1), (0, 0, 0), 0, (0, 0), 0, 0, 0, z, z, 401)
    assert t400 == 0 and t401 == oracle.TRUNC


def test_camera_level_drone_kat(oracle):
    # SURVEY §8a A7 [DERIVED]: level drone at (0,0,1): centre (0,-0.359), area 0.0173, ratio 1.415;
    # at (0,0,0.5): area 0.0157, ratio 1.458 (front face at x=4.98 shifts these in the 3rd digit)
    vis, c, area, ratio = oracle.camera((0, 0, 1), (0, 0, 0, 1), "analytic")
    assert vis == 1 and abs(c[0]) < 1e-12
    assert abs(c[1] + 0.359) < 5e-3 and abs(area - 0.0173) < 3e-4 and abs(ratio - 1.415) < 1e-2
    vis, c, area, ratio = oracle.camera((0, 0, 0.5), (0, 0, 0, 1), "analytic")
    assert vis == 1 and abs(area - 0.0157) < 3e-4 and abs(ratio - 1.458) < 1e-2


def test_camera_invisible_cases(oracle):
    # facing away (yaw pi): panel behind the camera
    assert oracle.camera((0, 0, 1), oracle.quat_from_euler((0, 0, math.pi)), "analytic")[0] == 0
    # pitched nose-down 60 deg: panel leaves the top of the frame
    assert oracle.camera((0, 0, 1), oracle.quat_from_euler((0, 1.05, 0)), "analytic")[0] == 0
    vis, c, area, ratio = oracle.camera((0, 0, 1), oracle.quat_from_euler((0, 1.05, 0)), "analytic")
    assert (c == 0).all() and area == 0 and ratio == 0


def test_reset_state_and_obs(oracle):
    st, obs = oracle.reset_states(3)
    s = st[0]
    assert np.allclose(s[0:3], (0, 0, 0.01)) and np.allclose(s[3:7], (0, 0, 0, 1))
    assert np.allclose(s[7:13], 0) and s[34] == 0 and s[35] == 0
    thr = 0.05 * (1 - (1 - (1 / 240) / 0.01) ** 20)
    assert np.allclose(s[13:17], thr, atol=1e-15)
    # first obs (hover.py:112-113): zero rates, identity quat, prev_* zero, action zero, target visible
    o = obs[0]
    assert np.allclose(o[0:3], 0) and np.allclose(o[3:7], (0, 0, 0, 1))
    assert o[13] == 1.0 and np.allclose(o[9:11], 0) and o[12] == 0 and o[15] == 0 and np.allclose(o[16:20], 0)
    # feature memory took the current values (hover.py:270-272)
    assert np.allclose(s[26:28], o[7:9]) and s[28] == o[11] and s[29] == o[14]
    assert (st[1] == st[0]).all() and (obs[2] == obs[0]).all()


def test_episode_bookkeeping(oracle):
    """Floor check first active on call 32; truncation on call 402 (ep_len min 32 / max 402)."""
    cfg = oracle.make_cfg(noise_ratio=0.0)
    st, _ = oracle.reset_states(1)
    a = np.array([0.0, 0.0, 0.0, -1.0])  # zero throttle: sits on the floor
    for call in range(1, 40):
        _, r, bits = oracle.step_one(cfg, st[0], a, 0, call)
        if bits:
            break
    assert call == 32 and bits == (oracle.TERM | oracle.FLOOR)

    # hovering-ish drone frozen in the air: never terminates, truncates on the 402nd call
    cfg2 = oracle.make_cfg(noise_ratio=0.0, floor_check=False)
    st, _ = oracle.reset_states(1)
    for call in range(1, 410):
        _, r, bits = oracle.step_one(cfg2, st[0], a, 0, call)
        if bits:
            break
    assert call == 402 and bits == oracle.TRUNC


def test_prev_action_survives_reset(oracle):
    # hover.py:101 resets only self.action; :357 sets prev_action
    cfg = oracle.make_cfg(noise_ratio=0.0)
    st, obs = oracle.reset_states(1)
    a = np.array([0.2, -0.3, 0.1, 0.5])
    oracle.step_one(cfg, st[0], a, 0, 0)
    oracle.reset_masked(st, obs, [1])
    assert np.allclose(st[0, 30:34], a) and st[0, 34] == 0


def test_airframe_constants_physics(oracle):
    """Hover throttle sqrt(0.981/4)=0.4952 holds altitude; max roll accel ~1867 rad/s^2."""
    cfg = oracle.make_cfg(noise_ratio=0.0, floor_check=False)
    st = oracle.synthetic_states(1, seed=1)
    st[0, 0:3] = (0, 0, 1.0); st[0, 3:7] = (0, 0, 0, 1); st[0, 7:13] = 0
    hover = math.sqrt(0.981 / 4)
    st[0, 13:17] = hover
    st[0, 23:26] = 0
    a = np.array([0, 0, 0, 2 * hover - 1.0])
    for t in range(20):
        oracle.step_one(cfg, st[0], a, 0, t)
    assert abs(st[0, 2] - 1.0) < 1e-6 and np.abs(st[0, 7:13]).max() < 1e-6
    # full roll command from rest: omega_x after one 1/240 s sub-step limited by torque 0.056 N m / 3e-5
    st[0, 10:13] = 0
    o, r, b = oracle.step_one(cfg, st[0], np.array([1.0, 0, 0, 2 * hover - 1.0]), 0, 99)
    assert 0 < st[0, 10] < 1867 * 0.05 + 1e-6

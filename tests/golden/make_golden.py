"""Generates tests/golden/hover_step_v1.npz (analytic detector, SPEC §6) and hover_step_v3.npz (the pixel detector of
SPEC §6b, the product's default since round 3; v3 = round 4, with cv2.findContours / approxPolyDP restated in full — v2 held
round 3's nearest-covered-pixel vertex rule and is gone) from the CPU oracle (oracle/fpv_oracle.c).

The reference ships no golden vectors and its PyFlyt/pybullet step cannot run here,
so these vectors pin the BUILD's spec (docs/SPEC.md), not PyFlyt: parity with PyFlyt is unpinned.
They freeze the oracle's outputs so that a later edit of either the oracle or the HIP kernels that
changes results is caught.  Run from the repo root:  python tests/golden/make_golden.py [v1|v3]   (default: v3;
v1 was written in round 1 and is kept as committed)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


def main(version="v3"):
    detector = {"v1": "analytic", "v3": "pixel"}[version]
    orc.build()
    out = {}
    # (a) one noisy step from 256 synthetic states with episode-end edge cases
    n = 256
    st = orc.synthetic_states(n, seed=21)
    st[:8, 0:3] = [2.99, 0, 0.5]; st[:8, 7:10] = [3.0, 0, 0]              # leaves the dome
    st[8:16, 2] = 0.105; st[8:16, 9] = -2.0; st[8:16, 34] = 200            # drops under the floor line
    st[16:24, 34] = 401                                                    # truncates
    st = st.astype(np.float32).astype(np.float64)
    act = np.random.Generator(np.random.Philox(22)).uniform(-1, 1, (n, 4)).astype(np.float32).astype(np.float64)
    out["a_state_in"], out["a_action"] = st.copy(), act
    cfg = orc.make_cfg(noise_ratio=0.02, seed=77, id_offset=1000, detector=detector)
    obs, rew, term, trunc, led = orc.step_batch(cfg, st, act, 5)
    out.update(a_state_out=st, a_obs=obs, a_reward=rew, a_term=term, a_trunc=trunc, a_led_idx=led.rows()[0],
               a_led_info=led.rows()[1], a_led_ret=led.rows()[2], a_led_obs=led.rows()[3],
               a_meta=np.array([0.02, 77, 1000, 5]))
    # (b) 60-step noise-free trajectory of 8 drones from reset (take-off, tumble, episode end, auto-reset)
    n, T = 8, 60
    st, obs0 = orc.reset_states(n, detector)
    rng = np.random.Generator(np.random.Philox(23))
    acts = rng.uniform(-1, 1, (T, n, 4))
    acts[:, :, 3] = np.abs(acts[:, :, 3]) * 0.6          # mostly climbing
    acts[:, :, 0:3] *= 0.1
    cfg = orc.make_cfg(noise_ratio=0.0, seed=0, detector=detector)
    tr_obs, tr_rew, tr_done = [], [], []
    for t in range(T):
        o, r, te, tu, _ = orc.step_batch(cfg, st, acts[t], t)
        tr_obs.append(o); tr_rew.append(r); tr_done.append(te | (tu << 1))
    out.update(b_actions=acts, b_obs0=obs0, b_obs=np.array(tr_obs), b_reward=np.array(tr_rew),
               b_flags=np.array(tr_done), b_state_out=st)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hover_step_%s.npz" % version)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main(*sys.argv[1:2])

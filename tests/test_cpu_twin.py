"""The oracle behind the product's C ABI (oracle/fpv_cpu.c, SURVEY.md §8b: "a parallel fpv_cpu_* set with identical signatures
over host pointers"): the twin exists for every entry point it names, takes the product's argument types, behaves like the
oracle's batch functions it wraps, and reports errors the product's way.  No GPU needed."""
import ctypes as C
import re
import os

import numpy as np
import pytest

from tests.abi_driver import TWIN, twin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_twin_mirrors_the_header_signatures():
    """Every fpv_cpu_X is declared in oracle/fpv_cpu.c with the parameter list include/fpv_env.h gives fpv_X."""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fpv_env.h")).read(), flags=re.S)
    source = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "oracle", "fpv_cpu.c")).read(), flags=re.S)

    def params(text, name):
        m = re.search(r"\b" + name + r"\s*\(([^)]*)\)\s*[;{]", text)
        assert m, name
        types = []
        for p in m.group(1).split(","):
            p = re.sub(r"\s+", " ", p.strip())
            if p in ("void", ""):
                continue
            types.append(re.sub(r"\s*\b[A-Za-z_][A-Za-z0-9_]*$", "", p).replace(" *", "*").strip())   # drop the parameter's name
        return types

    d = twin()
    for name in TWIN:
        cpu = "fpv_cpu_" + name[len("fpv_"):]
        assert hasattr(d.lib, cpu), cpu
        assert params(header, name) == params(source, cpu), name


def test_twin_steps_like_the_oracle_batch_functions(oracle):
    n, steps = 200, 40
    rng = np.random.default_rng(0)
    st = oracle.synthetic_states(n, seed=3)
    st[::7, 34] = 398                                       # truncations inside the window
    d = twin().create(n, precision=1, noise=0.02, seed=5, detector=0, ledger_capacity=4 * n)
    d.set_state(st)
    d.set_tick(1000)
    ref_state = np.ascontiguousarray(st.copy())
    cfg = oracle.make_cfg(noise_ratio=0.02, seed=5, detector="pixel")
    led = {"count": 0}
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, 4))
        obs, rew, term, trunc = d.step(a)
        o_obs, o_rew, o_term, o_trunc, _ = oracle.step_batch(cfg, ref_state, a, 1000 + t)
        assert np.array_equal(obs, o_obs) and np.array_equal(rew, o_rew)
        assert np.array_equal(term, o_term) and np.array_equal(trunc, o_trunc)
        led["count"] += int((o_term | o_trunc).sum())
    assert np.array_equal(d.get_state(), ref_state) and d.get_tick() == 1000 + steps
    got = d.ledger()
    assert got["count"] == led["count"] > 0 and len(got["idx"]) == got["count"]
    assert (got["tag"] >= 1000).all() and (got["tag"] < 1000 + steps).all()
    assert ((got["info"] & 3) != 0).all() and ((got["info"] >> 8) > 0).all()
    d.ledger_clear()
    assert d.ledger()["count"] == 0
    d.destroy()


def test_twin_ledger_pack_is_the_ledger_as_one_block(oracle):
    n = 256
    st = oracle.synthetic_states(n, seed=4)
    st[::3, 34] = 401
    d = twin().create(n, precision=0, noise=0.02, seed=5, ledger_capacity=2 * n)
    d.set_state(st)
    d.step(np.random.default_rng(0).uniform(-1, 1, (n, 4)))
    got = d.ledger()
    hdr, rows = d.ledger_pack(n)                           # fpv_ledger_pack: the same rows, dense, in (tile, row) order
    assert hdr[0] == hdr[1] == hdr[3] == got["count"] >= n // 3 and hdr[2] == 24
    assert np.array_equal(rows[:, 0].astype(np.int32), got["idx"]) and np.array_equal(rows[:, 1], got["info"])
    assert np.array_equal(np.ascontiguousarray(rows[:, 4:]).view(np.float32), got["obs"])
    d.ledger_clear()
    assert d.ledger_pack(8)[0][0] == 0
    d.destroy()


def test_twin_reset_keeps_prev_action_and_honours_the_mask(oracle):
    n = 64
    d = twin().create(n, precision=1, noise=0.0, detector=1)
    a = np.tile(np.array([0.1, -0.2, 0.3, 0.5]), (n, 1))
    d.step(a)
    before = d.get_state()
    mask = np.zeros(n, bool); mask[::2] = True
    obs = d.reset(mask)
    after = d.get_state()
    assert np.array_equal(after[1::2], before[1::2])                   # untouched drones
    assert np.allclose(after[::2, 30:34], before[::2, 30:34]) and (after[::2, 34] == 0).all()   # prev_action survives (hover.py:101)
    assert np.allclose(obs[::2, 16:20], 0.0)                           # the reset observation's action words are zero (hover.py:98)
    d.destroy()


def test_twin_reports_errors_the_products_way():
    d = twin()
    assert d.lib.fpv_cpu_create(None, None) == -1 and b"null" in d.lib.fpv_cpu_last_error()
    assert d.lib.fpv_cpu_version() == __import__("fpv_drone_rl_agent_amd")._lib.ABI_VERSION
    d.create(8, precision=1)
    with pytest.raises(RuntimeError, match="precision"):                # a float entry point on an fp64 handle
        d.call("step", d.h, d.ptr(np.zeros((8, 4), np.float32)), d.ptr(np.zeros((8, 20), np.float32)), d.ptr(np.zeros(8, np.float32)),
               d.ptr(d.term), d.ptr(d.trunc), None)
    d.destroy()
    assert d.lib.fpv_cpu_destroy(None) == 0


def test_twin_gae_matches_the_oracle(oracle):
    T, n = 9, 33
    rng = np.random.default_rng(1)
    r, v = rng.normal(size=(T, n)).astype(np.float32), rng.normal(size=(T, n)).astype(np.float32)
    s = (rng.random((T, n)) < 0.1).astype(np.uint8)
    lv, ld = rng.normal(size=n).astype(np.float32), (rng.random(n) < 0.1).astype(np.uint8)
    adv, ret, stats, ms = twin().gae(r, v, s, lv, ld)
    a64, r64 = oracle.gae(r.astype(np.float64), v.astype(np.float64), s, lv.astype(np.float64), ld, 0.99, 0.95)
    assert np.allclose(adv, a64, rtol=1e-6, atol=1e-6) and np.allclose(ret, r64, rtol=1e-6, atol=1e-6)
    assert stats[0] == T * n and np.isclose(ms[0], adv.astype(np.float64).mean()) and np.isclose(ms[1], adv.astype(np.float64).std())

"""The same call sequence through the same driver against libfpv_env.so (device buffers) and against the oracle's twin of the
C ABI (oracle/fpv_cpu.c, host buffers): create -> set_state -> set_tick -> step x K -> get_state / ledger_view -> reset(mask) -> gae.
fp64 handles agree to 1e-9 (masks and ledger bit-exact), fp32 handles to 1e-5 from identical states with one-pixel detector flips
allowed as everywhere (tests/parity.py)."""
import numpy as np
import pytest

from tests import parity
from tests.abi_driver import product, twin

pytestmark = pytest.mark.gpu


def _sequence(d, st, actions, precision, task=0, detector=0):
    n = st.shape[0]
    d.create(n, precision=precision, task=task, noise=0.02, seed=7, detector=detector, ledger_capacity=4 * n)
    d.set_state(st)
    d.set_tick(5000)
    out = {"steps": []}
    for a in actions:
        if precision == 0:                       # single steps from IDENTICAL states: the comparison the north_star states
            d.set_state(st)
        out["steps"].append(d.step(a))
        st = None if precision == 1 else st
    out["state"], out["tick"], out["ledger"] = d.get_state(), d.get_tick(), d.ledger()
    mask = np.zeros(n, bool); mask[::3] = True
    out["reset_obs"] = d.reset(mask)
    out["state_after_reset"] = d.get_state()
    d.destroy()
    return out


@pytest.mark.parametrize("detector", [0, 1])
def test_fp64_handles_agree_over_a_trajectory(oracle, detector):
    n, K = 1024, 24
    st = oracle.synthetic_states(n, seed=2)
    st[::9, 34] = 399
    rng = np.random.default_rng(0)
    actions = [rng.uniform(-1, 1, (n, 4)) for _ in range(K)]
    a, b = _sequence(product(), st, actions, 1, detector=detector), _sequence(twin(), st, actions, 1, detector=detector)
    for t, (x, y) in enumerate(zip(a["steps"], b["steps"])):
        assert np.array_equal(x[2], y[2]) and np.array_equal(x[3], y[3]), t          # done masks: bit-exact
        assert np.allclose(x[0], y[0], rtol=1e-9, atol=1e-9) and np.allclose(x[1], y[1], rtol=1e-9, atol=1e-9), t
    assert np.allclose(a["state"], b["state"], rtol=1e-9, atol=1e-9) and a["tick"] == b["tick"] == 5000 + K
    la, lb = a["ledger"], b["ledger"]
    assert la["count"] == lb["count"] > n // 20
    for k in ("idx", "info", "tag"):
        assert np.array_equal(la[k], lb[k]), k
    assert np.allclose(la["ep_return"], lb["ep_return"], rtol=1e-9, atol=1e-9) and np.allclose(la["obs"], lb["obs"], rtol=1e-9, atol=1e-9)
    assert np.allclose(a["reset_obs"][::3], b["reset_obs"][::3], atol=1e-12)
    assert np.allclose(a["state_after_reset"], b["state_after_reset"], rtol=1e-9, atol=1e-9)


def test_fp32_handles_agree_step_by_step_from_identical_states(oracle):
    n, K = 4096, 3
    st = oracle.synthetic_states(n, seed=5).astype(np.float32).astype(np.float64)    # values both precisions hold exactly
    st[::11, 34] = 401
    rng = np.random.default_rng(1)
    actions = [rng.uniform(-1, 1, (n, 4)).astype(np.float32) for _ in range(K)]
    a, b = _sequence(product(), st, actions, 0), _sequence(twin(), st, actions, 0)
    for t, (x, y) in enumerate(zip(a["steps"], b["steps"])):
        assert np.array_equal(x[2], y[2]) and np.array_equal(x[3], y[3]), t
        flips = parity.pixel_flips(x[0], y[0])
        scale = np.ones(20); scale[0:3] = 0.025                                      # obs[0:3] = wrap(d euler) / agent_dt (hover.py:230)
        assert parity.obs_close_except_flips(x[0], y[0], flips, 1e-5, scale).all()
        # (the reward sums several 1e-5 terms: 2e-5 as in tests/test_step_parity_gpu.py.  A row that auto-reset returns the reset
        # observation, so a one-pixel flip in its LAST observation shows in its reward only: such rows are bounded, not excluded)
        done = (x[2] | x[3]).astype(bool)
        bad = ~parity.close(x[1], y[1], 2e-5) & ~flips
        assert not (bad & ~done).any(), (t, np.where(bad & ~done)[0][:8], x[1][bad & ~done][:8], y[1][bad & ~done][:8])
        assert (bad & done).sum() <= max(2, int(0.01 * done.sum())), (t, int((bad & done).sum()), int(done.sum()))
    assert a["ledger"]["count"] == b["ledger"]["count"] > 0
    assert np.array_equal(a["ledger"]["idx"], b["ledger"]["idx"]) and np.array_equal(a["ledger"]["info"], b["ledger"]["info"])


def test_yaw_task_handles_agree(oracle):
    n, K = 512, 30
    rng = np.random.default_rng(2)
    outs = []
    for d in (product(), twin()):
        d.create(n, precision=1, task=1, noise=0.02, seed=3, domain_rand=1, ledger_capacity=4 * n)
        st0 = d.get_state()
        if outs:
            assert np.allclose(st0, outs[0]["state0"], rtol=1e-12, atol=1e-12)       # the Philox reset draws (heading, airframe)
        st = st0.copy()
        st[::5, 34] = 397
        d.set_state(st)
        rng_a = np.random.default_rng(4)
        steps = [d.step(rng_a.uniform(-1, 1, (n, 1))) for _ in range(K)]
        outs.append({"state0": st0, "steps": steps, "state": d.get_state(), "ledger": d.ledger()})
        d.destroy()
    a, b = outs
    for t, (x, y) in enumerate(zip(a["steps"], b["steps"])):
        assert np.array_equal(x[2], y[2]) and np.array_equal(x[3], y[3]), t
        assert np.allclose(x[0], y[0], rtol=1e-9, atol=1e-9) and np.allclose(x[1], y[1], rtol=1e-9, atol=1e-9), t
    assert np.allclose(a["state"], b["state"], rtol=1e-9, atol=1e-9)
    assert a["ledger"]["count"] == b["ledger"]["count"] > 0 and np.array_equal(a["ledger"]["info"], b["ledger"]["info"])


def test_gae_agrees(oracle):
    T, n = 32, 4096
    rng = np.random.default_rng(3)
    r, v = rng.normal(size=(T, n)).astype(np.float32), rng.normal(size=(T, n)).astype(np.float32)
    s = (rng.random((T, n)) < 0.05).astype(np.uint8)
    lv, ld = rng.normal(size=n).astype(np.float32), (rng.random(n) < 0.05).astype(np.uint8)
    a, b = product().gae(r, v, s, lv, ld), twin().gae(r, v, s, lv, ld)
    assert np.allclose(a[0], b[0], rtol=1e-5, atol=1e-5) and np.allclose(a[1], b[1], rtol=1e-5, atol=1e-5)
    assert a[2][0] == b[2][0] == T * n and np.allclose(a[2][1:], b[2][1:], rtol=1e-5) and np.allclose(a[3], b[3], rtol=1e-5, atol=1e-6)


def test_ledger_pack_is_the_ledger_as_one_block(oracle):
    """fpv_ledger_pack (the host-mode VecEnv's one copy per step): header + rows equal the ledger's own arrays in (tile, row) order,
    on the device and through the twin; with statistics the terminal observations come normalised as fpv_normalize does it; a
    max_rows below the row count packs the first rows and still reports every count."""
    n = 4096
    st = oracle.synthetic_states(n, seed=8).astype(np.float32).astype(np.float64)
    st[::5, 34] = 401                                                  # a truncation wave on top of the terminations
    a = np.random.default_rng(3).uniform(-1, 1, (n, 4)).astype(np.float32)
    rms = np.concatenate([np.linspace(-0.5, 0.5, 20), np.linspace(0.5, 2.0, 20), [100.0]])
    got = {}
    for name, d in (("product", product()), ("twin", twin())):
        d.create(n, precision=0, noise=0.02, seed=7, ledger_capacity=2 * n)
        d.set_state(st); d.set_tick(77)
        d.step(a)
        led = d.ledger()
        hdr, rows = d.ledger_pack(n)
        assert hdr[0] == hdr[1] == hdr[3] == led["count"] > n // 5 and hdr[2] == 24
        assert np.array_equal(rows[:, 0].astype(np.int32), led["idx"]) and np.array_equal(rows[:, 1], led["info"])
        assert np.array_equal(rows[:, 2], led["tag"]) and (np.diff(rows[:, 0].astype(np.int64)) > 0).all()
        assert np.array_equal(rows[:, 3].view(np.float32), led["ep_return"])
        assert np.array_equal(np.ascontiguousarray(rows[:, 4:]).view(np.float32), led["obs"])
        hdr_n, rows_n = d.ledger_pack(n, obs_rms=rms, clip=10.0, eps=1e-8)
        mu, inv = rms[:20].astype(np.float32), (1.0 / np.sqrt(rms[20:40] + np.float64(np.float32(1e-8)))).astype(np.float32)
        want = np.clip((led["obs"] - mu) * inv, -10.0, 10.0).astype(np.float32)
        assert np.allclose(np.ascontiguousarray(rows_n[:, 4:]).view(np.float32), want, rtol=0, atol=2e-6)
        hdr_s, rows_s = d.ledger_pack(100)
        assert hdr_s[0] == hdr_s[3] == led["count"] and hdr_s[1] == 100 and np.array_equal(rows_s, rows[:100])
        got[name] = (hdr, rows)
        d.destroy()
    (ha, ra), (hb, rb) = got["product"], got["twin"]
    assert np.array_equal(ha, hb) and np.array_equal(ra[:, :3], rb[:, :3])           # who ended, how, when: bit-exact

"""Edge cases of the C ABI on the device: ragged sizes around the 64-drone tile, argument errors, handle churn."""
import ctypes as C

import numpy as np
import pytest

from tests import parity

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 129, 1000])
def test_ragged_sizes_step_fused_and_ledger(oracle, n):
    """Sizes that do not fill the last tile: plain and fused step agree with the oracle, nothing is written past n."""
    from fpv_drone_rl_agent_amd import ops
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    st0 = oracle.synthetic_states(n, seed=n)
    st0[::3, 34] = 401                                   # a third of the drones truncate
    st0 = st0.astype(np.float32)
    act = np.random.default_rng(n).uniform(-1, 1, (n, 4)).astype(np.float32)
    ref = st0.astype(np.float64)
    cfg = oracle.make_cfg(noise_ratio=0.02, seed=n)
    o_r, r_r, te_r, tr_r, led = oracle.step_batch(cfg, ref, act.astype(np.float64), 3)
    for fused in (False, True):
        eng = FpvEngine(n, 0, seed=n)
        eng.set_state(torch.from_numpy(st0))
        eng.set_tick(3)
        guard = torch.full((n + 64, 20), -9.0, device="cuda")          # canary rows behind the real ones
        rew = torch.full((n + 64,), -9.0, device="cuda")
        a = torch.from_numpy(act).cuda()
        if fused:
            done = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
            ret = torch.zeros(n + 64, device="cuda")
            part = torch.zeros((eng.tiles, 44), dtype=torch.float64, device="cuda")
            eng.step_fused(a, done=done[:n], ret=ret[:n], partials=part, gamma=0.99, obs=guard[:n], reward=rew[:n])
            assert (done[:n].cpu().numpy() == ((te_r | tr_r) > 0)).all() and (done[n:] == 0).all() and (ret[n:] == 0).all()
            o = guard[:n].double()
            tot = part.sum(0).cpu().numpy()
            assert np.allclose(tot[:20], o.sum(0).cpu().numpy(), rtol=1e-9, atol=1e-9)
            assert np.allclose(tot[20:40], (o * o).sum(0).cpu().numpy(), rtol=1e-9, atol=1e-9)
            rms = torch.zeros(41, dtype=torch.float64, device="cuda"); rms[20:40] = 1; rms[40] = 1e-4
            ops.moments_merge(part, n, obs_rms=rms)
            assert np.isclose(float(rms[40]), n + 1e-4)
        else:
            eng.step(a, obs=guard[:n], reward=rew[:n])
        assert (guard[n:] == -9.0).all() and (rew[n:] == -9.0).all()    # no write past the last drone
        assert (eng.terminated.cpu().numpy() == te_r).all() and (eng.truncated.cpu().numpy() == tr_r).all()
        st = eng.get_state().cpu().numpy()
        flips = parity.pixel_flips(guard[:n].cpu().numpy(), o_r)             # one-pixel rasterisation flips (tests/parity.py)
        assert parity.state_close_except_flips(st, ref, flips).all()
        got = eng.ledger_read()
        assert (got["idx"] == led.rows()[0]).all() and got["count"] == len(led.rows()[0])
        eng.close()


def test_abi_argument_errors_on_device():
    from fpv_drone_rl_agent_amd import _lib as L
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    lib = L.load()
    eng = FpvEngine(128, 0)
    a = torch.zeros(128, 4, device="cuda")
    bad = torch.zeros(128 * 4 + 1, device="cuda")[1:].view(128, 4)        # 4-byte aligned only
    assert lib.fpv_step(eng._h, C.c_void_p(bad.data_ptr()), C.c_void_p(eng.obs.data_ptr()), C.c_void_p(eng.reward.data_ptr()),
                        C.c_void_p(eng.terminated.data_ptr()), C.c_void_p(eng.truncated.data_ptr()), None) == -1
    assert b"aligned" in lib.fpv_last_error()
    # fp32 handle through the fp64 entry point
    assert lib.fpv_step_f64(eng._h, C.c_void_p(a.data_ptr()), C.c_void_p(eng.obs.data_ptr()), C.c_void_p(eng.reward.data_ptr()),
                            C.c_void_p(eng.terminated.data_ptr()), C.c_void_p(eng.truncated.data_ptr()), None) == -1
    assert b"precision" in lib.fpv_last_error()
    cfg = L.FpvConfig(64, 99, 0, 1, 0, 0.02, 0, 0, 0, 0, 0)               # device that does not exist
    h = C.c_void_p()
    assert lib.fpv_create(C.byref(cfg), C.byref(h)) == -1 and b"not present" in lib.fpv_last_error()
    cfg = L.FpvConfig(64, 0, 0, 1, 7, 0.02, 0, 0, 0, 0, 0)                # unknown task
    assert lib.fpv_create(C.byref(cfg), C.byref(h)) == -1 and b"task" in lib.fpv_last_error()
    with pytest.raises(ValueError):                                       # yaw actions are [n, 1]
        FpvEngine(64, 0, task="yaw").step_fused(torch.zeros(64, 4, device="cuda"))
    with pytest.raises(ValueError):                                       # yaw partial rows are 28 words wide
        FpvEngine(64, 0, task="yaw").step_fused(torch.zeros(64, 1, device="cuda"),
                                                 partials=torch.zeros(1, 44, dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        FpvEngine(64, 0, precision="f64").step_fused(torch.zeros(64, 4, device="cuda", dtype=torch.float64))
    eng.step(a)                                                           # the handle is still usable after errors
    eng.close()


def test_handle_churn_does_not_leak():
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    torch.cuda.synchronize()
    a = torch.zeros(4096, 4, device="cuda")
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(40):
        eng = FpvEngine(4096, 0, task="yaw" if _ % 2 else "hover")
        eng.reset()
        eng.close()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < 64 * 2 ** 20           # nothing accumulates across create/destroy


@pytest.mark.parametrize("n", [65536, 12288 - 40])
def test_quad_kernel_is_bit_identical_to_the_other_launch_shapes(n):
    """step_quad_kernel (four tiles per workgroup, physics + helper wave per tile sharing the pixel detector's rows: the default
    for fp32 pixel-detector handles of 8 192 < n <= 65 536 drones) against the single-wave and the wave-pair kernel: fpv_step and
    fpv_step_fused, auto-resets and ledger included, a ragged last tile in the second size; the partial rows are per workgroup
    of four tiles in one and per tile / four tiles in the others: same totals."""
    from fpv_drone_rl_agent_amd import _lib as L
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.synthetic import synthetic_states
    lib = L.load()
    st0 = synthetic_states(n, seed=4).astype(np.float32)
    st0[::5, 34] = 399
    outs = []
    try:
        for quad, pair in (((0, 1 << 40), 32768), ((0, 0), 0), ((0, 0), 1 << 30)):
            lib.fpv_set_quad_range(*quad)
            lib.fpv_set_pair_threshold(pair)
            eng = FpvEngine(n, 0, seed=9, ledger_capacity=4 * n)
            assert eng.detector == "pixel"
            eng.set_state(torch.from_numpy(st0))
            g = torch.Generator(device="cuda").manual_seed(0)
            done = torch.zeros(n, dtype=torch.uint8, device="cuda"); ret = torch.zeros(n, device="cuda")
            part = torch.zeros((eng.partial_rows, 44), dtype=torch.float64, device="cuda")
            rews, sums = [], []
            for t in range(6):
                a = torch.rand(n, 4, device="cuda", generator=g) * 2 - 1
                if t % 2:
                    eng.step_fused(a, done=done, ret=ret, partials=part)
                    sums.append(part.sum(dim=0).clone())
                else:
                    eng.step(a)
                rews.append(eng.reward.clone())
            led = eng.ledger_read()
            outs.append({"state": eng.get_state().clone(), "obs": eng.obs.clone(), "rew": torch.stack(rews), "ret": ret.clone(), "done": done.clone(),
                         "term": eng.terminated.clone(), "trunc": eng.truncated.clone(), "sums": torch.stack(sums), "rows": eng.partial_rows,
                         "led": led})
            eng.close()
    finally:
        lib.fpv_set_quad_range(8192, 65536)
        lib.fpv_set_pair_threshold(32768)
    tiles = (n + 63) // 64
    assert outs[0]["rows"] == tiles // 4 and outs[2]["rows"] == tiles
    assert outs[0]["led"]["count"] > n // 10
    for other in outs[1:]:
        for k in ("state", "obs", "rew", "ret", "done", "term", "trunc"):
            assert torch.equal(outs[0][k], other[k]), k
        assert torch.allclose(outs[0]["sums"], other["sums"], rtol=1e-12, atol=1e-9)
        assert outs[0]["led"]["count"] == other["led"]["count"]
        for k in ("idx", "info", "tag", "ep_return", "obs"):
            assert np.array_equal(outs[0]["led"][k], other["led"][k]), k


@pytest.mark.parametrize("n,detector,rows", [(8192, "pixel", 128), (8448, "pixel", 33), (65536, "pixel", 256), (65792, "pixel", 257),
                                             (65536, "analytic", 256), (32768, "analytic", 512), (4160, "pixel", 65)])
def test_partial_rows_follow_the_launch_shape(n, detector, rows):
    """fpv_partial_rows mirrors launch_step's dispatch at its boundaries (wave-pair kernel: a row per tile up to 32 768 drones;
    step_quad_kernel: a row per four tiles for pixel-detector handles of 8 192 < n <= 65 536; else a row per four tiles when the
    tile count divides, per tile when not) — and a fused step writes exactly those rows: the rest of an oversized table stays
    untouched, the written rows add up to the drones' moment sums."""
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    eng = FpvEngine(n, 0, seed=1, detector=detector)
    assert eng.partial_rows == rows
    part = torch.full((rows + 8, 44), -7.0, dtype=torch.float64, device="cuda")
    done = torch.zeros(n, dtype=torch.uint8, device="cuda"); ret = torch.zeros(n, device="cuda")
    eng.step_fused(torch.rand(n, 4, device="cuda") * 2 - 1, done=done, ret=ret, partials=part)
    torch.cuda.synchronize()
    assert (part[rows:] == -7.0).all() and (part[:rows, 40] != -7.0).all()
    o = eng.obs.double()
    assert torch.allclose(part[:rows, :20].sum(dim=0), o.sum(dim=0), rtol=1e-9, atol=1e-6)
    assert torch.allclose(part[:rows, 20:40].sum(dim=0), (o * o).sum(dim=0), rtol=1e-9, atol=1e-6)
    eng.close()


def test_pair_and_single_wave_kernels_are_bit_identical():
    """fpv_set_pair_threshold only changes the schedule (physics wave + helper wave), never the results."""
    from fpv_drone_rl_agent_amd import _lib as L
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.synthetic import synthetic_states
    lib = L.load()
    n = 3000
    st0 = synthetic_states(n, seed=4).astype(np.float32)
    st0[::5, 34] = 401
    outs = []
    for thr in (0, 1 << 30):
        lib.fpv_set_pair_threshold(thr)
        eng = FpvEngine(n, 0, seed=9)
        eng.set_state(torch.from_numpy(st0))
        g = torch.Generator(device="cuda").manual_seed(0)
        done = torch.zeros(n, dtype=torch.uint8, device="cuda"); ret = torch.zeros(n, device="cuda")
        part = torch.zeros((eng.tiles, 44), dtype=torch.float64, device="cuda")
        rews = []
        for t in range(5):
            a = torch.rand(n, 4, device="cuda", generator=g) * 2 - 1
            if t % 2:
                eng.step_fused(a, done=done, ret=ret, partials=part)
            else:
                eng.step(a)
            rews.append(eng.reward.clone())
        led = eng.ledger_read()
        outs.append((eng.get_state().clone(), eng.obs.clone(), torch.stack(rews), ret.clone(), part.clone(), led["idx"], led["obs"]))
        eng.close()
    lib.fpv_set_pair_threshold(32768)
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(a, b)
    # the moment sums come per tile from the pair kernel and per four tiles from the single-wave kernel: same totals
    assert torch.allclose(outs[0][4].sum(0), outs[1][4].sum(0), rtol=1e-12, atol=1e-12)
    assert np.array_equal(outs[0][5], outs[1][5]) and np.array_equal(outs[0][6], outs[1][6])

"""GPU tests of the reference-shaped surfaces and of the whole on-device rollout
against an oracle replay (env via oracle/fpv_oracle.c, VecNormalize/GAE via the numpy/C restatements,
policy via the same torch module on the CPU)."""
import copy
import os

import numpy as np
import pytest

from tests import parity

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_single_env_protocol(oracle):
    from fpv_drone_rl_agent_amd.hover import QuadXHoverEnv
    env = QuadXHoverEnv(noise_ratio=0.0)
    obs, info = env.reset(seed=1)
    assert obs.shape == (20,) and obs.dtype == np.float64
    assert info == {"out_of_bounds": False, "collision": False, "env_complete": False, "on_floor": False}
    st_ref, obs_ref = oracle.reset_states(1)
    assert np.allclose(obs, obs_ref[0], atol=1e-5)
    cfg = oracle.make_cfg(noise_ratio=0.0)
    a = np.array([0.0, 0.0, 0.0, -1.0])
    for call in range(1, 33):
        o, r, term, trunc, info = env.step(a)
        o_ref, r_ref, bits = oracle.step_one(cfg, st_ref[0], a, 0, call - 1)
        assert isinstance(r, float) and isinstance(term, bool) and isinstance(trunc, bool)
        assert np.allclose(o, o_ref, atol=2e-3, rtol=1e-3) and abs(r - r_ref) < 2e-3
        assert term == bool(bits & 1) and trunc == bool(bits & 2)
    assert term and info["on_floor"] and not info["out_of_bounds"] and call == 32
    obs, info = env.reset()
    assert not info["on_floor"] and np.allclose(obs, obs_ref[0], atol=1e-5)
    with pytest.raises(ValueError):
        QuadXHoverEnv(flight_mode=6)
    env.close()


@pytest.mark.parametrize("n", [16, 4096])     # train_hover.py:41's n_envs; BASELINE.json configs[1]
def test_vec_env_host_mode_matches_oracle(oracle, n):
    from fpv_drone_rl_agent_amd.vec_env import make_vec_env
    env = make_vec_env(None, n_envs=n, noise_ratio=0.0)
    obs = env.reset()
    st, obs_ref = oracle.reset_states(n)
    assert obs.shape == (n, 20) and obs.dtype == np.float32 and np.allclose(obs, obs_ref, atol=1e-5)
    cfg = oracle.make_cfg(noise_ratio=0.0)
    rng = np.random.default_rng(0)
    saw_done = False
    prev_infos = None
    t_got, t_ref, ep_got, ep_ref = [], [], [], []
    for t in range(45):
        a = rng.uniform(-1, 1, (n, 4))
        a[:, 3] = -1.0 if t < 35 else 0.9
        if t >= 35:
            a[:, :3] *= 0.2        # gentle rates once airborne: at +-30 rad/s an fp32 trajectory leaves the fp64 one within steps
        o, r, d, infos = env.step(a)
        o_ref, r_ref, te, tr, led = oracle.step_batch(cfg, st, a.astype(np.float32).astype(np.float64), t)
        assert o.dtype == np.float32 and r.dtype == np.float32 and d.dtype == bool and len(infos) == n
        assert (d == ((te | tr) > 0)).all()
        # fp32 device trajectory vs fp64 oracle trajectory over 45 steps: the difference accumulates; with the pixel detector
        # (the default) a drone may rasterise one pixel differently at a step (tests/parity.py): at most two rows per step
        parity.rows_close(o, o_ref, atol=2e-3, rtol=1e-3, what=f"obs, step {t}")
        parity.rows_close(r, r_ref, atol=2e-3, rtol=1e-3, what=f"reward, step {t}")
        for k, i in enumerate(led.rows()[0]):
            saw_done = True
            inf = infos[i]
            t_got.append(np.delete(inf["terminal_observation"], parity.FEATURE_OBS + [9, 10, 12, 15]))
            t_ref.append(np.delete(led.rows()[3][k], parity.FEATURE_OBS + [9, 10, 12, 15]))
            bits = int(led.rows()[1][k])            # the oracle's ledger entry: FPV_INFO_* | ep_len << 8
            assert inf["TimeLimit.truncated"] is bool((bits & 2) and not (bits & 1)) and inf["episode"]["l"] == bits >> 8
            assert inf["on_floor"] == bool(bits & oracle.FLOOR)
            if n == 16 and t < 35:     # (these sixteen stay down at zero throttle: the floor rule ends them on call 32; among 4 096
                assert inf["episode"]["l"] == 32 and inf["on_floor"]      # some are lifted by the rate controller's saturated mixer)
            ep_got.append(inf["episode"]["r"]); ep_ref.append(led.rows()[2][k])
        for i in np.nonzero(~d)[0]:          # hover.py:102,358: a live env returns its four flags on every step
            assert infos[i] == {"out_of_bounds": False, "collision": False, "env_complete": False, "on_floor": False}
        assert len({id(x) for x in infos}) == n and (prev_infos is None or not ({id(x) for x in infos} & {id(x) for x in prev_infos}))
        for x in infos:                      # what a wrapper writes into this step's dicts (SB3's VecMonitor: "episode") ...
            x["written_by_a_wrapper"] = t
        prev_infos = infos                   # ... never shows in a later step's
    assert saw_done
    # terminal observations (without the camera words): airborne drones under +-30 rad/s rate commands tumble, and an fp32
    # trajectory leaves the fp64 one within a few steps — a small share of the late episode ends may differ
    parity.rows_close(np.array(t_got), np.array(t_ref), atol=2e-3, rtol=1e-3, max_bad_frac=0.02, what="terminal observations")
    parity.rows_close(np.array(ep_got), np.array(ep_ref), atol=1e-2, max_bad_frac=0.02, what="episode returns")
    assert env.get_attr("max_steps") == [400] * n and env.env_is_wrapped(object) == [False] * n
    env.close()


def test_vecnormalize_matches_sb3_arithmetic_and_roundtrips(oracle, tmp_path):
    from fpv_drone_rl_agent_amd.vec_env import FpvVecEnv, VecNormalize
    n = 32
    venv = VecNormalize(FpvVecEnv(n, noise_ratio=0.0), norm_obs=True, norm_reward=True)
    ref = oracle.VecNormalizeRef(n)
    st, obs_ref = oracle.reset_states(n)
    o = venv.reset()
    assert np.allclose(o, ref.reset(obs_ref), atol=1e-4)
    cfg = oracle.make_cfg(noise_ratio=0.0)
    rng = np.random.default_rng(1)
    seen_terminal, t_got, t_want = 0, [], []
    for t in range(40):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        o, r, d, infos = venv.step(a)
        led = oracle.Ledger(n)
        o_raw, r_raw, te, tr, _ = oracle.step_batch(cfg, st, a.astype(np.float64), t, ledger=led)
        o_ref, r_ref = ref.step(o_raw, r_raw, (te | tr) > 0)
        parity.rows_close(o, o_ref, atol=2e-3, what=f"normalised obs, step {t}")
        parity.rows_close(r, r_ref, atol=1e-4, what=f"normalised reward, step {t}")
        # SB3's VecNormalize.step_wait normalises infos[i]["terminal_observation"] with the statistics as they stand AFTER this
        # step's update; here that happens on the device, inside the launch that packs the done ledger (fpv_ledger_pack)
        for k, i in enumerate(led.rows()[0]):
            seen_terminal += 1
            keep = [j for j in range(20) if j not in parity.FEATURE_OBS]      # (one-pixel flips of the fp32 path aside)
            got = infos[i]["terminal_observation"]
            assert got.dtype == np.float32 and got.shape == (20,)
            t_got.append(got[keep]); t_want.append(ref.normalize_obs(led.rows()[3][k])[keep])
            assert set(infos[i]) >= {"terminal_observation", "TimeLimit.truncated", "episode", "out_of_bounds", "on_floor"}
        for i in np.nonzero(~d)[0]:
            assert infos[i] == {"out_of_bounds": False, "collision": False, "env_complete": False, "on_floor": False}
    assert seen_terminal > 0
    # (an fp32 trajectory of a tumbling drone leaves the fp64 one within tens of steps: a small share of the late episode ends may differ)
    parity.rows_close(np.array(t_got), np.array(t_want), atol=5e-3, max_bad_frac=0.1, what="normalised terminal observations")
    path = os.path.join(tmp_path, "hover.pkl")            # train_hover.py:27 naming
    with pytest.raises(ValueError, match="pickle"):        # a .pkl path gets this build's npz archive only on request
        venv.save(path)
    venv.save(path, format="npz")
    loaded = VecNormalize.load(path, venv.venv)            # test_hover.py:8
    import pickle
    real = os.path.join(tmp_path, "sb3_like.pkl")          # what stable-baselines3 itself would have written: a pickle
    with open(real, "wb") as f:
        pickle.dump({"obs_rms": None}, f)
    with pytest.raises(ValueError, match="Python pickle"):
        VecNormalize.load(real, venv.venv)
    loaded.training = False                                # test_hover.py:9-10
    loaded.norm_reward = False
    assert torch.equal(loaded.obs_rms.rms, venv.obs_rms.rms)
    before = loaded.obs_rms.rms.clone()
    o2, r2, d2, _ = loaded.step(np.zeros((n, 4), np.float32))
    assert torch.equal(before, loaded.obs_rms.rms)         # frozen statistics
    assert np.allclose(r2, venv.venv.engine.reward.cpu().numpy())  # raw rewards
    venv.close()


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("use_graph", [False, True])
def test_rollout_collector_matches_oracle_replay(oracle, use_graph, fused):
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    n, T, gamma, lam = 256, 8, 0.99, 0.95
    torch.manual_seed(3)
    pol = ActorCritic().cuda()
    pol_cpu = copy.deepcopy(pol).cpu().double()
    eng = FpvEngine(n, 0, noise_ratio=0.0)
    col = RolloutCollector(eng, pol, n_steps=T, gamma=gamma, lam=lam, use_graph=use_graph, fused=fused)
    col.start()
    ref_norm = oracle.VecNormalizeRef(n, gamma=gamma)
    _, obs_reset = oracle.reset_states(n)
    ref_norm.reset(obs_reset)
    st = oracle.synthetic_states(n, seed=9)
    st[:40, 34] = np.random.default_rng(0).integers(395, 402, 40)   # truncations inside the rollout
    st[40:60, 2] = 0.12; st[40:60, 9] = -1.0; st[40:60, 34] = 100   # floor terminations
    st = st.astype(np.float32).astype(np.float64)
    if use_graph:
        col._capture()          # the capture warm-up consumes one env step; load the test state after it
        tick0 = eng.get_tick()
    eng.set_state(torch.from_numpy(st.astype(np.float32)))
    if use_graph:
        # statistics were touched by the warm-up step: re-sync the reference normaliser to the device state
        rms = col.obs_norm.rms.cpu().numpy()
        ref_norm.obs_rms.mean, ref_norm.obs_rms.var, ref_norm.obs_rms.count = rms[:20].copy(), rms[20:40].copy(), float(rms[40])
        rr = col.rew_norm.rms.cpu().numpy()
        ref_norm.ret_rms.mean, ref_norm.ret_rms.var, ref_norm.ret_rms.count = float(rr[0]), float(rr[1]), float(rr[2])
        ref_norm.returns = col.rew_norm.ret.cpu().numpy().astype(np.float64)
    else:
        tick0 = eng.get_tick()
    start0 = col.episode_start.cpu().numpy().copy()
    out = col.collect()
    torch.cuda.synchronize()

    cfg = oracle.make_cfg(noise_ratio=0.0)
    acts = col.buf_act.cpu().numpy().astype(np.float64)
    rew_ref = np.zeros((T, n)); starts_ref = np.zeros((T, n), np.uint8); obs_n_ref = []
    starts = start0
    for t in range(T):
        starts_ref[t] = starts
        o_raw, r_raw, te, tr, led = oracle.step_batch(cfg, st, np.clip(acts[t], -1, 1), tick0 + t)
        done = (te | tr) > 0
        o_n, r_n = ref_norm.step(o_raw, r_raw, done)
        idx, info, _, tobs = led.rows()
        for k in range(len(idx)):                       # SB3 timeout bootstrap
            if (info[k] & 2) and not (info[k] & 1):
                tn = ref_norm.normalize_obs(tobs[k][None])
                r_n[idx[k]] += gamma * float(pol_cpu.value(torch.from_numpy(tn))[0])
        rew_ref[t] = r_n
        obs_n_ref.append(o_n)
        starts = done.astype(np.uint8)
    assert starts_ref[1:].any() and (rew_ref != 0).all()
    assert (col.buf_start.cpu().numpy() == starts_ref).all()
    got_obs = col.buf_obs.cpu().numpy()
    for t in range(1, T):
        parity.rows_close(got_obs[t], obs_n_ref[t - 1], atol=3e-3, what=f"rollout obs, step {t}")
    parity.rows_close(col.cur_obs.cpu().numpy(), obs_n_ref[-1], atol=3e-3, what="s_T")
    parity.rows_close(col.buf_rew.cpu().numpy().reshape(-1), rew_ref.reshape(-1), atol=2e-3, what="rollout rewards")
    # values / log-probs are the policy's; GAE and the advantage statistics through the oracle
    with torch.no_grad():
        v_cpu = pol_cpu.value(torch.from_numpy(got_obs.astype(np.float64))).numpy()
    assert np.allclose(col.buf_val.cpu().numpy(), v_cpu, atol=1e-4)
    adv_ref, ret_ref = oracle.gae(col.buf_rew.cpu().numpy(), col.buf_val.cpu().numpy(), starts_ref,
                                  col.last_val.cpu().numpy(), starts, gamma, lam)
    assert np.allclose(out["advantages"].cpu().numpy(), adv_ref, atol=1e-4)
    assert np.allclose(out["returns"].cpu().numpy(), ret_ref, atol=1e-4)
    assert abs(float(out["adv_mean"]) - adv_ref.mean()) < 1e-5 and abs(float(out["adv_std"]) - adv_ref.std()) < 1e-5
    eng.close()


def test_f64_engine_rollout_env_outputs_equal_the_oracle(oracle):
    """The rollout in the reference's own arithmetic type: its spaces are float64 (simulation/hover.py:61, :70), numpy and pybullet
    compute in double, SB3's torch policy in float32.  RolloutCollector on an fp64 engine = the fp32 policy kernel + fpv_step_f64
    with the conversions at the seam.  Replayed through the oracle with the collector's own (clipped, fp32-valued) actions, the
    env's raw outputs of every step agree to 1e-9 and the done masks exactly — over a 16-step rollout of 4 096 drones with motor
    noise, episode ends and auto-resets inside; the collector's normalised buffers agree with the reference normaliser fed with
    those outputs."""
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    n, T, gamma = 4096, 16, 0.99
    torch.manual_seed(5)
    pol = ActorCritic().cuda()
    eng = FpvEngine(n, 0, noise_ratio=0.02, seed=11, precision="f64")
    col = RolloutCollector(eng, pol, n_steps=T, gamma=gamma)
    col.start()
    ref_norm = oracle.VecNormalizeRef(n, gamma=gamma)
    _, obs_reset = oracle.reset_states(n)
    ref_norm.reset(obs_reset)
    st = oracle.synthetic_states(n, seed=19)
    st[:200, 34] = np.random.default_rng(1).integers(390, 402, 200)      # truncations inside the rollout
    st[200:300, 2] = 0.12; st[200:300, 9] = -1.0; st[200:300, 34] = 100  # floor terminations
    eng.set_state(torch.from_numpy(st))
    tick0 = eng.get_tick()
    start0 = col.episode_start.cpu().numpy().copy()
    out = col.collect()
    torch.cuda.synchronize()
    cfg = oracle.make_cfg(noise_ratio=0.02, seed=11)
    acts = np.clip(col.buf_act.cpu().numpy(), -1.0, 1.0).astype(np.float64)      # what crossed the seam: fp32 values, held in fp64
    env_obs, env_rew = col.env_obs.cpu().numpy(), col.env_rew.cpu().numpy()
    starts, ends = start0, 0
    for t in range(T):
        assert (col.buf_start[t].cpu().numpy() == starts).all()
        o_raw, r_raw, te, tr, _ = oracle.step_batch(cfg, st, acts[t], tick0 + t, nthreads=8)
        assert np.abs(env_obs[t] - o_raw).max() <= 1e-9 and np.abs(env_rew[t] - r_raw).max() <= 1e-9
        done = (te | tr) > 0
        ends += int(done.sum())
        o_n, _ = ref_norm.step(o_raw, r_raw, done)
        if t + 1 < T:
            assert np.abs(col.buf_obs[t + 1].cpu().numpy() - o_n).max() <= 1e-4      # fp32 normaliser on fp32-rounded observations
        starts = done.astype(np.uint8)
    assert ends > 250                                                  # episode ends and auto-resets were part of it
    assert np.abs(eng.get_state().cpu().numpy() - st).max() <= 1e-9
    assert torch.isfinite(out["advantages"]).all() and float(out["adv_std"]) > 0
    eng.close()


def test_ledger_scan_and_scatter_match_a_torch_restatement(oracle):
    """fpv_ledger_scan / fpv_ledger_scatter (SB3's timeout bootstrap, no host round trip) against the same rule
    written with torch indexing on the ledger read back to the host."""
    from fpv_drone_rl_agent_amd import ops
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.ops_policy import FusedPolicy
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    n, T, gamma = 4096, 6, 0.99
    torch.manual_seed(11)
    eng = FpvEngine(n, 0, seed=4, ledger_capacity=4 * n)
    st = oracle.synthetic_states(n, seed=4)
    rng = np.random.default_rng(2)
    st[: n // 3, 34] = rng.integers(396, 402, n // 3)                 # truncations inside the window
    st[n // 3: n // 2, 2] = 0.12; st[n // 3: n // 2, 9] = -1.0; st[n // 3: n // 2, 34] = 100   # floor terminations
    eng.set_state(torch.from_numpy(st.astype(np.float32)))
    eng.set_tick(1000)
    eng.ledger_clear()
    for t in range(T):
        eng.step(torch.rand(n, 4, device="cuda") * 2 - 1)
    led = eng.ledger_read()
    snap = torch.zeros((T, 44), dtype=torch.float64, device="cuda")
    snap[:, :20] = torch.randn(T, 20, dtype=torch.float64, device="cuda") * 0.3
    snap[:, 20:40] = torch.rand(T, 20, dtype=torch.float64, device="cuda") + 0.5
    pol = ActorCritic().cuda()
    fp = FusedPolicy(pol)
    cap = n
    tobs = torch.zeros((cap, 20), device="cuda"); t_idx = torch.zeros(cap, dtype=torch.int32, device="cuda")
    d_idx = torch.zeros(cap, dtype=torch.int32, device="cuda"); tv = torch.zeros(cap, device="cuda")
    counters = torch.full((4,), 77, dtype=torch.int32, device="cuda"); sums = torch.full((2,), 5.0, dtype=torch.float64, device="cuda")
    # status_out: the same totals written by the scan's last workgroup straight into pinned host memory (and, second call, into
    # device memory), no copy queued; the ticket it takes resets itself, so the launch repeats
    host = torch.full((4,), -1.0, dtype=torch.float64).pin_memory()
    ops.ledger_scan(eng, T, snapshots=snap, clip_obs=10.0, eps=1e-8, tobs=tobs, t_idx=t_idx, d_idx=d_idx, counters=counters, sums=sums,
                    status_out=host)
    torch.cuda.synchronize()
    assert host[:2].view(torch.int32).tolist() == counters.tolist() and host[2:4].tolist() == sums.tolist() and int(counters[1]) > 0
    dev_out = torch.zeros(4, dtype=torch.float64, device="cuda")
    ops.ledger_scan(eng, T, snapshots=snap, clip_obs=10.0, eps=1e-8, tobs=tobs, t_idx=t_idx, d_idx=d_idx, counters=counters, sums=sums,
                    status_out=dev_out)
    torch.cuda.synchronize()
    assert torch.equal(dev_out.cpu(), host)
    with pytest.raises(ValueError, match="pinned"):
        ops.ledger_scan(eng, T, snapshots=snap, clip_obs=10.0, eps=1e-8, tobs=tobs, t_idx=t_idx, d_idx=d_idx, counters=counters,
                        sums=sums, status_out=torch.zeros(4, dtype=torch.float64))
    # fpv_ledger_scan_finalize: the same scan with the rollout's deferred reward scaling in the same launch (what the collector's tail
    # calls) — the counters, the list as a set, the status words and the scaled rewards of fpv_ledger_scan + fpv_reward_finalize
    raw = torch.randn((T, n), device="cuda") * 3
    snap[:, 42] = torch.rand(T, dtype=torch.float64, device="cuda") * 0.05 + 0.01          # return variances: some rewards reach the clip
    want_rew = ops.reward_finalize(raw, snap, clip=10.0, eps=1e-8)
    c2 = torch.full((4,), 9, dtype=torch.int32, device="cuda"); s2 = torch.full((2,), 1.0, dtype=torch.float64, device="cuda")
    tobs2, t2, d2 = torch.zeros_like(tobs), torch.zeros_like(t_idx), torch.zeros_like(d_idx)
    got_rew = torch.zeros_like(raw)
    host2 = torch.full((4,), -1.0, dtype=torch.float64).pin_memory()
    ops.ledger_scan(eng, T, snapshots=snap, clip_obs=10.0, eps=1e-8, tobs=tobs2, t_idx=t2, d_idx=d2, counters=c2, sums=s2, status_out=host2,
                    finalize={"raw": raw, "snapshots": snap, "clip": 10.0, "eps": 1e-8, "out": got_rew})
    torch.cuda.synchronize()
    assert torch.equal(got_rew, want_rew) and float(got_rew.abs().max()) == 10.0
    assert torch.equal(c2, counters) and torch.equal(host2[:2], host[:2]) and host2[2:4].tolist() == s2.tolist()
    assert abs(float(s2[0] - sums[0])) <= 1e-9 * abs(float(sums[0])) and s2[1] == sums[1]       # (atomic adds: the order may differ)
    kk = int(counters[0])
    key = lambda t_, d_: set(zip(t_[:kk].tolist(), d_[:kk].tolist()))
    assert key(t2, d2) == key(t_idx, d_idx)
    o1 = {(int(a_), int(b_)): r_ for a_, b_, r_ in zip(t_idx[:kk].tolist(), d_idx[:kk].tolist(), tobs[:kk].cpu().numpy())}
    for a_, b_, r_ in zip(t2[:kk].tolist(), d2[:kk].tolist(), tobs2[:kk].cpu().numpy()):
        assert np.array_equal(o1[(a_, b_)], r_)
    fp.forward(tobs, value=tv, deterministic=True, nets="v", n_dev=counters[0:1])
    rew = torch.zeros((T, n), device="cuda")
    ops.ledger_scatter(counters, t_idx, d_idx, tv, gamma, rew)
    # the same addition inside the launch that evaluates V (fpv_policy_io.scatter_*): what the collector's tail uses
    rew_in, tv_in = torch.zeros((T, n), device="cuda"), torch.zeros(cap, device="cuda")
    fp.forward(tobs, value=tv_in, deterministic=True, nets="v", n_dev=counters[0:1],
               scatter={"t_idx": t_idx, "d_idx": d_idx, "out": rew_in, "scale": gamma})
    torch.cuda.synchronize()
    assert torch.equal(tv_in, tv) and torch.equal(rew_in, rew) and int((rew != 0).sum()) == int(counters[0])
    info, tag, idx = led["info"], led["tag"], led["idx"]
    mask = ((info & 2) != 0) & ((info & 1) == 0)
    k, valid, episodes, flags = counters.tolist()
    assert (k, valid, episodes, flags) == (int(mask.sum()), len(idx), led["count"], 0) and k > 100 and valid > k
    assert np.isclose(float(sums[0]), led["ep_return"].astype(np.float64).sum()) and float(sums[1]) == float((info >> 8).sum())
    tt = (tag[mask].astype(np.int64) - 1000)
    assert tt.min() >= 0 and tt.max() < T
    sn = snap.cpu().numpy()
    tn = np.clip((led["obs"][mask].astype(np.float64) - sn[tt, :20]) / np.sqrt(sn[tt, 20:40] + 1e-8), -10, 10).astype(np.float32)
    with torch.no_grad():
        v = pol.value(torch.from_numpy(tn).cuda()).cpu().numpy()
    ref = np.zeros((T, n), np.float32)
    ref[tt, idx[mask]] = gamma * v
    assert np.allclose(rew.cpu().numpy(), ref, atol=2e-5)
    eng.close()


def test_collector_turns_a_ledger_overflow_into_an_error():
    """One ledger row per 64-drone tile, every drone truncating inside the rollout: the scan raises flag 1 and the
    collector reports it when the status words are read."""
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    n = 1024
    eng = FpvEngine(n, 0, seed=0, ledger_capacity=n // 64)
    col = RolloutCollector(eng, ActorCritic().cuda(), n_steps=8, use_graph=False)
    col.start()
    st = eng.get_state()
    st[:, 34] = 399.0                                                   # step counters: all truncate within 8 steps
    eng.set_state(st)
    col.collect()
    with pytest.raises(RuntimeError, match="ledger overflow"):
        col.episodes
    # unread, the flag surfaces from collect() when the slot comes round again (two rollouts later) — at the END of that call, with
    # the collector's own bookkeeping advanced, so that the caller can go on (or stop) with a consistent object
    st[:, 34] = 399.0
    eng.set_state(st)
    col.collect()                                                       # slot 1: overflows again, nobody reads it
    col.collect()                                                       # slot 0: holds rollout 0's words, already read above
    r = col._rollouts
    with pytest.raises(RuntimeError, match="ledger overflow"):
        col.collect()                                                   # slot 1 comes round: rollout 1's flag
    assert col._rollouts == r + 1 and col.engine.get_tick() == 8 * (r + 1)
    eng.close()


@pytest.mark.parametrize("T", [7, 8])
@pytest.mark.both_libraries        # merge_in_step: a merge kernel behind the step in the product library, inside the step's launch in the other
@pytest.mark.parametrize("norm", [(True, True), (True, False), (False, True)])
def test_in_launch_statistics_merge_equals_the_merge_kernel(T, norm):
    """fpv_policy_io.merge_* (statistics folded into the next policy launch, ping-pong buffers: odd and even rollout
    lengths end in different buffers) against the stand-alone fpv_moments_merge launches, two rollouts each."""
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    n = 2048
    outs = []
    for mip in (True, False, "step"):
        torch.manual_seed(1)
        pol = ActorCritic(log_std_init=-1.0).cuda()
        eng = FpvEngine(n, 0, seed=5, ledger_capacity=4 * n)
        col = RolloutCollector(eng, pol, n_steps=T, norm_obs=norm[0], norm_reward=norm[1], merge_in_policy=mip is True,
                               merge_in_step=mip == "step")
        assert col._merge_in_policy == (mip is True) and col._merge_in_step == (mip == "step")
        col.collect()
        b = col.collect()
        torch.cuda.synchronize()
        outs.append({"obs": b["obs"].clone(), "rew": col.buf_rew.clone(), "adv": b["advantages"].clone(),
                     "obs_rms": col.obs_norm.rms.clone(), "ret_rms": col.rew_norm.rms.clone(), "snap": col.snap.clone(),
                     "episodes": col.episodes["count"]})
        eng.close()
    a, b, c = outs                     # in the policy prologue / merge kernel / inside fpv_step_fused (the default)
    for x in (b, c):
        assert a["episodes"] == x["episodes"] and a["episodes"] > 0
        for k in ("obs_rms", "ret_rms", "snap"):
            assert torch.allclose(a[k], x[k], rtol=1e-6, atol=1e-9), k
        for k in ("obs", "rew", "adv"):
            assert torch.allclose(a[k], x[k], rtol=1e-3, atol=2e-3), k


@pytest.mark.both_libraries
@pytest.mark.parametrize("n", [100, 4160, 70000])
def test_collector_at_ragged_sizes(n):
    """Drone counts that exercise the other dispatch branches of the rollout step: 100 (wave-pair kernel, one partial
    row per tile, ragged last tile), 4 160 = 65 tiles (not a multiple of four: one tile per workgroup), 70 000 = 1 094
    tiles (more than the in-launch merge takes: stand-alone merge kernel).  Both statistics paths must agree."""
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    outs = []
    for mip in (True, False, "step"):
        torch.manual_seed(4)
        pol = ActorCritic(log_std_init=-1.0).cuda()
        eng = FpvEngine(n, 0, seed=2, ledger_capacity=8 * n)
        col = RolloutCollector(eng, pol, n_steps=6, merge_in_policy=mip is True, merge_in_step=mip == "step")
        assert col._merge_in_policy == (mip is True and n <= 65536)
        col.collect()
        b = col.collect()
        torch.cuda.synchronize()
        assert torch.isfinite(b["advantages"]).all() and torch.isfinite(b["adv_std"])
        outs.append({"obs_rms": col.obs_norm.rms.clone(), "ret_rms": col.rew_norm.rms.clone(), "rew": col.buf_rew.clone(),
                     "episodes": col.episodes["count"], "rows": col.partials.shape[0]})
        eng.close()
    a, b, c = outs
    tiles = (n + 63) // 64
    assert a["rows"] == (tiles if (n <= 32768 or tiles % 4) else tiles // 4)
    for x in (b, c):                   # c: inside fpv_step_fused — at these sizes through the merge kernel behind the step
        assert a["episodes"] == x["episodes"]
        assert torch.allclose(a["obs_rms"], x["obs_rms"], rtol=1e-6, atol=1e-9) and torch.allclose(a["ret_rms"], x["ret_rms"], rtol=1e-6, atol=1e-9)
        assert torch.allclose(a["rew"], x["rew"], rtol=1e-3, atol=2e-3)

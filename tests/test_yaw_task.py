"""Yaw task (BASELINE cfg 5, simulation/yaw.py skeleton + docs/SPEC.md §10): oracle semantics on CPU,
HIP parity on GPU.  Reward, detector and randomisation ranges are [BUILD]-defined."""
import math

import numpy as np
import pytest


def test_yaw_oracle_semantics(oracle):
    cfg = oracle.make_cfg(noise_ratio=0.0, seed=3)
    st, obs = oracle.yaw_reset_states(cfg, 64, tick=0)
    psi = obs[:, 2] * math.pi
    assert (np.abs(obs[:, 2]) <= 1).all() and len(np.unique(np.round(psi, 6))) == 64   # random headings (yaw.py:79)
    assert np.allclose(st[:, 5], np.sin(psi / 2)) and np.allclose(st[:, 25], psi)
    assert np.allclose(st[:, 40:43], (0.1, 1.5, 1e-4)) and (obs[:, 5:] == 0).all()
    # action history: last four actions, oldest first (yaw.py:53-55, :69, :107)
    for t, a in enumerate([0.1, 0.2, 0.3, 0.4, 0.5]):
        o, r, te, tr, _ = oracle.yaw_step_batch(cfg, st, np.full(64, a), t)
    assert np.allclose(o[:, 8:12], (0.2, 0.3, 0.4, 0.5))
    assert (np.abs(o[:, 5:8]) <= 1).all()
    # truncation uses >= (yaw.py:138): the 401st call of an episode truncates
    st, _ = oracle.yaw_reset_states(cfg, 4, tick=0)
    for call in range(1, 405):
        o, r, te, tr, led = oracle.yaw_step_batch(cfg, st, np.zeros(4), call)
        if tr.any():
            break
    assert call == 401 and tr.all() and not te.any() and ((led.rows()[1] >> 8) == 401).all() and (st[:, 34] == 0).all()


def test_yaw_domain_randomisation_ranges(oracle):
    cfg = oracle.make_cfg(noise_ratio=0.0, seed=9, domain_rand=True)
    st, _ = oracle.yaw_reset_states(cfg, 4000, tick=7)
    m, cd, kp = st[:, 40], st[:, 41], st[:, 42]
    assert 0.08 <= m.min() < 0.081 and 0.119 < m.max() <= 0.12          # SURVEY §8d ranges
    assert 1.0 <= cd.min() < 1.01 and 1.99 < cd.max() <= 2.0
    assert 0.5e-4 <= kp.min() < 0.51e-4 and 1.99e-4 < kp.max() <= 2e-4
    assert abs(np.corrcoef(m, cd)[0, 1]) < 0.05
    # a heavier airframe spins up more slowly under the same command? no: mass does not enter the yaw axis, drag does
    st2 = st.copy()
    o, *_ = oracle.yaw_step_batch(cfg, st2, np.full(4000, 1.0), 0)
    assert (np.abs(st2[:, 12]) > 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("noise,dr", [(0.0, False), (0.02, True)])
def test_yaw_fp32_single_step_matches_oracle(oracle, noise, dr):
    import torch
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    n = 4096
    cfg = oracle.make_cfg(noise_ratio=noise, seed=21, id_offset=100, domain_rand=dr)
    eng = FpvEngine(n, 0, task="yaw", noise_ratio=noise, seed=21, id_offset=100, domain_rand=dr)
    assert (eng.state_words, eng.obs_words, eng.act_words) == (44, 12, 1)
    st_ref, obs_ref = oracle.yaw_reset_states(cfg, n, tick=0)
    obs0 = eng.reset()
    assert np.allclose(obs0.cpu().numpy(), obs_ref, atol=1e-5)
    assert np.allclose(eng.get_state().cpu().numpy(), st_ref, rtol=1e-5, atol=1e-6)
    # roll both forward a few steps, then compare ONE step from identical (fp32-rounded) states
    rng = np.random.default_rng(0)
    for t in range(6):
        a = rng.uniform(-1, 1, (n, 1)).astype(np.float32)
        eng.step(torch.from_numpy(a).cuda())
    st0 = eng.get_state().cpu().numpy()
    st0[: n // 16, 34] = 400                                           # these truncate on the next step
    st0[n // 16: n // 8, 0] = 2.99999                                  # these leave the dome
    eng.set_state(torch.from_numpy(st0))
    ref = st0.astype(np.float64)
    a = rng.uniform(-1, 1, (n, 1)).astype(np.float32)
    o_r, r_r, te_r, tr_r, led = oracle.yaw_step_batch(cfg, ref, a.astype(np.float64), 6)
    eng.ledger_clear()
    o, r, te, tr = eng.step(torch.from_numpy(a).cuda())
    assert (te.cpu().numpy() == te_r).all() and (tr.cpu().numpy() == tr_r).all() and tr_r.sum() == n // 16 and te_r.any()
    st = eng.get_state().cpu().numpy()
    assert (np.abs(st - ref) <= 1e-5 * np.maximum(1, np.abs(ref))).all()
    assert np.allclose(o.cpu().numpy(), o_r, atol=2e-5) and np.allclose(r.cpu().numpy(), r_r, atol=2e-5)
    got = eng.ledger_read()
    assert (got["idx"] == led.rows()[0]).all() and (got["info"] == led.rows()[1]).all()
    assert np.allclose(got["obs"], led.rows()[3], atol=2e-5)
    eng.close()


@pytest.mark.gpu
def test_yaw_fp64_trajectory_with_resets_matches_oracle(oracle):
    import torch
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    n, steps = 200, 450            # spans the 401-call truncation and the re-randomised restarts after it
    cfg = oracle.make_cfg(noise_ratio=0.02, seed=5, domain_rand=True)
    eng = FpvEngine(n, 0, task="yaw", noise_ratio=0.02, seed=5, domain_rand=True, precision="f64", ledger_capacity=8 * n)
    st_ref, obs_ref = oracle.yaw_reset_states(cfg, n, tick=0)
    assert np.allclose(eng.reset().cpu().numpy(), obs_ref, atol=1e-12)
    rng = np.random.default_rng(1)
    ends = 0
    for t in range(steps):
        a = rng.uniform(-0.3, 0.3, (n, 1))
        o_r, r_r, te_r, tr_r, _ = oracle.yaw_step_batch(cfg, st_ref, a, t)
        o, r, te, tr = eng.step(torch.from_numpy(a).cuda())
        assert (te.cpu().numpy() == te_r).all() and (tr.cpu().numpy() == tr_r).all(), t
        assert np.allclose(o.cpu().numpy(), o_r, atol=1e-8) and np.allclose(r.cpu().numpy(), r_r, atol=1e-8), t
        ends += int(te_r.sum() + tr_r.sum())
    assert ends >= n
    assert np.allclose(eng.get_state().cpu().numpy(), st_ref, atol=1e-8)
    eng.close()


@pytest.mark.gpu
def test_yaw_single_env_surface():
    from fpv_drone_rl_agent_amd.yaw import DroneEnv
    env = DroneEnv(noise_ratio=0.0)
    assert env.action_space.shape == (1,) and env.observation_space.shape == (12,)       # yaw.py:37-45
    obs, info = env.reset()
    assert obs.shape == (12,) and obs.dtype == np.float32 and info == {}
    o, r, term, trunc, info = env.step(np.array([0.5], np.float32))
    assert o.shape == (12,) and isinstance(r, float) and o[11] == 0.5 and not term and not trunc
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gemm", ["f32", "bf16x6"])
def test_yaw_fused_policy_matches_torch_and_collects(gemm):
    import copy
    import torch
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.ops_policy import FusedPolicy
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    torch.manual_seed(0)
    m = ActorCritic(obs_dim=12, act_dim=1).cuda()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.3)
    fp = FusedPolicy(m, gemm=gemm)
    n = 5000
    obs = torch.randn(n, 12, device="cuda")
    eps = torch.randn(n, 1, device="cuda")
    rms = torch.cat([torch.randn(12), torch.rand(12) + 0.5, torch.ones(1)]).double().cuda()
    out = dict(obs_n=torch.zeros(n, 12, device="cuda"), action=torch.zeros(n, 1, device="cuda"),
               clipped=torch.zeros(n, 1, device="cuda"), value=torch.zeros(n, device="cuda"), logp=torch.zeros(n, device="cuda"))
    fp.forward(obs, rms=rms, eps_in=eps, **out)
    with torch.no_grad():
        ref_obs = torch.clamp((obs.double() - rms[:12]) / torch.sqrt(rms[12:24] + 1e-8), -10, 10).float()
        a, v, lp = m.act(ref_obs, eps)
    assert torch.allclose(out["obs_n"], ref_obs, atol=1e-5) and torch.allclose(out["action"], a, atol=2e-4, rtol=1e-4)
    assert torch.allclose(out["value"], v, atol=2e-4, rtol=1e-4) and torch.allclose(out["logp"], lp, atol=1e-4)
    assert torch.equal(out["clipped"], out["action"].clamp(-1, 1))
    # every row against an fp64 evaluation of the module: fp32 rounding level on either arithmetic (the 6-term bf16 split at the
    # yaw widths is the same chain as at the hover widths, with the observation's K padded from 12 and one head column in use)
    m64 = copy.deepcopy(m).double()
    big = torch.randn(65536, 12, device="cuda") * 2
    with torch.no_grad():
        mu64, v64 = m64.mean_value(big.double())
    a1 = torch.zeros(65536, 1, device="cuda")
    r = fp.forward(big, action=a1, deterministic=True)
    assert float((a1.double() - mu64).abs().max()) <= 4e-6 * max(1.0, float(mu64.abs().max()))
    assert float((r["value"].double() - v64).abs().max()) <= 1e-5
    # nets = "pi" / "v" against the joint launch, bit for bit
    o2 = {k: torch.zeros_like(t) for k, t in out.items() if k in ("obs_n", "action", "clipped")}
    lp2 = torch.zeros(n, device="cuda")
    fp.forward(obs, rms=rms, eps_in=eps, logp=lp2, nets="pi", **o2)
    v2 = fp.forward(o2["obs_n"], nets="v", deterministic=True)["value"]
    assert all(torch.equal(o2[k], out[k]) for k in o2) and torch.equal(lp2, out["logp"]) and torch.equal(v2, out["value"])
    # the collector picks the fused policy for the yaw task (one launch per step) and stays finite
    eng = FpvEngine(2048, 0, task="yaw", domain_rand=True, ledger_capacity=8192)
    col = RolloutCollector(eng, m, n_steps=16, policy_gemm=gemm)
    assert col.fused_policy and col.fused and col._one_launch and col.policy_gemm == gemm
    b = col.collect()
    assert torch.isfinite(b["advantages"]).all() and torch.isfinite(b["returns"]).all() and float(b["adv_std"]) > 0
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T", [5, 8])
def test_yaw_fused_step_equals_the_stand_alone_normaliser_path(T):
    """cfg 5 through the two-kernel rollout step (VecNormalize bookkeeping inside yaw_step_kernel, 28-word partial rows,
    statistics merge in the policy launch) against the stand-alone env-step + normaliser kernels, two rollouts each."""
    import torch
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    n = 4096
    outs = []
    for fused_step in (True, False):
        torch.manual_seed(1)
        pol = ActorCritic(12, 1, log_std_init=-1.0).cuda()
        eng = FpvEngine(n, 0, seed=5, task="yaw", domain_rand=True, ledger_capacity=4 * n)
        col = RolloutCollector(eng, pol, n_steps=T, fused_step=fused_step)
        assert col.fused == fused_step
        col.start()
        st = eng.get_state()
        st[:, 34] = torch.randint(380, 400, (n,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)).to(st.dtype)
        eng.set_state(st)                       # step counters: truncations (step >= 400) inside the window
        col.collect()
        b = col.collect()
        torch.cuda.synchronize()
        outs.append({"obs": b["obs"].clone(), "rew": col.buf_rew.clone(), "adv": b["advantages"].clone(),
                     "obs_rms": col.obs_norm.rms.clone(), "ret_rms": col.rew_norm.rms.clone(),
                     "starts": col.buf_start.clone(), "episodes": dict(col.episodes)})
        eng.close()
    a, b = outs
    assert a["episodes"]["count"] == b["episodes"]["count"] > 0 and a["episodes"]["truncated"] == b["episodes"]["truncated"] > 0
    assert torch.equal(a["starts"], b["starts"])
    for k in ("obs_rms", "ret_rms"):
        assert torch.allclose(a[k], b[k], rtol=1e-6, atol=1e-9), k
    for k in ("obs", "rew", "adv"):
        assert torch.allclose(a[k], b[k], rtol=1e-3, atol=2e-3), k

"""GPU parity of the rollout primitives (GAE, VecNormalize moments) against the oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,n", [(1, 1), (5, 7), (32, 4096), (16, 70001)])
def test_gae_matches_oracle(oracle, T, n):
    from fpv_drone_rl_agent_amd import ops
    rng = np.random.default_rng(T * 1000 + n)
    r = rng.normal(size=(T, n)).astype(np.float32)
    v = rng.normal(size=(T, n)).astype(np.float32)
    es = (rng.random((T, n)) < 0.1).astype(np.uint8)
    lv = rng.normal(size=n).astype(np.float32)
    ld = (rng.random(n) < 0.1).astype(np.uint8)
    adv_r, ret_r = oracle.gae(r, v, es, lv, ld, 0.99, 0.95)
    stats = torch.zeros(3, dtype=torch.float64, device="cuda")
    adv, ret = ops.gae(*(torch.from_numpy(x).cuda() for x in (r, v, es, lv, ld)), 0.99, 0.95, stats=stats)
    assert np.allclose(adv.cpu().numpy(), adv_r, atol=2e-5, rtol=1e-5)
    assert np.allclose(ret.cpu().numpy(), ret_r, atol=2e-5, rtol=1e-5)
    s = stats.cpu().numpy()
    a64 = adv.cpu().numpy().astype(np.float64)
    assert s[0] == T * n and np.isclose(s[1], a64.sum(), rtol=1e-12, atol=1e-9)
    assert np.isclose(s[2], (a64 * a64).sum(), rtol=1e-12)


def test_gae_is_bitwise_reproducible():
    from fpv_drone_rl_agent_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    r = torch.randn(8, 50000, device="cuda", generator=g)
    v = torch.randn(8, 50000, device="cuda", generator=g)
    es = torch.zeros(8, 50000, dtype=torch.uint8, device="cuda")
    lv = torch.randn(50000, device="cuda", generator=g)
    ld = torch.zeros(50000, dtype=torch.uint8, device="cuda")
    s1 = torch.zeros(3, dtype=torch.float64, device="cuda")
    s2 = torch.zeros(3, dtype=torch.float64, device="cuda")
    a1, _ = ops.gae(r, v, es, lv, ld, stats=s1)
    a2, _ = ops.gae(r, v, es, lv, ld, stats=s2)
    assert torch.equal(a1, a2) and torch.equal(s1, s2)


@pytest.mark.parametrize("n,d", [(1, 20), (333, 20), (65536, 20), (1000, 12)])
def test_running_norm_matches_vecnormalize(oracle, n, d):
    from fpv_drone_rl_agent_amd import ops
    rng = np.random.default_rng(n + d)
    ref = oracle.RunningMeanStd((d,))
    rn = ops.RunningNorm(d, "cuda")
    for it in range(3):
        x = (rng.normal(size=(n, d)) * rng.uniform(0.1, 30, d) + rng.uniform(-5, 5, d)).astype(np.float32)
        ref.update(x)
        y_ref = np.clip((x - ref.mean) / np.sqrt(ref.var + 1e-8), -10, 10)
        y = rn(torch.from_numpy(x).cuda())
        assert np.allclose(rn.mean.cpu().numpy(), ref.mean, rtol=1e-10, atol=1e-10)
        assert np.allclose(rn.var.cpu().numpy(), ref.var, rtol=1e-9, atol=1e-12)
        assert np.isclose(float(rn.count), ref.count)
        assert np.allclose(y.cpu().numpy(), y_ref, atol=2e-5, rtol=1e-5)
    # frozen statistics (test_hover.py:9 training=False)
    x = rng.normal(size=(n, d)).astype(np.float32)
    before = rn.rms.clone()
    y = rn(torch.from_numpy(x).cuda(), update=False)
    assert torch.equal(before, rn.rms)
    assert np.allclose(y.cpu().numpy(), np.clip((x - ref.mean) / np.sqrt(ref.var + 1e-8), -10, 10), atol=2e-5)


def test_reward_norm_matches_vecnormalize(oracle):
    from fpv_drone_rl_agent_amd import ops
    n = 5000
    rng = np.random.default_rng(5)
    ref = oracle.VecNormalizeRef(n)
    rw = ops.RewardNorm(n, "cuda")
    for it in range(6):
        r = rng.normal(size=n).astype(np.float32) * 3
        done = (rng.random(n) < 0.2)
        _, r_ref = ref.step(np.zeros((n, 20)), r.astype(np.float64), done)
        out = rw(torch.from_numpy(r).cuda(), torch.from_numpy(done.astype(np.uint8)).cuda())
        assert np.allclose(out.cpu().numpy(), r_ref, atol=2e-5, rtol=1e-5)
        assert np.allclose(rw.ret.cpu().numpy(), ref.returns, atol=1e-4, rtol=1e-5)
        assert np.isclose(float(rw.rms[1]), ref.ret_rms.var, rtol=1e-6)


@pytest.mark.parametrize("n", [64, 1000, 65536, 200000])
def test_step_fused_bookkeeping_matches_unfused(oracle, n):
    """fpv_step_fused + fpv_moments_merge + fpv_reward_finalize == fpv_step + fpv_normalize(+_reward)."""
    from fpv_drone_rl_agent_amd import ops
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.synthetic import synthetic_states
    st = synthetic_states(n, seed=1).astype(np.float32)
    st[: n // 8, 34] = 401
    a_eng, b_eng = FpvEngine(n, 0, seed=4), FpvEngine(n, 0, seed=4)
    for e in (a_eng, b_eng):
        e.set_state(torch.from_numpy(st))
    obs_n_a, rn_a = ops.RunningNorm(20, "cuda"), ops.RewardNorm(n, "cuda")
    obs_rms_b = obs_n_a.rms.clone()
    ret_rms_b, ret_b = rn_a.rms.clone(), torch.zeros(n, device="cuda")
    partials = torch.zeros((a_eng.tiles, 44), dtype=torch.float64, device="cuda")
    T = 3
    snap = torch.zeros((T, 44), dtype=torch.float64, device="cuda")
    raw = torch.zeros((T, n), device="cuda")
    done_b = torch.zeros(n, dtype=torch.uint8, device="cuda")
    rew_a = []
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in range(T):
        act = torch.rand(n, 4, device="cuda", generator=g) * 2 - 1
        o, r, te, tr = a_eng.step(act)
        done_a = te | tr
        obs_n_a.update_only(o)
        rew_a.append(rn_a(r, done_a).clone())
        b_eng.step_fused(act, done=done_b, ret=ret_b, partials=partials, gamma=0.99, reward=raw[t])
        ops.moments_merge(partials, n, obs_rms=obs_rms_b, ret_rms=ret_rms_b, snapshot=snap[t])
        assert torch.equal(b_eng.obs, o) and torch.equal(raw[t], r) and torch.equal(done_b, done_a)
        assert torch.allclose(ret_b, rn_a.ret, atol=1e-5, rtol=1e-6)
        assert torch.allclose(obs_rms_b, obs_n_a.rms, rtol=1e-9, atol=1e-9)
        assert torch.allclose(ret_rms_b, rn_a.rms, rtol=1e-6, atol=1e-9)
        assert torch.equal(snap[t, :41], obs_rms_b) and torch.equal(snap[t, 41:], ret_rms_b)
    out = ops.reward_finalize(raw, snap)
    assert torch.allclose(out, torch.stack(rew_a), atol=1e-5, rtol=1e-5)
    a_eng.close(); b_eng.close()


@pytest.mark.parametrize("task,n", [("hover", 65536), ("hover", 262144), ("hover", 4096), ("hover", 1000), ("yaw", 65536), ("yaw", 1000)])
def test_statistics_update_inside_step_fused_matches_the_merge_kernel(task, n):
    """fpv_step_extras.obs_rms / ret_rms / snapshot: RunningMeanStd.update inside the env-step call — by the launch's last
    workgroup (write-through rows + ticket) at 65 536 / 262 144 drones, by the merge kernel behind the step at the
    other shapes — against fpv_step_fused + fpv_moments_merge as two calls.  Many consecutive launches, every one
    compared: a stale row or a lost ticket would show as a wrong or missing update."""
    from fpv_drone_rl_agent_amd import ops
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    yaw = task == "yaw"
    O, A = (12, 1) if yaw else (20, 4)
    engs = [FpvEngine(n, 0, seed=3, task=task, domain_rand=yaw, ledger_capacity=8 * n) for _ in range(2)]
    g = torch.Generator(device="cuda").manual_seed(1)
    stats = []
    for e in engs:
        stats.append({"done": torch.zeros(n, dtype=torch.uint8, device="cuda"), "ret": torch.zeros(n, device="cuda"),
                      "part": torch.zeros((e.partial_rows, 2 * O + 4), dtype=torch.float64, device="cuda"),
                      "raw": torch.zeros(n, device="cuda"),
                      "obs_rms": torch.cat([torch.zeros(O), torch.ones(O), torch.tensor([1e-4])]).double().cuda(),
                      "ret_rms": torch.tensor([0.0, 1.0, 1e-4], dtype=torch.float64, device="cuda"),
                      "snap": torch.zeros(44, dtype=torch.float64, device="cuda")})
    for it in range(40):
        act = torch.rand(n, A, device="cuda", generator=g) * 2 - 1
        a, b = stats
        engs[0].step_fused(act, done=a["done"], ret=a["ret"], partials=a["part"], gamma=0.99, reward=a["raw"],
                           obs_rms=a["obs_rms"], ret_rms=a["ret_rms"], snapshot=a["snap"])
        engs[1].step_fused(act, done=b["done"], ret=b["ret"], partials=b["part"], gamma=0.99, reward=b["raw"])
        ops.moments_merge(b["part"], n, obs_rms=b["obs_rms"], ret_rms=b["ret_rms"], snapshot=b["snap"])
        for k, tol in (("obs_rms", 1e-11), ("ret_rms", 1e-11)):
            assert torch.allclose(a[k], b[k], rtol=tol, atol=1e-13), (it, k, (a[k] - b[k]).abs().max())
        assert torch.allclose(a["snap"][:2 * O + 1], b["snap"][:2 * O + 1], rtol=1e-11, atol=1e-13)
        assert torch.allclose(a["snap"][41:44], b["snap"][41:44], rtol=1e-11, atol=1e-13)
        assert torch.equal(a["part"], b["part"]) and torch.equal(engs[0].obs, engs[1].obs)
    assert float(stats[0]["obs_rms"][2 * O]) == pytest.approx(1e-4 + 40 * n)
    for e in engs:
        e.close()


def test_rank_merge_kernel_matches_the_host_statement(oracle):
    """fpv_rank_merge (several ranks: VecNormalize's statistics over ALL ranks' drones from the one all-gather's payload) against
    dist.merge_rank_statistics, whose equality with a single process over the union is a CPU test (tests/test_dist_gloo.py)."""
    from fpv_drone_rl_agent_amd import dist as D
    from fpv_drone_rl_agent_amd import ops
    rng = np.random.default_rng(3)
    for d in (20, 12):
        G = 5
        base_o = oracle.RunningMeanStd((d,)); base_o.update(rng.normal(1.0, 3.0, (4000, d)))
        base_r = oracle.RunningMeanStd(()); base_r.update(rng.normal(-2.0, 5.0, 4000))
        b_o = np.concatenate([base_o.mean, base_o.var, [base_o.count]])
        b_r = np.array([base_r.mean, base_r.var, base_r.count])
        rows = []
        for r in range(G):
            o = oracle.RunningMeanStd((d,)); o.mean, o.var, o.count = base_o.mean.copy(), base_o.var.copy(), base_o.count
            q = oracle.RunningMeanStd(()); q.mean, q.var, q.count = base_r.mean, base_r.var, base_r.count
            if r != 2:                                             # rank 2 adds nothing
                for t in range(4):
                    o.update(rng.normal(r, 1 + r, (500, d))); q.update(rng.normal(r, 2.0, 500))
            rows.append(np.concatenate([[1.0, 2.0, 3.0], o.mean, o.var, [o.count], [q.mean, q.var, q.count]]))
        g = torch.from_numpy(np.stack(rows)).cuda()
        obs_rms, ret_rms = torch.from_numpy(b_o).cuda(), torch.from_numpy(b_r).cuda()
        ops.rank_merge(g, d, base_obs_rms=obs_rms.clone(), base_ret_rms=ret_rms.clone(), obs_rms=obs_rms, ret_rms=ret_rms)
        ref_o = D.merge_rank_statistics(torch.from_numpy(b_o), torch.from_numpy(np.stack(rows))[:, 3:3 + 2 * d + 1])
        ref_r = D.merge_rank_statistics(torch.from_numpy(b_r), torch.from_numpy(np.stack(rows))[:, 3 + 2 * d + 1:])
        assert torch.allclose(obs_rms.cpu(), ref_o, rtol=1e-13, atol=1e-13) and torch.allclose(ret_rms.cpu(), ref_r, rtol=1e-13, atol=1e-13)
        assert float(obs_rms[2 * d]) == base_o.count + 4 * 4 * 500
        # in place: bases that ARE the outputs (allowed: the kernel's four pointers are not restrict-qualified)
        o2, r2 = torch.from_numpy(b_o).cuda(), torch.from_numpy(b_r).cuda()
        ops.rank_merge(g, d, base_obs_rms=o2, base_ret_rms=r2, obs_rms=o2, ret_rms=r2)
        assert torch.equal(o2, obs_rms) and torch.equal(r2, ret_rms)
        # one set of statistics only — the shapes RolloutCollector.start() uses (observation statistics of the reset observations,
        # no return statistics yet: base_ret = null, the form that faulted at address 0 once in round 3) and its mirror
        o3, r3 = torch.from_numpy(b_o).cuda(), torch.full((3,), -7.0, dtype=torch.float64, device="cuda")
        ops.rank_merge(g, d, base_obs_rms=torch.from_numpy(b_o).cuda(), base_ret_rms=None, obs_rms=o3, ret_rms=None)
        assert torch.equal(o3, obs_rms) and (r3 == -7.0).all()
        o4, r4 = torch.full((2 * d + 1,), -7.0, dtype=torch.float64, device="cuda"), torch.from_numpy(b_r).cuda()
        ops.rank_merge(g, d, base_obs_rms=None, base_ret_rms=torch.from_numpy(b_r).cuda(), obs_rms=None, ret_rms=r4)
        assert torch.equal(r4, ret_rms) and (o4 == -7.0).all()
        # an output without its base is an argument error, not a read through a null pointer
        with pytest.raises(ValueError):
            ops.rank_merge(g, d, base_obs_rms=None, base_ret_rms=torch.from_numpy(b_r).cuda(), obs_rms=o4, ret_rms=r4.clone())

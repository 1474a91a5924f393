"""Size-independent properties at BASELINE.json's full single-GPU sizes (65 536 and 524 288 drones), where the
scalar oracle is too slow to replay every drone: determinism, shard independence, episode accounting,
state invariants, reset idempotence — plus a seeded sample of drones replayed through the oracle."""
import numpy as np
import pytest

from tests import parity

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _run(n, steps, seed, id_offset=0, state=None, actions=None):
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.synthetic import synthetic_states
    eng = FpvEngine(n, 0, seed=seed, id_offset=id_offset, ledger_capacity=4 * n)
    st0 = synthetic_states(n, seed=1).astype(np.float32) if state is None else state
    eng.set_state(torch.from_numpy(st0))
    g = torch.Generator(device="cuda").manual_seed(5)
    acts = actions if actions is not None else [torch.rand(n, 4, device="cuda", generator=g) * 2 - 1 for _ in range(steps)]
    dones = 0
    rew_sum = torch.zeros((), dtype=torch.float64, device="cuda")
    for a in acts:
        o, r, te, tr = eng.step(a)
        dones += int((te | tr).sum())
        rew_sum += r.double().sum()
    return eng, st0, acts, dones, float(rew_sum)


@pytest.mark.parametrize("n", [65536, 524288])
def test_fullsize_invariants_and_determinism(n):
    steps = 40
    eng, st0, acts, dones, rsum = _run(n, steps, seed=3)
    st = eng.get_state()
    # state invariants
    assert torch.isfinite(st).all() and torch.isfinite(eng.obs).all()
    assert torch.allclose(st[:, 3:7].norm(dim=1), torch.ones(n, device="cuda"), atol=1e-5)      # unit quaternions
    assert (st[:, 2] >= 0.01 - 1e-7).all()                                                       # nobody under the floor
    assert (st[:, 34] >= 0).all() and (st[:, 34] <= 402).all()                                   # hover.py:275: at most 402 calls
    assert (st[:, 7:10].abs() <= 50).all() and (st[:, 10:13].abs() <= 100).all()                 # SPEC state clamp
    assert (st[:, 13:17] > 0).all() and (st[:, 13:17] < 1.2).all()
    # episode accounting: every done flag has exactly one ledger row; lengths are legal (32..402 from reset)
    led = eng.ledger_read()
    assert led["count"] == dones == len(led["idx"]) and dones > n // 50
    ep_len = led["info"] >> 8
    assert ep_len.min() >= 1 and ep_len.max() <= 402
    trunc_rows = (led["info"] & 2) != 0
    assert (ep_len[trunc_rows] == 402).all()
    floor_rows = (led["info"] & 8) != 0
    assert (ep_len[floor_rows] >= 32).all()                      # floor rule cannot fire before call 32 (hover.py:283)
    assert ((led["info"] & 1) != 0).sum() + (trunc_rows & ((led["info"] & 1) == 0)).sum() == dones
    # determinism: an identical second run is bit-identical (state, ledger, reward checksum)
    eng2, _, _, dones2, rsum2 = _run(n, steps, seed=3, state=st0, actions=acts)
    assert dones2 == dones and rsum2 == rsum and torch.equal(eng2.get_state(), st)
    led2 = eng2.ledger_read()
    assert np.array_equal(led2["idx"], led["idx"]) and np.array_equal(led2["obs"], led["obs"])
    eng.close(); eng2.close()


def test_fullsize_shard_independence_and_oracle_sample(oracle):
    """65 536 drones as one handle == 8 shards of 8 192 with id offsets; a seeded sample of 256
    drones is replayed through the scalar oracle for one step from the same states."""
    n, shards = 65536, 8
    eng, st0, acts, _, _ = _run(n, 3, seed=11)
    whole = eng.get_state().cpu().numpy()
    per = n // shards
    for r in (0, 3, 7):
        part, *_ = _run(per, 3, seed=11, id_offset=r * per, state=st0[r * per:(r + 1) * per],
                        actions=[a[r * per:(r + 1) * per].contiguous() for a in acts])
        assert np.array_equal(part.get_state().cpu().numpy(), whole[r * per:(r + 1) * per])
        part.close()
    # oracle replay of a sample: one step from st0 with acts[0]
    pick = np.random.default_rng(0).choice(n, 256, replace=False)
    cfg = oracle.make_cfg(noise_ratio=0.02, seed=11)
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    one = FpvEngine(n, 0, seed=11)
    one.set_state(torch.from_numpy(st0))
    one.step(acts[0])
    got = one.get_state().cpu().numpy()[pick]
    a0 = acts[0].cpu().numpy().astype(np.float64)
    flipped = 0
    for k, i in enumerate(pick):
        row = st0[i].astype(np.float64).copy()
        _, _, bits = oracle.step_one(cfg, row, a0[i], int(i), 0)
        if bits & 3:
            continue                                             # auto-reset rows are covered by the parity tests
        ok = np.abs(got[k] - row) <= 1e-5 * np.maximum(1, np.abs(row))
        if not ok.all():                                         # a one-pixel flip of the pixel detector (tests/parity.py):
            assert set(np.where(~ok)[0]) <= set(parity.FEATURE_STATE), i      # only the feature memory / ep_return may differ
            flipped += 1
    assert flipped <= 2
    eng.close(); one.close()


@pytest.mark.parametrize("launch", ["rollout_step", "step_fused"])
def test_benchmarked_kernel_matches_oracle_sample_including_auto_reset(oracle, launch):
    """The kernels bench.py times at 65 536 drones with motor noise — "rollout_step": fpv_rollout_step, the default rollout's ONE
    kernel per step (policy forward, then the env step of the same drones; the actions the env step consumed are read back
    through io->clipped); "step_fused": step_kernel<float, true, true, 4>, the env-step kernel of the two-kernel shape —
    against the scalar oracle on a seeded sample of 256 drones from the same states, auto-reset rows INCLUDED (returned
    observation = reset observation, state = reset state with prev_action kept, terminal observation and episode statistics
    in the ledger).  The finite-difference rates obs[0:3] are compared as angle differences (x agent_dt, hover.py:230) at the
    1e-5 bar.  For "rollout_step" the policy half is checked too: V and log pi(a) against the torch module on the same rows."""
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.synthetic import synthetic_states
    n, seed, gamma = 65536, 13, 0.99
    st0 = synthetic_states(n, seed=2).astype(np.float32)
    rng = np.random.default_rng(3)
    pick = rng.choice(n, 256, replace=False)
    # make a third of the sample end its episode in this step: past max_steps, outside the dome, on the floor line
    k = len(pick) // 9
    st0[pick[:k], 34] = 401                                        # hover.py:275 truncation
    st0[pick[k:2 * k], 0:3] = np.array([2.2, 2.2, 1.2], np.float32)          # |pos| > 3 (hover.py:278-281)
    st0[pick[2 * k:3 * k], 2] = 0.05; st0[pick[2 * k:3 * k], 34] = 100       # z < 0.1 after call 31 (hover.py:283-290)
    act = (torch.rand(n, 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4)) * 2 - 1)
    ret_in = torch.rand(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    eng = FpvEngine(n, 0, seed=seed, noise_ratio=0.02, ledger_capacity=2 * n)
    assert eng.partial_rows == n // 256                            # the four-tiles-per-workgroup variant is the one launched
    eng.set_state(torch.from_numpy(st0))
    eng.set_tick(9)
    done = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ret = ret_in.clone()
    partials = torch.zeros((eng.partial_rows, 44), dtype=torch.float64, device="cuda")
    raw = torch.zeros(n, device="cuda")
    if launch == "step_fused":
        eng.step_fused(act, done=done, ret=ret, partials=partials, gamma=gamma, reward=raw)
    else:
        from fpv_drone_rl_agent_amd.ops_policy import FusedPolicy
        from fpv_drone_rl_agent_amd.policy import ActorCritic
        torch.manual_seed(6)
        pol = ActorCritic().cuda()
        with torch.no_grad():
            pol.log_std.fill_(-0.3)
        fp = FusedPolicy(pol, seed=21)
        obs0 = torch.randn((n, 20), device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
        eng.obs.copy_(obs0)
        sampled, val, logp = torch.zeros(n, 4, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        fp.rollout_step(eng, rms=None, action=sampled, clipped=act, value=val, logp=logp, done=done, ret=ret, partials=partials,
                        gamma=gamma, reward=raw)                 # `act` now holds the clipped actions the env step consumed
        torch.cuda.synchronize()
        with torch.no_grad():
            mu_all, v_all = pol.mean_value(obs0)
            lp_ref = pol.log_prob(mu_all, sampled)
            z = (sampled - mu_all) * torch.exp(-pol.log_std)     # the draws: standard normal over 262 144 values
        assert torch.allclose(val, v_all, atol=2e-5, rtol=1e-5) and torch.allclose(logp, lp_ref, atol=3e-4, rtol=1e-5)
        assert torch.equal(act, sampled.clamp(-1.0, 1.0)) and (sampled.abs() > 1).any()
        assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    torch.cuda.synchronize()
    got_st, got_obs = eng.get_state().cpu().numpy(), eng.obs.cpu().numpy()
    got_rew, got_done, got_ret = raw.cpu().numpy(), done.cpu().numpy(), ret.cpu().numpy()
    led = eng.ledger_read()
    led_row = {int(i): r for r, i in enumerate(led["idx"])}
    cfg = oracle.make_cfg(noise_ratio=0.02, seed=seed)
    a_np, r_in = act.cpu().numpy().astype(np.float64), ret_in.cpu().numpy().astype(np.float64)
    tol = lambda a, b, t=1e-5: (np.abs(a - b) <= t * np.maximum(1.0, np.abs(b))).all()
    ended = flipped = 0
    feat, other = parity.FEATURE_OBS, [c for c in range(3, 20) if c not in parity.FEATURE_OBS]
    st_other = [c for c in range(36) if c not in parity.FEATURE_STATE]
    for i in pick:
        row = st0[i].astype(np.float64).copy()
        obs_r, rew_r, bits = oracle.step_one(cfg, row, a_np[i], int(i), 9)
        assert got_done[i] == (1 if bits & 3 else 0)
        # the observation this step produced: in the ledger for a finished episode, else the returned one
        seen = led["obs"][led_row[int(i)]].astype(np.float64) if bits & 3 else got_obs[i].astype(np.float64)
        # pixel detector (the default): a one-pixel rasterisation flip between fp32 and the fp64 oracle (tests/parity.py) is
        # allowed on at most two drones of the sample; everything that does not derive from the camera features still holds
        flip = bool((np.abs(seen[feat] - obs_r[feat]) > 1e-5).any())
        flipped += flip
        assert flip or tol(got_rew[i], rew_r, 2e-5), i
        r_acc = r_in[i] * gamma + rew_r
        assert tol(seen[other], obs_r[other]) and tol(seen[0:3] * 0.025, obs_r[0:3] * 0.025), i
        assert flip or tol(seen[feat], obs_r[feat]), i
        if bits & 3:
            ended += 1
            j = led_row[int(i)]
            assert led["info"][j] & 0xFF == bits and led["info"][j] >> 8 == int(row[34])
            assert flip or tol(led["ep_return"][j], row[35], 2e-5)
            keep = row[30:34].copy()                               # prev_action survives reset (hover.py:101 vs :357)
            rs, ro = oracle.reset_states(1)
            rs[0, 30:34] = keep
            assert tol(got_st[i], rs[0]) and tol(got_obs[i], ro[0]), i
            assert got_ret[i] == 0.0
        else:
            assert tol(got_st[i][st_other], row[st_other]), i
            assert flip or (tol(got_st[i], row) and tol(got_ret[i], r_acc, 2e-5)), i
    assert ended >= 3 * k - 2 and flipped <= 2
    eng.close()


def test_reset_is_idempotent_and_total():
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    eng = FpvEngine(65536, 0)
    a = eng.reset().clone()
    s1 = eng.get_state().clone()
    b = eng.reset()
    assert torch.equal(a, b) and torch.equal(s1, eng.get_state())
    assert (a == a[0]).all()                                      # SPEC §8: the reset state is a constant
    eng.close()


def test_fullsize_rollout_fused_vs_torch_reference_collector():
    """65 536 drones, one rollout with thousands of truncations and terminations inside it: the two-kernel HIP path
    (statistics merge in the policy launch, ledger scan / scatter for the timeout bootstrap, everything graph-captured)
    against the torch reference collector (stand-alone kernels + torch indexing) on the same policy actions.
    The reference collector draws its own action noise, so the comparison replays the fused collector's actions:
    rewards, episode starts and GAE must agree."""
    from fpv_drone_rl_agent_amd import ops
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    from fpv_drone_rl_agent_amd.synthetic import synthetic_states
    n, T = 65536, 16
    torch.manual_seed(2)
    pol = ActorCritic(log_std_init=-1.5).cuda()
    st = synthetic_states(n, seed=3).astype(np.float32)
    st[:, 34] = np.random.default_rng(1).integers(380, 402, n)          # every drone truncates within ~22 steps
    eng = FpvEngine(n, 0, seed=7, ledger_capacity=4 * n)
    col = RolloutCollector(eng, pol, n_steps=T)
    col.start()
    col._capture()
    eng.set_state(torch.from_numpy(st))
    obs_rms0, ret_rms0, ret0 = col.obs_norm.rms.clone(), col.rew_norm.rms.clone(), col.rew_norm.ret.clone()
    start0 = col.episode_start.clone()
    tick0 = eng.get_tick()
    b = col.collect()
    ep = col.episodes
    assert ep["truncated"] > 0.5 * n * T / 22 and ep["count"] >= ep["truncated"]
    acts = b["actions"].clone()

    # torch-side replay of the same actions through the plain step + stand-alone normaliser ops
    eng2 = FpvEngine(n, 0, seed=7, ledger_capacity=4 * n)
    eng2.set_state(torch.from_numpy(st)); eng2.set_tick(tick0)
    obs_norm = ops.RunningNorm(20, eng2.device); obs_norm.rms.copy_(obs_rms0)
    rew_norm = ops.RewardNorm(n, eng2.device, gamma=0.99); rew_norm.rms.copy_(ret_rms0); rew_norm.ret.copy_(ret0)
    rew = torch.zeros((T, n), device="cuda"); starts = torch.zeros((T + 1, n), dtype=torch.uint8, device="cuda")
    starts[0] = start0
    snaps = []
    eng2.ledger_clear()
    for t in range(T):
        eng2.step(torch.clamp(acts[t], -1, 1))
        torch.bitwise_or(eng2.terminated, eng2.truncated, out=starts[t + 1])
        obs_norm.update_only(eng2.obs)
        snaps.append(obs_norm.rms.clone())
        rew_norm(eng2.reward, starts[t + 1], out=rew[t])
    led = eng2.ledger_read()
    info, tag, idx = led["info"], led["tag"], led["idx"]
    mask = ((info & 2) != 0) & ((info & 1) == 0)
    tt = tag[mask].astype(np.int64) - tick0
    sn = torch.stack(snaps).cpu().numpy()
    tn = np.clip((led["obs"][mask].astype(np.float64) - sn[tt, :20]) / np.sqrt(sn[tt, 20:40] + 1e-8), -10, 10).astype(np.float32)
    with torch.no_grad():
        v = pol.value(torch.from_numpy(tn).cuda())
    rew.index_put_((torch.from_numpy(tt).cuda(), torch.from_numpy(idx[mask].astype(np.int64)).cuda()), 0.99 * v, accumulate=True)
    torch.cuda.synchronize()
    assert int(mask.sum()) == ep["truncated"] and led["count"] == ep["count"]
    assert torch.equal(col.buf_start, starts[:T])
    assert torch.allclose(col.buf_rew, rew, atol=2e-4, rtol=1e-4)
    adv, ret = ops.gae(rew, b["values"], starts[:T], col.last_val, starts[T], 0.99, 0.95)[:2]
    assert torch.allclose(b["advantages"], adv, atol=5e-4, rtol=1e-4)
    assert abs(float(b["adv_mean"]) - float(adv.double().mean())) < 1e-5
    eng.close(); eng2.close()

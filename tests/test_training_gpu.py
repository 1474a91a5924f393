"""System-level check: PPO on the on-device rollout learns to stay airborne.
The reference's shipped logs show ep_len_mean rising from ~32 (floor rule, hover.py:283) to the truncation
limit 402 (hover.py:275); a correct env + rollout + update must reproduce that structure."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_ppo_learns_to_hover():
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.ppo import PPO
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    torch.manual_seed(0)
    n = 8192
    eng = FpvEngine(n, 0, seed=0)
    pol = ActorCritic(log_std_init=-2.0).cuda()
    col = RolloutCollector(eng, pol, n_steps=32)
    model = PPO(col, n_epochs=4, n_minibatches=8, target_kl=0.03)
    model.learn(32 * n * 10)
    early = model.logger["rollout/ep_len_mean"]
    model.learn(32 * n * 150)
    late, rew = model.logger["rollout/ep_len_mean"], model.logger["rollout/ep_rew_mean"]
    assert early < 80, early                      # untrained: dies at the floor check (call 32) or earlier
    assert late > 300 and rew > 0, (late, rew)    # trained: survives to (near) the 402-step truncation
    eng.close()


def test_deferred_logging_gives_the_synchronising_loops_numbers(monkeypatch):
    """PPO.train_on with one rank and the fused update leaves the epoch statistics as asynchronous copies in pinned host words and
    resolves model.logger when it is read (no host wait behind the last epoch, target_kl decided from the pinned words): the same
    parameters, the same logged numbers and the same early stops as the loop that synchronises after every epoch
    (FPV_PPO_SYNC_LOG=1)."""
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.ppo import PPO
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    runs, counted = [], []
    for sync in ("1", "0"):
        monkeypatch.setenv("FPV_PPO_SYNC_LOG", sync)
        torch.manual_seed(0)
        n = 4096
        eng = FpvEngine(n, 0, seed=0)
        pol = ActorCritic(log_std_init=-2.0).cuda()
        model = PPO(RolloutCollector(eng, pol, n_steps=16), n_epochs=4, n_minibatches=4, target_kl=0.002)   # (tight: epochs are cut short)
        logs = []
        for it in range(6):
            model.learn(16 * n * (it + 1))
            logs.append(dict(model.logger.items()))
            if sync == "0":                     # (target_kl decided on the device: how many of the four queued epochs counted)
                assert model._device_gate
                counted.append(int(model._gate_host[1]))
        runs.append((logs, torch.cat([p.detach().flatten() for p in pol.parameters()]).clone()))
        eng.close()
    (la, pa), (lb, pb) = runs
    assert torch.equal(pa, pb)
    assert "rollout/ep_len_mean" in la[-1] and "train/std" in la[-1]
    assert min(counted) < 4 and len(set(counted)) > 1, counted     # updates were cut short, at different epochs
    for a, b in zip(la, lb):
        assert a.keys() == b.keys() and len(a) >= 6
        for k in a:
            assert a[k] == pytest.approx(b[k], rel=1e-6, abs=1e-9), k


def test_ppo_learns_the_yaw_task_under_domain_randomisation():
    """BASELINE cfg 5 end to end: simulation/yaw.py's yaw-setpoint task with per-episode mass / drag randomisation;
    episodes always run to the truncation limit, so learning shows in the return."""
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.ppo import PPO
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    torch.manual_seed(0)
    n = 8192
    eng = FpvEngine(n, 0, seed=0, task="yaw", domain_rand=True, ledger_capacity=n * 4)
    pol = ActorCritic(12, 1, log_std_init=-2.0).cuda()
    col = RolloutCollector(eng, pol, n_steps=32)
    model = PPO(col, n_epochs=4, n_minibatches=8, target_kl=0.03, log_std_min=-3.5)
    model.learn(32 * n * 30)
    early = model.logger["rollout/ep_rew_mean"]
    model.learn(32 * n * 400)
    late = model.logger["rollout/ep_rew_mean"]
    assert late > early + 50 and late > 60, (early, late)
    eng.close()


def test_checkpoint_cadence_roundtrip_and_playback(tmp_path):
    """train_hover.py:8-34,62-63 cadence and naming; test_hover.py:8-22 style playback of the saved pair."""
    import os
    from fpv_drone_rl_agent_amd.checkpoint import CheckpointCallback, load_normalizer, load_policy_state, save_model
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.ppo import PPO
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    from fpv_drone_rl_agent_amd.vec_env import FpvVecEnv, VecNormalize
    torch.manual_seed(1)
    n = 1024
    eng = FpvEngine(n, 0)
    pol = ActorCritic(log_std_init=-2.0).cuda()
    model = PPO(RolloutCollector(eng, pol, n_steps=8), n_epochs=1, n_minibatches=2)
    cb = CheckpointCallback(save_path=str(tmp_path / "ppo_hover_checkpoint3"), name_prefix="ppo_hover", verbose=0,
                            min_steps_between_checkpoints=20000)
    model.learn(8 * n * 7, callback=cb)            # 7 iterations of 8192 steps: checkpoints after >= 20000 steps
    assert len(cb.saved) == 2 and "_24576_steps_" in cb.saved[0] and "_49152_steps_" in cb.saved[1]
    assert all(os.path.exists(p) and os.path.exists(p.replace(".zip", ".pkl")) for p in cb.saved)
    save_model(str(tmp_path / "hover"), model)     # train_hover.py:62
    state, data = load_policy_state(str(tmp_path / "hover"))
    assert data["num_timesteps"] == model.num_timesteps and data["algo"] == "PPO"
    pol2 = ActorCritic().cuda()
    pol2.load_state_dict(state)
    x = torch.randn(5, 20, device="cuda")
    assert torch.equal(pol2.mean_value(x)[0], pol.mean_value(x)[0])
    # the .pkl-named statistics load into the VecEnv-side normaliser (test_hover.py:8-10) and into a collector
    env = FpvVecEnv(4, noise_ratio=0.0)
    vn = VecNormalize.load(cb.saved[-1].replace(".zip", ".pkl"), venv=env)
    vn.training, vn.norm_reward = False, False
    col2 = RolloutCollector(FpvEngine(64, 0), pol2, n_steps=4)
    load_normalizer(cb.saved[-1].replace(".zip", ".pkl"), col2)
    assert torch.allclose(vn.obs_rms.rms, col2.obs_norm.rms)
    obs = vn.reset()
    for _ in range(5):                             # test_hover.py:15-26 loop shape
        action, _ = model.predict(torch.from_numpy(obs).cuda(), deterministic=True)
        obs, reward, done, info = vn.step(action.cpu().numpy())
    assert obs.shape == (4, 20) and reward.shape == (4,)
    env.close(); eng.close()


def test_splitk_linear_gradients_match_autograd():
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    torch.manual_seed(0)
    a, b = ActorCritic().cuda(), ActorCritic().cuda()
    b.load_state_dict(a.state_dict())
    obs = torch.randn(16384, 20, device="cuda")
    act = torch.randn(16384, 4, device="cuda")
    va, lpa, _ = a.evaluate(obs, act)                   # large batch -> split-K path
    (va.square().mean() + lpa.mean()).backward()
    for i in range(0, 16384, 2048):                     # reference: small chunks -> stock autograd path
        vb, lpb, _ = b.evaluate(obs[i:i + 2048], act[i:i + 2048])
        ((vb.square().sum() + lpb.sum()) / 16384).backward()
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.allclose(pa.grad, pb.grad, rtol=2e-3, atol=2e-5), n


def test_epoch_graph_equals_eager_minibatches_and_both_shuffles_run():
    """The fused update replays ONE HIP graph per epoch (every minibatch in it, on the row order drawn into a persistent buffer): two
    identical set-ups, one with the graph and one launching every kernel eagerly, hold the same parameters bit for bit after two
    iterations; `shuffle="randperm"` (torch's order) takes the same path and trains on every row once per epoch as well."""
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    from fpv_drone_rl_agent_amd.policy import ActorCritic
    from fpv_drone_rl_agent_amd.ppo import PPO
    from fpv_drone_rl_agent_amd.rollout import RolloutCollector
    n, T = 4096, 16
    flats = {}
    for name, kw in (("graph", {}), ("eager", {"use_graph": False}), ("randperm", {"shuffle": "randperm"})):
        torch.manual_seed(0)
        eng = FpvEngine(n, 0, seed=3)
        pol = ActorCritic(log_std_init=-2.0).cuda()
        model = PPO(RolloutCollector(eng, pol, n_steps=T, policy_seed=5), n_epochs=2, n_minibatches=4, **kw)
        assert model.fused_update and model.shuffle == ("randperm" if name == "randperm" else "keyed")
        model.learn(2 * T * n)
        flats[name] = model._flat.detach().clone()
        assert torch.isfinite(flats[name]).all()
        if name != "randperm":
            assert torch.equal(torch.sort(model._perm).values, torch.arange(T * n, device="cuda"))
        eng.close()
    assert torch.equal(flats["graph"], flats["eager"])
    assert not torch.equal(flats["graph"], flats["randperm"])          # another order, other minibatches
    assert float((flats["graph"] - flats["randperm"]).abs().max()) < 0.05

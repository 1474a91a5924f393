"""world_size-2 gloo run of the one collective on the rollout path: every rank ends up
with the same global advantage mean/std and merged VecNormalize moments as a single process over the
union of the shards."""
import os
import socket

import numpy as np
import torch
import torch.distributed as td
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fpv_drone_rl_agent_amd import dist as D
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world) and D.world_size() == world and D.rank() == rank
    rng = np.random.default_rng(100 + rank)
    adv = rng.normal(rank, 1.0 + rank, 1000 * (rank + 1))          # this shard's advantages
    obs = rng.normal(size=(200 * (rank + 1), 20)) * (1 + rank)      # this shard's observations
    payload = torch.tensor([adv.size, adv.sum(), (adv * adv).sum(), *obs.mean(0), *obs.var(0), obs.shape[0]],
                           dtype=torch.float64)                     # 44 words: a single collective
    g = D.allgather_payload(payload)
    mean, std = D.merge_advantage_stats(g)
    m, v, c = D.merge_moments(g[:, 3:23], g[:, 23:43], g[:, 43])
    q.put((rank, float(mean), float(std), m.numpy(), v.numpy(), float(c)))
    td.barrier()
    td.destroy_process_group()


def test_two_rank_allgather_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    advs = [np.random.default_rng(100 + r).normal(r, 1.0 + r, 1000 * (r + 1)) for r in range(world)]
    obss = []
    for r in range(world):
        rng = np.random.default_rng(100 + r)
        rng.normal(r, 1.0 + r, 1000 * (r + 1))
        obss.append(rng.normal(size=(200 * (r + 1), 20)) * (1 + r))
    a, o = np.concatenate(advs), np.concatenate(obss)
    for rank, mean, std, m, v, c in res:
        assert abs(mean - a.mean()) < 1e-12 and abs(std - a.std()) < 1e-10
        assert np.allclose(m, o.mean(0)) and np.allclose(v, o.var(0)) and c == o.shape[0]
    # bitwise identical on every rank (fixed summation order)
    assert res[0][1:3] == res[1][1:3] and np.array_equal(res[0][3], res[1][3])


def test_rank_merge_reference_equals_the_single_process_union():
    """dist.merge_rank_statistics (the host statement of fpv_rank_merge): G ranks start a rollout from common statistics, each
    folds its own samples in step by step (SB3 RunningMeanStd.update), the merge recovers each rank's batch and yields the
    statistics of ONE process that saw every rank's samples — to rounding, and the same bits whichever rank computes it."""
    from fpv_drone_rl_agent_amd import dist as D
    from oracle import oracle as orc
    rng = np.random.default_rng(7)
    d, G = 20, 4
    base = orc.RunningMeanStd((d,))
    base.update(rng.normal(0.5, 2.0, (5000, d)))                     # what all ranks hold at the start of the rollout
    b = torch.from_numpy(np.concatenate([base.mean, base.var, [base.count]]))
    union = orc.RunningMeanStd((d,)); union.mean, union.var, union.count = base.mean.copy(), base.var.copy(), base.count
    rows = []
    for r in range(G):
        mine = orc.RunningMeanStd((d,)); mine.mean, mine.var, mine.count = base.mean.copy(), base.var.copy(), base.count
        for t in range(8):                                           # eight steps of this rank's drones
            x = rng.normal(r - 1.0, 1.0 + 0.5 * r, (300 + 50 * r, d))
            mine.update(x)
            union.update(x)
        rows.append(np.concatenate([mine.mean, mine.var, [mine.count]]))
    merged = D.merge_rank_statistics(b, torch.from_numpy(np.stack(rows))).numpy()
    assert merged[2 * d] == union.count
    assert np.allclose(merged[:d], union.mean, rtol=1e-12, atol=1e-12) and np.allclose(merged[d:2 * d], union.var, rtol=1e-10)
    # a rank that added nothing changes nothing
    same = D.merge_rank_statistics(b, torch.stack([b, b])).numpy()
    assert np.array_equal(same, b.numpy())


def test_single_process_passthrough():
    from fpv_drone_rl_agent_amd import dist as D
    p = torch.tensor([10.0, 5.0, 30.0], dtype=torch.float64)
    g = D.allgather_payload(p)
    assert g.shape == (1, 3)
    mean, std = D.merge_advantage_stats(g)
    assert float(mean) == 0.5 and abs(float(std) - (3.0 - 0.25) ** 0.5) < 1e-12


def test_bench_launch_contract_under_torchrun_gloo():
    """bench.py exactly as the driver launches it for N > 1 — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` — in its GPU-free plumbing mode:
    rendezvous from the environment, WORLD_SIZE == --gpus, the rollout length that makes exactly K steps, barrier-bracketed
    timing with MAX over ranks, the one all-gather, ONE JSON line from rank 0 carrying the contract's keys."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "4",
           "--plumbing-only"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in rec
    assert rec["n_gpus"] == 2 and rec["steps"] == 20 and rec["scaling"] == "weak"
    assert rec["config"]["n_steps"] == 20 and rec["config"]["total_drones"] == 2 * 65536
    assert rec["plumbing"]["gathered_rows"] == 2 and rec["plumbing"]["max_elapsed_s"] >= 0.001   # rank 1's time won the MAX
    # a world size that contradicts --gpus must fail loudly on every rank
    bad = subprocess.run(cmd[:-7] + ["--gpus", "4", "--steps", "20", "--warmup", "4", "--plumbing-only"], cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stderr + bad.stdout)


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with a clean environment (no RANK / WORLD_SIZE): the parent touches no GPU, starts the two ranks
    through torch.distributed.run itself, relays rank 0's ONE line and the ranks' exit code — what the reference does with
    make_vec_env(..., n_envs=16, vec_env_cls=SubprocVecEnv) (simulation/train_hover.py:41-42).  GPU-free plumbing mode here; the
    -m gpu twin runs real kernels (tests/test_bench_gpu.py)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "4",
                          "--plumbing-only"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout                       # nothing but the one line on stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["total_drones"] == 2 * 65536 and rec["plumbing"]["gathered_rows"] == 2
    # a failing rank fails the parent: --n-steps 0 is refused by the ranks' own argument check
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--n-steps", "0",
                          "--plumbing-only"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0

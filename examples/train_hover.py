"""MI355X counterpart of simulation/train_hover.py: vectorised hover env + VecNormalize + PPO, all on device.
usage: python examples/train_hover.py [--drones 16384] [--timesteps 2e8]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpv_drone_rl_agent_amd import dist as D  # noqa: E402
from fpv_drone_rl_agent_amd.engine import FpvEngine  # noqa: E402
from fpv_drone_rl_agent_amd.policy import ActorCritic  # noqa: E402
from fpv_drone_rl_agent_amd.ppo import PPO  # noqa: E402
from fpv_drone_rl_agent_amd.rollout import RolloutCollector  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", choices=["hover", "yaw"], default="hover",
                    help="hover: simulation/hover.py; yaw: simulation/yaw.py's yaw-setpoint task with domain randomisation")
    ap.add_argument("--drones", type=int, default=16384)
    ap.add_argument("--timesteps", type=float, default=2e8)
    ap.add_argument("--n-steps", type=int, default=32)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--minibatches", type=int, default=8)
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--log-std-init", type=float, default=-2.0,
                    help="sigma0 = e^-2: the +-30 rad/s rate scaling (hover.py:338-340) makes SB3's sigma0 = 1 too violent to ever leave the floor")
    ap.add_argument("--target-kl", type=float, default=0.03)
    ap.add_argument("--ent-coef", type=float, default=0.0)
    ap.add_argument("--gamma", type=float, default=0.99)
    ap.add_argument("--log-std-min", type=float, default=-3.5, help="floor under the policy log-std (sigma >= 0.03)")
    ap.add_argument("--save", default=None, help="path stem: writes <stem>.zip (model) and <stem>.pkl (VecNormalize statistics), "
                                                 "as train_hover.py:62-63 does")
    ap.add_argument("--grad-storage", choices=["f32", "bf16"], default="f32",
                    help="how the update kernels pass activations through memory: bf16 halves that traffic")
    ap.add_argument("--update-path", choices=["one_kernel", "two_kernels", "two_kernels_bf16x6"], default=None,
                    help="which kernels compute the minibatch gradient (default: one_kernel with fp32 storage)")
    ap.add_argument("--autograd-update", action="store_true",
                    help="minibatch gradient through torch autograd instead of the fpv_ppo_grad kernels")
    args = ap.parse_args()
    rank, world, local = D.init_from_env(os.environ.get("FPV_DIST_BACKEND"))
    torch.cuda.set_device(local % torch.cuda.device_count())
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.manual_seed(args.seed)                       # same initial policy on every rank
    obs_dim, act_dim = (20, 4) if args.task == "hover" else (12, 1)
    policy = ActorCritic(obs_dim, act_dim, log_std_init=args.log_std_init).to(dev)
    eng = FpvEngine(args.drones, dev, seed=args.seed, id_offset=rank * args.drones, task=args.task,
                    domain_rand=args.task == "yaw", ledger_capacity=args.drones * max(2, args.n_steps // 8))
    col = RolloutCollector(eng, policy, n_steps=args.n_steps, policy_seed=args.seed, gamma=args.gamma)
    model = PPO(col, learning_rate=args.lr, n_epochs=args.epochs, n_minibatches=args.minibatches, ent_coef=args.ent_coef,
                target_kl=args.target_kl, fused_update=not args.autograd_update, log_std_min=args.log_std_min,
                grad_storage=args.grad_storage, update_path=args.update_path)
    t0 = time.time()
    model.learn(int(args.timesteps), log_every=args.log_every)
    if rank == 0:
        wall = time.time() - t0
        print({"wall_s": round(wall, 1), "timesteps": model.num_timesteps, "env_steps_per_s_with_updates": round(model.num_timesteps / wall),
               **model.logger})
        if args.save:
            from fpv_drone_rl_agent_amd.checkpoint import save_model, save_normalizer
            save_model(args.save, model)
            save_normalizer(args.save + ".pkl", col, format="npz")   # the reference's file name (train_hover.py:63), this build's npz archive
            print("saved", args.save + ".zip", args.save + ".pkl")


if __name__ == "__main__":
    main()

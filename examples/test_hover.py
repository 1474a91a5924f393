"""MI355X counterpart of simulation/test_hover.py: load the model + VecNormalize statistics that
examples/train_hover.py --save wrote and fly deterministic episodes through the single-env facade.
usage: python examples/test_hover.py --load hover [--episodes 3]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpv_drone_rl_agent_amd.checkpoint import load_policy_state  # noqa: E402
from fpv_drone_rl_agent_amd.policy import ActorCritic  # noqa: E402
from fpv_drone_rl_agent_amd.vec_env import FpvVecEnv, VecNormalize  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--load", required=True, help="path stem given to train_hover.py --save")
    ap.add_argument("--episodes", type=int, default=3)
    ap.add_argument("--envs", type=int, default=1, help="test_hover.py:7 uses one env; more fly in parallel")
    args = ap.parse_args()
    state, data = load_policy_state(args.load, map_location="cuda")
    policy = ActorCritic().cuda()
    policy.load_state_dict(state)
    env = FpvVecEnv(args.envs, render=True)                       # test_hover.py:7 (render=True: no floor rule)
    vec_env = VecNormalize.load(args.load + ".pkl", venv=env)     # test_hover.py:8
    vec_env.training = False                                      # test_hover.py:9
    vec_env.norm_reward = False                                   # test_hover.py:10
    obs = vec_env.reset()
    returns, lengths = [], []
    ep_ret, ep_len = np.zeros(args.envs), np.zeros(args.envs, dtype=int)
    while len(returns) < args.episodes:
        with torch.no_grad():                                     # test_hover.py:17 model.predict(obs, deterministic=True)
            mean, _ = policy.mean_value(torch.from_numpy(obs).cuda())
        obs, reward, done, info = vec_env.step(torch.clamp(mean, -1, 1).cpu().numpy())
        ep_ret += reward; ep_len += 1
        for i in np.nonzero(done)[0]:
            returns.append(float(ep_ret[i])); lengths.append(int(ep_len[i]))
            ep_ret[i], ep_len[i] = 0.0, 0
    print({"trained_timesteps": data["num_timesteps"], "episodes": len(returns), "ep_len": lengths[:args.episodes],
           "ep_return": [round(r, 1) for r in returns[:args.episodes]]})
    env.close()


if __name__ == "__main__":
    main()

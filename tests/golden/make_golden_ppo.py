"""Golden vector for the PPO minibatch gradient: SB3's PPO loss (clipped surrogate + vf_coef * MSE -
ent_coef * entropy) differentiated by torch autograd on the CPU in float64.  Run on any machine with torch:
    python tests/golden/make_golden_ppo.py   ->  tests/golden/ppo_grad_v1.npz
Inputs, parameters and gradients are stored in float64; the GPU test evaluates fpv_ppo_grad (fp32 MFMA) on them."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fpv_drone_rl_agent_amd.policy import ActorCritic  # noqa: E402


def main():
    torch.manual_seed(20240)
    rows, clip, vf, ent = 517, 0.2, 0.5, 0.01
    m = ActorCritic(log_std_init=-0.7).double()
    with torch.no_grad():
        m.action_net.weight.mul_(25.0)
        m.action_net.bias.normal_(0, 0.1)
        m.value_net.bias.fill_(0.2)
        m.log_std.add_(torch.randn(4, dtype=torch.float64) * 0.2)
    obs = torch.randn(rows, 20, dtype=torch.float64).clamp_(-10, 10)
    with torch.no_grad():
        mean, _ = m.mean_value(obs)
        act = mean + torch.exp(m.log_std) * torch.randn_like(mean)
        old_lp = m.log_prob(mean, act) + 0.3 * torch.randn(rows, dtype=torch.float64)
    adv = torch.randn(rows, dtype=torch.float64)
    ret = torch.randn(rows, dtype=torch.float64)
    # round the inputs to what the fp32 kernel will see, so that the reference differentiates the same numbers
    obs, act, old_lp, adv, ret = (t.float().double() for t in (obs, act, old_lp, adv, ret))
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.float().double())
    v, lp, en = m.evaluate(obs, act)
    ratio = torch.exp(lp - old_lp)
    pg = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
    vl = torch.nn.functional.mse_loss(v, ret)
    (pg + vf * vl - ent * en.mean()).backward()
    lr = (lp - old_lp).detach()
    out = {"obs": obs.numpy(), "act": act.numpy(), "old_lp": old_lp.numpy(), "adv": adv.numpy(), "ret": ret.numpy(),
           "clip": clip, "vf": vf, "ent": ent,
           "stats": np.array([pg.item(), vl.item(), ((torch.exp(lr) - 1) - lr).mean().item(),
                              ((ratio.detach() - 1).abs() > clip).double().mean().item()])}
    for n, p in m.named_parameters():
        out["param/" + n] = p.detach().numpy()
        out["grad/" + n] = p.grad.numpy()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ppo_grad_v1.npz"), **out)
    print("wrote ppo_grad_v1.npz:", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and k.startswith("grad/")})


if __name__ == "__main__":
    main()

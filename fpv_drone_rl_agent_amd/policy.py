"""Actor-critic policy of the PPO rollout (SB3 MlpPolicy shape).

The reference's only stated architecture is net_arch=[128, 128] (simulation/train_hover.py:57);
north_star asks for PPO, so this is SB3's PPO MlpPolicy: separate tanh trunks for pi and V,
a state-independent log-std, orthogonal init (gain sqrt 2 trunks, 0.01 action head, 1 value
head).  torch.nn holds the parameters (autograd for the update); the rollout-time forward can
run either through torch (rocBLAS GEMMs) or through the fused HIP kernel (ops_policy).
"""
from __future__ import annotations

import math

import torch
from torch import nn

LOG_2PI = math.log(2.0 * math.pi)


class _SplitKLinear(torch.autograd.Function):
    """y = x W^T + b whose weight gradient is a split-K product.  PPO minibatches here are tens of thousands of
    rows against 128 x 128 (or 128 x 20) weights: dW = dY^T X has a tiny output and K = rows, which rocBLAS runs on a
    handful of workgroups (210 us at 65 536 rows, profiles/r01_train_profile.txt).  Cutting the rows into chunks,
    multiplying the chunks as one batched GEMM and summing fills the chip."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g @ w if ctx.needs_input_grad[0] else None
        n = x.shape[0]
        s = 1
        while s < 128 and n % (2 * s) == 0 and n // (2 * s) >= 256:
            s *= 2
        if s > 1:
            gw = torch.bmm(g.view(s, n // s, -1).transpose(1, 2), x.view(s, n // s, -1)).sum(0)
        else:
            gw = g.t() @ x
        return gx, gw, g.sum(0)


def _linear(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """lin(x), with the split-K weight gradient when a large batch is being differentiated."""
    if torch.is_grad_enabled() and x.shape[0] >= 4096 and x.is_cuda:
        return _SplitKLinear.apply(x.contiguous(), lin.weight, lin.bias)
    return lin(x)


def _trunk(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """seq(x) with split-K weight gradients for the Linear layers when a large batch is being differentiated."""
    if not (torch.is_grad_enabled() and x.shape[0] >= 4096 and x.is_cuda):
        return seq(x)
    for m in seq:
        x = _SplitKLinear.apply(x.contiguous(), m.weight, m.bias) if isinstance(m, nn.Linear) else m(x)
    return x


class ActorCritic(nn.Module):
    def __init__(self, obs_dim: int = 20, act_dim: int = 4, hidden: int = 128, log_std_init: float = 0.0):
        super().__init__()
        self.obs_dim, self.act_dim, self.hidden = obs_dim, act_dim, hidden
        self.pi = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh())
        self.vf = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh())
        self.action_net = nn.Linear(hidden, act_dim)
        self.value_net = nn.Linear(hidden, 1)
        self.log_std = nn.Parameter(torch.full((act_dim,), float(log_std_init)))
        for seq in (self.pi, self.vf):
            for m in seq:
                if isinstance(m, nn.Linear):
                    nn.init.orthogonal_(m.weight, gain=math.sqrt(2.0))
                    nn.init.zeros_(m.bias)
        nn.init.orthogonal_(self.action_net.weight, gain=0.01)
        nn.init.zeros_(self.action_net.bias)
        nn.init.orthogonal_(self.value_net.weight, gain=1.0)
        nn.init.zeros_(self.value_net.bias)

    def mean_value(self, obs: torch.Tensor):
        return _linear(self.action_net, _trunk(self.pi, obs)), _linear(self.value_net, _trunk(self.vf, obs)).squeeze(-1)

    def flatten_parameters(self):
        """Re-home every parameter (and its gradient) as a view into ONE contiguous buffer, so that the optimiser
        step, the gradient clip, the zeroing and the cross-rank all-reduce are single-tensor operations instead of
        thirteen small ones each.  Returns (flat_param, flat_grad); state_dict keys and shapes are unchanged."""
        params = list(self.parameters())
        n = sum(p.numel() for p in params)
        dev, dt = params[0].device, params[0].dtype
        flat = nn.Parameter(torch.empty(n, device=dev, dtype=dt))
        flat.grad = torch.zeros(n, device=dev, dtype=dt)
        o = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                flat.data[o:o + k].copy_(p.data.reshape(-1))
                p.data = flat.data[o:o + k].view_as(p)
                p.grad = flat.grad[o:o + k].view_as(p)
                o += k
        self._flat = (flat, flat.grad)
        return self._flat

    def value(self, obs: torch.Tensor):
        return self.value_net(self.vf(obs)).squeeze(-1)

    def log_prob(self, mean: torch.Tensor, actions: torch.Tensor):
        z = (actions - mean) * torch.exp(-self.log_std)
        return (-0.5 * z * z - self.log_std - 0.5 * LOG_2PI).sum(-1)

    @torch.no_grad()
    def act(self, obs: torch.Tensor, eps: torch.Tensor):
        """Sample a = mu + sigma * eps.  Returns (action, value, log_prob); the env receives clip(a, -1, 1)."""
        mean, value = self.mean_value(obs)
        action = mean + torch.exp(self.log_std) * eps
        logp = (-0.5 * eps * eps - self.log_std - 0.5 * LOG_2PI).sum(-1)
        return action, value, logp

    def evaluate(self, obs: torch.Tensor, actions: torch.Tensor):
        mean, value = self.mean_value(obs)
        logp = self.log_prob(mean, actions)
        entropy = (0.5 + 0.5 * LOG_2PI + self.log_std).sum().expand_as(logp)
        return value, logp, entropy

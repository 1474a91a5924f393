"""Checkpoint I/O with the reference's cadence and file naming.

simulation/train_hover.py:8-34 saves, at the end of a rollout and at most every
`min_steps_between_checkpoints` timesteps, `<prefix>_<steps>_steps_<loss>_len<ep_len>_rew<ep_rew>.zip`
(model) plus the same name with `.pkl` (VecNormalize statistics); :62-63 saves `hover` / `hover` at the
end; simulation/test_hover.py:8-11 loads the pair for playback.

Contents: the `.zip` is a real zip archive holding `policy.pth` (torch state_dict, loadable with
weights_only=True), `optimizer.pth` and `data.json` (hyper-parameters, counters) — the member names SB3
uses — but SB3's own pickled `data` blob is not reproduced (SB3 is not installed here, so byte-level
compatibility cannot be checked and is not claimed).  The `.pkl`-named file holds the VecNormalize
statistics in npz format (vec_env.VecNormalize.save uses the same layout).
"""
from __future__ import annotations

import io
import json
import os
import zipfile

import numpy as np
import torch


def _fmt(v):
    # train_hover.py:21-23: round(x, 2) or the string "None"
    return round(v, 2) if v is not None else "None"


def checkpoint_name(prefix: str, num_timesteps: int, logger: dict) -> str:
    loss = _fmt(logger.get("train/loss"))
    ep_len = _fmt(logger.get("rollout/ep_len_mean"))
    ep_rew = _fmt(logger.get("rollout/ep_rew_mean"))
    return f"{prefix}_{num_timesteps}_steps_{loss}_len{ep_len}_rew{ep_rew}.zip"   # train_hover.py:24


def save_model(path: str, model) -> None:
    """model.save(path) — policy + optimiser + counters."""
    if not path.endswith(".zip"):
        path += ".zip"
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
        for name, obj in (("policy.pth", model.policy.state_dict()), ("optimizer.pth", model.opt.state_dict())):
            buf = io.BytesIO()
            torch.save(obj, buf)
            z.writestr(name, buf.getvalue())
        data = {"num_timesteps": model.num_timesteps, "n_epochs": model.n_epochs, "n_minibatches": model.n_minibatches,
                "clip_range": model.clip_range, "ent_coef": model.ent_coef, "vf_coef": model.vf_coef,
                "max_grad_norm": model.max_grad_norm, "gamma": model.col.gamma, "gae_lambda": model.col.lam,
                "n_steps": model.col.T, "algo": "PPO", "policy": "MlpPolicy[128,128]", "logger": dict(model.logger.items())}
        z.writestr("data.json", json.dumps(data))


def load_policy_state(path: str, map_location="cpu") -> tuple[dict, dict]:
    """Returns (policy state_dict, data dict).  Only tensors/JSON are read (weights_only=True)."""
    if not path.endswith(".zip"):
        path += ".zip"
    with zipfile.ZipFile(path) as z:
        state = torch.load(io.BytesIO(z.read("policy.pth")), map_location=map_location, weights_only=True)
        data = json.loads(z.read("data.json"))
    return state, data


_PKL_HELP = ("SB3's VecNormalize.save writes a Python pickle of the wrapper object at this path; this build stores the "
             "statistics as an npz archive (no SB3 in the image, and pickles execute code on load).  ")


def check_stats_save_path(path: str, fmt) -> None:
    """A `.pkl` path gets the npz archive only on request: a reference user who opens it with SB3 would otherwise meet an
    opaque unpickling error (train_hover.py:27,63 / test_hover.py:8 exchange real pickles)."""
    if fmt not in (None, "npz"):
        raise ValueError("format must be None or 'npz'")
    if path.endswith(".pkl") and fmt != "npz":
        raise ValueError(_PKL_HELP + "Pass format='npz' to write that archive under the reference's .pkl file name "
                         "(what CheckpointCallback does), or save to a .npz path.")


def open_stats_archive(path: str):
    """np.load of a statistics archive, with a clear error for a real pickle (SB3's own file)."""
    with open(path, "rb") as f:
        head = f.read(4)
    if head[:2] != b"PK":                                   # npz = zip container
        what = "a Python pickle" if head[:1] == b"\x80" else "not an npz archive"
        raise ValueError(f"{path} is {what}.  " + _PKL_HELP + "Re-export the statistics with VecNormalize.save(path, format='npz') "
                         "from this build; a file written by stable-baselines3 cannot be read here.")
    return np.load(path, allow_pickle=False)


def save_normalizer(path: str, collector, format=None) -> None:
    """model.env.save(path) — the rollout collector's VecNormalize statistics, vec_env.VecNormalize.load-compatible."""
    check_stats_save_path(path, format)
    collector.wait_stats()      # several ranks: the statistics over ALL ranks' drones are written behind the collective
    rms, rr = collector.obs_norm.rms.cpu().numpy(), collector.rew_norm.rms.cpu().numpy()
    d = collector.obs_norm.dim
    with open(path, "wb") as f:
        np.savez(f, obs_mean=rms[:d], obs_var=rms[d:2 * d], obs_count=rms[2 * d], ret_mean=rr[0], ret_var=rr[1],
                 ret_count=rr[2], clip_obs=collector.obs_norm.clip, clip_reward=collector.rew_norm.clip,
                 gamma=collector.gamma, epsilon=collector.obs_norm.eps, norm_obs=collector.norm_obs,
                 norm_reward=collector.norm_reward)


def load_normalizer(path: str, collector) -> None:
    z = open_stats_archive(path)
    rms = np.concatenate([z["obs_mean"], z["obs_var"], [z["obs_count"]]])
    collector.obs_norm.rms.copy_(torch.from_numpy(rms))
    collector.rew_norm.rms.copy_(torch.tensor([float(z["ret_mean"]), float(z["ret_var"]), float(z["ret_count"])],
                                              dtype=torch.float64))


class CheckpointCallback:
    """train_hover.py:8-34.  Call it after every rollout+update (PPO.learn(callback=...))."""

    def __init__(self, save_path: str, name_prefix: str = "rl_model", verbose: int = 0,
                 min_steps_between_checkpoints: int = 20000):
        self.save_path, self.name_prefix, self.verbose = save_path, name_prefix, verbose
        self.min_steps_between_checkpoints = min_steps_between_checkpoints
        os.makedirs(self.save_path, exist_ok=True)                      # train_hover.py:14
        self._last_checkpoint_step = 0
        self.saved: list[str] = []

    def __call__(self, model) -> bool:
        if model.num_timesteps - self._last_checkpoint_step >= self.min_steps_between_checkpoints:  # :19
            name = checkpoint_name(self.name_prefix, model.num_timesteps, model.logger)
            path = os.path.join(self.save_path, name)
            save_model(path, model)                                     # :26
            save_normalizer(path.replace(".zip", ".pkl"), model.col, format="npz")    # :27 (the reference's file name)
            self._last_checkpoint_step = model.num_timesteps
            self.saved.append(path)
            if self.verbose > 0:
                print(f"Saving model checkpoint to {path}\n")
        return True

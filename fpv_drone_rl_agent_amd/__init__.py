"""MI355X-native vectorised quadrotor hover environment + PPO rollout collector.

Drop-in for the hot path of Grzetan/FPV-drone-RL-agent (simulation/hover.py step/reset under
the gymnasium.Env / SB3 VecEnv surface of simulation/train_hover.py).  Compute lives in
libfpv_env.so (hand-written HIP for gfx950, C ABI in include/fpv_env.h).
"""
from . import _lib  # noqa: F401
from .build import build_extension  # noqa: F401

__all__ = ["build_extension"]

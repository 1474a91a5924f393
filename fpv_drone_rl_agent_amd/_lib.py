"""ctypes binding of libfpv_env.so — exactly the symbols include/fpv_env.h (and, apart, fpv_env_debug.h) declare.

The library is the product's only compute path: if it is missing or fails to load this
module raises; there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH, LIB_PATH_EXPERIMENTS, build_extension, variant_path

FPV_F32, FPV_F64 = 0, 1
TASK_HOVER, TASK_YAW = 0, 1
DETECTOR_PIXEL, DETECTOR_ANALYTIC = 0, 1   # fpv_detector: the reference's pixel rules are the default
ABI_VERSION = 7                              # FPV_ABI_VERSION of include/fpv_env.h this binding was written against
STATE_WORDS, OBS_WORDS, ACT_WORDS = 36, 20, 4
INFO_TERMINATED, INFO_TRUNCATED, INFO_OUT_OF_BOUNDS, INFO_ON_FLOOR = 1, 2, 4, 8


class FpvConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int64), ("device", C.c_int32), ("precision", C.c_int32),
        ("floor_check", C.c_int32), ("task", C.c_int32), ("noise_ratio", C.c_double),
        ("seed", C.c_uint64), ("id_offset", C.c_uint64), ("ledger_capacity", C.c_int64),
        ("domain_rand", C.c_int32), ("detector", C.c_int32),
    ]


class FpvLedger(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64), ("rows_per_tile", C.c_int64), ("count", C.c_void_p), ("idx", C.c_void_p), ("info", C.c_void_p),
        ("tag", C.c_void_p), ("ep_return", C.c_void_p), ("obs", C.c_void_p),
    ]


class FpvStepExtras(C.Structure):
    _fields_ = [("done", C.c_void_p), ("ret", C.c_void_p), ("partials", C.c_void_p), ("gamma", C.c_float),
                ("reserved", C.c_int32), ("obs_rms", C.c_void_p), ("ret_rms", C.c_void_p), ("snapshot", C.c_void_p)]


PARTIAL_WORDS = 44


class FpvRolloutIo(C.Structure):
    _fields_ = [
        ("T", C.c_int32), ("reserved", C.c_int32), ("obs", C.c_void_p), ("obs_n", C.c_void_p), ("action", C.c_void_p),
        ("value", C.c_void_p), ("logp", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p),
        ("terminated", C.c_void_p), ("truncated", C.c_void_p), ("ret", C.c_void_p),
        ("gamma", C.c_float), ("clip_obs", C.c_float), ("eps_norm", C.c_float), ("reserved1", C.c_float),
        ("seed", C.c_uint64), ("id_offset", C.c_uint64), ("rms", C.c_void_p), ("ret_rms", C.c_void_p),
        ("rms_out", C.c_void_p * 2), ("ret_rms_out", C.c_void_p * 2), ("snapshots", C.c_void_p),
        ("snapshot_stride", C.c_int64), ("partials", C.c_void_p * 2), ("status", C.c_void_p),
    ]


class FpvPolicyIo(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("obs", C.c_void_p), ("rms", C.c_void_p), ("eps_in", C.c_void_p), ("tick", C.c_void_p),
        ("obs_n", C.c_void_p), ("action", C.c_void_p), ("clipped", C.c_void_p), ("value", C.c_void_p),
        ("logp", C.c_void_p), ("eps_out", C.c_void_p), ("seed", C.c_uint64), ("id_offset", C.c_uint64),
        ("clip_obs", C.c_float), ("eps_norm", C.c_float), ("deterministic", C.c_int32), ("nets", C.c_int32),
        ("narrow", C.c_int32), ("reserved", C.c_int32),
        ("merge_partials", C.c_void_p), ("merge_rows", C.c_int64), ("merge_count", C.c_double), ("rms_out", C.c_void_p),
        ("ret_rms", C.c_void_p), ("ret_rms_out", C.c_void_p), ("merge_snapshot", C.c_void_p), ("n_dev", C.c_void_p),
        ("scatter_t", C.c_void_p), ("scatter_d", C.c_void_p), ("scatter_out", C.c_void_p), ("scatter_stride", C.c_int64),
        ("scatter_scale", C.c_float), ("reserved2", C.c_int32),
    ]


class FpvPpoIo(C.Structure):
    _fields_ = [
        ("rows", C.c_int64), ("index", C.c_void_p), ("obs", C.c_void_p), ("actions", C.c_void_p),
        ("old_logp", C.c_void_p), ("advantages", C.c_void_p), ("returns", C.c_void_p),
        ("clip_range", C.c_float), ("vf_coef", C.c_float), ("ent_coef", C.c_float), ("reserved0", C.c_float),
        ("weights", C.c_void_p * 13), ("grads", C.c_void_p * 13), ("stats", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("storage", C.c_int32), ("path", C.c_int32),
    ]


# every symbol of include/fpv_env.h: name -> (restype, argtypes)
_vp, _i64, _i32, _f32, _u64 = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_uint64
SIGNATURES = {
    "fpv_create": (C.c_int, [C.POINTER(FpvConfig), C.POINTER(_vp)]),
    "fpv_destroy": (C.c_int, [_vp]),
    "fpv_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "fpv_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fpv_step_fused": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(FpvStepExtras), _vp]),
    "fpv_moments_merge": (C.c_int, [_i64, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "fpv_partial_rows": (C.c_int, [_vp, C.POINTER(_i64)]),
    "fpv_moments_merge_to": (C.c_int, [_i64, _i32, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fpv_reward_finalize": (C.c_int, [_i64, _i64, _vp, _vp, _f32, _f32, _vp, _vp]),
    "fpv_dims": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "fpv_tick_ptr": (C.c_int, [_vp, C.POINTER(_vp)]),
    "fpv_step_f64": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fpv_get_state": (C.c_int, [_vp, _vp, _vp]),
    "fpv_set_state": (C.c_int, [_vp, _vp, _vp]),
    "fpv_set_tick": (C.c_int, [_vp, _u64, _vp]),
    "fpv_get_tick": (C.c_int, [_vp, C.POINTER(_u64), _vp]),
    "fpv_ledger_view": (C.c_int, [_vp, C.POINTER(FpvLedger)]),
    "fpv_ledger_clear": (C.c_int, [_vp, _vp]),
    "fpv_ledger_pack": (C.c_int, [_vp, _vp, _f32, _f32, _vp, _i64, _vp]),
    "fpv_rollout_begin": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "fpv_ledger_bind": (C.c_int, [_vp, C.POINTER(FpvLedger)]),
    "fpv_advantage_stats": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "fpv_rank_merge": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "fpv_ledger_scan": (C.c_int, [_vp, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp]),
    "fpv_ledger_scan_finalize": (C.c_int, [_vp, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _f32,
                                           _f32, _vp, _vp]),
    "fpv_ledger_scatter": (C.c_int, [_vp, _vp, _vp, _vp, _f32, _vp, _i64, _i64, _vp]),
    "fpv_gae": (C.c_int, [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "fpv_normalize": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _i32, _f32, _f32, _vp]),
    "fpv_normalize_reward": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _vp]),
    "fpv_policy_create": (C.c_int, [_i32, C.POINTER(_vp)]),
    "fpv_policy_destroy": (C.c_int, [_vp]),
    "fpv_policy_load": (C.c_int, [_vp, C.POINTER(_vp), _vp]),
    "fpv_policy_set_dims": (C.c_int, [_vp, _i32, _i32]),
    "fpv_policy_set_gemm": (C.c_int, [_vp, _i32]),
    "fpv_policy_forward": (C.c_int, [_vp, C.POINTER(FpvPolicyIo), _vp]),
    "fpv_rollout_step": (C.c_int, [_vp, _vp, C.POINTER(FpvPolicyIo), _vp, _vp, _vp, C.POINTER(FpvStepExtras), _vp]),
    "fpv_ppo_workspace_bytes": (C.c_int, [_vp, _i64, C.POINTER(_i64)]),
    "fpv_ppo_grad": (C.c_int, [_vp, C.POINTER(FpvPpoIo), _vp]),
    "fpv_adam_clip_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _vp]),
    "fpv_adam_clip_step_gated": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _vp, _vp]),
    "fpv_ppo_kl_gate": (C.c_int, [_vp, C.c_int32, _f32, _vp, _vp]),
    "fpv_shuffle_indices": (C.c_int, [_vp, _i64, C.c_uint64, C.c_uint64, _vp]),
    "fpv_last_error": (C.c_char_p, []),
    "fpv_version": (C.c_int, []),
}
# include/fpv_env_experiments.h: exported by libfpv_env_experiments.so only (FPV_EXPERIMENTS=1 in the environment selects it)
EXPERIMENT_SIGNATURES = {
    "fpv_rollout_steps": (C.c_int, [_vp, _vp, C.POINTER(FpvRolloutIo), _vp]),
}
# include/fpv_env_debug.h: diagnostic hooks (tools/, bench.py's phase report, tests) — not the product interface
DEBUG_SIGNATURES = {
    "fpv_set_pair_threshold": (C.c_int, [_i64]),
    "fpv_set_quad_range": (C.c_int, [_i64, _i64]),
    "fpv_noop": (C.c_int, [_vp]),
    "fpv_debug_rollout_stamps": (C.c_int, [_vp, _vp]),
    "fpv_debug_rollout_timeout": (C.c_int, [_vp, _u64]),
    "fpv_debug_memset": (C.c_int, [_vp, _i32, _i64, _vp]),
    "fpv_debug_atomic_inc": (C.c_int, [_vp, _i64, _vp]),
}

_lib = None
_lib_path = None
EXPERIMENTS = os.environ.get("FPV_EXPERIMENTS", "") not in ("", "0")    # which library this process loads
# FPV_LIB_VARIANT=name: this process runs on variants/name.so — an A/B or diagnostic build made by
# build.build_extension(variant=name, flags=...) — instead of the product library (tools/ab_*.sh; never copied over the product)
VARIANT = os.environ.get("FPV_LIB_VARIANT", "")


def experiments() -> bool:
    """True when this process runs on libfpv_env_experiments.so (FPV_EXPERIMENTS=1): the measured alternatives of DESIGN §5b."""
    return EXPERIMENTS


def require_experiments(what: str) -> None:
    if not EXPERIMENTS:
        raise RuntimeError(f"{what} is one of the measured alternatives that lost (DESIGN §5b): it is built into "
                           "libfpv_env_experiments.so only — set FPV_EXPERIMENTS=1 in the environment before importing the package")


def load(build_if_missing: bool = True):
    """Load libfpv_env.so — or libfpv_env_experiments.so when FPV_EXPERIMENTS=1 — (building it in-tree with hipcc when absent).
    Raises on failure."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it brings its own copy of the HIP runtime, and a process must run on ONE — a library that pulled in the system's
    # copy before torch loads its own leaves the second one without a device ("no ROCm-capable device is detected")
    import torch  # noqa: F401
    global _lib_path
    lib_path = LIB_PATH_EXPERIMENTS if EXPERIMENTS else LIB_PATH
    if VARIANT:
        lib_path = variant_path(VARIANT)
        if not os.path.exists(lib_path):
            raise RuntimeError(f"FPV_LIB_VARIANT={VARIANT}: {lib_path} does not exist (build.build_extension(variant=..., flags=...))")
    if not os.path.exists(lib_path):
        if not build_if_missing:
            raise RuntimeError(f"{lib_path} is missing: build it with fpv_drone_rl_agent_amd.build.build_extension()")
        build_extension(experiments=EXPERIMENTS)
    lib = C.CDLL(lib_path)
    for name, (res, args) in {**SIGNATURES, **DEBUG_SIGNATURES, **(EXPERIMENT_SIGNATURES if EXPERIMENTS else {})}.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the headers drift apart
        fn.restype, fn.argtypes = res, args
    built = lib.fpv_version()
    if built != ABI_VERSION:         # struct layouts / enum values differ between versions: refuse before the first real call
        raise RuntimeError(f"{lib_path} was built for FPV_ABI_VERSION {built}, this binding is version {ABI_VERSION}: "
                           "rebuild it (fpv_drone_rl_agent_amd.build.build_extension(force=True))")
    _lib, _lib_path = lib, lib_path
    return lib


def library_id() -> dict:
    """Which build this process runs on: path relative to the repository and the first 16 hex digits of its SHA-256
    (bench.py puts it on its JSON line, so a number can be traced to a binary)."""
    import hashlib
    load()
    with open(_lib_path, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()[:16]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return {"path": os.path.relpath(_lib_path, root), "sha16": sha, "variant": VARIANT or None}


def check(rc: int, what: str = "fpv call") -> None:
    if rc != 0:
        msg = load().fpv_last_error().decode("utf-8", "replace")
        exc = ValueError if rc == -1 else RuntimeError
        raise exc(f"{what} failed ({rc}): {msg}")

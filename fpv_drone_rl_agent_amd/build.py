"""In-tree build of the HIP extension (libfpv_env.so) for gfx950 with hipcc.

The shared library sits next to this file so that it travels with a snapshot of the repo;
nothing is installed into site-packages or a JIT cache.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libfpv_env.so")
SOURCES = ["fpv_step.hip", "fpv_yaw.hip", "fpv_rollout.hip", "fpv_policy.hip", "fpv_policy_w16.hip", "fpv_policy_split.hip", "fpv_rollstep.hip", "fpv_ppo.hip", "fpv_capi.hip"]
HEADERS = ["fpv_dynamics.h", "fpv_internal.h", "fpv_mlp.h", "fpv_policy_image.h", "fpv_step_parts.h", "fpv_yaw_parts.h", "fpv_pixel_tile.h",
           os.path.join("..", "..", "include", "fpv_env.h"), os.path.join("..", "..", "include", "fpv_env_debug.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X extension cannot be built")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in files)


def build_extension(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip into libfpv_env.so for gfx950.  Returns the library path."""
    if not force and not _stale():
        return LIB_PATH
    extra = os.environ.get("FPV_EXTRA_HIPCC_FLAGS", "").split()
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + extra
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for f in SOURCES:
        src = os.path.join(CSRC, f)
        if not os.path.exists(src):
            continue
        # Env kernels: -ffp-contract=on — a*b+c fuses only inside one source expression, so the same device function
        # yields the same bits in every kernel it is inlined into (pair == single-wave step, shard independence); the
        # default "fast" also fuses across statements, which depends on the surrounding kernel.  The policy kernels
        # have no such twin and keep "fast" (2 us quicker).
        contract = "fast" if f.startswith(("fpv_policy", "fpv_ppo")) else "on"
        obj = os.path.join(objdir, f.replace(".hip", ".o"))
        cmd = base + [f"-ffp-contract={contract}", "-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        objs.append(obj)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))

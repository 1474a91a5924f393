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
# The experiments library: the same sources with -DFPV_EXPERIMENTS=1 plus the 16-wave policy kernel — the measured alternatives that
# lost (persistent whole-rollout launch, 16-wave policy kernel, 256-thread policy launches, in-launch statistics merge of the env
# step).  Selected by FPV_EXPERIMENTS=1 in the environment (fpv_drone_rl_agent_amd._lib); the product library does
# not contain them.
LIB_PATH_EXPERIMENTS = os.path.join(_HERE, "libfpv_env_experiments.so")
SOURCES = ["fpv_step.hip", "fpv_yaw.hip", "fpv_rollout.hip", "fpv_policy.hip", "fpv_policy_split.hip", "fpv_rollstep.hip", "fpv_ppo.hip", "fpv_capi.hip"]
EXPERIMENT_SOURCES = ["fpv_policy_w16.hip"]
HEADERS = ["fpv_dynamics.h", "fpv_internal.h", "fpv_mlp.h", "fpv_policy_image.h", "fpv_step_parts.h", "fpv_yaw_parts.h", "fpv_pixel_tile.h", "fpv_polydp.h",
           "fpv_policy_split.h", os.path.join("..", "..", "include", "fpv_env.h"), os.path.join("..", "..", "include", "fpv_env_debug.h"),
           os.path.join("..", "..", "include", "fpv_env_experiments.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X extension cannot be built")


def _stale(lib_path: str) -> bool:
    if not os.path.exists(lib_path):
        return True
    t = os.path.getmtime(lib_path)
    files = [os.path.join(CSRC, f) for f in SOURCES + EXPERIMENT_SOURCES + HEADERS]
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in files)


VARIANTS_DIR = os.path.join(os.path.dirname(_HERE), "variants")      # A/B and diagnostic builds (git-ignored; they travel to the GPU box)


def variant_path(name: str) -> str:
    return os.path.join(VARIANTS_DIR, name + ".so")


def build_extension(force: bool = False, verbose: bool = False, experiments: bool = False, variant: str | None = None,
                    flags: tuple = ()) -> str:
    """Compile csrc/*.hip for gfx950 into libfpv_env.so, or with experiments=True into libfpv_env_experiments.so
    (-DFPV_EXPERIMENTS=1), or — variant="name", flags=("-DFPV_...=1", ...) — into variants/name.so: a diagnostic or A/B build that
    a process selects with FPV_LIB_VARIANT=name (fpv_drone_rl_agent_amd._lib); the product library is never overwritten.
    Returns the library path."""
    lib_path = variant_path(variant) if variant else (LIB_PATH_EXPERIMENTS if experiments else LIB_PATH)
    if not force and not _stale(lib_path):
        return lib_path
    extra = os.environ.get("FPV_EXTRA_HIPCC_FLAGS", "").split() + list(flags) + (["-DFPV_EXPERIMENTS=1"] if experiments else [])
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"] + extra
    objdir = os.path.join(_HERE, "build", os.path.join("variants", variant) if variant else ("experiments" if experiments else "product"))
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    objs, cmds = [], []
    for f in SOURCES + (EXPERIMENT_SOURCES if experiments else []):
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
        cmds.append(cmd)
        objs.append(obj)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(4, len(os.sched_getaffinity(0))))) as pool:     # a few hipcc processes side by side
        list(pool.map(lambda c: subprocess.check_call(c, cwd=CSRC), cmds))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return lib_path


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:            # python -m fpv_drone_rl_agent_amd.build NAME -DFLAG=1 ...  ->  variants/NAME.so
        print(build_extension(force=True, verbose=True, variant=sys.argv[1], flags=tuple(sys.argv[2:])))
    else:
        print(build_extension(force=True, verbose=True))
        print(build_extension(force=True, verbose=True, experiments=True))

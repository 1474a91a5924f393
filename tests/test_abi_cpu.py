"""The C-ABI library loads without a GPU and exports every symbol include/fpv_env.h declares;
the product path refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="fpv_env.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fpv_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    from fpv_drone_rl_agent_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fpv_env.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes binding and header drifted apart"
    debug = _declared_symbols("fpv_env_debug.h")            # diagnostic hooks live in their own header
    assert sorted(_lib.DEBUG_SIGNATURES) == debug and not set(debug) & set(declared)
    assert all(hasattr(lib, name) for name in debug)
    assert not [n for n in declared if "debug" in n or n in ("fpv_noop", "fpv_set_pair_threshold")]
    # the measured alternatives that lost live in a third header and a second library; the product library has none of them
    exper = _declared_symbols("fpv_env_experiments.h")
    assert sorted(_lib.EXPERIMENT_SIGNATURES) == exper and not set(exper) & set(declared)
    if not _lib.experiments():
        assert not any(hasattr(lib, name) for name in exper)
    from fpv_drone_rl_agent_amd.build import build_extension
    elib = ctypes.CDLL(build_extension(experiments=True))
    assert all(hasattr(elib, name) for name in declared + debug + exper)
    elib.fpv_version.restype = ctypes.c_int
    assert elib.fpv_version() == _lib.ABI_VERSION
    # the version handshake (a host built against another header would pass structs of another size): header macro ==
    # library == binding; _lib.load() refuses a stale library
    macro = int(re.search(r"#define FPV_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "fpv_env.h")).read()).group(1))
    assert lib.fpv_version() == macro == _lib.ABI_VERSION


def test_config_struct_layout_matches_header():
    from fpv_drone_rl_agent_amd import _lib
    assert ctypes.sizeof(_lib.FpvConfig) == 64  # i64, 4 x i32, f64, 2 x u64, i64, 2 x i32
    assert _lib.FpvConfig.noise_ratio.offset == 24 and _lib.FpvConfig.ledger_capacity.offset == 48
    assert ctypes.sizeof(_lib.FpvLedger) == 64


def test_every_struct_layout_matches_the_header(tmp_path):
    """sizeof and every field offset of the ctypes mirrors against what gcc makes of include/fpv_env.h."""
    import subprocess
    from fpv_drone_rl_agent_amd import _lib
    pairs = {"fpv_config": _lib.FpvConfig, "fpv_ledger": _lib.FpvLedger, "fpv_step_extras": _lib.FpvStepExtras,
             "fpv_policy_io": _lib.FpvPolicyIo, "fpv_ppo_io": _lib.FpvPpoIo, "fpv_rollout_io": _lib.FpvRolloutIo}
    header = open(os.path.join(ROOT, "include", "fpv_env.h")).read() + open(os.path.join(ROOT, "include", "fpv_env_experiments.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "fpv_env.h"', '#include "fpv_env_experiments.h"', 'int main(void) {']
    fields = {}
    for cname, cls in pairs.items():
        body = re.search(r"typedef struct\s*\{([^}]*)\}\s*" + cname + r"\s*;", header, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *rest = decl.split(",")
            names.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[[^\]]*\])?\s*$", first.strip())[0])
            names += [re.findall(r"([A-Za-z_][A-Za-z0-9_]*)", r)[0] for r in rest]
        fields[cname] = names
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        lines += [f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));' for f in names]
    lines += ["return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        assert [n for n, *_ in cls._fields_] == fields[cname], f"{cname}: field names / order differ from the header"
        for f in fields[cname]:
            assert int(got[f"{cname}.{f}"]) == getattr(cls, f).offset, f"{cname}.{f}"


def test_argument_errors_reported_without_gpu():
    from fpv_drone_rl_agent_amd import _lib
    lib = _lib.load()
    assert lib.fpv_create(None, None) == -1
    assert b"null" in lib.fpv_last_error()
    assert lib.fpv_step(None, None, None, None, None, None, None) == -1
    assert lib.fpv_gae(0, 0, None, None, None, None, None, 0.99, 0.95, None, None, None, None, None) == -1
    assert lib.fpv_destroy(None) == 0
    assert lib.fpv_rollout_step(None, None, None, None, None, None, None, None) == -1 and b"fpv_rollout_step" in lib.fpv_last_error()
    if _lib.experiments():
        assert lib.fpv_rollout_steps(None, None, None, None) == -1 and b"fpv_rollout_steps" in lib.fpv_last_error()
    assert lib.fpv_rollout_begin(None, None, None, None, None, None) == -1
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
    with pytest.raises(RuntimeError):
        _lib.check(-2, "x")


def test_engine_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FpvEngine(64, 0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "fpv_drone_rl_agent_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "fpv_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f

"""The C-ABI library loads without a GPU and exports every symbol include/fpv_env.h declares;
the product path refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fpv_env.h")).read()
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
    assert lib.fpv_version() == 1


def test_config_struct_layout_matches_header():
    from fpv_drone_rl_agent_amd import _lib
    assert ctypes.sizeof(_lib.FpvConfig) == 64  # i64, 4 x i32, f64, 2 x u64, i64, 2 x i32
    assert _lib.FpvConfig.noise_ratio.offset == 24 and _lib.FpvConfig.ledger_capacity.offset == 48
    assert ctypes.sizeof(_lib.FpvLedger) == 64


def test_argument_errors_reported_without_gpu():
    from fpv_drone_rl_agent_amd import _lib
    lib = _lib.load()
    assert lib.fpv_create(None, None) == -1
    assert b"null" in lib.fpv_last_error()
    assert lib.fpv_step(None, None, None, None, None, None, None) == -1
    assert lib.fpv_gae(0, 0, None, None, None, None, None, 0.99, 0.95, None, None, None, None) == -1
    assert lib.fpv_destroy(None) == 0
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

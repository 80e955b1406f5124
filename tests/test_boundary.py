"""The drop-in boundary (CPU, no GPU compute): the C-ABI library loads and exports
every symbol include/sgmcmc_hip.h declares; the product never touches oracle/;
the extension is required (no silent fallback)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sgmcmc_hip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgmcmc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = _declared_symbols()
    for name in ("sgmcmc_sghmc_step_f32", "sgmcmc_sghmc_step_f64", "sgmcmc_sgld_step_f32", "sgmcmc_sgld_step_f64",
                 "sgmcmc_rsghmc_step_f32", "sgmcmc_rsghmc_step_f64", "sgmcmc_philox_normal_f32",
                 "sgmcmc_moments_update_f32", "sgmcmc_rhat_pack_f32", "sgmcmc_rhat_finish_f32",
                 "sgmcmc_summary_f32", "sgmcmc_last_error", "sgmcmc_abi_version"):
        assert name in syms


def test_library_loads_and_exports_every_declared_symbol():
    from pysgmcmc_amd import _lib
    _lib.build()
    handle = ctypes.CDLL(_lib.lib_path())
    for name in _declared_symbols():
        assert hasattr(handle, name), "libsgmcmc_hip.so does not export %s" % name
    lib = _lib.lib()
    assert lib.sgmcmc_abi_version() == 1
    assert lib.sgmcmc_summary_workspace_bytes() >= 1024 * 32
    # launch-config knobs are host-only: usable without a GPU
    assert lib.sgmcmc_set_launch_config(-1, 1, 1 << 20, 2) == 0
    assert lib.sgmcmc_set_launch_config(100, 0, 0, -1) != 0
    assert b"block_threads" in lib.sgmcmc_last_error()


def test_library_contains_gfx950_code_object():
    from pysgmcmc_amd import _lib
    out = subprocess.run(["strings", "-n", "6", _lib.lib_path()], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pysgmcmc_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|sgmcmc_oracle|oracle_shim|libsgmcmc_oracle", text, re.M):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from pysgmcmc_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SgmcmcLibraryError):
        _lib.lib()


def test_cpu_tensor_is_refused():
    """No CPU fallback: stepping a sampler whose state is on the CPU raises."""
    import torch
    from pysgmcmc_amd import kernels
    from pysgmcmc_amd._lib import SgmcmcLibraryError
    t = torch.zeros(8)
    with pytest.raises(SgmcmcLibraryError):
        kernels.sghmc_step(t, t.clone(), t.clone(), t.clone(), t.clone(), t.clone(), t.clone(), None,
                           0.01, 1.0, 0.05, True)

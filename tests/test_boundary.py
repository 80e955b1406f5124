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
                 "sgmcmc_rhat_pack_f64", "sgmcmc_rhat_finish_f64",
                 "sgmcmc_summary_f32", "sgmcmc_last_error", "sgmcmc_abi_version"):
        assert name in syms


def test_library_loads_and_exports_every_declared_symbol():
    from pysgmcmc_amd import _lib
    _lib.build()
    handle = ctypes.CDLL(_lib.lib_path())
    for name in _declared_symbols():
        assert hasattr(handle, name), "libsgmcmc_hip.so does not export %s" % name
    lib = _lib.lib()
    assert lib.sgmcmc_abi_version() == _lib.ABI_VERSION == 6
    assert lib.sgmcmc_summary_workspace_bytes() >= 1024 * 32
    # the per-call launch geometry is validated on the host before anything is launched: checkable without a
    # GPU (the output pointer is a dummy that is never dereferenced because the call fails first)
    bad = _lib.LaunchStruct(100, 0, 0, -1)
    rc = lib.sgmcmc_philox_normal_f32(ctypes.c_void_p(4096), 8, 1, 0, None, ctypes.byref(bad), None)
    assert rc == -1 and b"block_threads" in lib.sgmcmc_last_error()
    bad = _lib.LaunchStruct(0, 3, 0, -1)
    rc = lib.sgmcmc_moments_update_f32(ctypes.c_void_p(4096), ctypes.c_void_p(4096), ctypes.c_void_p(4096), 8, 1,
                                       ctypes.byref(bad), None)
    assert rc == -1 and b"quads_per_thread" in lib.sgmcmc_last_error()


def _contract_map():
    text = open(HEADER).read()
    block = text[text.index("---- Contract map"):text.index("---- end of the contract map")]
    groups, cur = {}, None
    for line in block.splitlines():
        m = re.match(r"\s*\*\s*\[([a-z-]+)\]", line)
        if m:
            cur = groups.setdefault(m.group(1), [])
        if cur is not None:
            cur.extend(re.findall(r"\bsgmcmc_[a-z0-9_]+\b(?![*_])", line))
    return groups


def test_contract_map_partitions_the_declared_entry_points():
    """the header says which entry points are the SURVEY 8(b) boundary and which are the cost path's
    internals; the map names every declared entry point exactly once, and nothing that is not declared (or not exported)."""
    from pysgmcmc_amd import _lib
    groups = _contract_map()
    assert sorted(groups) == ["boundary", "cost-path", "svgd", "whole-step"]
    named = [n for g in groups.values() for n in g]
    assert len(named) == len(set(named)), sorted(n for n in set(named) if named.count(n) > 1)
    assert sorted(named) == _declared_symbols()
    handle = ctypes.CDLL(_lib.build())
    assert all(hasattr(handle, n) for n in named)
    # the five step functions, moments, R-hat pack / finish, Philox, statistics: the boundary section 8(b) asked for
    for name in ("sgmcmc_sghmc_step_f32", "sgmcmc_sgld_step_f64", "sgmcmc_rsghmc_step_f32", "sgmcmc_moments_update_f32",
                 "sgmcmc_rhat_pack_f32", "sgmcmc_rhat_finish_f64", "sgmcmc_philox_normal_f32", "sgmcmc_step_stats_finish"):
        assert name in groups["boundary"]
    assert not [n for n in groups["boundary"] if "bnn" in n or "tanh" in n or "svgd" in n]
    assert not hasattr(handle, "sgmcmc_tanh_rowdot_f32")        # superseded by sgmcmc_bias_tanh_rowdot_*, dropped in ABI v6
    assert len(named) == 70


def test_abi_exports_no_experiment_knobs():
    """the ABI a maintainer binds must not carry experiment switches. v4 dropped the hand-written GEMM entry
    points (tile-variant / timing-probe / de-phasing arguments); they were removed with the experiments in round 5."""
    from pysgmcmc_amd import _lib
    handle = ctypes.CDLL(_lib.build())
    header = open(HEADER).read()
    assert not [name for name in _declared_symbols() if "gemm" in name]
    for name in ("sgmcmc_gemm_tn_f32", "sgmcmc_gemm_tn_sghmc_f32", "sgmcmc_gemm_tn_sghmc_blocks"):
        assert not hasattr(handle, name)
    for word in ("phase_counters", "phase_sleep", "int variant", "probe"):
        assert word not in header, "include/sgmcmc_hip.h mentions %r" % word


def test_abi_holds_no_process_wide_state():
    """no global mutable state. The launch-geometry setter of ABI v1 is gone; the header must not
    declare, and the library must not export, any set/get configuration entry point."""
    from pysgmcmc_amd import _lib
    handle = ctypes.CDLL(_lib.build())
    for name in ("sgmcmc_set_launch_config", "sgmcmc_get_launch_config"):
        assert name not in _declared_symbols()
        assert not hasattr(handle, name)
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.lib_path()], capture_output=True, text=True).stdout
    # exported data symbols (B/D = bss/data) would be process-wide state; the one allowed is the THREAD-LOCAL
    # error text behind sgmcmc_last_error() (sgmcmc_host::g_err, a TLS symbol)
    data = [l for l in nm.splitlines() if l.split()[1:2] and l.split()[1] in ("B", "D") and "sgmcmc" in l.lower()
            and "g_err" not in l]
    assert not data, data


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


def test_plain_graph_launch_is_a_call_not_a_default(monkeypatch):
    """The HIP-runtime setting that helps device-bound chains and hurts host-bound ones is never set by importing the package."""
    import pysgmcmc_amd
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    assert pysgmcmc_amd.runtime_env() == {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": None}
    import importlib
    importlib.reload(pysgmcmc_amd)
    assert pysgmcmc_amd.runtime_env() == {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": None}
    in_time = pysgmcmc_amd.prefer_plain_graph_launch()
    assert pysgmcmc_amd.runtime_env() == {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}
    import torch
    assert in_time == (not torch.cuda.is_initialized())
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    # under a profiler's preloaded tool library the HIP runtime is up before Python runs: setting the variable here is too late ...
    monkeypatch.setattr(pysgmcmc_amd, "_PLAIN_LAUNCH_SET_IN_TIME", False)
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert pysgmcmc_amd.prefer_plain_graph_launch() is False
    # ... and stays too late: the variable that call left in the environment is not mistaken for the caller's
    assert pysgmcmc_amd.runtime_env() == {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}
    assert pysgmcmc_amd.prefer_plain_graph_launch() is False
    assert pysgmcmc_amd.configure_for_device_bound_chains(gemm_tuning=False)["plain_graph_launch"] is False
    # ... unless whoever started the process exported it (recorded when the package was imported)
    monkeypatch.setattr(pysgmcmc_amd, "_PLAIN_LAUNCH_EXPORTED_AT_IMPORT", True)
    assert pysgmcmc_amd.prefer_plain_graph_launch() is True
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    importlib.reload(pysgmcmc_amd)
    assert pysgmcmc_amd._PLAIN_LAUNCH_EXPORTED_AT_IMPORT is True
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    importlib.reload(pysgmcmc_amd)
    assert pysgmcmc_amd._PLAIN_LAUNCH_EXPORTED_AT_IMPORT is False


def test_device_bound_switch_reports_what_took_effect(monkeypatch):
    """``configure_for_device_bound_chains()`` is the one documented call behind bench.py's `value`: it returns what took effect
    (the graph launch path only before the first HIP call; GEMM tuning only where TunableOp exists -- not on a box without a
    device, where it must not raise either)."""
    import pysgmcmc_amd
    import torch
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    try:
        took = pysgmcmc_amd.configure_for_device_bound_chains()
        assert took["plain_graph_launch"] == (not torch.cuda.is_initialized()) and set(took) == {"gemm_tuning", "plain_graph_launch"}
        assert took["gemm_tuning"] or not torch.cuda.is_available()
        assert pysgmcmc_amd.runtime_env() == {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}
        monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
        assert pysgmcmc_amd.configure_for_device_bound_chains(gemm_tuning=False, plain_graph_launch=False) == {
            "gemm_tuning": False, "plain_graph_launch": False}
        assert pysgmcmc_amd.runtime_env() == {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": None}
    finally:
        if torch.cuda.is_available():
            from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning
            enable_gemm_tuning(False)


def test_dense_layer_column_plan_without_a_device():
    """``sgmcmc_bnn_dense_tanh_dot_parts`` is host arithmetic (which columns of an M x N layer run as 32 x 64 tiles and which as
    half tiles): without a device it assumes 256 compute units, MI355X's count."""
    from pysgmcmc_amd import _lib
    lib = _lib.lib()
    parts = lib.sgmcmc_bnn_dense_tanh_dot_parts
    assert parts(256, 2048) == 32                                # one tile per CU: 32 column tiles of 64
    assert parts(512, 2048) == 32                                # two full rounds
    assert parts(256, 4864) == 64 + 24                           # configs[4]: 512 full tiles + the last 12 column tiles as 24 half tiles
    assert parts(256, 4096 + 64 * 20) == (4096 + 64 * 20) // 64  # a last round more than half full stays on full tiles
    assert parts(20, 2048) == 0 and parts(256, 100) == 0         # shapes the launch refuses

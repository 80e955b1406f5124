"""ctypes binding of ``libsgmcmc_hip.so`` (C ABI in ``include/sgmcmc_hip.h``).

This is the only way the package reaches the GPU update kernels. There is no
CPU fallback: if the shared library is missing or a call fails, a
``SgmcmcLibraryError`` is raised.
"""
import ctypes
import os

__all__ = ["SgmcmcLibraryError", "lib", "lib_path", "check", "build"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# PYSGMCMC_AMD_LIB: load an alternative build of the same ABI (kernel experiments)
_LIB_PATH = os.environ.get("PYSGMCMC_AMD_LIB") or os.path.join(_CSRC, "libsgmcmc_hip.so")


class SgmcmcLibraryError(RuntimeError):
    """libsgmcmc_hip.so is missing, failed to load, or a call into it failed."""


def lib_path():
    return _LIB_PATH


def build(force=False):
    """Compile ``csrc/libsgmcmc_hip.so`` for gfx950 with hipcc (no GPU needed)."""
    import subprocess
    deps = [os.path.join(_CSRC, f) for f in ("sgmcmc_kernels.hip", "sgmcmc_sghmc.hip", "sgmcmc_sgld.hip", "sgmcmc_rsghmc.hip", "sgmcmc_toy.hip", "sgmcmc_bnn_gemm.hip", "sgmcmc_stream.hpp", "sgmcmc_bnn_fused.hip", "sgmcmc_svgd.hip", "sgmcmc_device.hpp",
                                             "sgmcmc_host.hpp")]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "sgmcmc_hip.h"))
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(d) for d in deps))
    if force or stale:
        subprocess.check_call(["make", "-s", "-j4", "-C", _CSRC, "libsgmcmc_hip.so"])
    return _LIB_PATH


_lib = None
ABI_VERSION = 6               # SGMCMC_ABI_VERSION of include/sgmcmc_hip.h

_u64 = ctypes.c_uint64
_sz = ctypes.c_size_t
_ci = ctypes.c_int
_vp = ctypes.c_void_p


class LaunchStruct(ctypes.Structure):
    """``sgmcmc_launch_t``: per-call launch geometry (0 / -1 = default)."""
    _fields_ = [("block_threads", _ci), ("quads_per_thread", _ci), ("max_blocks", _ci), ("nontemporal", _ci),
                ("start_event", _vp), ("stop_event", _vp)]


_lp = ctypes.POINTER(LaunchStruct)


class StepOptsStruct(ctypes.Structure):
    """``sgmcmc_step_opts_t``: optional extras of one step call (slices, statistics selection, fused moments, ...)."""
    _fields_ = [("first_element", ctypes.c_uint64), ("stats_record_base", ctypes.c_uint32),
                ("stats_record_total", ctypes.c_uint32), ("stats_select", ctypes.c_int), ("flags", ctypes.c_uint),
                ("moments_mean", ctypes.c_void_p), ("moments_m2", ctypes.c_void_p), ("moments_count", ctypes.c_uint64),
                ("scalars_dev", ctypes.c_void_p),
                ("gather_x", ctypes.c_void_p), ("gather_y", ctypes.c_void_p), ("gather_x_out", ctypes.c_void_p),
                ("gather_y_out", ctypes.c_void_p), ("gather_start", ctypes.c_uint64), ("gather_batch", ctypes.c_uint32),
                ("gather_dim", ctypes.c_uint32), ("gather_x_out_ld", ctypes.c_uint32), ("reserved0", ctypes.c_uint32)]


_op = ctypes.POINTER(StepOptsStruct)
STATS_THETA_SQ = 1
STEP_HBM_RESIDENT = 1
STEP_SKIP_MINV_STORE = 2


def _declare(lib):
    lib.sgmcmc_abi_version.restype = _ci
    lib.sgmcmc_last_error.restype = ctypes.c_char_p
    lib.sgmcmc_device_count.restype = _ci
    lib.sgmcmc_event_create.argtypes = [ctypes.POINTER(_vp)]
    lib.sgmcmc_event_create.restype = _ci
    lib.sgmcmc_event_destroy.argtypes = [_vp]
    lib.sgmcmc_event_destroy.restype = _ci
    lib.sgmcmc_event_elapsed_ms.argtypes = [_vp, _vp, ctypes.POINTER(ctypes.c_float)]
    lib.sgmcmc_event_elapsed_ms.restype = _ci
    lib.sgmcmc_event_synchronize.argtypes = [_vp]
    lib.sgmcmc_event_synchronize.restype = _ci
    for sfx, real in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
        f = getattr(lib, "sgmcmc_sghmc_step_" + sfx)
        f.argtypes = [_vp] * 8 + [_sz, real, real, real, real, _ci, _vp, _u64, _u64, _vp, _vp, _op, _lp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_sgld_step_" + sfx)
        f.argtypes = [_vp] * 7 + [_sz, real, real, real, real, _ci, _vp, _u64, _u64, _vp, _vp, _op, _lp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_rsghmc_step_" + sfx)
        f.argtypes = [_vp] * 3 + [_sz, real, real, real, real, real, real, _vp, _u64, _u64, _vp, _vp, _op, _lp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_sghmc_scalars_" + sfx)
        f.argtypes = [real, real, real, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_sgld_scalars_" + sfx)
        f.argtypes = [real, real, real, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_rsghmc_scalars_" + sfx)
        f.argtypes = [real, real, real, real, real, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_toy_chains_" + sfx)
        f.argtypes = [_ci, _ci, ctypes.POINTER(ctypes.c_double), _ci] + [_vp] * 6 + [_sz, _ci, ctypes.POINTER(ctypes.c_double),
                      _vp, _u64, _u64, ctypes.c_int64, _u64, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_philox_normal_" + sfx)
        f.argtypes = [_vp, _sz, _u64, _u64, _vp, _lp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_moments_update_" + sfx)
        f.argtypes = [_vp, _vp, _vp, _sz, _u64, _lp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_rhat_pack_" + sfx)
        f.argtypes = [_vp, _vp, _sz, _u64, _sz, _sz, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_rhat_finish_" + sfx)
        f.argtypes = [_vp, _sz, _sz, _ci, _u64, _vp, _vp, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_bnn_head_" + sfx)
        f.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _sz] + [ctypes.c_double] * 6 + [_ci, _vp, _vp, _vp, _vp, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_bnn_last_layer_backward_" + sfx)
        f.argtypes = [_vp, _vp, _vp, _sz, _sz, _vp, real, _vp, _vp, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_tanh_backward_colsum_" + sfx)
        f.argtypes = [_vp, _vp, _sz, _sz, _vp, real, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_tanh_backward_" + sfx)
        f.argtypes = [_vp, _vp, _sz, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_bnn_fused_sghmc_steps_" + sfx)
        f.argtypes = ([_vp] * 7 + [_sz, _sz, _ci, ctypes.POINTER(_ci), _ci, _vp, _vp, _sz, _vp, _ci]
                      + [ctypes.c_double] * 5 + [real, real, real, _u64, _u64, _u64, _u64, _vp, _vp, _vp])
        f.restype = _ci
        f = getattr(lib, "sgmcmc_bias_tanh_rowdot_" + sfx)
        f.argtypes = [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_bias_tanh_" + sfx)
        f.argtypes = [_vp, _vp, _sz, _sz, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_bnn_head_last_layer_backward_" + sfx)
        f.argtypes = ([_vp, _sz] + [_vp] * 4 + [_sz, _sz] + [ctypes.c_double] * 6 + [_ci, _vp, _vp, _vp, real] + [_vp] * 7 + [_vp])
        f.restype = _ci
        f = getattr(lib, "sgmcmc_window_gather_" + sfx)
        f.argtypes = [_vp, _vp, _sz, _sz, _sz, _sz, _vp, _sz, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_bnn_fused_sgld_steps_" + sfx)
        f.argtypes = ([_vp] * 6 + [_sz, _sz, _ci, ctypes.POINTER(_ci), _ci, _vp, _vp, _sz, _vp, _ci]
                      + [ctypes.c_double] * 5 + [real, real, real, _u64, _u64, _u64, _u64, _vp, _vp, _vp])
        f.restype = _ci
        f = getattr(lib, "sgmcmc_svgd_step_" + sfx)
        f.argtypes = [_vp, _vp, _vp, _sz, _sz, _sz, real, ctypes.c_double, real, _ci, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_svgd_kernel_" + sfx)
        f.argtypes = [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _sz, _vp, _vp]
        f.restype = _ci
        f = getattr(lib, "sgmcmc_summary_" + sfx)
        f.argtypes = [_vp, _sz, _vp, _vp, _vp]
        f.restype = _ci
    lib.sgmcmc_bnn_dense_tanh_f32.argtypes = [_vp] * 4 + [_ci] * 6 + [_vp] * 5
    lib.sgmcmc_bnn_dense_tanh_f32.restype = _ci
    lib.sgmcmc_bnn_dense_tanh_dot_parts.argtypes = [_ci, _ci]
    lib.sgmcmc_bnn_dense_tanh_dot_parts.restype = _ci
    lib.sgmcmc_bnn_dense_tanh_backward_f32.argtypes = [_vp] * 5 + [_ci] * 7 + [_vp, _ci, _ci, _vp, ctypes.c_float, _vp, _vp]
    lib.sgmcmc_bnn_dense_tanh_backward_f32.restype = _ci
    lib.sgmcmc_colsum_finish_f32.argtypes = [_vp, _ci, _ci, _vp, ctypes.c_float, _vp, _vp]
    lib.sgmcmc_colsum_finish_f32.restype = _ci
    lib.sgmcmc_bnn_planes_bytes.argtypes = [_ci, _ci]
    lib.sgmcmc_bnn_planes_bytes.restype = _sz
    lib.sgmcmc_bnn_split_planes_f32.argtypes = [_vp, _ci, _sz, _ci, _ci, _ci, _vp, _sz, _vp]
    lib.sgmcmc_bnn_split_planes_f32.restype = _ci
    lib.sgmcmc_bnn_gw_planes_f32.argtypes = [_vp, _sz, _vp, _sz, _vp, _sz, _ci, _ci, _ci, _ci, _ci, _vp]
    lib.sgmcmc_bnn_gw_planes_f32.restype = _ci
    lib.sgmcmc_philox_bits_u32.argtypes = [_vp, _sz, _u64, _u64, _vp, _vp]
    lib.sgmcmc_counter_add_u64.argtypes = [_vp, _u64, _vp]
    lib.sgmcmc_counter_add_u64.restype = _ci
    lib.sgmcmc_philox_bits_u32.restype = _ci
    lib.sgmcmc_summary_workspace_bytes.restype = _sz
    lib.sgmcmc_svgd_workspace_bytes.argtypes = [_sz, _sz]
    lib.sgmcmc_svgd_workspace_bytes.restype = _sz
    lib.sgmcmc_svgd_max_particles.restype = _ci
    lib.sgmcmc_step_stats_workspace_bytes.argtypes = [_sz]
    lib.sgmcmc_step_stats_workspace_bytes.restype = _sz
    lib.sgmcmc_step_stats_records.argtypes = [_sz, _lp]
    lib.sgmcmc_step_stats_records.restype = _sz
    lib.sgmcmc_step_stats_finish.argtypes = [_vp, _vp, _vp]
    lib.sgmcmc_step_stats_finish.restype = _ci


def lib():
    """The loaded library (argtypes declared). Raises SgmcmcLibraryError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise SgmcmcLibraryError(
            "pysgmcmc_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pysgmcmc_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % _LIB_PATH)
    try:
        handle = ctypes.CDLL(_LIB_PATH)
    except OSError as exc:
        raise SgmcmcLibraryError("pysgmcmc_amd: cannot load %s: %s" % (_LIB_PATH, exc))
    _declare(handle)
    if handle.sgmcmc_abi_version() != ABI_VERSION:
        raise SgmcmcLibraryError("pysgmcmc_amd: ABI version mismatch in %s" % _LIB_PATH)
    _lib = handle
    return handle


def check(rc, what):
    """Turn a non-zero return code into an exception carrying sgmcmc_last_error()."""
    if rc != 0:
        msg = lib().sgmcmc_last_error()
        raise SgmcmcLibraryError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))

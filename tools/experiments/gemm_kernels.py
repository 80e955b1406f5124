"""Torch front end of the round-3 GEMM experiments (``csrc/sgmcmc_gemm_experiments.h``): moved out of
``pysgmcmc_amd.kernels`` in round 4 because neither kernel beat the library products in the sampler's step.

Needs ``libsgmcmc_hip_experiments.so`` (``make -C tools/experiments``); importing this module points ``PYSGMCMC_AMD_LIB`` at it
when the variable is unset -- import it BEFORE anything loads the product library."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("PYSGMCMC_AMD_LIB", os.path.join(_HERE, "libsgmcmc_hip_experiments.so"))

import torch

from pysgmcmc_amd import _lib
from pysgmcmc_amd._lib import check
from pysgmcmc_amd.kernels import _ctr, _on, _ptr, _stream

_declared = False


def lib():
    global _declared
    handle = _lib.lib()
    if not _declared:
        if not hasattr(handle, "sgmcmc_gemm_tn_f32"):
            raise _lib.SgmcmcLibraryError("%s does not export the GEMM experiments: build tools/experiments and load it "
                                          "through PYSGMCMC_AMD_LIB" % _lib.lib_path())
        _vp, _ci, _sz, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint64
        handle.sgmcmc_gemm_tn_f32.argtypes = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _vp, _ci, _vp]
        handle.sgmcmc_gemm_tn_f32.restype = _ci
        handle.sgmcmc_gemm_tn_sghmc_f32.argtypes = [_vp, _vp, _ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _sz, _vp] + [ctypes.c_float] * 4 + [
            _u64, _u64, _vp, _u64, _vp, ctypes.c_uint32, ctypes.c_uint32, _ci, _vp, _ci, _vp]
        handle.sgmcmc_gemm_tn_sghmc_f32.restype = _ci
        handle.sgmcmc_gemm_tn_sghmc_blocks.argtypes = [_ci, _ci, _sz, _ci]
        handle.sgmcmc_gemm_tn_sghmc_blocks.restype = _ci
        _declared = True
    return handle


def gemm_tn(a, b, out, variant=0, phase_counters=None, phase_sleep=0):
    """``out[M, N] = a[K, M]^T @ b[K, N]`` (fp32, matrix cores): the weight-gradient product of a dense layer."""
    K, M = a.shape
    N = b.shape[1]
    if a.dtype != torch.float32 or b.dtype != torch.float32 or out.dtype != torch.float32:
        raise TypeError("gemm_tn is fp32")
    if b.shape[0] != K or tuple(out.shape) != (M, N) or a.stride(1) != 1 or b.stride(1) != 1 or out.stride(1) != 1:
        raise ValueError("gemm_tn: shapes / strides do not match")
    with _on(a):
        rc = lib().sgmcmc_gemm_tn_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, a.stride(0), b.stride(0), out.stride(0),
                                      int(variant), _ptr(phase_counters), int(phase_sleep), _stream(a))
    check(rc, "sgmcmc_gemm_tn_f32")
    return out


def gemm_tn_sghmc(a, b, theta, V, minv, grad_tail, eps, scale_grad, mdecay, grad_decay=0.0, seed=0, step=0, step_dev=None,
                  first_element=0, stats=None, stats_base=0, stats_total=0, grad_out=None, gemm_blocks=0,
                  phase_counters=None, phase_sleeps=0):
    """Weight-gradient product ``a[K, M]^T @ b[K, N]`` with the frozen SGHMC update of the layer as its epilogue
    (``sgmcmc_gemm_tn_sghmc_f32``). ``theta`` / ``V`` / ``minv``: the layer's slice of the arena rows -- the ``M * N``
    weights followed by ``grad_tail.numel()`` more parameters whose gradient ``grad_tail`` already holds."""
    K, M = a.shape
    N = b.shape[1]
    n_tail = 0 if grad_tail is None else grad_tail.numel()
    for t in (a, b, theta, V, minv):
        if t.dtype != torch.float32:
            raise TypeError("gemm_tn_sghmc is fp32")
    if b.shape[0] != K or a.stride(1) != 1 or b.stride(1) != 1 or theta.numel() != M * N + n_tail:
        raise ValueError("gemm_tn_sghmc: shapes / strides do not match")
    with _on(a):
        rc = lib().sgmcmc_gemm_tn_sghmc_f32(
            a.data_ptr(), b.data_ptr(), M, N, K, a.stride(0), b.stride(0), _ptr(theta), _ptr(V, theta), _ptr(minv, theta),
            _ptr(grad_tail), n_tail, _ptr(grad_out), float(eps), float(scale_grad), float(mdecay), float(grad_decay),
            int(seed), int(step), _ctr(step_dev), int(first_element), None if stats is None else _ptr(stats.workspace),
            int(stats_base), int(stats_total), int(gemm_blocks), _ptr(phase_counters), int(phase_sleeps), _stream(a))
    check(rc, "sgmcmc_gemm_tn_sghmc_f32")


def gemm_tn_sghmc_blocks(M, N, n_tail, gemm_blocks=0):
    return int(lib().sgmcmc_gemm_tn_sghmc_blocks(int(M), int(N), int(n_tail), int(gemm_blocks)))



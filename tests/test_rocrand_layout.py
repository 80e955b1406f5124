"""The ABI contract says the library's noise stream is "identical to rocRAND's philox4x32_10 stream with subsequence = quad,
offset = 4 * step" (``include/sgmcmc_hip.h``; north_star: "rocRAND Philox state carried in registers"). Checked word for word
against rocRAND's OWN engine (``/opt/rocm/include/rocrand/rocrand_kernel.h``), built into a small probe with hipcc:

* on the host (the engine is ``__host__ __device__``): rocRAND's words == the oracle's words (which the device words are
  tested bit-exactly against in ``test_hip_parity.py``) -- runs without a GPU;
* on the device: ``rocrand_init`` + ``rocrand4`` in a kernel == ``sgmcmc_philox_bits_u32`` through the C ABI.

Replaces ``tf.random_normal``'s Philox stream, ``pysgmcmc/samplers/base_classes.py:218-220``."""
import ctypes
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "probes", "rocrand_philox_probe.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# (seed, step, quad): both halves of every 64-bit field exercised, quads beyond 2^32 (arenas beyond 2^34 elements)
CASES = [(0, 0, 0), (1, 0, 0), (1234, 7, 3), (0xDEADBEEFCAFEF00D, 5, 2_500_608), ((1 << 63) + 11, (1 << 32) + 9, 17),
         (42, 123_456_789, (1 << 32) + 5), (99, (1 << 40) + 1, (1 << 45) + 123), (2 ** 64 - 1, 2 ** 62 - 1, 2 ** 64 - 1)]


@pytest.fixture(scope="module")
def probe():
    if not os.path.exists(HIPCC) or not os.path.exists("/opt/rocm/include/rocrand/rocrand_kernel.h"):
        pytest.skip("hipcc / rocRAND headers not available")
    out = os.path.join(tempfile.mkdtemp(prefix="rocrand_probe_"), "librocrand_probe.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", out, SRC],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lib = ctypes.CDLL(out)
    p64, p32 = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
    for f in (lib.rocrand_words_host, lib.rocrand_words_device):
        f.argtypes = [ctypes.c_uint64, p64, p64, ctypes.c_int, p32]
    lib.rocrand_words_host.restype = None
    lib.rocrand_words_device.restype = ctypes.c_int
    return lib


def _rocrand_words(fn, seed, steps_quads):
    sub = np.array([q for _, q in steps_quads], dtype=np.uint64)
    off = np.array([(4 * s) % (1 << 64) for s, _ in steps_quads], dtype=np.uint64)
    out = np.zeros(4 * len(sub), dtype=np.uint32)
    rc = fn(ctypes.c_uint64(seed), sub.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
            off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), len(sub), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    assert not rc
    return out.reshape(-1, 4)


def test_rocrand_host_engine_equals_the_oracle_stream(probe):
    """subsequence = quad, offset = 4 * step: rocRAND's counter is (offset / 4, subsequence) = (step, quad), key = seed."""
    from oracle import sgmcmc_oracle as O
    for seed, step, quad in CASES:
        if step >= (1 << 62):
            continue                                          # 4 * step must fit rocRAND's 64-bit offset
        got = _rocrand_words(probe.rocrand_words_host, seed, [(step, quad)])[0]
        want = O.c_philox4x32_10((step & 0xFFFFFFFF, step >> 32, quad & 0xFFFFFFFF, quad >> 32), (seed & 0xFFFFFFFF, seed >> 32))
        assert tuple(int(w) for w in got) == tuple(int(w) for w in want), (seed, step, quad)
    # a run of consecutive quads of one step = consecutive elements of the oracle's word stream
    quads = list(range(40))
    got = _rocrand_words(probe.rocrand_words_host, 77, [(3, q) for q in quads]).reshape(-1)
    assert np.array_equal(got, O.c_philox_bits(77, 3, 160))


@pytest.mark.gpu
def test_rocrand_device_engine_equals_the_library_stream(probe, gpu):
    """The same on the device, against ``sgmcmc_philox_bits_u32`` through the C ABI: word for word, including quads >= 2^32
    (reached through the step kernels' ``first_element`` offset; here through rocRAND's subsequence and the oracle)."""
    import torch
    from oracle import sgmcmc_oracle as O
    from pysgmcmc_amd import kernels
    for seed, step in ((1234, 7), (0xDEADBEEFCAFEF00D, (1 << 40) + 1), (5, 0)):
        n_quads = 4096 + 3
        lib_words = torch.empty(4 * n_quads, dtype=torch.int32, device=gpu)
        kernels.philox_bits(lib_words, seed, step)
        want = lib_words.cpu().numpy().view(np.uint32).reshape(-1, 4)
        got = _rocrand_words(probe.rocrand_words_device, seed, [(step, q) for q in range(n_quads)])
        assert np.array_equal(got, want), (seed, step)
    for seed, step, quad in CASES:
        if step >= (1 << 62):
            continue
        got = _rocrand_words(probe.rocrand_words_device, seed, [(step, quad)])[0]
        want = O.c_philox4x32_10((step & 0xFFFFFFFF, step >> 32, quad & 0xFFFFFFFF, quad >> 32), (seed & 0xFFFFFFFF, seed >> 32))
        assert tuple(int(w) for w in got) == tuple(int(w) for w in want), (seed, step, quad)

"""CPU oracle for the SG-MCMC update path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module. The product package ``pysgmcmc_amd`` never
does (tests/test_boundary.py greps for it).

Two independent restatements of the reference arithmetic live under oracle/:

* ``sgmcmc_oracle.c`` -- fused, one pass per step (what a sane CPU port looks
  like; also the ``"port"`` CPU baseline of bench.py). Loaded here via ctypes.
* this file -- an **op-by-op numpy mirror of the reference's TensorFlow graph**:
  one numpy call per TF op, every temporary materialised, in the order the
  ``tf.control_dependencies`` chain forces. It is the closest executable proxy
  of the TF-CPU sampler available in an image without TensorFlow ("baseline A"
  of BASELINE.md section 3) and the cross-check that pins the fused C code.

PARITY PINNING STATUS: pinned STATISTICALLY against outputs of the reference itself for the
relativistic sampler (the ESS-vs-stepsize data file the reference holds under docs/, reproduced
within 1-3 %; tests/test_reference_outputs.py) and by ONE printed ``next(sampler)`` of the
quickstart notebook for SGHMC; SGHMC / SGLD trajectories are otherwise *unpinned* (TensorFlow
cannot be run here and the reference tests hold no golden trajectories --
pysgmcmc/tests/samplers/sampler_testing.py:55-59). Also pinned: safe_divide/safe_sqrt doctests,
the BNN prior golden constants, Philox KATs, C-vs-numpy bit equality. See sgmcmc_oracle.c header
and DESIGN.md section 4.

Reference citations are relative to the pysgmcmc repository root.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsgmcmc_oracle.so")


# --------------------------------------------------------------------------
# C oracle loader
# --------------------------------------------------------------------------

def build_c(force=False):
    """Compile oracle/libsgmcmc_oracle.so with gcc (no GPU needed)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(os.path.join(_HERE, f))
            for f in ("sgmcmc_oracle.c", "sgmcmc_oracle_body.inc"))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsgmcmc_oracle.so"])
    return _LIB_PATH


_c_lib = None


def load_c():
    """ctypes handle to the fused C oracle with argtypes set."""
    global _c_lib
    if _c_lib is not None:
        return _c_lib
    lib = ctypes.CDLL(build_c())
    u64, sz, ci = ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int
    vp = ctypes.c_void_p
    for sfx, real in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
        f = getattr(lib, "oracle_sghmc_step_" + sfx)
        f.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, real, real, real, real, ci, vp, u64, u64]
        f.restype = ci
        f = getattr(lib, "oracle_sgld_step_" + sfx)
        f.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, real, real, real, real, ci, vp, u64, u64]
        f.restype = ci
        f = getattr(lib, "oracle_rsghmc_step_" + sfx)
        f.argtypes = [vp, vp, vp, sz, real, real, real, real, real, real, vp, u64, u64]
        f.restype = ci
        f = getattr(lib, "oracle_moments_update_" + sfx)
        f.argtypes = [vp, vp, vp, sz, u64]
        f.restype = ci
        f = getattr(lib, "oracle_rsghmc_toy_chain_" + sfx)
        f.argtypes = [ci, vp, ci, vp, vp, ci, real, real, real, real, real, u64, u64, u64, u64, vp]
        f.restype = ci
        f = getattr(lib, "oracle_rhat_pack_" + sfx)
        f.argtypes = [vp, vp, sz, u64, sz, sz, vp]
        f.restype = ci
        f = getattr(lib, "oracle_rhat_finish_" + sfx)
        f.argtypes = [vp, sz, sz, ci, u64, vp]
        f.restype = ci
        f = getattr(lib, "oracle_philox_normal_" + sfx)
        f.argtypes = [u64, u64, sz, vp]
        f.restype = None
        f = getattr(lib, "oracle_philox_normal_range_" + sfx)
        f.argtypes = [u64, u64, u64, sz, vp]
        f.restype = None
        f = getattr(lib, "oracle_sghmc_consts_" + sfx)
        f.argtypes = [real, real, real, vp]
        f.restype = None
        f = getattr(lib, "oracle_safe_divide_" + sfx)
        f.argtypes = [real, real]
        f.restype = real
        f = getattr(lib, "oracle_safe_sqrt_" + sfx)
        f.argtypes = [real]
        f.restype = real
    lib.oracle_philox4x32_10.argtypes = [vp, vp, vp]
    lib.oracle_philox4x32_10.restype = None
    lib.oracle_philox_uniform_bits.argtypes = [u64, u64, sz, vp]
    lib.oracle_philox_uniform_bits.restype = None
    lib.oracle_set_num_threads.argtypes = [ci]
    lib.oracle_max_threads.restype = ci
    _c_lib = lib
    return lib


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(dtype)


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


class CState(object):
    """Flat per-chain state for the C oracle (same arrays the HIP kernels use)."""

    def __init__(self, theta0, dtype=np.float32):
        dt = np.dtype(dtype)
        self.dtype = dt
        self.theta = np.ascontiguousarray(np.asarray(theta0, dtype=dt).ravel()).copy()
        n = self.theta.size
        self.n = n
        # pysgmcmc/samplers/sghmc.py:126-155 initial values
        self.V = np.zeros(n, dt)
        self.tau = np.ones(n, dt)
        self.g = np.ones(n, dt)
        self.v_hat = np.ones(n, dt)
        self.minv = np.ones(n, dt)
        self.r = np.full(n, 0.5, dt)
        self.p = np.zeros(n, dt)     # relativistic momentum


def c_sghmc_step(st, grad, eps, scale_grad, mdecay, adapt, xi=None, seed=0, step=0, grad_decay=0.0):
    lib = load_c()
    grad = np.ascontiguousarray(grad, dtype=st.dtype).ravel()
    if xi is not None:
        xi = np.ascontiguousarray(xi, dtype=st.dtype).ravel()
    f = getattr(lib, "oracle_sghmc_step_" + _sfx(st.dtype))
    rc = f(_p(st.theta), _p(st.V), _p(grad), _p(st.tau), _p(st.g), _p(st.v_hat),
           _p(st.minv), _p(st.r), st.n, eps, scale_grad, mdecay, grad_decay, int(adapt),
           _p(xi), seed, step)
    assert rc == 0


def baseline_sghmc_frozen_step(st, grad, eps, scale_grad, mdecay, seed=0, step=0, grad_decay=0.0):
    """bench.py's CPU-baseline update (f32 only): frozen SGHMC step with ONE Philox call per quad and single-precision
    Box-Muller -- what a CPU port would run. NOT a parity function (its noise differs from the checked stream in the last
    bits); see ``oracle_baseline_sghmc_frozen_step_f32`` in sgmcmc_oracle.c."""
    lib = load_c()
    assert st.dtype == np.float32
    grad = np.ascontiguousarray(grad, dtype=np.float32).ravel()
    f = lib.oracle_baseline_sghmc_frozen_step_f32
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t] + [ctypes.c_float] * 4 + [ctypes.c_uint64] * 2
    rc = f(_p(st.theta), _p(st.V), _p(grad), _p(st.minv), st.n, eps, scale_grad, mdecay, grad_decay, seed, step)
    assert rc == 0


def c_sgld_step(st, grad, eps, A, scale_grad, adapt, xi=None, seed=0, step=0, grad_decay=0.0):
    lib = load_c()
    grad = np.ascontiguousarray(grad, dtype=st.dtype).ravel()
    if xi is not None:
        xi = np.ascontiguousarray(xi, dtype=st.dtype).ravel()
    f = getattr(lib, "oracle_sgld_step_" + _sfx(st.dtype))
    rc = f(_p(st.theta), _p(grad), _p(st.tau), _p(st.g), _p(st.v_hat),
           _p(st.minv), _p(st.r), st.n, eps, A, scale_grad, grad_decay, int(adapt),
           _p(xi), seed, step)
    assert rc == 0


def c_rsghmc_step(st, grad_cost, eps, mass, c, D, b_hat, xi=None, seed=0, step=0, grad_decay=0.0):
    lib = load_c()
    grad_cost = np.ascontiguousarray(grad_cost, dtype=st.dtype).ravel()
    if xi is not None:
        xi = np.ascontiguousarray(xi, dtype=st.dtype).ravel()
    f = getattr(lib, "oracle_rsghmc_step_" + _sfx(st.dtype))
    rc = f(_p(st.theta), _p(st.p), _p(grad_cost), st.n, eps, mass, c, D, b_hat, grad_decay,
           _p(xi), seed, step)
    assert rc == 0


def c_moments_update(theta, mean, m2, count):
    lib = load_c()
    f = getattr(lib, "oracle_moments_update_" + _sfx(theta.dtype))
    rc = f(_p(theta), _p(mean), _p(m2), theta.size, count)
    assert rc == 0


GMM_TARGETS = {   # pysgmcmc/diagnostics/objective_functions.py:62-98
    "gmm1": ((-5., 0., 5.), (1., 1., 1.), (1 / 3., 1 / 3., 1 / 3.)),
    "gmm2": ((-5., 0., 5.), (1. / 0.5, 0.5, 1. / 0.5), (1 / 3., 1 / 3., 1 / 3.)),
    "gmm3": ((-5., 0., 5.), (1. / 0.3, 0.3, 1. / 0.3), (1 / 3., 1 / 3., 1 / 3.)),
}


def c_rsghmc_toy_chain(target, theta, p, eps, n_steps, keep_every=1, first_step=0, seed=0, mass=1.0, c=1.0, D=1.0,
                       b_hat=0.0):
    """Run ``n_steps`` relativistic-SGHMC steps on a toy target ("gmm1/2/3" or "banana") entirely in C, updating
    ``theta`` / ``p`` (1-D float arrays) in place; returns the kept samples ``[ceil(n_steps / keep_every), dim]``
    (theta after steps first_step, first_step + keep_every, ...: ``islice(sampler, 0, n_steps, keep_every)``)."""
    lib = load_c()
    dt = theta.dtype
    dim = theta.size
    if target == "banana":
        tid, tp, k = 1, np.zeros(1, dt), 0
    else:
        mu, var, w = GMM_TARGETS[target]
        tid, tp, k = 0, np.ascontiguousarray(np.concatenate([mu, var, w]).astype(dt)), len(mu)
    kept = np.empty(((n_steps + keep_every - 1) // keep_every, dim), dt)
    rc = getattr(lib, "oracle_rsghmc_toy_chain_" + _sfx(dt))(tid, _p(tp), k, _p(theta), _p(p), dim, eps, mass, c, D, b_hat,
                                                             seed, first_step, n_steps, keep_every, _p(kept))
    assert rc == 0
    return kept


def c_rhat_pack(mean, m2, count, n_shards=1, shard_len=None):
    """[mean | mean^2 | m2/(count-1)] of one chain in the arrays' dtype: the all-reduce payload (``n_shards = 1``) or
    ``n_shards`` chunks of ``[3][shard_len]`` for a reduce-scatter (zero beyond ``n``)."""
    lib = load_c()
    shard_len = mean.size if shard_len is None else int(shard_len)
    out3 = np.empty(3 * n_shards * shard_len, mean.dtype)
    rc = getattr(lib, "oracle_rhat_pack_" + _sfx(mean.dtype))(_p(mean), _p(m2), mean.size, int(count), int(n_shards),
                                                              shard_len, _p(out3))
    assert rc == 0
    return out3


def c_rhat_finish(sum3, m_chains, count, n=None, ld=None):
    """R-hat per parameter from the chain-summed rows (row pitch ``ld``, ``n`` valid), in the rows' dtype."""
    lib = load_c()
    ld = sum3.size // 3 if ld is None else int(ld)
    n = ld if n is None else int(n)
    rhat = np.empty(n, sum3.dtype)
    rc = getattr(lib, "oracle_rhat_finish_" + _sfx(sum3.dtype))(_p(sum3), n, ld, int(m_chains), int(count), _p(rhat))
    assert rc == 0
    return rhat


def c_philox_normal(seed, step, n, dtype=np.float32):
    lib = load_c()
    out = np.empty(n, dtype)
    getattr(lib, "oracle_philox_normal_" + _sfx(dtype))(seed, step, n, _p(out))
    return out


def c_philox_normal_range(seed, step, start, n, dtype=np.float32):
    """xi(seed, step, i) for i in [start, start + n)."""
    lib = load_c()
    out = np.empty(n, dtype)
    getattr(lib, "oracle_philox_normal_range_" + _sfx(dtype))(seed, step, start, n, _p(out))
    return out


def c_philox_bits(seed, step, n):
    lib = load_c()
    out = np.empty(n, np.uint32)
    lib.oracle_philox_uniform_bits(seed, step, n, _p(out))
    return out


def c_philox4x32_10(ctr, key):
    lib = load_c()
    c = np.asarray(ctr, np.uint32).copy()
    k = np.asarray(key, np.uint32).copy()
    out = np.empty(4, np.uint32)
    lib.oracle_philox4x32_10(_p(c), _p(k), _p(out))
    return out


# --------------------------------------------------------------------------
# Pure-Python Philox4x32-10 (third independent statement; small cases only)
# --------------------------------------------------------------------------

def py_philox4x32_10(ctr, key):
    """Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw, SC'11)."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = [int(x) for x in ctr]
    k = [int(x) for x in key]
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF,
             ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
    return c


# --------------------------------------------------------------------------
# Op-by-op numpy mirror of the reference TF graph
# --------------------------------------------------------------------------

def safe_divide(x, y, small_constant=1e-16):
    """pysgmcmc/tensor_utils.py:269 (dtype follows the array operands)."""
    dt = np.result_type(x, y) if isinstance(y, np.ndarray) or isinstance(x, np.ndarray) else np.float64
    dt = np.dtype(dt).type
    two, sc = dt(2.0), dt(small_constant)
    t0 = two * np.sign(y)          # 2. * tf.sign(y)
    t1 = t0 * sc                   #   * small_constant
    t2 = t1 + sc                   #   + small_constant
    t3 = y + t2                    # y + (...)
    return np.divide(x, t3)        # tf.divide


def safe_sqrt(x, clip_value_min=0.0, clip_value_max=float("inf")):
    """pysgmcmc/tensor_utils.py:319-323."""
    dt = np.asarray(x).dtype.type
    t0 = np.maximum(x, dt(clip_value_min))
    t1 = np.minimum(t0, dt(clip_value_max))
    return np.sqrt(t1)


class OpByOpState(object):
    """Per-parameter-tensor state as the reference keeps it: one (n,1) column
    per array (pysgmcmc/tensor_utils.py:87-98, samplers/sghmc.py:126-155)."""

    def __init__(self, theta0, dtype=np.float32):
        dt = np.dtype(dtype)
        self.dtype = dt
        self.theta = np.asarray(theta0, dtype=dt).reshape(-1, 1).copy()
        ones = np.ones_like(self.theta)
        self.tau = ones.copy()
        self.r = (dt.type(1.0) / (self.tau + dt.type(1.0)))
        self.g = ones.copy()
        self.v_hat = ones.copy()
        self.minv = np.divide(dt.type(1.0), np.sqrt(self.v_hat))
        self.V = np.zeros_like(self.theta)
        self.p = np.zeros_like(self.theta)


def _burn_in_ops(st, grad):
    """sghmc.py:168-196 / sgld.py:154-180. Returns minv_t; assigns r, tau, minv,
    g, v_hat in the control-dependency order (all reads see OLD values)."""
    T = st.dtype.type
    tau0, g0, vh0 = st.tau, st.g, st.v_hat
    r_t = T(1.0) / (tau0 + T(1.0))                       # tf.assign(r, 1./(tau+1))
    a0 = -g0                                             # -g
    a1 = a0 * g0                                         # -g * g
    a2 = a1 * tau0                                       # -g * g * tau
    a3 = safe_divide(a2, vh0)
    a4 = a3 + T(1.0)
    tau_t = tau0 + a4                                    # assign_add(tau, ...)
    minv_t = safe_divide(T(1.0), safe_sqrt(vh0))         # assign(minv, ...)
    b0 = (-r_t) * g0
    b1 = r_t * grad
    g_t = g0 + (b0 + b1)                                 # assign_add(g, ...)
    c0 = (-r_t) * vh0
    c1 = r_t * (grad * grad)                             # grad ** 2
    v_hat_t = vh0 + (c0 + c1)                            # assign_add(v_hat, ...)
    st.r, st.tau, st.minv, st.g, st.v_hat = r_t, tau_t, minv_t, g_t, v_hat_t
    return minv_t


def opbyop_sghmc_step(st, grad, eps, scale_grad, mdecay, xi, frozen_minv=None,
                      update_stats_when_frozen=True):
    """One `session.run` of the SGHMC graph, samplers/sghmc.py:165-251.

    ``frozen_minv`` plays the role of the feed at base_classes.py:454: when
    given, it replaces the *output tensor* of ``minv_t`` for everything
    downstream. The statistics ops still execute in TF (they are control
    dependencies of ``v_t``); ``update_stats_when_frozen`` keeps that side
    effect, which no output ever observes.
    """
    T = st.dtype.type
    grad = np.asarray(grad, st.dtype).reshape(-1, 1)
    xi = np.asarray(xi, st.dtype).reshape(-1, 1)
    eps = T(eps)
    noise = T(0.0)                                       # :111
    scale_grad = T(scale_grad)                           # :113
    eps_s = eps / np.sqrt(scale_grad)                    # :115
    mdecay = T(mdecay)                                   # :117
    if frozen_minv is None:
        minv_t = _burn_in_ops(st, grad)
    else:
        if update_stats_when_frozen:
            _burn_in_ops(st, grad)
        minv_t = np.asarray(frozen_minv, st.dtype).reshape(-1, 1)
    # :211-217
    n0 = T(2.0) * np.power(eps_s, T(2.0))
    n1 = n0 * mdecay
    n2 = n1 * minv_t
    n3 = T(2.0) * np.power(eps_s, T(3.0))
    n4 = n3 * np.square(minv_t)
    n5 = n4 * noise
    n6 = n2 - n5
    n7 = np.power(eps_s, T(4.0))
    noise_scale = n6 - n7
    sigma = np.sqrt(np.maximum(noise_scale, T(1e-16)))   # :220
    sample = sigma * xi                                  # base_classes.py:218
    # :233-238
    v0 = -np.power(eps, T(2.0))
    v1 = v0 * minv_t
    v2 = v1 * grad
    v3 = mdecay * st.V
    v4 = v2 - v3
    v5 = v4 + sample
    v_t = st.V + v5
    st.V = v_t
    st.theta = st.theta + v_t                            # :241-243
    return st.theta


def opbyop_sgld_step(st, grad, eps, A, scale_grad, xi, frozen_minv=None,
                     update_stats_when_frozen=True):
    """samplers/sgld.py:149-211."""
    T = st.dtype.type
    grad = np.asarray(grad, st.dtype).reshape(-1, 1)
    xi = np.asarray(xi, st.dtype).reshape(-1, 1)
    eps, A, scale_grad, noise = T(eps), T(A), T(scale_grad), T(0.0)
    if frozen_minv is None:
        minv_t = _burn_in_ops(st, grad)
    else:
        if update_stats_when_frozen:
            _burn_in_ops(st, grad)
        minv_t = np.asarray(frozen_minv, st.dtype).reshape(-1, 1)
    s0 = T(2.0) * eps
    s1 = minv_t * (A - noise)
    s2 = safe_divide(s1, np.asarray(scale_grad))
    sigma = safe_sqrt(s0 * s2)
    sample = sigma * xi
    u0 = (-eps) * minv_t
    u1 = u0 * A
    u2 = u1 * grad
    st.theta = st.theta + (u2 + sample)
    return st.theta


def opbyop_rsghmc_step(st, grad_cost, eps, mass, c, D, b_hat, xi):
    """samplers/relativistic_sghmc.py:100-140, elementwise."""
    T = st.dtype.type
    gl = -np.asarray(grad_cost, st.dtype).reshape(-1, 1)   # tf.gradients(-cost)
    xi = np.asarray(xi, st.dtype).reshape(-1, 1)
    eps, m, c, D, b_hat = T(eps), T(mass), T(c), T(D), T(b_hat)
    m2c2 = np.square(m) * np.square(c)
    p0 = st.p
    pg = (eps * p0) / (m * np.sqrt((p0 * p0) / m2c2 + T(1.0)))
    n = np.sqrt(eps * (T(2.0) * D - eps * b_hat)) * xi
    p1 = p0 + (((eps * gl) + n) - (D * pg))
    pg1 = (eps * p1) / (m * np.sqrt((p1 * p1) / m2c2 + T(1.0)))
    st.p = p1
    st.theta = st.theta + pg1
    return st.theta


# --------------------------------------------------------------------------
# Toy targets (pysgmcmc/diagnostics/objective_functions.py:49-98), fp64 numpy
# --------------------------------------------------------------------------

def banana_log_likelihood(x):
    return -0.5 * (0.01 * x[0] ** 2 + (x[1] + 0.1 * x[0] ** 2 - 10) ** 2)


def banana_cost_grad(x):
    """cost = -loglik; analytic gradient."""
    x0, x1 = float(x[0]), float(x[1])
    t = x1 + 0.1 * x0 ** 2 - 10
    cost = 0.5 * (0.01 * x0 ** 2 + t ** 2)
    return cost, np.array([0.01 * x0 + t * 0.2 * x0, t])


def gmm_log_likelihood(x, mu=(-5, 0, 5), var=(1., 1., 1.), weights=(1 / 3., 1 / 3., 1 / 3.)):
    x = float(np.asarray(x).ravel()[0])
    terms = [np.log(w) - 0.5 * np.log(2.0 * np.pi * v) - 0.5 * ((x - m) ** 2) / v
             for m, v, w in zip(mu, var, weights)]
    mx = max(terms)
    return mx + np.log(sum(np.exp(t - mx) for t in terms))


def gmm_cost_grad(x, mu=(-5, 0, 5), var=(1., 1., 1.), weights=(1 / 3., 1 / 3., 1 / 3.)):
    xv = float(np.asarray(x).ravel()[0])
    terms = np.array([np.log(w) - 0.5 * np.log(2.0 * np.pi * v) - 0.5 * ((xv - m) ** 2) / v
                      for m, v, w in zip(mu, var, weights)])
    ll = gmm_log_likelihood(x, mu, var, weights)
    resp = np.exp(terms - ll)
    dll = sum(rk * (-(xv - m) / v) for rk, m, v in zip(resp, mu, var))
    return -ll, np.array([-dll])


# --------------------------------------------------------------------------
# BNN cost path (pysgmcmc/models/bayesian_neural_network.py), numpy
# --------------------------------------------------------------------------

def log_variance_prior_log_like(log_var, mean=1e-6, var=0.01):
    """bayesian_neural_network.py:102-107 (fp64)."""
    log_var = np.asarray(log_var, np.float64)
    mean, var = np.float64(mean), np.float64(var)
    inner = safe_divide(-np.square(log_var - np.log(mean)), np.asarray(2.0 * var)) - 0.5 * np.log(var)
    return np.mean(np.sum(inner, axis=1))


def weight_prior_log_like(parameters, wdecay=1.0):
    """bayesian_neural_network.py:131-141 (fp64)."""
    log_like = np.float64(0.0)
    n_params = np.float64(0.0)
    for p in parameters:
        p = np.asarray(p, np.float64)
        log_like = log_like + np.sum(-wdecay * 0.5 * np.square(p))
        n_params = n_params + np.float64(p.size)
    return safe_divide(np.asarray(log_like), np.asarray(n_params))


def bnn_forward(params, X):
    """Default net, bayesian_neural_network.py:28-69. params = [W1,b1,W2,b2,W3,b3,W4,b4,output_bias]."""
    W1, b1, W2, b2, W3, b3, W4, b4, ob = params
    h = np.tanh(X @ W1 + b1)
    h = np.tanh(h @ W2 + b2)
    h = np.tanh(h @ W3 + b3)
    mean = h @ W4 + b4
    log_var = np.ones_like(mean) * ob
    return np.concatenate([mean, log_var], axis=1)


def bnn_negative_log_likelihood(params, X, Y, batch_size, n_examples):
    """bayesian_neural_network.py:365-388. Returns (nll, mse)."""
    out = bnn_forward(params, X)
    f_mean = out[:, 0].reshape(-1, 1)
    f_log_var = out[:, 1].reshape(-1, 1)
    f_var_inv = 1.0 / (np.exp(f_log_var) + 1e-16)
    mse = np.square(Y - f_mean)
    log_like = np.sum(np.sum(-mse * (0.5 * f_var_inv) - 0.5 * f_log_var, axis=1))
    log_like = log_like / batch_size
    log_like = log_like + log_variance_prior_log_like(f_log_var) / n_examples
    log_like = log_like + weight_prior_log_like(params) / n_examples
    return -log_like, np.mean(mse)


def bnn_cost_and_grad(params, X, Y, batch_size, n_examples, wdecay=1.0, prior_mean=1e-6, prior_var=0.01):
    """NLL of bayesian_neural_network.py:365-388 and its analytic gradient w.r.t. every parameter, in the
    dtype of ``params`` (numpy; BLAS matmuls). Used as the CPU full-step baseline and cross-checked against
    autograd in tests. Returns (nll, [grad per parameter])."""
    n_layers = (len(params) - 1) // 2
    dt = params[0].dtype.type
    hs, h = [], X
    for l in range(n_layers):
        a = h @ params[2 * l] + params[2 * l + 1]
        h = np.tanh(a) if l < n_layers - 1 else a
        hs.append(h)
    mean = hs[-1]
    s = float(params[-1].ravel()[0])
    es = np.exp(s)
    inv = 1.0 / (es + 1e-16)
    resid = (Y - mean).astype(np.float64)
    sse = float((resid * resid).sum())
    B = X.shape[0]
    n_params = float(sum(p.size for p in params))
    wp_den, lvp_den = n_params + 3e-16, 2.0 * prior_var + 3e-16
    sumsq = float(sum((p.astype(np.float64) ** 2).sum() for p in params))
    log_like = (-(sse * 0.5 * inv) - 0.5 * s * B) / batch_size
    lvp = -(s - np.log(prior_mean)) ** 2 / lvp_den - 0.5 * np.log(prior_var)
    wp = -0.5 * wdecay * sumsq / wp_den
    cost = -(log_like + lvp / n_examples + wp / n_examples)
    coef = wdecay / (wp_den * n_examples)
    grads = [None] * len(params)
    ds = -((sse * 0.5 * es * inv * inv - 0.5 * B) / batch_size + (-2.0 * (s - np.log(prior_mean)) / lvp_den) / n_examples)
    grads[-1] = np.full(params[-1].shape, ds + coef * s, dtype=params[-1].dtype)
    delta = (resid * (-(inv / batch_size))).astype(params[0].dtype)
    for l in range(n_layers - 1, -1, -1):
        h_in = X if l == 0 else hs[l - 1]
        grads[2 * l] = (h_in.T @ delta + dt(coef) * params[2 * l]).astype(params[2 * l].dtype)
        grads[2 * l + 1] = (delta.sum(axis=0) + dt(coef) * params[2 * l + 1]).astype(params[2 * l + 1].dtype)
        if l > 0:
            delta = (delta @ params[2 * l].T) * (dt(1.0) - hs[l - 1] * hs[l - 1])
    return cost, grads


# --------------------------------------------------------------------------
# Cross-chain diagnostics (formulas of pymc3 3.1, unpinned)
# --------------------------------------------------------------------------

def gelman_rubin(chains):
    """chains: (m, n, P) -> R-hat (P,). B = n var_c(mean_c), W = mean_c(var_c),
    Vhat = W (n-1)/n + B/n, Rhat = sqrt(Vhat / W)   (ddof=1 throughout)."""
    x = np.asarray(chains, np.float64)
    m, n = x.shape[0], x.shape[1]
    B = n * np.var(x.mean(axis=1), axis=0, ddof=1)
    W = np.mean(np.var(x, axis=1, ddof=1), axis=0)
    Vhat = W * (n - 1) / n + B / n
    return np.sqrt(Vhat / W)


def effective_n(chains):
    """chains: (m, n) scalar trace per chain -> n_eff
    (pysgmcmc/diagnostics/sampler_diagnostics.py:76-82: n_eff = mn / (1 + 2 sum rho_t),
    truncated at the first odd T with rho_{T+1} + rho_{T+2} < 0; variogram form)."""
    x = np.asarray(chains, np.float64)
    m, n = x.shape
    B = n * np.var(x.mean(axis=1), ddof=1)
    W = np.mean(np.var(x, axis=1, ddof=1))
    Vhat = W * (n - 1) / n + B / n

    def vario(t):
        d = x[:, t:] - x[:, :n - t]
        return np.sum(d * d) / (m * (n - t))

    rho = np.ones(n)
    negative_autocorr = False
    t = 1
    while not negative_autocorr and t < n:
        rho[t] = 1.0 - vario(t) / (2.0 * Vhat)
        if not t % 2:
            negative_autocorr = (rho[t - 1] + rho[t]) < 0
        t += 1
    return int(m * n / (1.0 + 2.0 * rho[1:t].sum()))


# ---------------------------------------------------------------------------------------------
# Stein variational gradient descent (pysgmcmc/samplers/svgd.py) -- op-by-op numpy restatement.
# Parity UNPINNED against reference outputs (no TensorFlow here; the reference has no SVGD test).
# ---------------------------------------------------------------------------------------------

def svgd_median(values):
    """pysgmcmc/tensor_utils.py:197-209: middle value of the flattened, sorted tensor; for an even
    count the mean of the two middle values."""
    v = np.sort(np.asarray(values).reshape(-1))
    n = v.shape[0]
    mid = n // 2
    if n % 2 == 1:
        return v[mid]
    return (v[mid - 1] + v[mid]) / v.dtype.type(2)


def svgd_pairwise_sqdist(X):
    """``squareform(pdist(X)) ** 2`` (pysgmcmc/samplers/svgd.py:165-166): pdist is
    ``tf.norm(X[i] - X[j])`` for every i < j (tensor_utils.py:397-408), squareform mirrors it into a
    symmetric matrix with a zero diagonal (tensor_utils.py:466-565), then the element-wise square."""
    X = np.asarray(X)
    n = X.shape[0]
    dist = np.zeros((n, n), X.dtype)
    for i in range(n):
        diff = X[i][None, :] - X[i + 1:]
        dist[i, i + 1:] = np.sqrt(np.sum(diff * diff, axis=1, dtype=X.dtype))
    dist = dist + dist.T
    return dist * dist


def svgd_kernel(X):
    """RBF kernel with the median bandwidth and its summed gradients, pysgmcmc/samplers/svgd.py:149-181.
    Returns (kernel_matrix, kernel_gradients, h, pairwise_sq_distances), all in X.dtype."""
    X = np.asarray(X)
    T = X.dtype.type
    n = X.shape[0]
    D = svgd_pairwise_sqdist(X)
    h = np.sqrt(T(0.5) * svgd_median(D) / np.log(T(n) + T(1.0)))       # :169-171
    with np.errstate(divide="ignore", invalid="ignore"):                # one particle: h = 0 and 0 / 0 = nan, as in the reference
        K = np.exp(-D / (h * h) / T(2))                                 # :173
    ksum = np.sum(K, axis=1, dtype=X.dtype)                             # :174
    kgrad = -(K @ X) + X * ksum[:, None]                                # :176-179
    with np.errstate(divide="ignore", invalid="ignore"):
        return K, kgrad / (h * h), h, D                                 # :181


def svgd_step(X, G, hist, eps, alpha=0.9, fudge=1e-6, repulsion_sign=1.0):
    """One SVGD step in place on (X, hist), pysgmcmc/samplers/svgd.py:118-147. G[i] = d cost / d X[i].
    ``repulsion_sign = +1`` is the reference as written: the kernel-gradient term is ADDED to
    ``K @ grad(cost)`` and the sum is SUBTRACTED from the particles, which makes the term attract
    particles instead of repelling them (quirk Q10); ``-1`` is Liu & Wang's update."""
    T = X.dtype.type
    n = X.shape[0]
    K, kgrad, h, _ = svgd_kernel(X)
    gt = (K @ G + T(repulsion_sign) * kgrad) / T(n)                     # :124-127
    hist[...] = T(alpha) * hist + T(1. - alpha) * (gt * gt)             # :129-132
    adj = gt / (T(fudge) + np.sqrt(hist))                               # :134-137
    X -= T(eps) * adj                                                   # :139-143
    return K, h

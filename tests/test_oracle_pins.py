"""Pin the oracle (CPU, no GPU): every golden value / known answer the reference
holds for this path, the published Philox KATs, C-vs-numpy bit equality, and the
committed golden trajectories."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Random123 v1.14 known-answer vectors for philox4x32-10 (kat_vectors): (counter, key, expected)
PHILOX_KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,want", PHILOX_KAT)
def test_philox_known_answers(oracle, ctr, key, want):
    assert tuple(int(x) for x in oracle.c_philox4x32_10(ctr, key)) == want
    assert tuple(oracle.py_philox4x32_10(ctr, key)) == want


def test_philox_stream_layout(oracle):
    """element i uses word i%4 of Philox(counter=(step, quad=i//4), key=seed)."""
    seed, step = (7 << 32) | 3, (5 << 32) | 9
    bits = oracle.c_philox_bits(seed, step, 20)
    for i in range(20):
        x = oracle.py_philox4x32_10((9, 5, i // 4, 0), (3, 7))
        assert int(bits[i]) == x[i % 4]


def test_philox_normal_moments(oracle):
    z = np.concatenate([oracle.c_philox_normal(s, 3, 1 << 18) for s in range(8)]).astype(np.float64)
    n = z.size
    assert abs(z.mean()) < 4 / np.sqrt(n)
    assert abs(z.var() - 1) < 6 * np.sqrt(2.0 / n)
    assert abs((z ** 3).mean()) < 0.02 and abs((z ** 4).mean() - 3) < 0.05
    z64 = oracle.c_philox_normal(1, 0, 1 << 16, np.float64)
    z32 = oracle.c_philox_normal(1, 0, 1 << 16, np.float32)
    assert np.abs(z64 - z32).max() < 1e-3       # same words, different uniform mapping


def test_safe_divide_safe_sqrt_doctest_answers(oracle):
    """pysgmcmc/tensor_utils.py:241-265, :304-316."""
    lib = oracle.load_c()
    with np.errstate(divide="ignore"):                     # the plain divisions the doctests contrast safe_divide with
        assert np.isinf(np.float32(1.0) / np.float32(0.0))
        assert np.isinf(np.float32(1.0) / (np.float32(-1e-16) + np.float32(1e-16)))
    assert np.isfinite(lib.oracle_safe_divide_f32(1.0, 0.0))
    assert np.isfinite(lib.oracle_safe_divide_f32(1.0, -1e-16))
    assert np.isfinite(lib.oracle_safe_divide_f64(1.0, 0.0)) and np.isfinite(lib.oracle_safe_divide_f64(1.0, -1e-16))
    assert lib.oracle_safe_sqrt_f32(-1e-16) == 0.0 and lib.oracle_safe_sqrt_f64(-1e-16) == 0.0
    assert lib.oracle_safe_sqrt_f32(4.0) == 2.0
    # numpy mirror agrees
    assert np.isfinite(oracle.safe_divide(np.float32(1.0), np.asarray(np.float32(0.0))))
    assert oracle.safe_sqrt(np.asarray(np.float32(-1e-16))) == 0.0


def test_bnn_prior_golden_constants(oracle):
    """The reference's own golden values (tests/bayesian_neural_network/test_priors.py:20-81).
    Reduction order differs from TF's, so 1 ulp of fp64 is allowed (rtol 1e-14)."""
    d = np.load(os.path.join(GOLDEN, "bnn_priors.npz"))
    assert float(d["log_variance_expected"]) == -325.5744411137498
    assert float(d["weights_expected"]) == -0.01895130158314839
    got = oracle.log_variance_prior_log_like(d["log_variance_input"])
    assert np.isclose(got, d["log_variance_expected"], rtol=1e-14, atol=0)
    weights = [d["weights_input_%d" % k] for k in range(9)]
    assert [w.shape for w in weights] == [(1, 50), (50,), (50, 50), (50,), (50, 50), (50,), (50, 1), (1,), (1, 1)]
    got = oracle.weight_prior_log_like(weights)
    assert np.isclose(got, d["weights_expected"], rtol=1e-14, atol=0)


def test_toy_target_known_answers(oracle):
    """banana optimum (objective_functions.py:54-56) and the quickstart notebook's first
    cost: cost at (0, 0) of the notebook's banana is 50 in magnitude (api_quickstart.ipynb cell 13)."""
    assert np.isclose(oracle.banana_log_likelihood((0, 10)), 0.0)
    cost, grad = oracle.banana_cost_grad(np.array([0.0, 0.0]))
    assert np.isclose(cost, 50.0)
    eps = 1e-6
    for fn, x0 in ((oracle.banana_cost_grad, np.array([0.3, 6.0])), (oracle.gmm_cost_grad, np.array([1.7]))):
        c0, g = fn(x0)
        for k in range(x0.size):
            xp = x0.copy(); xp[k] += eps
            xm = x0.copy(); xm[k] -= eps
            assert np.isclose((fn(xp)[0] - fn(xm)[0]) / (2 * eps), g[k], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fused_c_equals_opbyop_numpy(oracle, dtype):
    """The fused C restatement and the op-by-op numpy mirror of the TF graph agree bit for bit."""
    rng = np.random.default_rng(0)
    n = 777
    th0 = rng.normal(size=n)
    for sampler in ("sghmc", "sgld", "rsghmc"):
        cs, ns = oracle.CState(th0, dtype), oracle.OpByOpState(th0, dtype)
        if sampler == "rsghmc":
            p0 = rng.normal(size=n).astype(dtype)
            cs.p[:] = p0
            ns.p[:] = p0.reshape(-1, 1)
        frozen = None
        for t in range(40):
            grad = (rng.normal(size=n) * 3).astype(dtype)
            xi = rng.normal(size=n).astype(dtype)
            adapt = t < 20
            if sampler == "sghmc":
                oracle.c_sghmc_step(cs, grad, 0.01, 100.0, 0.05, adapt, xi)
                if adapt:
                    oracle.opbyop_sghmc_step(ns, grad, 0.01, 100.0, 0.05, xi)
                    frozen = ns.minv.copy()
                else:
                    oracle.opbyop_sghmc_step(ns, grad, 0.01, 100.0, 0.05, xi, frozen_minv=frozen)
                assert np.array_equal(cs.V, ns.V.ravel())
            elif sampler == "sgld":
                oracle.c_sgld_step(cs, grad, 0.01, 1.0, 100.0, adapt, xi)
                if adapt:
                    oracle.opbyop_sgld_step(ns, grad, 0.01, 1.0, 100.0, xi)
                    frozen = ns.minv.copy()
                else:
                    oracle.opbyop_sgld_step(ns, grad, 0.01, 1.0, 100.0, xi, frozen_minv=frozen)
            else:
                oracle.c_rsghmc_step(cs, grad, 0.001, 1.0, 1.0, 1.0, 0.0, xi)
                oracle.opbyop_rsghmc_step(ns, grad, 0.001, 1.0, 1.0, 1.0, 0.0, xi)
                assert np.array_equal(cs.p, ns.p.ravel())
            assert np.array_equal(cs.theta, ns.theta.ravel()), (sampler, t)
            if adapt and sampler != "rsghmc":
                for name in ("tau", "g", "v_hat", "minv", "r"):
                    assert np.array_equal(getattr(cs, name), getattr(ns, name).ravel()), (sampler, t, name)


def test_frozen_statistics_are_unobservable(oracle):
    """After burn-in TF keeps running the statistics ops as control dependencies, but no
    output depends on them: theta/V are identical whether or not they run."""
    rng = np.random.default_rng(2)
    n = 50
    a, b = oracle.OpByOpState(rng.normal(size=n), np.float32), None
    b = oracle.OpByOpState(a.theta.ravel(), np.float32)
    frozen = np.abs(rng.normal(size=(n, 1))).astype(np.float32) + 0.5
    for t in range(10):
        grad, xi = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
        oracle.opbyop_sghmc_step(a, grad, 0.01, 10.0, 0.05, xi, frozen_minv=frozen, update_stats_when_frozen=True)
        oracle.opbyop_sghmc_step(b, grad, 0.01, 10.0, 0.05, xi, frozen_minv=frozen, update_stats_when_frozen=False)
        assert np.array_equal(a.theta, b.theta) and np.array_equal(a.V, b.V)


def test_golden_trajectories_reproduce(oracle):
    """The committed fixtures are what the C oracle produces today (guards the fixtures and the oracle)."""
    d = np.load(os.path.join(GOLDEN, "trajectories.npz"))
    for key in d["cases"]:
        sampler, target, dtname, eps, burn = str(key).split("|")
        dt, eps, burn = np.dtype(dtname), float(eps), int(burn)
        theta0 = {"gmm1": [0.0], "banana": [0.0, 6.0]}[target]
        st = oracle.CState(theta0, dt)
        if sampler == "rsghmc":
            st.p[:] = d[key + "|p0"]
        grads, xis, thetas = d[key + "|grad"], d[key + "|xi"], d[key + "|theta"]
        assert grads.dtype == dt and thetas.dtype == dt
        for t in range(thetas.shape[0]):
            adapt = t < burn or burn <= 0
            with np.errstate(all="ignore"):
                if sampler == "sghmc":
                    oracle.c_sghmc_step(st, grads[t], eps, 1.0, 0.05, adapt, xis[t])
                elif sampler == "sgld":
                    oracle.c_sgld_step(st, grads[t], eps, 1.0, 1.0, adapt, xis[t])
                else:
                    oracle.c_rsghmc_step(st, grads[t], eps, 1.0, 1.0, 1.0, 0.0, xis[t])
            assert np.array_equal(st.theta, thetas[t], equal_nan=True), (key, t)


def test_welford_and_rhat_formulas(oracle):
    rng = np.random.default_rng(0)
    x = rng.normal(size=(50, 33))
    mean, m2 = np.zeros(33), np.zeros(33)
    for c in range(50):
        oracle.c_moments_update(np.ascontiguousarray(x[c]), mean, m2, c + 1)
    assert np.allclose(mean, x.mean(axis=0)) and np.allclose(m2 / 49, x.var(axis=0, ddof=1))
    # identical chains -> B = 0 -> Rhat = sqrt((n-1)/n); offset chains -> Rhat > 1
    ch = np.stack([x, x])
    assert np.allclose(oracle.gelman_rubin(ch), np.sqrt(49 / 50.0))
    ch2 = np.stack([x, x + 3.0])
    assert (oracle.gelman_rubin(ch2) > 1.5).all()
    iid = rng.normal(size=(4, 2000))
    assert 4000 < oracle.effective_n(iid) <= 8000 * 1.3
    ar = np.zeros((4, 2000))
    for t in range(1, 2000):
        ar[:, t] = 0.9 * ar[:, t - 1] + rng.normal(size=4)
    assert oracle.effective_n(ar) < 1200          # ~ 8000 * (1-0.9)/(1+0.9) = 421


def test_svgd_oracle_known_answers_and_golden(oracle):
    """SVGD restatement (pysgmcmc/samplers/svgd.py): the reference's doctest answers for `median`
    (tensor_utils.py:183-194) and `pdist`/`squareform` (tensor_utils.py:352-364,442-451: equal to scipy's),
    structural properties of the kernel, and the committed trajectories."""
    from scipy.spatial.distance import pdist, squareform
    assert oracle.svgd_median(np.array([1., 3., 5.])) == 3.0
    assert oracle.svgd_median(np.array([1., 3., 5., 7.])) == 4.0
    X = np.array([[0.77228064, 0.09543156], [0.3918973, 0.96806584], [0.66008144, 0.22163063]])
    np.testing.assert_allclose(oracle.svgd_pairwise_sqdist(X), squareform(pdist(X)) ** 2, rtol=1e-14)
    K, kg, h, D = oracle.svgd_kernel(X)
    assert np.array_equal(K, K.T) and np.all(np.diag(K) == 1)
    np.testing.assert_allclose(h, np.sqrt(0.5 * np.median(D) / np.log(4.0)), rtol=1e-15)
    # kernel gradients are sum_j K_ij (x_i - x_j) / h^2
    ref = np.stack([sum(K[i, j] * (X[i] - X[j]) for j in range(3)) for i in range(3)]) / h ** 2
    np.testing.assert_allclose(kg, ref, rtol=1e-12, atol=1e-14)
    d = np.load(os.path.join(GOLDEN, "svgd.npz"))
    for key in d["cases"]:
        key = str(key)
        target, n, dtname, sign = key.split("|")
        dt = np.dtype(dtname)
        Xc, H = d[key + "|x0"].copy(), np.zeros_like(d[key + "|x0"])
        assert Xc.dtype == dt
        tol = 1e-5 if dt == np.float32 else 1e-12          # BLAS summation order may differ between hosts
        K0, kg0, h0, D0 = oracle.svgd_kernel(Xc)
        np.testing.assert_allclose(K0, d[key + "|K0"], rtol=tol, atol=tol)
        np.testing.assert_allclose(h0, d[key + "|bw0"][1], rtol=tol)
        for t in range(d[key + "|x"].shape[0]):
            oracle.svgd_step(Xc, d[key + "|grad"][t], H, 0.1, 0.9, 1e-6, float(sign))
            np.testing.assert_allclose(Xc, d[key + "|x"][t], rtol=100 * tol, atol=100 * tol)
            Xc[...] = d[key + "|x"][t]                      # re-anchor: one-step pins, no drift
        # the reference's sign contracts the cloud, the repulsive sign keeps it spread
    spread = {s: d["banana|10|float64|%d|x" % s][-1].std(axis=0).max() for s in (1, -1)}
    assert spread[1] < spread[-1]


def test_c_oracle_is_clean_under_address_and_ub_sanitizers():
    """SURVEY section 5 (race detection / sanitizers): every entry point of the C oracle, ragged sizes, sharded R-hat
    layout and the toy chains, built with -fsanitize=address,undefined (errors abort) and run once."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    res = subprocess.run(["make", "-s", "-C", here, "sanitize"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "sanitize: ok" in res.stdout, (res.stdout[-2000:], res.stderr[-2000:])

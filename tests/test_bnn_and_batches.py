"""BNN cost path, model driver and minibatch generator: what the reference's own tests pin
(tests/bayesian_neural_network/*, tests/test_data_batches.py), CPU parts here, GPU parts marked."""
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from pysgmcmc_amd.data_batches import Placeholder, generate_batches, generate_shuffled_batches
from pysgmcmc_amd.models.bayesian_neural_network import (
    BayesianNeuralNetwork, BNNCost, init_mlp_params, log_variance_prior_log_like, weight_prior_log_like)
from pysgmcmc_amd.sampling import Sampler

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------- priors (golden values of the reference)

def test_prior_golden_values():
    """tests/bayesian_neural_network/test_priors.py:20-81; 1 ulp of fp64 allowed (reduction order)."""
    d = np.load(os.path.join(GOLDEN, "bnn_priors.npz"))
    got = float(log_variance_prior_log_like(d["log_variance_input"], mean=1e-6, var=0.01, dtype=torch.float64))
    assert np.isclose(got, -325.5744411137498, rtol=1e-14, atol=0)
    got = float(weight_prior_log_like([d["weights_input_%d" % k] for k in range(9)]))
    assert np.isclose(got, -0.01895130158314839, rtol=1e-14, atol=0)


def test_bnn_cost_matches_oracle_and_fused_path(oracle):
    """torch NLL == numpy restatement of bayesian_neural_network.py:365-388; fused analytic backward == autograd."""
    rng = np.random.default_rng(0)
    params = init_mlp_params(1, seed=4, dtype=torch.float64)
    for p in params[1::2]:
        p.normal_()
    X, Y = rng.normal(size=(20, 1)), rng.normal(size=(20, 1))
    xp, yp = Placeholder().feed(torch.tensor(X)), Placeholder().feed(torch.tensor(Y))
    c = BNNCost(xp, yp, batch_size=20, n_examples=100)
    nll, mse = c.negative_log_likelihood(params, xp.value, yp.value)
    want_nll, want_mse = oracle.bnn_negative_log_likelihood([p.numpy() for p in params], X, Y, 20, 100)
    assert np.isclose(float(nll), want_nll, rtol=1e-13) and np.isclose(float(mse), want_mse, rtol=1e-13)
    ps = [p.clone().requires_grad_(True) for p in params]
    grads = torch.autograd.grad(c(ps), ps)
    gv = [torch.empty_like(p) for p in params]
    from pysgmcmc_amd._lib import SgmcmcLibraryError
    with pytest.raises(SgmcmcLibraryError):
        c.cost_and_grad(params, gv)             # HIP kernels by default: CPU tensors are refused, no fallback
    c.use_hip_kernels = False                   # explicit opt-in to the torch-op formulation
    cost2 = c.cost_and_grad(params, gv)
    assert np.isclose(float(cost2), want_nll, rtol=1e-13)
    for a, b in zip(grads, gv):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-14)
    # the numpy analytic gradient of the oracle (CPU full-step baseline) agrees too
    c_np, g_np = oracle.bnn_cost_and_grad([p.numpy() for p in params], X, Y, 20, 100)
    assert np.isclose(c_np, want_nll, rtol=1e-13)
    for a, b in zip(grads, g_np):
        assert np.allclose(a.numpy(), b, rtol=1e-10, atol=1e-14)


def test_fused_layer_launches_are_device_only():
    """The one-launch layer kernels (forward / backward) and the pitched feed buffer belong to the HIP path: on host tensors the
    shape checks say no, the launches refuse loudly, and the cost function offers no buffer -- nothing falls back silently."""
    from pysgmcmc_amd import kernels
    from pysgmcmc_amd._lib import SgmcmcLibraryError
    h, W, out = torch.zeros(32, 64), torch.zeros(64, 64), torch.zeros(32, 64)
    assert not kernels.bnn_dense_tanh_fits(h, W, out)
    assert not kernels.bnn_dense_tanh_backward_fits(h, W, out, torch.zeros(32, 64))
    with pytest.raises((ValueError, SgmcmcLibraryError)):
        kernels.bnn_dense_tanh(h, W, torch.zeros(64), out)
    with pytest.raises((ValueError, SgmcmcLibraryError)):
        kernels.bnn_dense_tanh_backward(h, W, out, torch.zeros(32, 64))
    with pytest.raises((TypeError, SgmcmcLibraryError)):
        kernels.colsum_finish(torch.zeros(2, 64), torch.zeros(64))
    xp, yp = Placeholder().feed(torch.zeros(32, 64)), Placeholder().feed(torch.zeros(32, 1))
    c = BNNCost(xp, yp, batch_size=32, n_examples=100)
    assert c.wants_static_feeds                                  # HIP path (the default): feed me through my own buffers
    assert c.static_feed_buffer(xp, xp.value) is None            # ... which exist on the device only
    c.use_hip_kernels = False
    assert not c.wants_static_feeds


def test_cost_plan_is_decided_once_and_has_three_switches():
    """The launch sequence of the HIP cost path is a stored plan (``BNNCost._plan``), one per configuration; the only switches
    are ``use_hip_kernels``, ``fold_prior`` and ``fused_layers``. On host tensors no layer fits a device launch: the plan names
    library products only (and running it raises, see above) -- the mixed states of the old per-launch booleans do not exist."""
    params = init_mlp_params(64, hidden=(128, 128), seed=1, dtype=torch.float32)
    xp, yp = Placeholder().feed(torch.zeros(64, 64)), Placeholder().feed(torch.zeros(64, 1))
    c = BNNCost(xp, yp, batch_size=64, n_examples=100)
    for gone in ("fuse_tanh_rowdot", "fuse_head", "fused_dense", "fused_dense_backward", "bias_gradient_from_product"):
        assert not hasattr(c, gone)
    grads = [torch.zeros_like(p) for p in params]
    plan = c.plan_summary(params, grads)
    assert plan == {"forward": ["mm+bias_tanh", "mm+bias_tanh_rowdot", "by rowdot"], "head": "head",
                    "backward": {2: "last_layer_backward", 1: "mm+tanh_backward_colsum"},
                    "first_layer_bias_gradient": "column sums"}
    assert c._plan(params, grads, xp.value, c._buffers(params, 64), False) is c._plan(params, grads, xp.value, c._buffers(params, 64), False)
    assert c.plan_summary(params, grads, theta_sumsq_partials=torch.zeros(8))["head"] == "head+last_layer_backward"
    # a multi-output last layer: generic products all the way, the loss head on its own
    wide = init_mlp_params(64, hidden=(128,), seed=1, dtype=torch.float32)
    wide[2] = torch.zeros(128, 3)
    wide[3] = torch.zeros(3)
    plan = c.plan_summary(wide, [torch.zeros_like(p) for p in wide])
    assert plan["forward"] == ["mm+bias_tanh", "addmm"] and plan["head"] == "head" and plan["backward"] == {1: "mm+tanh_backward_colsum"}


def test_predict_evaluates_every_kept_network_in_one_pass():
    """``predict`` (pysgmcmc/models/bayesian_neural_network.py:599-630): ensemble mean / variance of the kept networks' means, or the
    individual means and noise variances -- from ONE batched evaluation, equal to the net-by-net loop (host tensors here)."""
    bnn = BayesianNeuralNetwork(session="cpu", dtype=torch.float64, n_nets=7, normalize_input=False, normalize_output=False)
    bnn.is_trained = True
    for k in range(7):
        net = init_mlp_params(3, hidden=(8, 5), seed=k, dtype=torch.float64)
        net[1].normal_()
        net[-1].fill_(-2.0 - 0.1 * k)
        bnn.samples.append(net)
    X = np.random.RandomState(0).randn(11, 3)
    out = bnn._network_outputs(X)
    ref = np.stack([bnn.compute_network_output(params=net, input_data=X) for net in bnn.samples])
    assert out.shape == (7, 11, 2) and np.allclose(out, ref, rtol=1e-13, atol=1e-13)
    mean, var = bnn.predict(X)
    assert np.allclose(mean, ref[:, :, 0].mean(0)) and np.allclose(var, ref[:, :, 0].var(0))
    f, noise = bnn.predict(X, return_individual_predictions=True)
    assert np.allclose(f, ref[:, :, 0]) and np.allclose(noise, np.exp(ref[:, :, 1]))
    #  the rows go through in chunks under a byte budget -- here 2 rows per pass -- with the same numbers; and kept
    # samples may be numpy arrays, as compute_network_output accepts them
    bnn.PREDICT_ACTIVATION_BYTES = 2 * 2 * 7 * 8 * 8
    assert np.allclose(bnn._network_outputs(X), out, rtol=1e-13, atol=1e-13)
    bnn.samples = type(bnn.samples)([[p.numpy() for p in net] for net in bnn.samples], maxlen=7)
    assert np.allclose(bnn._network_outputs(X), out, rtol=1e-13, atol=1e-13)


def test_init_seeding_and_shapes():
    """tests/bayesian_neural_network/test_seeding.py: same seed => identical initial weights."""
    a, b, c = init_mlp_params(1, seed=7), init_mlp_params(1, seed=7), init_mlp_params(1, seed=8)
    assert [tuple(p.shape) for p in a] == [(1, 50), (50,), (50, 50), (50,), (50, 50), (50,), (50, 1), (1,), (1, 1)]
    assert sum(p.numel() for p in a) == 5252
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and not torch.equal(a[0], c[0])
    assert float(a[-1]) == pytest.approx(np.log(1e-3)) and all(float(p.abs().sum()) == 0 for p in a[1:-1:2])
    w = init_mlp_params(400, hidden=(300,), seed=0)[0]
    assert abs(float(w.var()) * 400 - 1.0) < 0.03 and float(w.abs().max()) <= 2.3 / 20


# ---------------------------------------------------------------- constructor checks
# tests/bayesian_neural_network/test_invalid_inputs.py:17-100

@pytest.mark.parametrize("kw", [dict(n_nets=0), dict(n_nets=1.5), dict(n_iters=-3), dict(n_iters="a"),
                                dict(batch_size=0), dict(batch_size=2.0), dict(sample_steps=0),
                                dict(burn_in_steps=-1), dict(burn_in_steps=1.0), dict(dtype=torch.int32)])
def test_invalid_constructor_arguments(kw):
    with pytest.raises(AssertionError):
        BayesianNeuralNetwork(**kw)


def test_unsupported_sampling_method_and_predict_before_train():
    for bad in (Sampler.RelativisticSGHMC, Sampler.SVGD, "SGHMC", 0):
        with pytest.raises(ValueError):
            BayesianNeuralNetwork(sampling_method=bad)
    with pytest.raises(ValueError):
        BayesianNeuralNetwork().predict(np.zeros((3, 1)))


# ---------------------------------------------------------------- batches (tests/test_data_batches.py:79-209)

@given(st.integers(max_value=0))
@settings(max_examples=20, deadline=None)
def test_invalid_batch_size(batch_size):
    X, y = np.zeros((10, 2)), np.zeros(10)
    with pytest.raises(AssertionError):
        next(generate_batches(X, y, Placeholder(), Placeholder(), batch_size=batch_size))


@pytest.mark.parametrize("seed", [-1, 2 ** 32, 1.5, "a"])
def test_invalid_seed(seed):
    X, y = np.zeros((10, 2)), np.zeros(10)
    with pytest.raises(AssertionError):
        next(generate_batches(X, y, Placeholder(), Placeholder(), batch_size=2, seed=seed))


@given(st.integers(1, 60), st.integers(1, 80), st.integers(1, 5), st.integers(0, 2 ** 32 - 1))
@settings(max_examples=40, deadline=None)
def test_batch_shapes_clamping_and_seed_reproducibility(n, batch_size, d, seed):
    rng = np.random.RandomState(0)
    X, y = rng.rand(n, d), rng.rand(n)
    xp, yp = Placeholder(), Placeholder()
    g1 = generate_batches(X, y, xp, yp, batch_size=batch_size, seed=seed)
    g2 = generate_batches(X, y, xp, yp, batch_size=batch_size, seed=seed)
    ref = np.random.RandomState(seed)
    eff = min(batch_size, n)
    for _ in range(4):
        b1, b2 = next(g1), next(g2)
        assert tuple(b1[xp].shape) == (eff, d) and tuple(b1[yp].shape) == (eff, 1)
        assert torch.equal(b1[xp], b2[xp]) and torch.equal(b1[yp], b2[yp])
        start = ref.randint(0, n - eff + 1)                   # the reference's window stream
        assert np.array_equal(b1[xp].numpy(), X[start:start + eff])
        assert np.array_equal(b1[yp].numpy().ravel(), y[start:start + eff])


def test_shuffled_batches_keep_rows_paired():
    X = np.arange(50, dtype=np.float64).reshape(50, 1)
    y = np.arange(50, dtype=np.float64)
    xp, yp = Placeholder(), Placeholder()
    g = generate_shuffled_batches(X, y, xp, yp, batch_size=10, seed=3)
    for _ in range(5):
        b = next(g)
        assert torch.equal(b[xp].ravel(), b[yp].ravel())
        assert sorted(b[xp].ravel().tolist()) == list(range(int(b[xp].min()), int(b[xp].min()) + 10))


# ---------------------------------------------------------------- end to end on the GPU

@pytest.mark.gpu
@pytest.mark.parametrize("method,dtype", [(Sampler.SGHMC, torch.float64), (Sampler.SGHMC, torch.float32),
                                           (Sampler.SGLD, torch.float32)])
def test_train_predict_sinc(gpu, method, dtype):
    """tests/bayesian_neural_network/test_train_predict.py:12-48: 100 sinc points, burn_in_steps=1000,
    n_nets=10, defaults otherwise -> test MSE < 0.1; individual predictions have n_nets rows (:75-115)."""
    rng = np.random.RandomState(1)
    X = rng.rand(100, 1)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    X_test = np.linspace(0, 1, 100)[:, None]
    y_test = np.sinc(X_test * 10 - 5).sum(axis=1)
    bnn = BayesianNeuralNetwork(session=gpu, sampling_method=method, dtype=dtype, burn_in_steps=1000, n_nets=10,
                                seed=1)
    bnn.train(X, y)
    assert bnn.is_trained and len(bnn.samples) == 10
    mean, var = bnn.predict(X_test)
    assert mean.shape == (100,) and var.shape == (100,) and (var >= 0).all()
    assert np.mean((y_test - mean) ** 2) < 0.1
    f_out, noise = bnn.predict(X_test, return_individual_predictions=True)
    assert f_out.shape == (10, 100) and noise.shape == (10, 100)
    assert bnn.sampler.use_hip_graph and bnn.sampler.n_iterations >= 1900
    # predict evaluates all kept networks in ONE batched pass: same numbers as the reference's net-by-net loop (:599-607)
    xn = (X_test - bnn.x_mean) / bnn.x_std
    one_by_one = np.stack([bnn.compute_network_output(params=net, input_data=xn) for net in bnn.samples])
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    assert np.allclose(one_by_one[:, :, 0] * bnn.y_std + bnn.y_mean, f_out, rtol=tol, atol=tol)
    assert np.allclose(np.exp(one_by_one[:, :, 1]) * bnn.y_std ** 2, noise, rtol=tol, atol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("normalize", [True, False])
def test_train_predict_sinc_with_and_without_normalisation(gpu, normalize):
    """The reference's own loop (tests/bayesian_neural_network/test_train_predict.py:27-48): the same training with
    ``normalize_input = normalize_output`` True and False, float64 like the reference, test MSE within 0.1."""
    rng = np.random.RandomState(7)
    X = rng.rand(100, 1)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    X_test = np.linspace(0, 1, 100)[:, None]
    y_test = np.sinc(X_test * 10 - 5).sum(axis=1)
    bnn = BayesianNeuralNetwork(session=gpu, dtype=torch.float64, burn_in_steps=1000, n_nets=10, seed=2,
                                normalize_input=normalize, normalize_output=normalize)
    assert not bnn.is_trained
    with pytest.raises(ValueError):
        bnn.predict(X_test)                                   # test_predict_before_train_error, :51-72
    bnn.train(X, y)
    assert bnn.is_trained
    mean, var = bnn.predict(X_test)
    assert np.mean((y_test - mean) ** 2) < 0.1


@pytest.mark.gpu
def test_train_is_seed_reproducible_and_graph_equals_eager(gpu):
    rng = np.random.RandomState(2)
    X, y = rng.rand(60, 2), rng.rand(60)

    def run(graph):
        bnn = BayesianNeuralNetwork(session=gpu, dtype=torch.float32, burn_in_steps=40, sample_steps=10, n_nets=5,
                                    seed=3, hidden=(16, 16))
        bnn.use_hip_graph = graph
        bnn.train(X, y)
        return bnn.predict(X[:7])
    (m1, v1), (m2, v2), (m3, v3) = run(True), run(True), run(False)
    assert np.array_equal(m1, m2) and np.array_equal(v1, v2)
    assert np.allclose(m1, m3, rtol=1e-5, atol=1e-6)


def test_cost_plan_marks_its_first_evaluation():
    """``BNNCost.auto_gemm_tuning`` acts on the first evaluation of a plan: the plan object carries the mark (slots class)."""
    from pysgmcmc_amd.models.bayesian_neural_network import _CostPlan, BNNCost, AUTO_GEMM_TUNING_MIN_PARAMS
    plan = _CostPlan(["addmm"], "head", {}, False, None, False)
    assert plan.fresh is True
    plan.fresh = False
    assert plan.fresh is False and _CostPlan(["addmm"], "head", {}, False, None, False).fresh is True
    cost = BNNCost(None, None, batch_size=4, n_examples=10)
    assert cost.auto_gemm_tuning is True and cost.gemm_tuning_applied is None and AUTO_GEMM_TUNING_MIN_PARAMS == 1000000


def test_auto_gemm_tuning_has_an_environment_opt_out(monkeypatch):
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost
    monkeypatch.setenv("PYSGMCMC_AMD_AUTO_GEMM_TUNING", "0")
    assert BNNCost(None, None, batch_size=4, n_examples=10).auto_gemm_tuning is False

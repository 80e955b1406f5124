"""Host-side sampler logic on CPU (no GPU): the reference's API surface and
per-step driver semantics, with the oracle standing in for the HIP kernels
(tests/oracle_shim.py). What the reference pins for this layer:
seed reproducibility (tests/samplers/sampler_testing.py:29-59), factory
behaviour and error texts (sampling.py doctests, tests/test_sampling.py),
iterator protocol / return convention (samplers/base_classes.py:226-310)."""
from itertools import islice

import numpy as np
import pytest
import torch

import oracle_shim
from oracle import sgmcmc_oracle as O
from pysgmcmc_amd.diagnostics.objective_functions import (
    banana_log_likelihood, gmm1_log_likelihood, to_negative_log_likelihood)
from pysgmcmc_amd.sampling import Sampler
from pysgmcmc_amd.samplers import RelativisticSGHMCSampler, SGHMCSampler, SGLDSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule, StepsizeSchedule

TARGETS = {
    "gmm1": (gmm1_log_likelihood, lambda: [torch.tensor(0., dtype=torch.float32)]),
    "banana": (banana_log_likelihood, lambda: [torch.tensor(0., dtype=torch.float32),
                                               torch.tensor(6., dtype=torch.float32)]),
}


def cost_function(log_likelihood_function):
    return to_negative_log_likelihood(log_likelihood_function)


@pytest.fixture
def shim(monkeypatch):
    return oracle_shim.install(monkeypatch)


@pytest.mark.parametrize("ctor,kwargs", [(SGHMCSampler, {}), (SGLDSampler, {}), (RelativisticSGHMCSampler, {})])
@pytest.mark.parametrize("target", ["gmm1", "banana"])
def test_seed_reproducibility(shim, ctor, kwargs, target):
    """The reference's own sampler test: two fresh chains, same seed -> same samples and costs."""
    fn, make = TARGETS[target]
    seed = int(np.random.randint(0, 2 ** 31 - 1))
    n_samples = int(np.random.randint(1, 100))

    def fresh_chain():
        sampler = ctor(params=make(), cost_fun=cost_function(fn), seed=seed, session="cpu",
                       dtype=torch.float32, **kwargs)
        return list(islice(sampler, n_samples))

    chain1, chain2 = fresh_chain(), fresh_chain()
    assert len(chain1) == n_samples
    for (s1, c1), (s2, c2) in zip(chain1, chain2):
        assert np.allclose(c1, c2)
        assert np.allclose(s1, s2)


def test_iterator_protocol_and_return_convention(shim):
    x = torch.tensor(1.0)
    sampler = SGHMCSampler(params=[x], cost_fun=lambda p: 0.5 * p[0] ** 2, session="cpu",
                           dtype=torch.float32, burn_in_steps=5, seed=1)
    assert iter(sampler) is sampler
    assert sampler.n_iterations == 0 and sampler.is_burning_in
    sample, cost = next(sampler)
    # single parameter -> bare array, not a list (base_classes.py:302-304)
    assert isinstance(sample, np.ndarray) and sample.shape == () and sample.dtype == np.float32
    # cost is U(theta_{t-1}): the cost at the INITIAL point, not at the returned sample
    assert np.isclose(cost, 0.5)
    assert sampler.n_iterations == 1
    theta1 = float(sample)
    sample2, cost2 = next(sampler)
    assert np.isclose(cost2, 0.5 * theta1 ** 2, rtol=1e-6)
    # the user's tensor aliases the arena: it holds the current sample
    assert np.isclose(float(x.detach()), float(sample2))
    burn = list(islice(sampler, 3))
    assert len(burn) == 3 and not sampler.is_burning_in
    # two parameters -> list in the parameters' shapes
    a, b = torch.zeros(2, 3), torch.zeros(4)
    s2 = SGLDSampler(params=[a, b], cost_fun=lambda p: (p[0] ** 2).sum() + (p[1] ** 2).sum(), session="cpu",
                     dtype=torch.float64, seed=3)
    sample, cost = next(s2)
    assert isinstance(sample, list) and [v.shape for v in sample] == [(2, 3), (4,)]
    assert sample[0].dtype == np.float64


def test_burn_in_switch_and_perpetual_adaptation(shim):
    mk = lambda burn: SGHMCSampler(params=[torch.zeros(3)], cost_fun=lambda p: (p[0] ** 2).sum(), session="cpu",
                                   dtype=torch.float32, burn_in_steps=burn, seed=0)
    s = mk(4)
    list(islice(s, 7))
    assert [c[1] for c in shim] == [True] * 4 + [False] * 3        # adapt for steps 0..3, frozen from step 4
    assert [c[3] for c in shim] == list(range(7))                    # Philox step index = n_iterations
    minv_frozen = [m.copy() for m in s.minv]
    next(s)
    assert all(np.array_equal(a, b) for a, b in zip(minv_frozen, s.minv))
    assert s.minv[0].shape == (3, 1)
    del shim[:]
    s0 = mk(0)                      # burn_in_steps == 0 -> never frozen (base_classes.py:449)
    list(islice(s0, 5))
    assert [c[1] for c in shim] == [True] * 5
    with pytest.raises(AssertionError):
        mk(3.5)


def test_stepsize_schedule_is_fed_and_updated(shim):
    class Decay(StepsizeSchedule):
        def __init__(self):
            super().__init__(0.1)
            self.t, self.updates = 0, []

        def __next__(self):
            self.t += 1
            return 0.1 / self.t

        def update(self, params, cost):
            self.updates.append((np.asarray(params).copy(), float(cost)))
    sched = Decay()
    s = SGHMCSampler(params=[torch.tensor(2.0)], cost_fun=lambda p: p[0] ** 2, stepsize_schedule=sched,
                     session="cpu", dtype=torch.float64, seed=0)
    out = list(islice(s, 3))
    assert np.allclose([c[2] for c in shim], [0.1, 0.05, 0.1 / 3])
    assert len(sched.updates) == 3 and np.isclose(sched.updates[0][1], 4.0)
    assert np.allclose(sched.updates[-1][0], out[-1][0])


def test_sgld_forwards_schedule_unless_strict(shim):
    mk = lambda: SGLDSampler(params=[torch.tensor(0.0)], cost_fun=lambda p: p[0] ** 2,
                             stepsize_schedule=ConstantStepsizeSchedule(0.5), session="cpu", seed=0)
    fixed, strict = mk(), mk()
    strict.strict_reference_quirks = True          # reference bug sgld.py:96-100: schedule dropped (per instance)
    next(fixed)
    assert shim[-1][2] == 0.5 and not fixed.strict_reference_quirks
    next(strict)
    assert shim[-1][2] == 0.01 and strict.strict_reference_quirks
    next(fixed)                                    # side by side in one process
    assert shim[-1][2] == 0.5
    strict.strict_reference_quirks = False
    next(strict)
    assert shim[-1][2] == 0.5
    # the factory's samplers take the attribute the same way (it is not a constructor keyword)
    from pysgmcmc_amd.sampling import Sampler
    s = Sampler.get_sampler(Sampler.SGLD, params=[torch.tensor(0.0)], cost_fun=lambda p: p[0] ** 2,
                            stepsize_schedule=ConstantStepsizeSchedule(0.5), session="cpu", seed=0)
    s.strict_reference_quirks = True
    next(s)
    assert shim[-1][2] == 0.01


def test_batches_are_fed_through_placeholders(shim):
    from pysgmcmc_amd.data_batches import Placeholder, generate_batches
    X = np.arange(40, dtype=np.float64).reshape(20, 2)
    y = np.arange(20, dtype=np.float64)
    xp, yp = Placeholder(dtype=torch.float64), Placeholder(dtype=torch.float64)
    seen = []

    def cost_fun(params):
        seen.append((xp.value.clone(), yp.value.clone()))
        return ((xp.value @ params[0] - yp.value) ** 2).mean()
    s = SGHMCSampler(params=[torch.zeros(2, 1, dtype=torch.float64)], cost_fun=cost_fun,
                     batch_generator=generate_batches(X, y, xp, yp, batch_size=5, seed=3),
                     session="cpu", seed=0, scale_grad=20.0)
    list(islice(s, 4))
    rng = np.random.RandomState(3)                 # the reference's window stream, data_batches.py:104-118
    for bx, by in seen:
        start = rng.randint(0, 20 - 5 + 1)
        assert np.array_equal(bx.numpy(), X[start:start + 5]) and np.array_equal(by.numpy().ravel(), y[start:start + 5])
        assert by.shape == (5, 1)
    assert s._next_batch().keys() == {xp, yp}
    assert SGHMCSampler(params=[torch.zeros(1)], cost_fun=lambda p: p[0].sum(), session="cpu")._next_batch() == {}


def test_constructor_assertions_like_the_reference():
    ok = dict(params=[torch.zeros(1)], cost_fun=lambda p: p[0].sum(), session="cpu")
    with pytest.raises(AssertionError):
        SGHMCSampler(**dict(ok, seed=1.5))
    with pytest.raises(AssertionError):
        SGHMCSampler(**dict(ok, cost_fun=3))
    with pytest.raises(AssertionError):
        SGHMCSampler(**dict(ok, batch_generator=[1, 2]))
    with pytest.raises(AssertionError):
        SGHMCSampler(**dict(ok, stepsize_schedule=0.01))
    with pytest.raises(AssertionError):
        SGHMCSampler(**dict(ok, dtype=torch.int32))
    s = SGHMCSampler(**dict(ok, dtype=np.float32))
    assert s.dtype == np.float32 and s.arena.dtype == torch.float32
    # default dtype is float64 like the reference (base_classes.py:25)
    assert SGHMCSampler(**ok).arena.dtype == torch.float64


def test_factory_classes_defaults_and_error_texts(shim):
    params = [torch.tensor(0.)]
    cost_fun = lambda params: sum(p.sum() for p in params)
    for method, cls in ((Sampler.SGHMC, SGHMCSampler), (Sampler.SGLD, SGLDSampler),
                        (Sampler.RelativisticSGHMC, RelativisticSGHMCSampler)):
        s = Sampler.get_sampler(method, params=[torch.tensor(0.)], cost_fun=cost_fun, dtype=torch.float32,
                                session="cpu")
        assert type(s) is cls and s.dtype == torch.float32
    s = Sampler.get_sampler(Sampler.SGHMC, params=params, cost_fun=cost_fun, session="cpu")
    assert s.dtype == torch.float64 and s.burn_in_steps == 3000 and s.mdecay == 0.05
    assert Sampler.get_sampler(Sampler.RelativisticSGHMC, params=[torch.tensor(0.)], cost_fun=cost_fun,
                               session="cpu").stepsize_schedule.initial_value == 0.001
    with pytest.raises(ValueError) as e:
        Sampler.get_sampler(Sampler.SGHMC, dtype=torch.float32)
    assert str(e.value) == ("sampling.Sampler.get_sampler: params was not overwritten as sampler argument in "
                            "`sampler_args` and does not have any default value in SGHMCSampler.__init__"
                            "Please pass an explicit value for this parameter.")
    with pytest.raises(ValueError) as e:
        Sampler.get_sampler(Sampler.SGLD, unknown_argument=None, params=params, cost_fun=cost_fun)
    assert "'SGLDSampler' does not take any parameter with name 'unknown_argument'" in str(e.value)
    assert str(e.value).endswith("-params\n-cost_fun\n-batch_generator\n-stepsize_schedule\n-burn_in_steps\n-A\n"
                                 "-scale_grad\n-session\n-dtype\n-seed")
    with pytest.raises(ValueError) as e:           # SVGD takes `particles`, not `params` (svgd.py:24)
        Sampler.get_sampler(Sampler.SVGD, params=params, cost_fun=cost_fun)
    assert "'SVGDSampler' does not take any parameter with name 'params'" in str(e.value)
    assert str(e.value).endswith("-particles\n-cost_fun\n-batch_generator\n-stepsize_schedule\n-alpha\n"
                                 "-fudge_factor\n-session\n-dtype\n-seed")
    s = Sampler.get_sampler(Sampler.SVGD, particles=[torch.zeros(2), torch.ones(2)], cost_fun=lambda p: (p ** 2).sum(),
                            session="cpu")
    assert type(s).__name__ == "SVGDSampler" and s.stepsize_schedule.initial_value == 0.1
    assert s.alpha == 0.9 and s.fudge_factor == 1e-6 and s.dtype == torch.float64
    assert [m.value for m in Sampler] == ["SGHMC", "RelativisticSGHMC", "SGLD", "SVGD"]
    assert Sampler.is_supported(Sampler.SGLD) and not Sampler.is_supported(Sampler.RelativisticSGHMC)


def test_sample_formats_and_checkpoint_resume(shim):
    mk = lambda: SGHMCSampler(params=[torch.zeros(5), torch.ones(2, 2)],
                              cost_fun=lambda p: (p[0] ** 2).sum() + (p[1] ** 2).sum(),
                              session="cpu", dtype=torch.float32, seed=11, burn_in_steps=3)
    s = mk()
    s.sample_format = "view"
    sample, cost = next(s)
    assert isinstance(sample[0], torch.Tensor) and isinstance(cost, torch.Tensor)
    assert sample[0].data_ptr() == s.arena.row("theta").data_ptr()       # zero-copy view of the arena
    s.sample_format = "device"
    sample, _ = next(s)
    assert sample[0].data_ptr() != s.arena.row("theta").data_ptr()
    s.sample_format = "numpy"
    list(islice(s, 3))
    state = s.state_dict()
    tail = [smp for smp, _ in islice(s, 4)]
    s2 = mk()
    s2.load_state_dict(state)
    tail2 = [smp for smp, _ in islice(s2, 4)]
    for a, b in zip(tail, tail2):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    with pytest.raises(ValueError):
        s.sample_format = "bogus"
        next(s)


def test_injected_noise_matches_golden_trajectory(shim):
    """Sampler driver + autograd gradients reproduce the committed oracle trajectory (fp64: tight)."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectories.npz"))
    for key, ctor, kw in (("sghmc|banana|float64|0.01|5", SGHMCSampler, dict(burn_in_steps=5)),
                          ("sgld|banana|float64|0.01|5", SGLDSampler, dict(burn_in_steps=5)),
                          ("sghmc|gmm1|float64|0.1|0", SGHMCSampler, dict(burn_in_steps=0))):
        fn, make = TARGETS[key.split("|")[1]]
        eps = float(key.split("|")[3])
        xi = d[key + "|xi"]
        s = ctor(params=[p.double() for p in make()], cost_fun=cost_function(fn), session="cpu",
                 stepsize_schedule=ConstantStepsizeSchedule(eps), dtype=torch.float64, seed=0, **kw)
        s.noise_source = lambda step, n: xi[step]
        for t, (sample, cost) in enumerate(islice(s, 100)):
            got = np.array([float(v) for v in sample]) if isinstance(sample, list) else np.array([float(sample)])
            assert np.allclose(got, d[key + "|theta"][t], rtol=1e-9, atol=1e-12), (key, t)
            assert np.isclose(float(cost), d[key + "|cost"][t], rtol=1e-9, atol=1e-12)


def test_relativistic_initial_momentum_law():
    from pysgmcmc_amd.samplers.relativistic_sghmc import _sample_relativistic_momentum
    p = _sample_relativistic_momentum(1.0, 1.0, 200000, seed=1).numpy()
    assert abs(p.mean()) < 0.02 and abs(p.std() - 1.6430) < 0.02     # sd of exp(-sqrt(p^2+1)) law
    p2 = _sample_relativistic_momentum(1.0, 1.0, 1000, seed=1).numpy()
    assert np.array_equal(p[:0], p2[:0]) and len(p2) == 1000
    with pytest.raises(AssertionError):
        _sample_relativistic_momentum(1, 1.0, 3)


def test_svgd_host_logic_matches_oracle_and_fixes_the_sign(shim):
    """SVGDSampler through the CPU shim: next() returns (list of particles, vector of costs), the step is the
    oracle's, particles of a standard normal target spread out with the fixed sign and collapse with the
    reference's (quirk Q10)."""
    import pysgmcmc_amd.samplers.svgd as svgd_mod
    from pysgmcmc_amd.samplers import SVGDSampler
    rng = np.random.RandomState(3)
    x0 = rng.normal(size=(10, 2))

    def run(strict, steps=200):
        s = SVGDSampler(particles=[torch.tensor(r) for r in x0], cost_fun=lambda p: 0.5 * (p ** 2).sum(),
                        session="cpu", dtype=torch.float64)
        s.strict_reference_quirks = strict
        assert s.repulsion_sign == (1 if strict else -1)
        assert iter(s) is s and s.n_particles == 10 and s.particle_dim == 2
        X, H = x0.copy(), np.zeros_like(x0)
        for t in range(steps):
            sample, cost = next(s)
            assert isinstance(sample, list) and len(sample) == 10 and sample[0].shape == (2,)
            assert cost.shape == (10,)
            np.testing.assert_allclose(cost, 0.5 * (X ** 2).sum(axis=1), rtol=1e-12)
            O.svgd_step(X, X.copy(), H, 0.1, 0.9, 1e-6, 1.0 if strict else -1.0)
            np.testing.assert_allclose(np.stack(sample), X, rtol=1e-12, atol=1e-14)
        return np.stack(sample)

    fixed = run(False)
    strict = run(True)
    assert 0.5 < fixed.std(axis=0).min() and fixed.std(axis=0).max() < 1.2
    assert strict.std(axis=0).max() < 0.2
    with pytest.raises(AssertionError):
        SVGDSampler(particles=[torch.zeros(2), torch.zeros(3)], cost_fun=lambda p: p.sum(), session="cpu")

    # batched cost function: one call with the [n, d] matrix, same trajectory
    batched = lambda P: 0.5 * (P ** 2).sum(dim=1)
    batched.batched = True
    s = SVGDSampler(particles=[torch.tensor(r) for r in x0], cost_fun=batched, session="cpu", dtype=torch.float64)
    for _ in range(5):
        sample_b, cost_b = next(s)
    s2 = SVGDSampler(particles=[torch.tensor(r) for r in x0], cost_fun=lambda p: 0.5 * (p ** 2).sum(),
                     session="cpu", dtype=torch.float64)
    for _ in range(5):
        sample_l, cost_l = next(s2)
    np.testing.assert_allclose(np.stack(sample_b), np.stack(sample_l), rtol=1e-13)
    K, kg = s.svgd_kernel()
    K_ref, kg_ref, _, _ = O.svgd_kernel(np.stack(sample_b))
    np.testing.assert_allclose(K.numpy(), K_ref, rtol=1e-13)
    np.testing.assert_allclose(kg.numpy(), kg_ref, rtol=1e-12, atol=1e-14)


def test_svgd_checkpoint_resume(shim):
    """state_dict / load_state_dict carry the particles and the running squared updates: a resumed SVGD run
    continues exactly like the uninterrupted one."""
    from pysgmcmc_amd.samplers import SVGDSampler
    x0 = np.random.RandomState(8).normal(size=(6, 3))
    mk = lambda: SVGDSampler(particles=[torch.tensor(r) for r in x0], cost_fun=lambda p: 0.5 * (p ** 2).sum(),
                             session="cpu", dtype=torch.float64)
    full = mk()
    for _ in range(9):
        ref, _ = next(full)
    first = mk()
    for _ in range(4):
        next(first)
    state = first.state_dict()
    assert "historical_grad" in state["arena"]
    resumed = mk()
    resumed.load_state_dict(state)
    assert resumed.n_iterations == 4
    for _ in range(5):
        got, _ = next(resumed)
    np.testing.assert_array_equal(np.stack(got), np.stack(ref))


def test_arena_rebind_and_param_alignment():
    """FlatArena.rebind moves a chain's state into caller-provided memory (several chains back to back in one
    allocation) and keeps the parameters aliased; param_align pads every parameter's start."""
    from pysgmcmc_amd.arena import FlatArena
    a, b = torch.arange(5.0), torch.arange(6.0).reshape(2, 3) + 10
    arena = FlatArena([a, b], ("V",), torch.float32, "cpu")
    arena.row("V").fill_(3.0)
    pool = torch.full((2 * arena.storage.numel(),), -1.0)
    arena.rebind(pool[arena.storage.numel():])
    assert arena.storage.data_ptr() == pool[arena.storage.numel():].data_ptr()
    assert a.data_ptr() == arena.row("theta").data_ptr() and torch.equal(b, torch.arange(6.0).reshape(2, 3) + 10)
    assert torch.all(arena.row("V") == 3.0) and torch.all(pool[:arena.storage.numel()] == -1.0)
    a.data.add_(1.0)                                     # writes through to the arena row
    assert torch.equal(arena.row("theta")[:5], torch.arange(5.0) + 1)
    assert arena.grad_views[1].data_ptr() == arena.row("grad")[5:].data_ptr()
    padded = FlatArena([torch.zeros(5), torch.zeros(70), torch.zeros(3)], (), torch.float32, "cpu", param_align=64)
    assert padded.offsets == [0, 64, 192] and padded.n == 256


def test_get_sampler_import_missing():
    """The reference's tests/test_sampling.py:8-23: a method the enum accepts but `get_sampler` has no import for raises
    ValueError (the reference's message, format placeholder and all)."""
    from pysgmcmc_amd.sampling import Sampler
    with pytest.raises(ValueError, match="missing an `import` statement"):
        Sampler.get_sampler("NEW_SAMPLER")


def test_every_public_name_of_the_reference_packages_imports_from_the_namesake():
    """Drop-in surface: every name in the ``__all__`` of ``pysgmcmc.samplers`` / ``.diagnostics`` / ``.models`` (minus ``BaseModel``,
    out of scope) and the public names of ``pysgmcmc.sampling`` import from the ``pysgmcmc_amd`` namesake. The name lists are fixture
    data written by tests/golden/make_reference_api_names.py."""
    import importlib
    import json
    import os
    names = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_api_names.json")))
    assert names["models"] == ["BaseModel", "BayesianNeuralNetwork", "log_variance_prior_log_like", "weight_prior_log_like"]
    assert names["tensor_utils"] == ["vectorize", "unvectorize", "median", "safe_divide", "safe_sqrt", "pdist", "squareform", "uninitialized_params"]
    for package in ("samplers", "diagnostics", "models", "sampling", "tensor_utils", "stepsize_schedules", "data_batches"):
        module = importlib.import_module("pysgmcmc_amd." + package)
        for name in names[package]:
            if (package, name) == ("models", "BaseModel"):
                continue
            assert hasattr(module, name), "pysgmcmc_amd.%s lacks %s" % (package, name)
            if hasattr(module, "__all__") and package in ("samplers", "diagnostics", "models", "tensor_utils"):
                assert name in module.__all__, "pysgmcmc_amd.%s.__all__ lacks %s" % (package, name)
    from pysgmcmc_amd.models import weight_prior_log_like, log_variance_prior_log_like       # the import a user of the reference writes
    assert callable(weight_prior_log_like) and callable(log_variance_prior_log_like)

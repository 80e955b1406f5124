"""SURVEY 8(a) row a5 -- initial relativistic momenta (``pysgmcmc/samplers/relativistic_sghmc.py:143-223``).

The reference calls ``arspy.ars.adaptive_rejection_sampling`` (un-vendored, unpinned). oracle/ars_oracle.py restates
Gilks & Wild's derivative-free sampler with the reference's call (a=-10, b=10, unbounded domain, the log-density of
``:208-216``). ARS is exact, so parity is distributional:
  oracle ARS  ~  quadrature CDF of the law  ~  the product's inverse-CDF sampler (host and device).
Tolerance: Kolmogorov-Smirnov at the stated sample sizes, p > 1e-3 with FIXED seeds (deterministic tests).
"""
import numpy as np
import pytest
import torch
from scipy import stats

from oracle import ars_oracle as A
from pysgmcmc_amd.samplers import RelativisticSGHMCSampler
from pysgmcmc_amd.samplers.relativistic_sghmc import _sample_relativistic_momentum
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

P_MIN = 1e-3


@pytest.fixture
def shim(monkeypatch):
    """CPU stand-in for the step kernels (tests/oracle_shim.py): host-logic tests only."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_shim
    return oracle_shim.install(monkeypatch)


def _cdf(m, c, p_max):
    p, cdf = A.relativistic_cdf_table(m, c, p_max=p_max)
    return lambda v: np.interp(v, p, cdf)


@pytest.mark.parametrize("m,c,p_max", [(1.0, 1.0, 60.0), (2.0, 0.5, 250.0), (0.5, 3.0, 40.0)])
def test_ars_oracle_samples_the_reference_law(m, c, p_max):
    info = {}
    x = np.array(A.adaptive_rejection_sampling(A.relativistic_logpdf(m, c), -10.0, 10.0, (-np.inf, np.inf), 20000,
                                               seed=11, stats=info))
    assert len(x) == 20000 and np.isfinite(x).all()
    assert stats.kstest(x, _cdf(m, c, p_max)).pvalue > P_MIN
    # adaptive: the hull tightens, so almost every proposal is accepted and few density evaluations are needed
    assert info["proposals"] < 1.02 * 20000 and info["evaluations"] < 200
    # symmetric law, variance from the quadrature table
    p, cdf = A.relativistic_cdf_table(m, c, p_max=p_max)
    pdf = np.gradient(cdf, p)
    var = np.trapezoid(pdf * p * p, p)
    assert abs(x.mean()) < 4 * np.sqrt(var / len(x)) and abs(x.var() / var - 1) < 0.05


def test_ars_oracle_reference_call_shape_and_seed():
    # the reference's doctest (relativistic_sghmc.py:192-195) and seed reproducibility
    a = A.sample_relativistic_momentum(m=1.0, c=1.0, n_params=10, seed=3)
    b = A.sample_relativistic_momentum(m=1.0, c=1.0, n_params=10, seed=3)
    assert len(a) == 10 and a == b
    assert A.sample_relativistic_momentum(m=1.0, c=1.0, n_params=10, seed=4) != a
    with pytest.raises(AssertionError):
        A.sample_relativistic_momentum(m=1, c=1.0, n_params=1)            # floats only, :203-204
    # bounded domain: draws stay inside
    x = A.adaptive_rejection_sampling(A.relativistic_logpdf(1.0, 1.0), -1.0, 2.0, (-1.5, 2.5), 2000, seed=0)
    assert min(x) >= -1.5 and max(x) <= 2.5
    # a log-density that does not fall at the right start point is refused for an unbounded domain
    with pytest.raises(ValueError):
        A.adaptive_rejection_sampling(lambda p: -(p - 50.0) ** 2, -10.0, 10.0, (-np.inf, np.inf), 1, seed=0)


def test_product_host_sampler_matches_ars_oracle_in_law():
    ars = np.array(A.sample_relativistic_momentum(m=1.0, c=1.0, n_params=20000, seed=5))
    got = _sample_relativistic_momentum(m=1.0, c=1.0, n_params=100000, seed=7).numpy()
    assert stats.ks_2samp(got, ars).pvalue > P_MIN
    assert stats.kstest(got, _cdf(1.0, 1.0, 60.0)).pvalue > P_MIN


def test_strict_mode_is_one_momentum_per_parameter_tensor(shim):
    params = [torch.tensor(0.0), torch.tensor([6.0]), torch.tensor(1.0)]
    mk = lambda ps: RelativisticSGHMCSampler(params=ps, cost_fun=lambda p: sum((q ** 2).sum() for q in p),
                                             stepsize_schedule=ConstantStepsizeSchedule(0.001), session="cpu",
                                             dtype=torch.float64, seed=9)
    s = mk([p.clone() for p in params])
    default = s.arena.row("p").clone()
    s.strict_reference_quirks = True
    strict = s.arena.row("p").clone()
    assert strict.numel() == len(params) and s.strict_reference_quirks
    # reference semantics: len(params) host draws with the sampler's seed, one per tensor
    want = _sample_relativistic_momentum(m=1.0, c=1.0, n_params=3, seed=9)
    assert torch.equal(strict, want)
    s2 = mk([p.clone() for p in params])
    s2.strict_reference_quirks = True
    assert torch.equal(s2.arena.row("p"), strict)                          # seed-reproducible
    s.strict_reference_quirks = False
    assert torch.equal(s.arena.row("p"), default)
    next(s)
    with pytest.raises(AssertionError):
        s.strict_reference_quirks = True                                   # initial momenta only
    # tensors with more than one element are undefined in the reference (quirk Q5): refused in strict mode
    big = mk([torch.zeros(3)])
    with pytest.raises(AssertionError):
        big.strict_reference_quirks = True


@pytest.mark.gpu
def test_device_sampler_matches_ars_oracle_in_law():
    dev = torch.device("cuda:0")
    ars = np.array(A.sample_relativistic_momentum(m=1.0, c=1.0, n_params=20000, seed=5))
    for dtype in (torch.float32, torch.float64):
        got = _sample_relativistic_momentum(m=1.0, c=1.0, n_params=400000, seed=123, device=dev, dtype=dtype).cpu().numpy()
        assert stats.ks_2samp(got.astype(np.float64), ars).pvalue > P_MIN
        assert stats.kstest(got.astype(np.float64), _cdf(1.0, 1.0, 60.0)).pvalue > P_MIN
    # other constants, through the sampler: the momentum row of a 200 000-element chain
    s = RelativisticSGHMCSampler(params=[torch.zeros(200000, device=dev)], cost_fun=lambda p: (p[0] ** 2).sum(),
                                 mass=2.0, speed_of_light=0.5, session=dev, dtype=torch.float32, seed=77)
    p0 = s.arena.row("p").cpu().numpy().astype(np.float64)
    ars2 = np.array(A.sample_relativistic_momentum(m=2.0, c=0.5, n_params=20000, seed=6))
    assert stats.ks_2samp(p0, ars2).pvalue > P_MIN
    assert stats.kstest(p0, _cdf(2.0, 0.5, 250.0)).pvalue > P_MIN
    # strict mode on the device: scalar parameters, one host draw per tensor
    ps = [torch.tensor(0.0, device=dev), torch.tensor(6.0, device=dev)]
    s = RelativisticSGHMCSampler(params=ps, cost_fun=lambda p: p[0] ** 2 + p[1] ** 2, session=dev,
                                 dtype=torch.float32, seed=5)
    s.strict_reference_quirks = True
    want = _sample_relativistic_momentum(m=1.0, c=1.0, n_params=2, seed=5, dtype=torch.float32)
    assert torch.equal(s.arena.row("p").cpu(), want)
    next(s)
    assert torch.isfinite(s.arena.row("theta")).all()

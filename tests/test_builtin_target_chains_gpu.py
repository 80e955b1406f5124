"""The n-steps-per-launch toy path (``sgmcmc_toy_chains_*`` / ``BuiltinTargetChains``) against the public sampler API
and the CPU oracle, and BASELINE.json configs[0] (SGLD on the 2-D Gaussian mixture) against the oracle."""
from itertools import islice

import numpy as np
import pytest
import torch

from pysgmcmc_amd.diagnostics.objective_functions import (
    banana_log_likelihood, gmm1_log_likelihood, gmm2_log_likelihood, gmm2d_log_likelihood, to_negative_log_likelihood)
from pysgmcmc_amd.samplers import RelativisticSGHMCSampler, SGHMCSampler, SGLDSampler
from pysgmcmc_amd.samplers.builtin_target_chains import BuiltinTargetChains, builtin_target_of
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

pytestmark = pytest.mark.gpu

TARGETS = {"banana": (banana_log_likelihood, [0.0, 6.0]), "gmm1": (gmm1_log_likelihood, [0.0]),
           "gmm2": (gmm2_log_likelihood, [0.0]), "gmm2d": (gmm2d_log_likelihood, [0.5, -0.5])}


def _samplers(gpu, ctor, target, dt, n_chains, eps, **kw):
    fn, start = TARGETS[target]
    out = []
    for c in range(n_chains):
        params = [torch.tensor(v + 0.1 * c, dtype=dt, device=gpu) for v in start]      # scalar parameters, sampler_testing.py:14-18
        out.append(ctor(params=params, cost_fun=to_negative_log_likelihood(fn), stepsize_schedule=ConstantStepsizeSchedule(eps),
                        session=gpu, dtype=dt, seed=100 + c, **kw))
        out[-1].sample_format = "view"
    return out


@pytest.mark.parametrize("target", ["banana", "gmm1", "gmm2d"])
@pytest.mark.parametrize("ctor,kw", [(SGHMCSampler, dict(burn_in_steps=40)), (SGLDSampler, dict(burn_in_steps=40)),
                                     (RelativisticSGHMCSampler, {})])
def test_one_launch_equals_stepping_the_samplers(gpu, ctor, kw, target):
    """120 steps of 5 chains in ONE launch leave every sampler where 120 ``next(sampler)`` calls leave it (f64: the
    analytic and the autograd gradient agree to rounding), across the burn-in -> frozen switch; the kept trace equals
    the yielded samples; afterwards both sets of samplers continue identically through the public API."""
    dt = torch.float64
    eps = 0.05 if ctor is not RelativisticSGHMCSampler else 0.1
    a = _samplers(gpu, ctor, target, dt, 5, eps, **kw)
    b = _samplers(gpu, ctor, target, dt, 5, eps, **kw)
    assert all(torch.equal(x.arena.storage, y.arena.storage) for x, y in zip(a, b))     # same seeds -> same initial momenta
    as_list = lambda smp: smp if isinstance(smp, list) else [smp]                       # single parameter: the bare array
    ref = torch.stack([torch.stack([torch.stack([v.reshape(()) for v in as_list(smp)]).clone()
                                    for smp, _ in islice(s, 0, 120, 3)]) for s in a], dim=1)   # [kept, chain, dim]
    chains = BuiltinTargetChains(b)
    kept = chains.run(118, keep_every=3)                                                # what islice(s, 0, 120, 3) consumes
    assert kept.shape == ref.shape == (40, 5, chains.dim)
    assert torch.allclose(kept, ref, rtol=1e-9, atol=1e-10), (ctor.__name__, target, (kept - ref).abs().max())
    assert all(s.n_iterations == 118 for s in b)
    for s in b:
        next(s), next(s)
    for x, y in zip(a, b):
        assert x.n_iterations == y.n_iterations == 120
        assert torch.allclose(x.arena.storage, y.arena.storage, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("target", ["gmm2", "banana"])
def test_relativistic_toy_chains_equal_the_c_oracle(gpu, oracle, target):
    """f64 chains against oracle_rsghmc_toy_chain_f64 (same update chain, same Philox stream; the two noise streams
    differ by ~1e-12, which a chaotic chain amplifies: 2 000 steps on the mixture, 400 on the banana, 1e-7)."""
    eps = 0.3
    n_steps = 2000 if target != "banana" else 400
    ss = _samplers(gpu, RelativisticSGHMCSampler, target, torch.float64, 4, eps)
    theta0 = [s.arena.row("theta").cpu().numpy().copy() for s in ss]
    p0 = [s.arena.row("p").cpu().numpy().copy() for s in ss]
    kept = BuiltinTargetChains(ss).run(n_steps, keep_every=7).cpu().numpy()
    for c, s in enumerate(ss):
        want = oracle.c_rsghmc_toy_chain(target, theta0[c], p0[c], eps, n_steps, 7, first_step=0, seed=s._philox_seed)
        assert np.allclose(kept[:, c, :], want, rtol=1e-7, atol=1e-7), (target, c, np.abs(kept[:, c, :] - want).max())
        assert np.allclose(s.arena.row("theta").cpu().numpy(), theta0[c], rtol=1e-7, atol=1e-7)   # the oracle advanced theta0 in place


def test_builtin_target_lookup_and_refusals(gpu):
    assert builtin_target_of(to_negative_log_likelihood(banana_log_likelihood))[0] == 1
    with pytest.raises(ValueError, match="not the negative log likelihood of a built-in"):
        builtin_target_of(lambda p: (p[0] ** 2).sum())
    a = _samplers(gpu, SGLDSampler, "gmm1", torch.float32, 2, 0.01, burn_in_steps=5)
    b = _samplers(gpu, SGLDSampler, "gmm1", torch.float32, 1, 0.02, burn_in_steps=5)
    with pytest.raises(ValueError, match="share hyper-parameters"):
        BuiltinTargetChains(a + b)
    from pysgmcmc_amd import kernels
    th = torch.zeros(3, 2, device=gpu)
    with pytest.raises(Exception, match="dim"):
        kernels.toy_chains(1, 0, [0, 1, 1], th, None, th.clone(), th.clone(), th.clone(), th.clone(), (0.01, 1.0, 1.0),
                           torch.zeros(3, dtype=torch.int64, device=gpu), 0, 10, 5)


def _gmm2d_grad(x, centers):
    """d cost / d x of the 2-D mixture in float32, numpy (cost = -logsumexp_i[-0.5 |x - c_i|^2] + const)."""
    d = x[None, :].astype(np.float64) - centers
    t = -0.5 * (d * d).sum(1)
    w = np.exp(t - t.max())
    w /= w.sum()
    return (w[:, None] * d).sum(0)


def test_config0_sgld_on_the_2d_mixture_against_the_oracle(gpu, oracle):
    """BASELINE.json configs[0]: SGLD on the 2-D Gaussian mixture with the reference's defaults (A = 1,
    scale_grad = 1, burn_in_steps = 3000, eps = 0.01 -- pysgmcmc/samplers/sgld.py:32-35), f32, through the public API with
    injected noise: all 3 500 steps (burn-in, the switch at 3 000, frozen steps) bit-equal to ``oracle.c_sgld_step`` fed
    the gradients the kernel consumed -- theta, the statistics and the preconditioner."""
    x = torch.tensor([0.3, -0.2], dtype=torch.float32)
    s = SGLDSampler(params=[x], cost_fun=to_negative_log_likelihood(gmm2d_log_likelihood), session=gpu,
                    dtype=torch.float32, seed=1)
    s.sample_format = "view"
    n_steps = 3500
    xi = np.random.default_rng(7).normal(size=(n_steps, 2)).astype(np.float32)
    xi_dev = torch.from_numpy(xi).to(gpu)
    s.noise_source = lambda step, n: xi_dev[step]
    st = oracle.CState(s.arena.row("theta").cpu().numpy(), np.float32)
    assert (s.burn_in_steps, s.A, s.scale_grad, s.stepsize_schedule.initial_value) == (3000, 1.0, 1.0, 0.01)
    centers = np.array([[-5.0, 0.0], [0.0, 0.0], [5.0, 0.0]])
    for t in range(n_steps):
        adapt = s._adapting
        before = s.arena.row("theta").cpu().numpy().copy()
        next(s)
        grad = s.arena.row("grad").cpu().numpy()
        if t % 500 == 0:                          # the autograd gradient is the analytic one
            assert np.allclose(grad, _gmm2d_grad(before, centers), rtol=2e-5, atol=1e-6)
        oracle.c_sgld_step(st, grad, 0.01, 1.0, 1.0, adapt, xi[t])
        if t % 50 == 0 or 2990 <= t <= 3010 or t == n_steps - 1:
            for name in ("theta", "tau", "g", "v_hat", "minv"):
                assert np.array_equal(s.arena.row(name).cpu().numpy(), getattr(st, name)), (name, t)
    assert not s._adapting and s.n_iterations == n_steps


def test_config0_sgld_visits_every_mode_with_the_right_moments(gpu):
    """... and the long run, at a stepsize that hops (eps = 0.1, 64 chains x 1e6 steps in one launch after the
    3 000-step burn-in; every 20th state kept): every chain visits all three modes; pooled over chains the mode weights
    are 1/3 each and every mode has the target's moments (mean = centre, unit variance in x and y)."""
    n_chains = 64
    ss = []
    for c in range(n_chains):
        x = torch.tensor([0.0, 0.0], dtype=torch.float32)
        s = SGLDSampler(params=[x], cost_fun=to_negative_log_likelihood(gmm2d_log_likelihood),
                        stepsize_schedule=ConstantStepsizeSchedule(0.1), burn_in_steps=3000, session=gpu, dtype=torch.float32,
                        seed=10 + c)
        ss.append(s)
    chains = BuiltinTargetChains(ss)
    chains.run(3000, keep=False)
    assert not ss[0]._adapting
    kept = chains.run(1_000_000, keep_every=20).double()                 # [50 000, 64, 2]
    assert torch.isfinite(kept).all()
    centers = torch.tensor([-5.0, 0.0, 5.0], dtype=torch.float64, device=gpu)
    mode = (kept[:, :, 0:1] - centers).abs().argmin(dim=2)               # nearest centre
    per_chain = torch.stack([(mode == k).double().mean(dim=0) for k in range(3)])       # [3, chains]
    assert (per_chain > 0.02).all(), per_chain.min()                     # every chain spent time in every mode
    w = per_chain.mean(dim=1).cpu().numpy()
    assert np.allclose(w, 1.0 / 3.0, atol=0.04), w
    for k in range(3):
        sel = kept[mode == k]
        mx, my = sel[:, 0].mean().item(), sel[:, 1].mean().item()
        vx, vy = sel[:, 0].var().item(), sel[:, 1].var().item()
        assert abs(mx - centers[k].item()) < 0.1 and abs(my) < 0.05, (k, mx, my)
        # the nearest-centre cut truncates the x tails of neighbouring modes: variance a little below 1 in x
        assert 0.85 < vx < 1.1 and 0.9 < vy < 1.15, (k, vx, vy)

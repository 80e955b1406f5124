"""The fused small-model path (csrc/sgmcmc_bnn_fused.hip): whole SGHMC steps of the reference's default
3x50 BNN in one launch. Parity bar: same windows, same Philox stream and the same update operator as the
GEMM-based path, so chains agree to the rounding of the matrix products; chunking and multi-chain
launches are bit-exact re-arrangements of the same computation."""
import os
from itertools import islice

import numpy as np
import pytest
import torch

from pysgmcmc_amd.data_batches import Placeholder, generate_batches
from pysgmcmc_amd.models.bayesian_neural_network import BayesianNeuralNetwork, BNNCost, init_mlp_params
from pysgmcmc_amd.samplers import SGHMCSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _chain(gpu, dt, seed=5, hidden=(50, 50, 50), n_in=1, burn=6, X=None, y=None, batch=20):
    rng = np.random.RandomState(1)
    if X is None:
        X = rng.rand(100, n_in)
        y = np.sinc(X * 10 - 5).sum(axis=1)
    xp, yp = Placeholder(dtype=dt, device=gpu), Placeholder(dtype=dt, device=gpu)
    params = init_mlp_params(n_in, hidden=hidden, seed=3, dtype=dt, device=gpu)
    s = SGHMCSampler(params=params, cost_fun=BNNCost(xp, yp, batch_size=batch, n_examples=X.shape[0]),
                     batch_generator=generate_batches(X, y, xp, yp, batch, seed=1),
                     stepsize_schedule=ConstantStepsizeSchedule(0.01), burn_in_steps=burn, mdecay=0.05,
                     scale_grad=float(X.shape[0]), session=gpu, dtype=dt, seed=seed)
    s.sample_format = "view"
    return s


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_fused_steps_track_the_gemm_path(gpu, dt):
    a, b = _chain(gpu, dt), _chain(gpu, dt)
    assert b.fused_bnn_available()
    costs_a = torch.stack([c.reshape(()).clone() for _, c in islice(a, 14)])   # the cost buffer is reused every step
    costs_b = b.fused_bnn_steps(14)
    tol = 2e-4 if dt == torch.float32 else 1e-9
    ta, tb = a.arena.row("theta"), b.arena.row("theta")
    assert float((ta - tb).abs().max()) <= tol * float(ta.abs().max())
    assert torch.allclose(costs_a, costs_b, rtol=1e-4 if dt == torch.float32 else 1e-9)
    assert float((a.arena.row("minv") - b.arena.row("minv")).abs().max()) <= tol * float(a.arena.row("minv").abs().max())
    assert b.n_iterations == 14 and not b.is_burning_in
    # the chain continues seamlessly on the per-step path
    next(a); next(b)
    assert float((a.arena.row("theta") - b.arena.row("theta")).abs().max()) <= 2 * tol * float(ta.abs().max())


def test_chunking_is_bit_exact(gpu):
    a, b, c = (_chain(gpu, torch.float32) for _ in range(3))
    a.fused_bnn_steps(12)
    b.fused_bnn_steps(5); b.fused_bnn_steps(7)
    for _ in range(12):
        c.fused_bnn_steps(1)
    for other in (b, c):
        for row in ("theta", "V", "minv", "tau", "g", "v_hat"):
            assert torch.equal(a.arena.row(row), other.arena.row(row)), row


@pytest.mark.parametrize("dtname", ["float32", "float64"])
def test_golden_bnn_trajectory_through_the_fused_kernel(gpu, dtname):
    """The committed 12-step sinc-BNN trajectory (fp64 CPU gradients, injected noise, seed-matched windows)."""
    from pysgmcmc_amd import kernels
    d = np.load(os.path.join(GOLDEN, "bnn_trajectory.npz"))
    dt = torch.float32 if dtname == "float32" else torch.float64
    P = d["theta0"].size
    rows = {k: torch.zeros(P, dtype=dt, device=gpu) for k in ("theta", "V", "grad", "tau", "g", "v_hat", "minv")}
    rows["theta"].copy_(torch.tensor(d["theta0"], dtype=dt))
    for k in ("tau", "g", "v_hat", "minv"):
        rows[k].fill_(1.0)
    nrng = np.random.default_rng(4321)
    xi = torch.tensor(np.stack([nrng.normal(size=P).astype(dtname) for _ in range(12)]), dtype=dt, device=gpu)
    X = torch.tensor(d["X"], dtype=dt, device=gpu).contiguous()
    y = torch.tensor(d["y"], dtype=dt, device=gpu).contiguous()
    starts = torch.tensor(d["starts"], dtype=torch.int32, device=gpu)
    costs = torch.empty(12, dtype=dt, device=gpu)
    tol = 2e-4 if dt == torch.float32 else 1e-9
    for t in range(12):                                       # one step per launch: compare every step
        kernels.bnn_fused_sghmc_steps(rows["theta"], rows["V"], rows["grad"], rows["tau"], rows["g"], rows["v_hat"],
                                      rows["minv"], [1, 50, 50, 50, 1], X, y, starts[t:t + 1].contiguous(), 20,
                                      20, 100, 1.0, 1e-6, 0.01, 0.01, 100.0, 0.05, t, 1, 6, 0, costs[t:t + 1],
                                      xi=xi[t:t + 1].contiguous())
        want = d[dtname + "|theta"][t]
        assert np.abs(rows["theta"].cpu().numpy() - want).max() <= tol * np.abs(want).max(), t
    assert np.allclose(costs.cpu().numpy(), d[dtname + "|cost"], rtol=1e-4 if dt == torch.float32 else 1e-9)


def test_many_chains_in_one_launch(gpu):
    """blockIdx = chain: 5 chains in one launch == 5 single-chain launches with seed_base + c (bit-exact)."""
    from pysgmcmc_amd import kernels
    sizes, P, n_steps, B = [2, 16, 16, 1], 2 * 16 + 16 + 16 * 16 + 16 + 16 + 1 + 1, 9, 8
    stride = ((P + 63) // 64) * 64
    g = torch.Generator(device=gpu).manual_seed(0)
    X = torch.randn(64, 2, device=gpu, generator=g)
    y = torch.randn(64, device=gpu, generator=g)
    theta0 = torch.randn(5, stride, device=gpu, generator=g) * 0.3
    starts = torch.randint(0, 64 - B + 1, (5, n_steps), device=gpu, generator=g, dtype=torch.int32)

    def fresh(n):
        r = {k: torch.zeros(n * stride, device=gpu) for k in ("V", "grad")}
        r.update({k: torch.ones(n * stride, device=gpu) for k in ("tau", "g", "v_hat", "minv")})
        return r
    multi = fresh(5)
    multi["theta"] = theta0.reshape(-1).clone()
    cm = torch.empty(5 * n_steps, device=gpu)
    args = (B, 8, 64, 1.0, 1e-6, 0.01, 0.02, 64.0, 0.05, 0, n_steps, 4)
    kernels.bnn_fused_sghmc_steps(multi["theta"], multi["V"], multi["grad"], multi["tau"], multi["g"], multi["v_hat"],
                                  multi["minv"], sizes, X, y, starts.reshape(-1).contiguous(), *args, 100, cm,
                                  n_chains=5, chain_stride=stride)
    for c in range(5):
        one = fresh(1)
        one["theta"] = theta0[c].clone()
        c1 = torch.empty(n_steps, device=gpu)
        kernels.bnn_fused_sghmc_steps(one["theta"], one["V"], one["grad"], one["tau"], one["g"], one["v_hat"],
                                      one["minv"], sizes, X, y, starts[c].contiguous(), *args, 100 + c, c1,
                                      n_chains=1, chain_stride=stride)
        assert torch.equal(multi["theta"][c * stride:c * stride + P], one["theta"][:P])
        assert torch.equal(cm[c * n_steps:(c + 1) * n_steps], c1)
    assert not torch.equal(multi["theta"][:P], multi["theta"][stride:stride + P])


def test_bnn_train_uses_the_fused_path(gpu):
    rng = np.random.RandomState(1)
    X = rng.rand(100, 1)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    Xt = np.linspace(0, 1, 100)[:, None]
    yt = np.sinc(Xt * 10 - 5).sum(axis=1)
    res = {}
    for fused in (True, False):
        bnn = BayesianNeuralNetwork(session=gpu, dtype=torch.float32, burn_in_steps=1000, n_nets=10, seed=1)
        bnn.use_fused_steps = fused
        bnn.train(X, y)
        assert bnn.used_fused_steps == fused and len(bnn.samples) == 10
        m, v = bnn.predict(Xt)
        res[fused] = (m, bnn.sampler.n_iterations)
        assert np.mean((yt - m) ** 2) < 0.1
    assert res[True][1] == res[False][1]                       # same number of sampler iterations
    assert np.abs(res[True][0] - res[False][0]).max() < 0.2    # chaotic divergence over 2 000 steps, same posterior


def test_fused_chain_group_equals_individual_chains(gpu):
    """FusedBNNChains: 6 SGHMC chains re-homed in one allocation and advanced by ONE launch per chunk are, bit
    for bit, the 6 chains advanced one by one; each member stays a working sampler afterwards."""
    from pysgmcmc_amd.samplers.fused_chains import FusedBNNChains
    rng = np.random.RandomState(1)
    X = rng.rand(100, 1)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    kw = dict(hidden=(50, 50, 50), batch_size=20, seed=40, dtype=torch.float32, device=gpu, burn_in_steps=7)
    group = FusedBNNChains.for_dataset(X, y, 6, **kw)
    solo = FusedBNNChains.for_dataset(X, y, 6, **kw).samplers       # same construction, stepped individually
    assert group.n_chains == 6 and group.theta().shape == (6, 5252)
    costs = torch.cat([group.steps(5), group.steps(9)], dim=1)      # burn-in switch inside the second chunk
    for c, s in enumerate(solo):
        c1 = torch.cat([s.fused_bnn_steps(5), s.fused_bnn_steps(9)])
        assert torch.equal(group.samplers[c].arena.row("theta"), s.arena.row("theta"))
        assert torch.equal(group.samplers[c].arena.row("minv"), s.arena.row("minv"))
        assert torch.equal(costs[c], c1)
        assert torch.equal(group.theta()[c], s.arena.row("theta"))
    assert group.n_iterations == 14 and not group.samplers[0].is_burning_in
    assert not torch.equal(group.theta()[0], group.theta()[1])
    # thinned snapshots of all chains -> R-hat across the chains (finite, and > 1 this early in the run)
    from pysgmcmc_amd.diagnostics.sampler_diagnostics import gelman_rubin_from_chains
    snaps = group.collect(6, every=10)
    assert snaps.shape == (6, 6, 5252) and group.n_iterations == 14 + 60
    assert torch.equal(snaps[:, -1], group.theta())
    rhat = gelman_rubin_from_chains(snaps)
    assert rhat.shape == (5252,) and torch.isfinite(rhat).all() and float(rhat.median()) > 1.0
    # members keep working as ordinary samplers on the shared memory (GEMM path), and so do their parameters
    a, b = group.samplers[2], solo[2]
    for _ in range(6):
        b.fused_bnn_steps(10)                             # bring the solo twin to iteration 74 as well
    a.sample_format = b.sample_format = "view"
    next(a); next(b)
    assert torch.allclose(a.arena.row("theta"), b.arena.row("theta"), rtol=1e-5, atol=1e-6)
    assert a.params[0].data_ptr() == a.arena.row("theta").data_ptr()
    with pytest.raises(ValueError):                       # the group insists on lockstep
        group.steps(1)
    # mismatched chains are refused
    other = FusedBNNChains.for_dataset(X, y, 2, **dict(kw, seed=90)).samplers
    with pytest.raises(ValueError):
        FusedBNNChains([group.samplers[0], other[0]])


def _sgld_chain(gpu, dt, seed=5, burn=6):
    from pysgmcmc_amd.samplers import SGLDSampler
    rng = np.random.RandomState(1)
    X = rng.rand(100, 1)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    xp, yp = Placeholder(dtype=dt, device=gpu), Placeholder(dtype=dt, device=gpu)
    params = init_mlp_params(1, hidden=(50, 50, 50), seed=3, dtype=dt, device=gpu)
    s = SGLDSampler(params=params, cost_fun=BNNCost(xp, yp, batch_size=20, n_examples=100),
                    batch_generator=generate_batches(X, y, xp, yp, 20, seed=1),
                    stepsize_schedule=ConstantStepsizeSchedule(1e-3), burn_in_steps=burn, A=1.0,
                    scale_grad=100.0, session=gpu, dtype=dt, seed=seed)
    s.sample_format = "view"
    return s


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_fused_sgld_steps_track_the_gemm_path(gpu, dt):
    """The fused kernel with K2's operator (preconditioned SGLD): same chain as next() up to the rounding of the
    matrix products; chunking is bit-exact; the burn-in switch happens inside a launch."""
    a, b, c = _sgld_chain(gpu, dt), _sgld_chain(gpu, dt), _sgld_chain(gpu, dt)
    assert b.fused_bnn_available()
    costs_a = torch.stack([cst.reshape(()).clone() for _, cst in islice(a, 14)])
    costs_b = b.fused_bnn_steps(14)
    costs_c = torch.cat([c.fused_bnn_steps(4), c.fused_bnn_steps(10)])
    tol = 2e-4 if dt == torch.float32 else 1e-9
    ta, tb = a.arena.row("theta"), b.arena.row("theta")
    assert float((ta - tb).abs().max()) <= tol * float(ta.abs().max())
    assert torch.allclose(costs_a, costs_b, rtol=5e-4 if dt == torch.float32 else 1e-9)
    assert float((a.arena.row("minv") - b.arena.row("minv")).abs().max()) <= tol * float(a.arena.row("minv").abs().max())
    assert torch.equal(tb, c.arena.row("theta")) and torch.equal(costs_b, costs_c)
    assert b.n_iterations == 14 and not b.is_burning_in
    next(a); next(b)
    assert float((a.arena.row("theta") - b.arena.row("theta")).abs().max()) <= 2 * tol * float(ta.abs().max())


def test_bnn_train_with_sgld_uses_the_fused_path_and_groups_work(gpu):
    from pysgmcmc_amd.sampling import Sampler
    from pysgmcmc_amd.samplers.fused_chains import FusedBNNChains
    rng = np.random.RandomState(1)
    X = rng.rand(100, 1)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    Xt = np.linspace(0, 1, 100)[:, None]
    yt = np.sinc(Xt * 10 - 5).sum(axis=1)
    bnn = BayesianNeuralNetwork(session=gpu, sampling_method=Sampler.SGLD, dtype=torch.float32, burn_in_steps=1000,
                                n_nets=10, seed=1)
    bnn.train(X, y)
    assert bnn.used_fused_steps and type(bnn.sampler).__name__ == "SGLDSampler"
    m, v = bnn.predict(Xt)
    assert np.isfinite(m).all() and np.mean((yt - m) ** 2) < 0.2
    # a group of SGLD chains == the chains one by one
    chains = [_sgld_chain(gpu, torch.float32, seed=70 + c) for c in range(3)]
    solo = [_sgld_chain(gpu, torch.float32, seed=70 + c) for c in range(3)]
    # the group wants ONE resident dataset: rebuild the members over chain 0's buffers
    for s in chains[1:]:
        s.batch_generator.x_dev, s.batch_generator.y_dev = chains[0].batch_generator.x_dev, chains[0].batch_generator.y_dev
    group = FusedBNNChains(chains)
    costs = group.steps(11)
    for c, s in enumerate(solo):
        assert torch.equal(s.fused_bnn_steps(11), costs[c])
        assert torch.equal(s.arena.row("theta"), group.samplers[c].arena.row("theta"))


def test_bnn_train_with_several_chains(gpu):
    """BayesianNeuralNetwork(n_chains=4): the chains advance together, every collection point yields one network
    per chain, so the same number of networks needs a quarter of the sampling iterations."""
    rng = np.random.RandomState(1)
    X = rng.rand(100, 1)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    Xt = np.linspace(0, 1, 100)[:, None]
    yt = np.sinc(Xt * 10 - 5).sum(axis=1)
    kw = dict(session=gpu, dtype=torch.float32, burn_in_steps=1000, sample_steps=100, n_nets=20, seed=1)
    one = BayesianNeuralNetwork(**kw)
    one.train(X, y)
    four = BayesianNeuralNetwork(n_chains=4, **kw)
    four.train(X, y)
    assert four.used_fused_steps and four.chains.n_chains == 4 and len(four.samples) == 20
    assert one.sampler.n_iterations == 1000 + 20 * 100 + 1 and four.sampler.n_iterations == 1000 + 5 * 100 + 1
    m, v = four.predict(Xt)
    assert np.mean((yt - m) ** 2) < 0.1
    # chain 0 is the single-chain run; the other chains are different draws
    assert torch.equal(four.samples[0][0], one.samples[0][0])
    assert not torch.equal(four.samples[0][0], four.samples[1][0])
    # a configuration the fused kernel does not take (here: switched off): the chains advance as concurrent streams
    # (ConcurrentChains), chain 0 is again the single-chain run of the same configuration
    kw2 = dict(kw, burn_in_steps=200, sample_steps=20, n_nets=6)
    solo = BayesianNeuralNetwork(**kw2)
    solo.use_fused_steps = False
    solo.train(X, y)
    duo = BayesianNeuralNetwork(n_chains=2, **kw2)
    duo.use_fused_steps = False
    duo.train(X, y)
    from pysgmcmc_amd.samplers import ConcurrentChains
    assert isinstance(duo.chains, ConcurrentChains) and not duo.used_fused_steps and len(duo.samples) == 6
    assert solo.sampler.n_iterations == 200 + 6 * 20 + 1 and duo.sampler.n_iterations == 200 + 3 * 20 + 1
    for a, b in zip(duo.samples[0], solo.samples[0]):
        assert torch.equal(a, b)
    assert torch.equal(duo.samples[2][0], solo.samples[1][0]) and not torch.equal(duo.samples[0][0], duo.samples[1][0])
    assert np.all(np.isfinite(duo.predict(Xt)[0]))
    with pytest.raises(ValueError, match="needs an AMD GPU"):
        BayesianNeuralNetwork(n_chains=2, **dict(kw, session="cpu")).train(X, y)

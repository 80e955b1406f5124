"""BASELINE.json configs[4] under ``-m gpu``: relativistic SGHMC and preconditioned SGLD on the 49 826 818-parameter BNN
(512-4864-4864-4864-1 + log-variance) with the burn-in stepsize ramp, THROUGH THE PUBLIC SAMPLER API.

Parity (bit-exact, f32): the sampler is driven with injected noise; windows of the arena (start, middle across a layer
boundary, the ragged end) are carried through the CPU oracle with the same gradients, the same per-step stepsize and the
same noise, and must stay bit-equal to the device state at every step (the pattern of
``test_hip_parity.py::test_large_array_64bit_indexing``, but over a trajectory and through ``next(sampler)``).
Also: the stepsize really ramps, the burn-in -> frozen switch happens at the right step, and Philox chains are
deterministic. Reference: ``pysgmcmc/samplers/relativistic_sghmc.py:120-140``, ``sgld.py:149-211``,
``stepsize_schedules.py:4-34``, ``base_classes.py:393-456``.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LAYERS = (512, 4864, 4864, 4864)
N_PARAMS = 49_826_818
BATCH, N_DATA = 256, 4096


def _chain(kind, dev, seed, burn):
    from pysgmcmc_amd.data_batches import Placeholder, generate_batches
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
    from pysgmcmc_amd.samplers import RelativisticSGHMCSampler, SGLDSampler
    from pysgmcmc_amd.stepsize_schedules import BurnInRampStepsizeSchedule
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(N_DATA, LAYERS[0], device=dev, generator=g)
    y = torch.randn(N_DATA, device=dev, generator=g)
    xp, yp = Placeholder(dtype=torch.float32, device=dev), Placeholder(dtype=torch.float32, device=dev)
    params = init_mlp_params(LAYERS[0], hidden=LAYERS[1:], seed=7, dtype=torch.float32, device=dev)
    cost = BNNCost(xp, yp, batch_size=BATCH, n_examples=N_DATA)
    common = dict(params=params, cost_fun=cost, batch_generator=generate_batches(X, y, xp, yp, batch_size=BATCH, seed=3),
                  stepsize_schedule=BurnInRampStepsizeSchedule(1e-4, 1e-3, burn_in_steps=burn),
                  session=dev, dtype=torch.float32, seed=seed)
    if kind == "sgld":
        s = SGLDSampler(A=1.0, scale_grad=float(N_DATA), burn_in_steps=burn, **common)
    else:
        s = RelativisticSGHMCSampler(mass=1.0, speed_of_light=1.0, D=1.0, Bhat=0.0, **common)
    s.sample_format = "view"
    assert s.arena.n == N_PARAMS
    return s


def _windows(s):
    n = s.arena.n
    b1 = s.arena.offsets[2]                        # first element of the second weight matrix: a layer boundary
    return [(0, 2003), (b1 - 1001, b1 + 1002), (n // 2 - 999, n // 2 + 1000), (n - 2050, n)]


def _host(t, a, b):
    return t[a:b].cpu().numpy().copy()


@pytest.mark.parametrize("kind", ["rsghmc", "sgld"])
def test_config4_trajectory_windows_bit_equal_to_oracle(gpu, oracle, kind):
    burn, steps = 3, 6
    s = _chain(kind, gpu, seed=11, burn=burn)
    g = torch.Generator(device=gpu).manual_seed(99)
    xi_buf = torch.empty(s.arena.n, device=gpu)
    s.noise_source = lambda step, n: xi_buf.normal_(generator=g)        # fresh injected noise every step
    wins = _windows(s)
    a = s.arena
    # oracle state of every window, initialised from the device state before the first step
    states = []
    for (lo, hi) in wins:
        st = oracle.CState(_host(a.row("theta"), lo, hi), np.float32)
        if kind == "rsghmc":
            st.p[:] = _host(a.row("p"), lo, hi)
        states.append(st)
    expected_eps = [1e-4 + (1e-3 - 1e-4) * (t / float(burn)) if t < burn else 1e-3 for t in range(steps)]
    seen_eps, phases = [], []
    for t in range(steps):
        adapting = bool(getattr(s, "_adapting", False))
        phases.append(adapting)
        sample, cost = next(s)
        seen_eps.append(float(s.epsilon))
        assert torch.isfinite(cost).item()
        for (lo, hi), st in zip(wins, states):
            grad = _host(a.row("grad"), lo, hi)                          # the gradient the kernel consumed
            xi = _host(xi_buf, lo, hi)
            if kind == "rsghmc":
                oracle.c_rsghmc_step(st, grad, seen_eps[-1], 1.0, 1.0, 1.0, 0.0, xi, grad_decay=s._grad_decay)
                names = ("theta", "p")
            else:
                oracle.c_sgld_step(st, grad, seen_eps[-1], 1.0, float(N_DATA), adapting, xi, grad_decay=s._grad_decay)
                names = ("theta", "minv") + (("tau", "g", "v_hat") if adapting else ())
            for name in names:
                got = _host(a.row(name), lo, hi)
                assert np.array_equal(got.view(np.uint32), getattr(st, name).view(np.uint32)), (
                    "%s step %d window [%d, %d) row %s differs from the oracle" % (kind, t, lo, hi, name))
    assert np.allclose(seen_eps, expected_eps, rtol=1e-12), seen_eps      # the stepsize really ramps, then stays
    assert len(set(seen_eps[:burn + 1])) == burn + 1 and seen_eps[burn] == seen_eps[-1] == 1e-3
    if kind == "sgld":
        assert phases == [True] * burn + [False] * (steps - burn)         # burn-in -> frozen switch, base_classes.py:393-456
        assert s._grad_decay > 0                                          # weight prior folded into the update kernel
        # frozen phase leaves the preconditioner statistics alone
        tau_before = _host(a.row("tau"), 0, 4096)
        next(s)
        assert np.array_equal(tau_before, _host(a.row("tau"), 0, 4096))
    assert torch.isfinite(a.row("theta")).all()


@pytest.mark.parametrize("kind", ["rsghmc", "sgld"])
def test_config4_philox_chains_are_deterministic(gpu, kind):
    """In-register Philox noise at 49.8 M parameters: same seed -> bit-equal chain (the reference's own criterion,
    tests/samplers/sampler_testing.py:55-59), another seed -> another chain; hipGraph stepping == eager stepping."""
    def run(seed, graph):
        s = _chain(kind, gpu, seed=seed, burn=2)
        s.use_hip_graph = graph
        costs = []
        for _ in range(4):
            _, c = next(s)
            costs.append(float(c))
        theta = s.arena.row("theta").clone()
        del s
        torch.cuda.empty_cache()
        return theta, costs
    t1, c1 = run(5, False)
    t2, c2 = run(5, True)                               # cost pipeline replayed from a hipGraph, update launched directly
    assert torch.equal(t1, t2) and c1 == c2
    del t2
    t3, _ = run(6, False)
    assert not torch.equal(t1, t3)
    # the two chains share theta_0 and gradients at step 0, so they differ only through the noise: about N(0, tiny)
    assert float((t1 - t3).abs().max()) > 0.0


# ------------------------------------------------------------------------------------------------------------------
# configs[2] (the headline workload): SGHMC on the 10 002 434-parameter BNN, through the public API
# ------------------------------------------------------------------------------------------------------------------

def test_config2_sghmc_trajectory_windows_bit_equal_to_oracle(gpu, oracle):
    """The bench workload (784-2048-2048-2048-1 + log-variance = 10 002 434 parameters, batch 256, eps 0.01, mdecay 0.05,
    scale_grad = N) stepped through ``next(sampler)`` with injected noise across the burn-in -> frozen switch: four windows
    of every state row stay bit-equal to the oracle (``sghmc.py:165-251``, ``base_classes.py:393-456``), in the eager mode
    and with the cost pipeline replayed from a hipGraph; the fused step statistics equal the recomputed sums."""
    from pysgmcmc_amd.data_batches import Placeholder, generate_batches
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
    from pysgmcmc_amd.samplers import SGHMCSampler
    from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
    layers, n_data, burn, steps = (784, 2048, 2048, 2048), 8192, 3, 7
    g0 = torch.Generator(device=gpu).manual_seed(0)
    X = torch.randn(n_data, layers[0], device=gpu, generator=g0)
    y = torch.randn(n_data, device=gpu, generator=g0)

    def chain():
        xp, yp = Placeholder(dtype=torch.float32, device=gpu), Placeholder(dtype=torch.float32, device=gpu)
        params = init_mlp_params(layers[0], hidden=layers[1:], seed=1000, dtype=torch.float32, device=gpu)
        s = SGHMCSampler(params=params, cost_fun=BNNCost(xp, yp, batch_size=BATCH, n_examples=n_data),
                         batch_generator=generate_batches(X, y, xp, yp, batch_size=BATCH, seed=0),
                         stepsize_schedule=ConstantStepsizeSchedule(0.01), mdecay=0.05, scale_grad=float(n_data),
                         burn_in_steps=burn, session=gpu, dtype=torch.float32, seed=1234)
        s.sample_format = "view"
        return s
    s = chain()
    assert s.arena.n == 10_002_434
    a = s.arena
    gx = torch.Generator(device=gpu).manual_seed(5)
    xi_buf = torch.empty(a.n, device=gpu)
    s.noise_source = lambda step, n: xi_buf.normal_(generator=gx)
    wins = _windows(s)
    states = [oracle.CState(_host(a.row("theta"), lo, hi), np.float32) for lo, hi in wins]
    for t in range(steps):
        adapting = s._adapting
        assert adapting == (t < burn)
        next(s)
        for (lo, hi), st in zip(wins, states):
            oracle.c_sghmc_step(st, _host(a.row("grad"), lo, hi), 0.01, float(n_data), 0.05, adapting, _host(xi_buf, lo, hi),
                                grad_decay=s._grad_decay)
            for name in ("theta", "V", "minv") + (("tau", "g", "v_hat") if adapting else ()):
                got = _host(a.row(name), lo, hi)
                assert np.array_equal(got.view(np.uint32), getattr(st, name).view(np.uint32)), (t, lo, name)
    st = s.stats
    assert np.isclose(st["theta_sq"], float((a.row("theta").double() ** 2).sum()), rtol=5e-7)
    assert np.isclose(st["momentum_sq"], float((a.row("V").double() ** 2).sum()), rtol=5e-7)
    assert np.isclose(st["minv_sum"], float(a.row("minv").double().sum()), rtol=5e-7)
    # Philox noise: graph-stepped chain == eager chain, bit for bit, across the switch
    e, g = chain(), chain()
    g.use_hip_graph = True
    for _ in range(steps):
        ce = next(e)[1]
        cg = next(g)[1]
        assert float(ce) == float(cg)
    for row in ("theta", "V", "minv", "tau"):
        assert torch.equal(e.arena.row(row), g.arena.row(row)), row

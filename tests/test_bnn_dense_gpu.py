"""``sgmcmc_bnn_dense_tanh_f32``: a hidden layer of the BNN forward pass (``pysgmcmc/models/bayesian_neural_network.py:30-56``)
as ONE launch -- fp32 matrix-core product with bias + tanh (and the output unit's partial dot products) as its epilogue -- against
an fp64 product, against the library path of ``BNNCost`` (GEMM + activation launch), and in the sampler's chain."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N", [(256, 2048, 2048), (256, 784, 2048), (32, 64, 64), (64, 80, 128), (96, 272, 192), (32, 1040, 64)])
def test_dense_tanh_equals_an_fp64_product(gpu, M, K, N):
    """Every K tail (K % 64 in {0, 16, 32, 48}), ragged tile counts (XCD map on and off), pitched operands; the output unit's
    partial dot products add up to the GEMV; two launches give the same bits (no atomics, fixed summation order)."""
    from pysgmcmc_amd import kernels
    g = torch.Generator(device=gpu).manual_seed(M + K + N)
    hbuf = torch.randn(M, K + 8, device=gpu, generator=g)
    h = hbuf[:, :K]                                              # row pitch K + 8: ld != K
    W = torch.randn(K, N, device=gpu, generator=g) / K ** 0.5
    b = torch.randn(N, device=gpu, generator=g) * 0.3
    w_next = torch.randn(N, device=gpu, generator=g)
    out = torch.full((M + 1, N), -7.0, device=gpu)
    parts = torch.zeros(N // 64, M, device=gpu)
    assert kernels.bnn_dense_tanh_fits(h, W, out[:M])
    kernels.bnn_dense_tanh(h, W, b, out[:M], w_next=w_next, dot_parts=parts)
    ref = torch.tanh(h.double() @ W.double() + b.double())
    assert (out[:M].double() - ref).abs().max().item() < 4e-6 and torch.all(out[M] == -7.0)
    assert torch.allclose(parts.double().sum(dim=0), out[:M].double() @ w_next.double(), rtol=0, atol=1e-4)
    out2, parts2 = torch.empty(M, N, device=gpu), torch.zeros_like(parts)
    kernels.bnn_dense_tanh(h, W, b, out2, w_next=w_next, dot_parts=parts2)
    assert torch.equal(out2, out[:M]) and torch.equal(parts2, parts)
    # without the dot product: same activations
    out3 = torch.empty(M, N, device=gpu)
    kernels.bnn_dense_tanh(h, W, b, out3)
    assert torch.equal(out3, out2)


def test_dense_tanh_refuses_what_it_cannot_take(gpu):
    from pysgmcmc_amd import kernels
    h, W = torch.randn(256, 2048, device=gpu), torch.randn(2048, 2048, device=gpu)
    out = torch.empty(256, 2048, device=gpu)
    assert kernels.bnn_dense_tanh_fits(h, W, out)
    assert not kernels.bnn_dense_tanh_fits(h[:20], W, out[:20])                    # the reference BNN's batch of 20
    assert not kernels.bnn_dense_tanh_fits(h.double(), W.double(), out.double())   # f32 only
    assert not kernels.bnn_dense_tanh_fits(h[:, :50].contiguous(), W[:50], out)    # K < 64
    assert not kernels.bnn_dense_tanh_fits(torch.randn(512, 2048, device=gpu), W, torch.empty(512, 2048, device=gpu))   # > one tile per CU
    assert not kernels.bnn_dense_tanh_fits(h[:, 1:65], W[:64], out)                # rows not 16-byte aligned
    with pytest.raises(ValueError, match="bnn_dense_tanh"):
        kernels.bnn_dense_tanh(h[:20], W, torch.randn(2048, device=gpu), out[:20])
    with pytest.raises(ValueError, match="tsq_parts"):                             # the slicing side job needs >= 16 tiles
        kernels.bnn_dense_tanh(h[:32, :64].contiguous(), W[:64, :64].contiguous(), torch.zeros(64, device=gpu),
                               torch.empty(32, 64, device=gpu), stats_workspace=torch.zeros(64, dtype=torch.float64, device=gpu),
                               tsq_parts=torch.zeros(16, dtype=torch.float64, device=gpu))


def _cost(gpu, fused, batch, hidden, n_in=64, seed=3):
    from pysgmcmc_amd.data_batches import Placeholder
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
    xp, yp = Placeholder(dtype=torch.float32, device=gpu), Placeholder(dtype=torch.float32, device=gpu)
    g = torch.Generator(device=gpu).manual_seed(seed)
    xp.feed(torch.randn(batch, n_in, device=gpu, generator=g))
    yp.feed(torch.randn(batch, 1, device=gpu, generator=g))
    params = init_mlp_params(n_in, hidden=hidden, seed=seed, dtype=torch.float32, device=gpu)
    for p in params[1:-1:2]:
        p.normal_(0.0, 0.2, generator=g)                         # non-zero biases
    cost = BNNCost(xp, yp, batch_size=batch, n_examples=1000)
    cost.fused_dense = fused
    return cost, params


@pytest.mark.parametrize("batch,hidden", [(256, (128, 128, 128)), (64, (128, 64)), (32, (64, 64, 64))])
def test_cost_path_with_fused_dense_layers_equals_the_library_path(gpu, batch, hidden):
    """BNNCost.fused_dense: same cost, mse and gradients as GEMM + activation launches to matrix-product rounding -- with the
    output unit's dot product and the sum(theta^2) slices riding in the last hidden layer's launch (256 x 128: 16 tiles) and
    without (fewer tiles: that layer keeps the library product + rowdot launch)."""
    from pysgmcmc_amd import kernels
    res = []
    for fused in (False, True):
        cost, params = _cost(gpu, fused, batch, hidden)
        n = sum(p.numel() for p in params)
        st = kernels.StepStats(n, gpu)                           # sum(theta^2) records as a step kernel leaves them
        flat = torch.cat([p.reshape(-1) for p in params])
        kernels.sghmc_step(flat.clone(), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu), None, None, None,
                           torch.ones(n, device=gpu), None, 0.0, 1.0, 0.0, False, xi=torch.zeros(n, device=gpu), stats=st,
                           opts=dict(theta_sq_only=True))
        grads = [torch.full_like(p, float("nan")) for p in params]
        c = cost.cost_and_grad(params, grads, theta_sumsq_partials=st.workspace)
        res.append((float(c), float(cost.last_mse), [g.clone() for g in grads]))
    (c0, m0, g0), (c1, m1, g1) = res
    assert abs(c1 - c0) <= 2e-6 * abs(c0) and abs(m1 - m0) <= 2e-6 * abs(m0)
    for a, b in zip(g0, g1):
        assert torch.isfinite(b).all() and float((a - b).abs().max()) <= 3e-5 * float(a.abs().max()) + 1e-9


def test_chain_with_fused_dense_layers_tracks_the_library_chain(gpu):
    """A 14-step SGHMC chain (burn-in switch inside) on a net whose layers all take the fused launch agrees with the chain on
    the library products to accumulated matrix-product rounding (f32 2e-4, the bar of the fused small-model kernel), in eager
    and hipGraph stepping; the fused chain is bit-reproducible."""
    from pysgmcmc_amd.data_batches import Placeholder, generate_batches
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
    from pysgmcmc_amd.samplers import SGHMCSampler
    from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
    rng = np.random.RandomState(0)
    X, y = rng.rand(2000, 64), rng.rand(2000)

    def chain(fused, graph):
        xp, yp = Placeholder(dtype=torch.float32, device=gpu), Placeholder(dtype=torch.float32, device=gpu)
        params = init_mlp_params(64, hidden=(128, 128, 128), seed=5, dtype=torch.float32, device=gpu)
        cost = BNNCost(xp, yp, batch_size=256, n_examples=2000)
        cost.fused_dense = fused
        s = SGHMCSampler(params=params, cost_fun=cost, batch_generator=generate_batches(X, y, xp, yp, batch_size=256, seed=2),
                         stepsize_schedule=ConstantStepsizeSchedule(0.01), burn_in_steps=5, scale_grad=2000.0, session=gpu,
                         dtype=torch.float32, seed=9)
        s.sample_format, s.use_hip_graph, s.collect_stats = "view", graph, "theta_sq"
        costs = [float(next(s)[1]) for _ in range(14)]
        return s.arena.row("theta").clone(), costs
    ref, cref = chain(False, False)
    for graph in (False, True):
        th, c = chain(True, graph)
        assert torch.allclose(th, ref, rtol=2e-4, atol=2e-5) and np.allclose(c, cref, rtol=2e-5)
    a, _ = chain(True, True)
    b, _ = chain(True, True)
    assert torch.equal(a, b)

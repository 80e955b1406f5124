"""``sgmcmc_bnn_dense_tanh_f32``: a hidden layer of the BNN forward pass (``pysgmcmc/models/bayesian_neural_network.py:30-56``)
as ONE launch -- fp32 matrix-core product with bias + tanh (and the output unit's partial dot products) as its epilogue -- and
``sgmcmc_bnn_dense_tanh_backward_f32``, the backward step through such a layer (delta W^T with tanh' and the bias gradient as the
epilogue), against fp64 products, against the library path of ``BNNCost`` (GEMM + activation / tanh' launches), and in the
sampler's chain."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


# (256, 4864, 4864): configs[4]'s hidden layer, 608 tiles = 2 rounds of full tiles + 192 HALF tiles on 256 compute units;
# (512, 2048, 2048): two full rounds; (64, 144, 8704) / (32, 80, 8320): half tiles with K tails, XCD map on / off for the halves
BIG_SHAPES = [(256, 4864, 4864), (512, 2048, 2048), (64, 144, 8704), (32, 80, 8320)]


@pytest.mark.parametrize("M,K,N", [(256, 2048, 2048), (256, 784, 2048), (32, 64, 64), (64, 80, 128), (96, 272, 192), (32, 1040, 64)] + BIG_SHAPES)
def test_dense_tanh_equals_an_fp64_product(gpu, M, K, N):
    """Every K tail (K % 64 in {0, 16, 32, 48}), ragged tile counts (XCD map on and off), pitched operands, more tiles than compute
    units (rounds of workgroups, a thin last round as half tiles); the output unit's partial dot products add up to the GEMV;
    two launches give the same bits (no atomics, fixed summation order)."""
    from pysgmcmc_amd import kernels
    g = torch.Generator(device=gpu).manual_seed(M + K + N)
    hbuf = torch.randn(M, K + 8, device=gpu, generator=g)
    h = hbuf[:, :K]                                              # row pitch K + 8: ld != K
    W = torch.randn(K, N, device=gpu, generator=g) / K ** 0.5
    b = torch.randn(N, device=gpu, generator=g) * 0.3
    w_next = torch.randn(N, device=gpu, generator=g)
    out = torch.full((M + 1, N), -7.0, device=gpu)
    n_parts = kernels.bnn_dense_tanh_dot_parts(M, N, gpu)
    cus = torch.cuda.get_device_properties(gpu).multi_processor_count
    tiles = (M // 32) * (N // 64)
    if tiles <= cus or tiles % cus == 0:
        assert n_parts == N // 64
    elif (M, N) in ((256, 4864), (64, 8704), (32, 8320)) and cus == 256:
        assert n_parts > N // 64                                 # the thin last round runs as half tiles
    parts = torch.zeros(n_parts, M, device=gpu)
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
    assert kernels.bnn_dense_tanh_fits(torch.randn(512, 2048, device=gpu), W, torch.empty(512, 2048, device=gpu))   # more than one tile per CU: rounds
    assert not kernels.bnn_dense_tanh_fits(h[:, 1:65], W[:64], out)                # rows not 16-byte aligned
    with pytest.raises(ValueError, match="bnn_dense_tanh"):
        kernels.bnn_dense_tanh(h[:20], W, torch.randn(2048, device=gpu), out[:20])
    with pytest.raises(ValueError, match="tsq_parts"):                             # the slicing side job needs >= 16 tiles
        kernels.bnn_dense_tanh(h[:32, :64].contiguous(), W[:64, :64].contiguous(), torch.zeros(64, device=gpu),
                               torch.empty(32, 64, device=gpu), stats_workspace=torch.zeros(64, dtype=torch.float64, device=gpu),
                               tsq_parts=torch.zeros(16, dtype=torch.float64, device=gpu))


@pytest.mark.parametrize("M,K,N", [(256, 2048, 2048), (256, 2048, 784 + 48), (32, 64, 64), (64, 80, 128), (96, 272, 192), (32, 1040, 64),
                                   (160, 64, 320)] + BIG_SHAPES)
def test_dense_tanh_backward_equals_an_fp64_product(gpu, M, K, N):
    """out = (delta W^T) (1 - act^2) and its column sums per 32-row tile: every K tail, one and several row tiles, XCD map on and
    off, pitched operands. The row tiles are added up -- in order -- by a small launch or by the NEXT backward launch on the
    side: same bits either way, and from launch to launch."""
    from pysgmcmc_amd import kernels
    g = torch.Generator(device=gpu).manual_seed(M + 3 * K + N)
    dbuf = torch.randn(M, K + 4, device=gpu, generator=g)
    delta = dbuf[:, :K]
    Wbuf = torch.randn(N, K + 12, device=gpu, generator=g) / K ** 0.5
    W = Wbuf[:, :K]                                              # [fan-in N of the layer above][its fan-out K], pitched
    act = torch.tanh(torch.randn(M, N, device=gpu, generator=g))
    bias = torch.randn(N, device=gpu, generator=g)
    out = torch.full((M + 1, N), -7.0, device=gpu)
    parts = torch.full((M // 32 + 1, N), -3.0, device=gpu)
    assert kernels.bnn_dense_tanh_backward_fits(delta, W, act, out[:M])
    kernels.bnn_dense_tanh_backward(delta, W, act, out[:M], colsum_parts=parts[:M // 32])
    ref = (delta.double() @ W.double().t()) * (1.0 - act.double() ** 2)
    scale = max(float(ref.abs().max()), 1.0)
    assert float((out[:M].double() - ref).abs().max()) < 4e-6 * scale and torch.all(out[M] == -7.0)
    tiles = out[:M].double().view(M // 32, 32, N).sum(1)         # the partial rows are sums of the fp32 outputs
    assert torch.allclose(parts[:M // 32].double(), tiles, rtol=0, atol=1e-4 * scale) and torch.all(parts[M // 32] == -3.0)
    colsum = torch.full((N + 4,), -5.0, device=gpu)
    kernels.colsum_finish(parts[:M // 32], colsum[:N])
    assert torch.allclose(colsum[:N].double(), ref.sum(0), rtol=0, atol=2e-5 * M ** 0.5 * scale) and torch.all(colsum[N:] == -5.0)
    c_beta = torch.empty(N, device=gpu)
    kernels.colsum_finish(parts[:M // 32], c_beta, bias=bias, beta=0.25)
    assert torch.allclose(c_beta, colsum[:N] + 0.25 * bias, rtol=1e-6, atol=1e-6 * scale)
    # the same sums as the side job of a following launch (which leaves its own partial rows elsewhere), bit for bit
    for beta, want in ((0.0, colsum[:N]), (0.25, c_beta)):
        o2, p2, c2 = torch.empty(M, N, device=gpu), torch.empty(M // 32, N, device=gpu), torch.full((N,), 9.0, device=gpu)
        kernels.bnn_dense_tanh_backward(delta, W, act, o2, colsum_parts=p2, finish=(parts[:M // 32], c2, bias, beta))
        assert torch.equal(o2, out[:M]) and torch.equal(p2, parts[:M // 32]) and torch.equal(c2, want)
    # a finish job wider than 64 columns per workgroup of the launch that carries it
    wide = torch.randn(3, 64 * (M // 32) * (N // 64) + 200, device=gpu, generator=g)
    cw = torch.empty(wide.shape[1], device=gpu)
    kernels.bnn_dense_tanh_backward(delta, W, act, torch.empty(M, N, device=gpu), finish=(wide, cw, None, 0.0))
    assert torch.equal(cw, (wide[0] + wide[1]) + wide[2])
    # without column sums: same outputs
    o3 = torch.empty(M, N, device=gpu)
    kernels.bnn_dense_tanh_backward(delta, W, act, o3)
    assert torch.equal(o3, out[:M])


def test_dense_tanh_backward_refuses_what_it_cannot_take(gpu):
    from pysgmcmc_amd import kernels
    d, W = torch.randn(256, 2048, device=gpu), torch.randn(2048, 2048, device=gpu)
    act, out = torch.zeros(256, 2048, device=gpu), torch.empty(256, 2048, device=gpu)
    parts = torch.zeros(8, 2048, device=gpu)
    assert kernels.bnn_dense_tanh_backward_fits(d, W, act, out)
    assert not kernels.bnn_dense_tanh_backward_fits(d[:20], W, act[:20], out[:20])
    assert not kernels.bnn_dense_tanh_backward_fits(d.double(), W.double(), act.double(), out.double())
    assert not kernels.bnn_dense_tanh_backward_fits(d[:, :48].contiguous(), W[:, :48].contiguous(), act, out)      # K < 64
    assert not kernels.bnn_dense_tanh_backward_fits(d, W[:50], act[:, :50].contiguous(), out[:, :50].contiguous())  # N % 64
    with pytest.raises(ValueError, match="bnn_dense_tanh_backward"):
        kernels.bnn_dense_tanh_backward(d[:20], W, act[:20], out[:20])
    with pytest.raises(ValueError, match="colsum_parts"):
        kernels.bnn_dense_tanh_backward(d, W, act, out, colsum_parts=torch.zeros(4, 2048, device=gpu))
    with pytest.raises(ValueError, match="other than this launch's own"):
        kernels.bnn_dense_tanh_backward(d, W, act, out, colsum_parts=parts, finish=(parts, torch.empty(2048, device=gpu), None, 0.0))
    with pytest.raises(ValueError, match="finish job"):
        kernels.bnn_dense_tanh_backward(d, W, act, out, finish=(torch.zeros(8, 2048, device=gpu), torch.empty(2048, device=gpu), None, 0.5))
    with pytest.raises(ValueError, match="finish job"):
        kernels.colsum_finish(torch.zeros(8, 100, device=gpu), torch.empty(64, device=gpu))


def test_window_gather_into_a_pitched_buffer(gpu):
    """The static feed buffer of BNNCost is pitched (a column of ones behind the data): the gather fills the data columns only."""
    from pysgmcmc_amd import kernels
    for dt, dim in ((torch.float32, 784), (torch.float32, 13), (torch.float64, 6)):
        X = torch.randn(500, dim, device=gpu, dtype=dt)
        y = torch.randn(500, device=gpu, dtype=dt)
        ext = torch.full((64, dim + 4), 7.0, device=gpu, dtype=dt)
        yb = torch.empty(64, device=gpu, dtype=dt)
        kernels.window_gather(X, y, 101, ext[:, :dim], yb)
        assert torch.equal(ext[:, :dim], X[101:165]) and torch.all(ext[:, dim:] == 7.0) and torch.equal(yb, y[101:165])


def _cost(gpu, fused, batch, hidden, n_in=64, seed=3, own_feed_buffer=False):
    from pysgmcmc_amd.data_batches import Placeholder
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
    xp, yp = Placeholder(dtype=torch.float32, device=gpu), Placeholder(dtype=torch.float32, device=gpu)
    g = torch.Generator(device=gpu).manual_seed(seed)
    x = torch.randn(batch, n_in, device=gpu, generator=g)
    yp.feed(torch.randn(batch, 1, device=gpu, generator=g))
    params = init_mlp_params(n_in, hidden=hidden, seed=seed, dtype=torch.float32, device=gpu)
    for p in params[1:-1:2]:
        p.normal_(0.0, 0.2, generator=g)                         # non-zero biases
    cost = BNNCost(xp, yp, batch_size=batch, n_examples=1000)
    cost.fused_layers = fused
    if own_feed_buffer:                                          # as the sampler's hipGraph modes feed x: pitched, ones behind the data
        buf = cost.static_feed_buffer(xp, x)
        assert buf is not None and buf.stride(0) == n_in + 4
        buf.copy_(x)
        xp.value = buf
    else:
        xp.feed(x)
    return cost, params


@pytest.mark.parametrize("fold_prior", [True, False])
@pytest.mark.parametrize("batch,hidden,own_feed_buffer", [(256, (128, 128, 128), False), (256, (128, 128, 128), True), (64, (128, 64), True),
                                                          (32, (64, 64, 64), False), (32, (64, 64, 64), True),
                                                          (20, (48, 40, 24), True)])     # nothing fits the fused launches: library products, bias row still there
def test_cost_path_with_fused_dense_layers_equals_the_library_path(gpu, batch, hidden, own_feed_buffer, fold_prior):
    """BNNCost.fused_layers: same cost, mse and gradients as GEMM + activation / tanh' launches to
    matrix-product rounding -- with the output unit's dot product and the sum(theta^2) slices riding in the last hidden layer's
    launch (256 x 128: 16 tiles) and without (fewer tiles: that layer keeps the library product + rowdot launch); with the first
    layer's bias gradient from the [x | 1]^T delta product (the cost function's own pitched feed buffer, gradients in one arena)
    and from the column sums a last small launch adds up; weight prior folded into the update or in the gradients."""
    from pysgmcmc_amd import kernels
    res = []
    for fused in (False, True):
        cost, params = _cost(gpu, fused, batch, hidden, own_feed_buffer=own_feed_buffer and fused)
        cost.fold_prior = fold_prior
        n = sum(p.numel() for p in params)
        flat = torch.cat([p.reshape(-1) for p in params])
        offs = np.cumsum([0] + [p.numel() for p in params])
        params = [flat[offs[k]:offs[k + 1]].view(p.shape) for k, p in enumerate(params)]      # one arena, as in the sampler
        st = kernels.StepStats(n, gpu)                           # sum(theta^2) records as a step kernel leaves them
        kernels.sghmc_step(flat.clone(), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu), None, None, None,
                           torch.ones(n, device=gpu), None, 0.0, 1.0, 0.0, False, xi=torch.zeros(n, device=gpu), stats=st,
                           opts=dict(theta_sq_only=True))
        gflat = torch.full((n,), float("nan"), device=gpu)
        grads = [gflat[offs[k]:offs[k + 1]].view(p.shape) for k, p in enumerate(params)]
        c = cost.cost_and_grad(params, grads, theta_sumsq_partials=st.workspace)
        res.append((float(c), float(cost.last_mse), [g.clone() for g in grads]))
    (c0, m0, g0), (c1, m1, g1) = res
    assert abs(c1 - c0) <= 2e-6 * abs(c0) and abs(m1 - m0) <= 2e-6 * abs(m0)
    for a, b in zip(g0, g1):
        assert torch.isfinite(b).all() and float((a - b).abs().max()) <= 3e-5 * float(a.abs().max()) + 1e-9


def test_chain_with_fused_dense_layers_tracks_the_library_chain(gpu):
    """A 14-step SGHMC chain (burn-in switch inside) on a net whose layers all take the fused launch agrees with the chain on
    the library products to accumulated matrix-product rounding (f32 2e-4, the bar of the fused small-model kernel), in eager
    and both hipGraph stepping modes -- which give the SAME bits as each other (eager steps are fed through the cost function's pitched
    buffer too); the fused chain is bit-reproducible."""
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
        cost.fused_layers = fused
        s = SGHMCSampler(params=params, cost_fun=cost, batch_generator=generate_batches(X, y, xp, yp, batch_size=256, seed=2),
                         stepsize_schedule=ConstantStepsizeSchedule(0.01), burn_in_steps=5, scale_grad=2000.0, session=gpu,
                         dtype=torch.float32, seed=9)
        s.sample_format, s.use_hip_graph, s.collect_stats = "view", graph, "theta_sq"
        costs = [float(next(s)[1]) for _ in range(14)]
        return s.arena.row("theta").clone(), costs
    ref, cref = chain(False, False)
    got = {}
    for graph in (False, True, "full"):
        got[graph], c = chain(True, graph)
        assert torch.allclose(got[graph], ref, rtol=2e-4, atol=2e-5) and np.allclose(c, cref, rtol=2e-5)
    # eager stepping feeds the cost function through the same pitched buffer as the graph modes: same arithmetic, same bits
    assert torch.equal(got[False], got[True]) and torch.equal(got[True], got["full"])
    b, _ = chain(True, True)
    assert torch.equal(got[True], b)


@pytest.mark.parametrize("name,batch,n_in,hidden,partials,want", [
    # every hidden layer on the fused launches, the output unit's dot product in the last one, bias gradient of layer 0 from its product
    ("all fused", 256, 64, (128, 128, 128), True,
     {"forward": ["dense_tanh", "dense_tanh", "dense_tanh+dot", "by rowdot"], "head": "head+last_layer_backward",
      "backward": {3: "in head launch", 2: "dense_tanh_backward", 1: "dense_tanh_backward"},
      "first_layer_bias_gradient": "from the [x | 1]^T delta product", "weight_gradients_in_one_batched_product": [1, 2]}),
    # the reference's own net (3 x 50, batch 20): nothing fits, library products + the small launches
    ("library only", 20, 1, (50, 50, 50), True,
     {"forward": ["mm+bias_tanh", "mm+bias_tanh", "mm+bias_tanh_rowdot", "by rowdot"], "head": "head+last_layer_backward",
      "backward": {3: "in head launch", 2: "mm+tanh_backward_colsum", 1: "mm+tanh_backward_colsum"},
      "first_layer_bias_gradient": "column sums", "weight_gradients_in_one_batched_product": [1, 2]}),
    # mixed by shape: the 80-wide layer stays on the library, its neighbours are fused
    ("mixed by shape", 256, 64, (128, 80, 128), True,
     {"forward": ["dense_tanh", "mm+bias_tanh", "dense_tanh+dot", "by rowdot"], "head": "head+last_layer_backward",
      "backward": {3: "in head launch", 2: "mm+tanh_backward_colsum", 1: "dense_tanh_backward"},
      "first_layer_bias_gradient": "from the [x | 1]^T delta product"}),
    # no statistics records from a step kernel (a bare cost_and_grad call): separate loss head, no dot product in the layer launch
    ("no partials", 256, 64, (128, 128, 128), False,
     {"forward": ["dense_tanh", "dense_tanh", "mm+bias_tanh_rowdot", "by rowdot"], "head": "head",
      "backward": {3: "last_layer_backward", 2: "dense_tanh_backward", 1: "dense_tanh_backward"},
      "first_layer_bias_gradient": "from the [x | 1]^T delta product", "weight_gradients_in_one_batched_product": [1, 2]}),
])
@pytest.mark.parametrize("fold_prior", [True, False])
def test_every_reachable_plan_against_autograd_and_the_oracle(gpu, name, batch, n_in, hidden, partials, want, fold_prior):
    """BNNCost decides its launch sequence once per configuration (``_plan``; consecutive hidden layers of one shape get their weight
    gradients from ONE strided batched product); each reachable kind of plan is checked here
    against autograd through the torch restatement of the cost and against the oracle's numpy NLL
    (pysgmcmc/models/bayesian_neural_network.py:365-388), to matrix-product rounding."""
    from oracle import sgmcmc_oracle as O
    from pysgmcmc_amd import kernels
    pad = n_in % 4 == 0
    cost, params = _cost(gpu, True, batch, hidden, n_in=n_in, own_feed_buffer=pad)
    cost.fold_prior = fold_prior
    n = sum(p.numel() for p in params)
    flat = torch.cat([p.reshape(-1) for p in params])
    offs = np.cumsum([0] + [p.numel() for p in params])
    params = [flat[offs[k]:offs[k + 1]].view(p.shape) for k, p in enumerate(params)]
    gflat = torch.full((n,), float("nan"), device=gpu)
    grads = [gflat[offs[k]:offs[k + 1]].view(p.shape) for k, p in enumerate(params)]
    ws = None
    if partials:
        st = kernels.StepStats(n, gpu)
        kernels.sghmc_step(flat.clone(), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu), None, None, None,
                           torch.ones(n, device=gpu), None, 0.0, 1.0, 0.0, False, xi=torch.zeros(n, device=gpu), stats=st,
                           opts=dict(theta_sq_only=True))
        ws = st.workspace
    got_plan = cost.plan_summary(params, grads, ws)
    if not pad:
        want = dict(want, first_layer_bias_gradient="column sums")
    if "weight_gradients_in_one_batched_product" in want:        # small layers: the batched product stays on the library
        want = dict(want, batched_weight_gradient_arithmetic="library fp32 product")
    assert got_plan == want, (name, got_plan)
    c = float(cost.cost_and_grad(params, grads, theta_sumsq_partials=ws))
    if fold_prior:                                               # the update kernel adds coef * theta: put it back for the comparison
        gflat += cost.grad_theta_coef * flat
    # autograd through the torch restatement
    leaves = [p.detach().clone().double().requires_grad_(True) for p in params]
    X, Y = cost.x_placeholder.value.double(), cost.y_placeholder.value.double()
    nll, _ = cost.negative_log_likelihood(leaves, X, Y)
    auto = torch.autograd.grad(nll, leaves)
    nll_v = float(nll.detach())
    assert abs(c - nll_v) <= 2e-6 * abs(nll_v)
    for g, a in zip(grads, auto):
        assert torch.isfinite(g).all() and float((g.double() - a).abs().max()) <= 3e-5 * float(a.abs().max()) + 1e-9, name
    # the oracle's numpy restatement
    c_ref, g_ref = O.bnn_cost_and_grad([p.cpu().numpy().astype(np.float64) for p in params], X.cpu().numpy(), Y.cpu().numpy(),
                                       batch, 1000)
    assert abs(c - c_ref) <= 2e-6 * abs(c_ref)
    for g, w in zip(grads, g_ref):
        assert np.allclose(g.cpu().numpy(), w, rtol=3e-4, atol=3e-5 * float(np.abs(w).max()) + 1e-9), name


def test_plan_keeps_multi_round_layers_on_the_library_unless_forced(gpu):
    """configs[4]'s 256 x 4864 layers are 608 tiles on 256 compute units: the fused launches take them (rounds + half tiles,
    tested above against fp64) but the step is faster on the library's products (profiles/r05_dense_rounds.txt), so the default
    plan leaves them there; ``fused_layers = "all"`` forces the fused launches, and both give the same cost and gradients to
    matrix-product rounding."""
    from pysgmcmc_amd import kernels
    cus = torch.cuda.get_device_properties(gpu).multi_processor_count
    wide = 64 * (cus // 8 + 12)                                  # 8 row tiles x (cus / 8 + 12) column tiles > cus
    res = {}
    for mode in (True, "all"):
        cost, params = _cost(gpu, mode, 256, (wide, wide), n_in=64, own_feed_buffer=True)
        n = sum(p.numel() for p in params)
        flat = torch.cat([p.reshape(-1) for p in params])
        offs = np.cumsum([0] + [p.numel() for p in params])
        params = [flat[offs[k]:offs[k + 1]].view(p.shape) for k, p in enumerate(params)]
        gflat = torch.full((n,), float("nan"), device=gpu)
        grads = [gflat[offs[k]:offs[k + 1]].view(p.shape) for k, p in enumerate(params)]
        st = kernels.StepStats(n, gpu)
        kernels.sghmc_step(flat.clone(), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu), None, None, None,
                           torch.ones(n, device=gpu), None, 0.0, 1.0, 0.0, False, xi=torch.zeros(n, device=gpu), stats=st,
                           opts=dict(theta_sq_only=True))
        plan = cost.plan_summary(params, grads, st.workspace)
        if mode is True:
            assert plan["forward"][:2] == ["mm+bias_tanh", "mm+bias_tanh_rowdot"] and plan["backward"][1] == "mm+tanh_backward", plan
        else:
            assert plan["forward"][:2] == ["dense_tanh", "dense_tanh+dot"] and plan["backward"][1] == "dense_tanh_backward", plan
        c = float(cost.cost_and_grad(params, grads, theta_sumsq_partials=st.workspace))
        res[mode] = (c, gflat.clone())
    (c0, g0), (c1, g1) = res[True], res["all"]
    assert abs(c0 - c1) <= 2e-6 * abs(c0)
    assert torch.isfinite(g1).all() and float((g0 - g1).abs().max()) <= 3e-5 * float(g0.abs().max()) + 1e-9


# ---- the batched weight gradients on the bf16 matrix pipe (csrc/sgmcmc_bnn_gw.hip)

@pytest.mark.parametrize("M,nA,nB,count", [(256, 2048, 2048, 2), (256, 4864, 4864, 2), (256, 785, 2048, 1), (64, 200, 72, 3), (16, 128, 136, 1),
                                           (32, 1000, 1304, 2)])
def test_gw_planes_equals_an_fp64_product_at_least_as_well_as_the_library(gpu, M, nA, nB, count):
    """``gW_z = h_z^T delta_z`` from three exact bf16 planes per operand and six bf16 MFMA products: against fp64 the maximal and the
    rms error are no larger than those of the library's fp32 product on the same operands (VERDICT r05 item 1's accuracy bar) and
    within the 4e-6 bound of the layer launches; cut tiles (features that are no multiple of 128), several products per launch,
    pitched gradient slices; bit-reproducible; nothing written outside the slices. Operands: tanh outputs and deltas whose rows
    spread over six decades (what a backward pass produces)."""
    from pysgmcmc_amd import kernels
    g = torch.Generator(device=gpu).manual_seed(M + nA + nB)
    h = torch.tanh(1.5 * torch.randn(count, M, nA, device=gpu, generator=g))
    d = torch.randn(count, M, nB, device=gpu, generator=g) * 0.05 * 10 ** (-6 * torch.rand(count, M, 1, device=gpu, generator=g))
    ldc = nB + 8
    cbuf = torch.full((count, nA + 1, ldc), -7.0, device=gpu)
    outs = [cbuf[z, :nA, :nB] for z in range(count)]
    pa = torch.empty(count * kernels.bnn_planes_bytes(M, nA), dtype=torch.uint8, device=gpu)
    pb = torch.empty(count * kernels.bnn_planes_bytes(M, nB), dtype=torch.uint8, device=gpu)
    assert kernels.bnn_planes_bytes(M, nA) == 3 * M * nA * 2 and kernels.bnn_planes_bytes(20, nA) == 0
    kernels.bnn_split_planes([h[z] for z in range(count)], pa)
    kernels.bnn_split_planes([d[z] for z in range(count)], pb)
    kernels.bnn_gw_planes(pa, pb, outs, M)
    ref = torch.bmm(h.double().transpose(1, 2), d.double())
    lib = torch.bmm(h.transpose(1, 2), d)
    got = torch.stack(outs)
    err, err_lib = (got.double() - ref).abs(), (lib.double() - ref).abs()
    assert torch.isfinite(got).all()
    assert err.max().item() <= 1.05 * err_lib.max().item() + 1e-12, (err.max().item(), err_lib.max().item())
    assert (err ** 2).mean().sqrt().item() <= (err_lib ** 2).mean().sqrt().item() + 1e-15
    assert err.max().item() < 4e-6
    assert torch.all(cbuf[:, nA] == -7.0) and torch.all(cbuf[:, :, nB:] == -7.0)          # pitch columns and the row behind untouched
    # the planes are an EXACT split: their sum gives the operand back bit for bit
    planes = pa[:kernels.bnn_planes_bytes(M, nA)].view(torch.bfloat16).view(3, M // 8, nA, 8).float()
    back = planes.sum(0).permute(0, 2, 1).reshape(M, nA)                                  # (p0 + p1) + p2 in fp32: exact
    assert torch.equal(back, h[0])
    again = torch.empty_like(cbuf)
    kernels.bnn_gw_planes(pa, pb, [again[z, :nA, :nB] for z in range(count)], M)
    assert torch.equal(again[:, :nA, :nB], got)


def test_gw_planes_refuses_what_it_cannot_take(gpu):
    from pysgmcmc_amd import kernels
    from pysgmcmc_amd._lib import SgmcmcLibraryError
    h, d = torch.randn(24, 128, device=gpu), torch.randn(24, 128, device=gpu)
    with pytest.raises(ValueError):
        kernels.bnn_split_planes([h], torch.empty(3 * 24 * 128 * 2, dtype=torch.uint8, device=gpu))      # batch % 16 != 0
    h = torch.randn(32, 128, device=gpu)
    with pytest.raises(ValueError):
        kernels.bnn_split_planes([h], torch.empty(100, dtype=torch.uint8, device=gpu))                   # buffer too small
    with pytest.raises(ValueError):
        kernels.bnn_split_planes([h.double()], torch.empty(3 * 32 * 128 * 2, dtype=torch.uint8, device=gpu))
    with pytest.raises(ValueError):
        kernels.bnn_split_planes([h, torch.randn(32, 64, device=gpu)], torch.empty(2 * 3 * 32 * 128 * 2, dtype=torch.uint8, device=gpu))
    assert SgmcmcLibraryError is not None


@pytest.mark.parametrize("hidden,setting,want_planes", [((128, 128, 128), True, True), ((128, 128, 128), "auto", False),
                                                        ((2944, 2944, 2944), "auto", True), ((2944, 2944, 2944), False, False)])
def test_cost_path_with_the_weight_gradients_on_bf16_planes(gpu, hidden, setting, want_planes):
    """``BNNCost.gw_on_bf16_planes``: "auto" moves the batched weight gradients of equally shaped hidden layers to the bf16 matrix
    pipe when they have at least 4 tiles of 128 x 128 per compute unit (2 x 23 x 23 tiles at 2944 wide on 256 CUs), True wherever
    the shapes fit, False never; cost and every gradient against autograd through the torch restatement (fp64)."""
    from pysgmcmc_amd import kernels
    cus = torch.cuda.get_device_properties(gpu).multi_processor_count
    if setting == "auto" and want_planes and 2 * 23 * 23 < 4 * cus:
        pytest.skip("more compute units than the shape was sized for")
    cost, params = _cost(gpu, True, 256, hidden, n_in=64, own_feed_buffer=True)
    cost.gw_on_bf16_planes = setting
    n = sum(p.numel() for p in params)
    flat = torch.cat([p.reshape(-1) for p in params])
    offs = np.cumsum([0] + [p.numel() for p in params])
    params = [flat[offs[k]:offs[k + 1]].view(p.shape) for k, p in enumerate(params)]
    gflat = torch.full((n,), float("nan"), device=gpu)
    grads = [gflat[offs[k]:offs[k + 1]].view(p.shape) for k, p in enumerate(params)]
    st = kernels.StepStats(n, gpu)
    kernels.sghmc_step(flat.clone(), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu), None, None, None,
                       torch.ones(n, device=gpu), None, 0.0, 1.0, 0.0, False, xi=torch.zeros(n, device=gpu), stats=st,
                       opts=dict(theta_sq_only=True))
    plan = cost.plan_summary(params, grads, st.workspace)
    assert plan["weight_gradients_in_one_batched_product"] == [1, 2]
    assert plan["batched_weight_gradient_arithmetic"].startswith("3 exact bf16 planes" if want_planes else "library fp32")
    c = float(cost.cost_and_grad(params, grads, theta_sumsq_partials=st.workspace))
    gflat += cost.grad_theta_coef * flat                        # fold_prior: the update kernel adds coef * theta
    leaves = [p.detach().clone().double().requires_grad_(True) for p in params]
    X, Y = cost.x_placeholder.value.double(), cost.y_placeholder.value.double()
    nll, _ = cost.negative_log_likelihood(leaves, X, Y)
    auto = torch.autograd.grad(nll, leaves)
    assert abs(c - float(nll.detach())) <= 2e-6 * abs(float(nll.detach()))
    for g, a in zip(grads, auto):
        assert torch.isfinite(g).all() and float((g.double() - a).abs().max()) <= 3e-5 * float(a.abs().max()) + 1e-9
    # fold_prior = False needs beta * W in the product's epilogue: that plan keeps the library
    cost.fold_prior = False
    assert cost.plan_summary(params, grads, st.workspace)["batched_weight_gradient_arithmetic"].startswith("library fp32")

"""Tests of the round-3 experiments that live outside the product (run by hand on a GPU box, not by the driver):

    make -C tools/experiments && python -m pytest tools/experiments/test_experiments_gpu.py -q -p no:cacheprovider

* the hand-written fp32 matrix-core weight-gradient GEMM (``sgmcmc_gemm_tn_f32``) and its fused form with the frozen SGHMC
  update as epilogue (``sgmcmc_gemm_tn_sghmc_f32``);
* the two stepping modes built on them / on slice launches (``tools/experiments/stepping.py``): bit-equal chains."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.experiments import gemm_kernels          # noqa: E402  (sets PYSGMCMC_AMD_LIB before the library is loaded)

import numpy as np                                   # noqa: E402
import pytest                                        # noqa: E402
import torch                                         # noqa: E402

from tools.experiments.stepping import HookedBNNCost, experimental    # noqa: E402


@pytest.fixture
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an AMD GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("M,N,K", [(2048, 256, 64), (784, 128, 256), (132, 256, 64), (4, 128, 32)])
def test_gemm_tn_equals_a_k_ordered_fp32_product(gpu, M, N, K):
    """fp32 MFMA is an exact fmaf chain in k order: the tile variants agree with each other BIT FOR BIT and with an fp64
    product to fp32 roundoff; rows beyond M of a ragged last tile are never written."""
    from pysgmcmc_amd import kernels
    g = torch.Generator(device=gpu).manual_seed(1)
    a, b = torch.randn(K, M, device=gpu, generator=g), torch.randn(K, N, device=gpu, generator=g)
    exact = a.double().t() @ b.double()
    outs = []
    for variant in range(13):                               # 0, 10, 11: direct-to-LDS operand loads; the others stage through registers
        if K % (16, 32, 64, 32, 64, 32, 32, 32, 32, 16, 32, 16, 16)[variant]:
            continue
        out = torch.full((M + 3, N), -7.0, device=gpu)
        gemm_kernels.gemm_tn(a, b, out[:M], variant=variant)
        assert (out[:M].double() - exact).abs().max().item() <= 2e-6 * K and torch.all(out[M:] == -7.0)
        outs.append(out[:M].clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    with pytest.raises(Exception, match="gemm_tn"):
        gemm_kernels.gemm_tn(a, b[:, :100].contiguous(), torch.empty(M, 100, device=gpu))


@pytest.mark.parametrize("M,N,K,n_tail,first", [(256, 256, 64, 256, 1024), (132, 128, 32, 7, 64), (64, 128, 256, 0, 0),
                                                (784, 256, 256, 258, 4)])
def test_fused_gemm_update_equals_k1_on_the_gradient_it_computed(gpu, M, N, K, n_tail, first):
    """The epilogue IS kernel K1: given the gradient the fused kernel formed (written out through ``grad_out``) the
    streaming SGHMC step on the same slice -- same Philox stream (first_element), same grad_decay -- produces the same
    theta', V' bit for bit, for the weights and for the parameters that follow them; sum theta'^2 lands in the records."""
    from pysgmcmc_amd import kernels
    g = torch.Generator(device=gpu).manual_seed(2)
    a = torch.randn(K, M, device=gpu, generator=g)
    b = torch.randn(K, N, device=gpu, generator=g) * 0.01
    n = M * N + n_tail
    theta0, V0 = torch.randn(n, device=gpu, generator=g) * 0.05, torch.randn(n, device=gpu, generator=g) * 0.01
    minv = torch.rand(n, device=gpu, generator=g) + 0.5
    gtail = torch.randn(n_tail, device=gpu, generator=g) * 0.1 if n_tail else None
    # default grid; 3 persistent workgroups walking over the tiles; the two flavours that request the state before the K loop
    for step, blocks in ((5, 0), (6, 3), (7, 3 | (1 << 16)), (8, 2 << 16)):
        th, V = theta0.clone(), V0.clone()
        gout = torch.full((M, N), float("nan"), device=gpu)
        st = kernels.StepStats(n, gpu)
        gemm_kernels.gemm_tn_sghmc(a, b, th, V, minv, gtail, 0.01, 1e4, 0.05, grad_decay=1e-5, seed=11, step=step,
                              first_element=first, stats=st, grad_out=gout, gemm_blocks=blocks)
        th2, V2 = theta0.clone(), V0.clone()
        grad = gout.reshape(-1) if n_tail == 0 else torch.cat([gout.reshape(-1), gtail])
        kernels.sghmc_step(th2, V2, grad, None, None, None, minv, None, 0.01, 1e4, 0.05, False, seed=11, step=step,
                           grad_decay=1e-5, opts=dict(first_element=first))
        assert torch.equal(th, th2) and torch.equal(V, V2)
        assert (gout.double() - a.double().t() @ b.double()).abs().max().item() < 1e-5
        assert np.isclose(kernels.step_stats_finish(st)[0].item(), (th.double() ** 2).sum().item(), rtol=1e-6)
        assert int(st.workspace.view(torch.int64)[0]) == gemm_kernels.gemm_tn_sghmc_blocks(M, N, n_tail, blocks)


def _bnn_sghmc(gpu, fused, gw_gemm, graph=True, steps=12, moments_every=0):
    from pysgmcmc_amd.data_batches import Placeholder, generate_batches
    from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
    from pysgmcmc_amd.samplers import SGHMCSampler
    from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
    rng = np.random.RandomState(0)
    X, y = rng.rand(600, 20), rng.rand(600)
    xp, yp = Placeholder(dtype=torch.float32, device=gpu), Placeholder(dtype=torch.float32, device=gpu)
    params = init_mlp_params(20, hidden=(128, 256, 128), seed=5, dtype=torch.float32, device=gpu)      # 20->128->256->128->1
    cost = HookedBNNCost(xp, yp, batch_size=64, n_examples=600)
    cost.gw_gemm = gw_gemm
    s = experimental(SGHMCSampler)(params=params, cost_fun=cost, batch_generator=generate_batches(X, y, xp, yp, batch_size=64, seed=2),
                     stepsize_schedule=ConstantStepsizeSchedule(0.01), burn_in_steps=4, scale_grad=600.0, session=gpu,
                     dtype=torch.float32, seed=9)
    s.sample_format = "view"
    s.use_hip_graph = graph
    s.collect_stats = "theta_sq"
    s.fuse_update_into_gemm = fused
    m = ChainMoments(s.arena.n, gpu)
    if moments_every:
        s.attach_moments(m, moments_every)
    costs = [float(next(s)[1]) for _ in range(steps)]
    torch.cuda.synchronize()
    return s, m, costs


def test_sampler_with_the_update_fused_into_the_weight_gradient_gemms(gpu):
    """``fuse_update_into_gemm``: after burn-in every hidden layer's weight-gradient GEMM carries the update of that layer's
    slice (no update launch). The chain equals, BIT FOR BIT, the un-fused sampler whose weight-gradient products come from
    the same matrix-core kernel (``gw_gemm = "mfma"``) -- through the burn-in switch and moments steps (which step un-fused)
    -- and agrees with the library-GEMM sampler to summation-order rounding."""
    fused, mf, cf = _bnn_sghmc(gpu, True, "mfma", moments_every=5)
    plain, mp, cp = _bnn_sghmc(gpu, False, "mfma", moments_every=5)
    blas, _, cb = _bnn_sghmc(gpu, False, "blas", moments_every=5)
    assert any(k[0] == "fused_gemm" for k in fused._graphs) and not any(k[0] == "fused_gemm" for k in plain._graphs)
    plan, total = fused._fused_plan
    assert [p["layer"] for p in plan] == [0, 1, 2] and plan[-1]["hi"] == fused.arena.n and plan[0]["lo"] == 0
    for row in ("theta", "V", "minv"):
        assert torch.equal(fused.arena.row(row), plain.arena.row(row)), row
    assert torch.equal(mf.mean, mp.mean) and torch.equal(mf.m2, mp.m2) and mf.count == mp.count == 2
    assert np.allclose(cf, cp, rtol=1e-6) and np.allclose(cf, cb, rtol=1e-4)
    assert torch.allclose(fused.arena.row("theta"), blas.arena.row("theta"), rtol=1e-3, atol=1e-5)
    st = fused.stats
    assert np.isclose(st["theta_sq"], (fused.arena.row("theta").double() ** 2).sum().item(), rtol=1e-6)
    # a model the kernel does not fit (fan_out not a multiple of 128) steps the usual way
    from pysgmcmc_amd.samplers import SGHMCSampler
    s = experimental(SGHMCSampler)(params=[torch.zeros(8, device=gpu)], cost_fun=lambda p: (p[0] ** 2).sum(), burn_in_steps=1, session=gpu,
                     dtype=torch.float32, seed=1)
    s.use_hip_graph = True
    s.fuse_update_into_gemm = True
    for _ in range(4):
        next(s)
    assert s._fused_plan is False


def _bnn_chain(gpu, ctor, overlap, graph, moments_every=0, fused_moments=True, steps=14, **kw):
    from pysgmcmc_amd.data_batches import Placeholder, generate_batches
    from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
    from pysgmcmc_amd.profiling import UpdateKernelTimer
    from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
    rng = np.random.RandomState(0)
    X, y = rng.rand(400, 16), rng.rand(400)
    xp, yp = Placeholder(dtype=torch.float32, device=gpu), Placeholder(dtype=torch.float32, device=gpu)
    params = init_mlp_params(16, hidden=(96, 128, 64), seed=5, dtype=torch.float32, device=gpu)
    cost = HookedBNNCost(xp, yp, batch_size=32, n_examples=400)
    cost.OVERLAP_MIN_WEIGHTS = 1024                        # every hidden layer announces its gradient
    s = experimental(ctor)(params=params, cost_fun=cost, batch_generator=generate_batches(X, y, xp, yp, batch_size=32, seed=2),
             stepsize_schedule=ConstantStepsizeSchedule(0.01), session=gpu, dtype=torch.float32, seed=9, **kw)
    s.sample_format = "view"
    s.use_hip_graph = graph
    s.overlap_update = overlap
    s.collect_stats = "theta_sq"
    s.kernel_timer = UpdateKernelTimer()
    s.kernel_timer.enabled = True
    m = ChainMoments(s.arena.n, gpu)
    if moments_every and fused_moments:
        s.attach_moments(m, moments_every)
    costs = []
    for i in range(steps):
        costs.append(float(next(s)[1]))
        if moments_every and not fused_moments and (i + 1) % moments_every == 0:
            m.update(s.arena.row("theta"))
    torch.cuda.synchronize()
    return s, m, costs


def test_overlapped_update_gives_the_same_chain(gpu):
    """overlap_update: the cost pipeline is replayed as graph segments and every finished slice of the arena is updated
    on a side stream under the rest of the backward pass. Chain, costs, statistics-fed weight prior and the fused
    Welford moments equal the single-launch sampler's bit for bit (SGHMC across the burn-in switch, SGLD, relativistic)."""
    from pysgmcmc_amd.samplers import RelativisticSGHMCSampler, SGHMCSampler, SGLDSampler
    for ctor, kw, rows in ((SGHMCSampler, dict(burn_in_steps=5, scale_grad=400.0), ("theta", "V", "minv", "grad")),
                           (SGLDSampler, dict(burn_in_steps=5, scale_grad=400.0), ("theta", "minv")),
                           (RelativisticSGHMCSampler, {}, ("theta", "p"))):
        base, mb, cb = _bnn_chain(gpu, ctor, overlap=False, graph=True, moments_every=3, fused_moments=False, **kw)
        over, mo, co = _bnn_chain(gpu, ctor, overlap=True, graph=True, moments_every=3, **kw)
        eager, me, ce = _bnn_chain(gpu, ctor, overlap=False, graph=False, moments_every=3, **kw)
        assert len(over._graphs[("cost_segments",)][0]) == 3          # three graph segments: two announced layers + the tail
        assert len(over.kernel_timer.kevents) == 3 * 14 and len(base.kernel_timer.kevents) == 14
        assert over.kernel_timer.per_step_kernel_us().shape == (14,) and (over.kernel_timer.kernel_us() > 0).all()
        for other, m, c in ((over, mo, co), (eager, me, ce)):
            for row in rows:
                assert torch.equal(base.arena.row(row), other.arena.row(row)), (ctor.__name__, row)
            assert np.allclose(c, cb, rtol=1e-6, atol=0)
            assert m.count == mb.count == 4 and torch.equal(m.mean, mb.mean) and torch.equal(m.m2, mb.m2)




def test_fused_gemm_mode_declines_what_the_kernel_cannot_take(gpu):
    """with a batch that is not a multiple of 16 (the reference BNN's default is 20) or with fold_prior = False the
    fused GEMM + update mode must step un-fused -- the chain keeps moving and equals the plain sampler's."""
    from pysgmcmc_amd.data_batches import Placeholder, generate_batches
    from pysgmcmc_amd.models.bayesian_neural_network import init_mlp_params
    from pysgmcmc_amd.samplers import SGHMCSampler
    from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
    rng = np.random.RandomState(0)
    X, y = rng.rand(400, 16), rng.rand(400)

    def chain(fuse, batch, fold):
        xp, yp = Placeholder(dtype=torch.float32, device=gpu), Placeholder(dtype=torch.float32, device=gpu)
        params = init_mlp_params(16, hidden=(128, 128), seed=5, dtype=torch.float32, device=gpu)
        cost = HookedBNNCost(xp, yp, batch_size=batch, n_examples=400, fold_prior=fold)
        s = experimental(SGHMCSampler)(params=params, cost_fun=cost, batch_generator=generate_batches(X, y, xp, yp, batch_size=batch, seed=2),
                                       stepsize_schedule=ConstantStepsizeSchedule(0.01), burn_in_steps=2, scale_grad=400.0,
                                       session=gpu, dtype=torch.float32, seed=9)
        s.sample_format = "view"
        s.use_hip_graph = True
        s.fuse_update_into_gemm = fuse
        s.collect_stats = "theta_sq"
        for _ in range(8):
            next(s)
        return s
    for batch, fold in ((20, True), (32, False)):
        a, b = chain(True, batch, fold), chain(False, batch, fold)
        assert a._fused_plan is False
        for row in ("theta", "V"):
            assert torch.equal(a.arena.row(row), b.arena.row(row)), (batch, fold, row)

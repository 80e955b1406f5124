"""The hand-written fp32 matrix-core weight-gradient GEMM (``sgmcmc_gemm_tn_f32``) and its fused form with the frozen
SGHMC update as epilogue (``sgmcmc_gemm_tn_sghmc_f32``; experimental, not on the sampler's default path)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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
        kernels.gemm_tn(a, b, out[:M], variant=variant)
        assert (out[:M].double() - exact).abs().max().item() <= 2e-6 * K and torch.all(out[M:] == -7.0)
        outs.append(out[:M].clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    with pytest.raises(Exception, match="gemm_tn"):
        kernels.gemm_tn(a, b[:, :100].contiguous(), torch.empty(M, 100, device=gpu))


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
        kernels.gemm_tn_sghmc(a, b, th, V, minv, gtail, 0.01, 1e4, 0.05, grad_decay=1e-5, seed=11, step=step,
                              first_element=first, stats=st, grad_out=gout, gemm_blocks=blocks)
        th2, V2 = theta0.clone(), V0.clone()
        grad = gout.reshape(-1) if n_tail == 0 else torch.cat([gout.reshape(-1), gtail])
        kernels.sghmc_step(th2, V2, grad, None, None, None, minv, None, 0.01, 1e4, 0.05, False, seed=11, step=step,
                           grad_decay=1e-5, opts=dict(first_element=first))
        assert torch.equal(th, th2) and torch.equal(V, V2)
        assert (gout.double() - a.double().t() @ b.double()).abs().max().item() < 1e-5
        assert np.isclose(kernels.step_stats_finish(st)[0].item(), (th.double() ** 2).sum().item(), rtol=1e-6)
        assert int(st.workspace.view(torch.int64)[0]) == kernels.gemm_tn_sghmc_blocks(M, N, n_tail, blocks)


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
    cost = BNNCost(xp, yp, batch_size=64, n_examples=600)
    cost.gw_gemm = gw_gemm
    s = SGHMCSampler(params=params, cost_fun=cost, batch_generator=generate_batches(X, y, xp, yp, batch_size=64, seed=2),
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
    s = SGHMCSampler(params=[torch.zeros(8, device=gpu)], cost_fun=lambda p: (p[0] ** 2).sum(), burn_in_steps=1, session=gpu,
                     dtype=torch.float32, seed=1)
    s.use_hip_graph = True
    s.fuse_update_into_gemm = True
    for _ in range(4):
        next(s)
    assert s._fused_plan is False

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
    for variant in range(9):
        if K % (16, 32, 64, 32, 64, 32, 64, 32, 64)[variant]:
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
    for step, blocks in ((5, 0), (6, 3)):                   # default grid, and 3 persistent workgroups walking over the tiles
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

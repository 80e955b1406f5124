"""The fused weight-gradient GEMM + SGHMC update against (library GEMM, then K1 on the layer's slice): bit-exactness of
theta', V' given the gradient the fused kernel computed, and microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.experiments import gemm_kernels      # sets PYSGMCMC_AMD_LIB to the experiments build
import torch
from pysgmcmc_amd import kernels
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning

dev = torch.device("cuda:0")
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)
for (M, N, K, n_tail, first) in ((2048, 2048, 256, 2048, 1607680), (784, 2048, 256, 2048, 0), (2048, 2048, 256, 4098, 5804032), (132, 128, 64, 7, 64)):
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(K, M, device=dev, generator=g)
    b = torch.randn(K, N, device=dev, generator=g) * 0.01
    n = M * N + n_tail
    theta0 = torch.randn(n, device=dev, generator=g) * 0.05
    V0 = torch.randn(n, device=dev, generator=g) * 0.01
    minv = torch.rand(n, device=dev, generator=g) + 0.5
    gtail = torch.randn(n_tail, device=dev, generator=g) * 0.1
    # fused
    th, V = theta0.clone(), V0.clone()
    gout = torch.full((M, N), float("nan"), device=dev)
    st = kernels.StepStats(n, dev)
    gemm_kernels.gemm_tn_sghmc(a, b, th, V, minv, gtail, 0.01, 1e5, 0.05, grad_decay=1e-6, seed=11, step=5, first_element=first,
                          stats=st, grad_out=gout)
    # reference: K1 on the gradient the fused kernel wrote
    th2, V2 = theta0.clone(), V0.clone()
    grad = torch.cat([gout.reshape(-1), gtail])
    kernels.sghmc_step(th2, V2, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=11, step=5, grad_decay=1e-6,
                       opts=dict(first_element=first))
    exact = a.double().t() @ b.double()
    print("M=%d N=%d K=%d tail=%d: theta bit-equal %s, V bit-equal %s, gW max|err| %.2e, sum theta^2 %.9g vs %.9g" % (
        M, N, K, n_tail, torch.equal(th, th2), torch.equal(V, V2), (gout.double() - exact).abs().max().item(),
        kernels.step_stats_finish(st)[0].item(), (th.double() ** 2).sum().item()), flush=True)
    th_ref, V_ref = th.clone(), V.clone()
    if M < 500:
        continue
    def t(fn, nrep=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nrep):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / nrep * 1e3
    gw = torch.empty(M * N + n_tail, device=dev)
    gw[M * N:] = gtail
    def separate():
        torch.mm(a.t(), b, out=gw[:M * N].view(M, N))
        kernels.sghmc_step(th2, V2, gw, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=11, step=5, stats=st)
    def fused(blocks):
        gemm_kernels.gemm_tn_sghmc(a, b, th, V, minv, gtail, 0.01, 1e5, 0.05, seed=11, step=5, first_element=first, stats=st,
                              gemm_blocks=blocks)
    print("    library GEMM + K1 slice: %.1f us" % t(separate), flush=True)
    names = {0: "no prefetch, 4 WG/CU", 1: "2 of 4 quads prefetched, 4 WG/CU", 2: "4 of 4 prefetched, 4 WG/CU (spills)",
             3: "4 of 4 prefetched, 3 WG/CU", 4: "no prefetch, 5 WG/CU (spills)", 5: "1 of 4 prefetched, 4 WG/CU",
             6: "2 of 4 prefetched, 3 WG/CU"}
    for fl in range(7):
        th3, V3 = theta0.clone(), V0.clone()
        gemm_kernels.gemm_tn_sghmc(a, b, th3, V3, minv, gtail, 0.01, 1e5, 0.05, grad_decay=1e-6, seed=11, step=5, first_element=first,
                              gemm_blocks=1024 | (fl << 16))
        ok = torch.equal(th3, th_ref) and torch.equal(V3, V_ref)
        print("    flavour %d (%s): bit-equal %s, %s us" % (fl, names[fl], ok, " / ".join(
            "%.1f" % t(lambda: fused(blocks | (fl << 16))) for blocks in (768, 1024, 2048))), flush=True)
    pc = torch.zeros(2048, dtype=torch.int32, device=dev)
    for sl in (1, 2, 3):
        def fused_dephased():
            gemm_kernels.gemm_tn_sghmc(a, b, th, V, minv, gtail, 0.01, 1e5, 0.05, seed=11, step=5, first_element=first, stats=st,
                                  phase_counters=pc, phase_sleeps=sl)
        print("    fused, de-phased by %d x 3.4 us: %.1f us   (CU keys seen: %d)" % (sl, t(fused_dephased), int((pc > 0).sum())), flush=True)
    th.copy_(theta0); V.copy_(V0); th2.copy_(theta0); V2.copy_(V0)

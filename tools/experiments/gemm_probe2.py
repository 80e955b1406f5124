"""The hand-written fp32 MFMA weight-gradient GEMM against the library's (torch.mm -> rocBLAS / hipBLASLt, TunableOp on):
correctness against an fp64 product and microseconds per call at the bench's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.experiments import gemm_kernels      # sets PYSGMCMC_AMD_LIB to the experiments build
import torch
from pysgmcmc_amd import kernels
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning

dev = torch.device("cuda:0")
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)
for (M, N, K) in ((2048, 2048, 256), (784, 2048, 256), (132, 128, 64)):
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(K, M, device=dev, generator=g)
    b = torch.randn(K, N, device=dev, generator=g)
    out = torch.empty(M, N, device=dev)
    ref = torch.empty(M, N, device=dev)
    gemm_kernels.gemm_tn(a, b, out)
    torch.mm(a.t(), b, out=ref)
    exact = (a.double().t() @ b.double())
    e_mine = (out.double() - exact).abs().max().item()
    e_lib = (ref.double() - exact).abs().max().item()
    def t(fn, n=300):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    us_lib = t(lambda: torch.mm(a.t(), b, out=ref))
    fl = 2.0 * M * N * K
    print("M=%d N=%d K=%d  max|err| mine %.2e lib %.2e | lib %.1f us (%.0f TF/s)" % (M, N, K, e_mine, e_lib, us_lib, fl / us_lib / 1e6), flush=True)
    for v in (0, 9, 11, 12):
        out.zero_()
        gemm_kernels.gemm_tn(a, b, out, variant=v)
        err = (out.double() - exact).abs().max().item()
        us = t(lambda: gemm_kernels.gemm_tn(a, b, out, variant=v))
        ph = " | frags once %.1f | bare MFMA chain %.1f" % (t(lambda: gemm_kernels.gemm_tn(a, b, out, variant=v | 0x400)), t(lambda: gemm_kernels.gemm_tn(a, b, out, variant=v | 0xC00)))
        us_ns = t(lambda: gemm_kernels.gemm_tn(a, b, out, variant=v | 0x100))
        us_q = t(lambda: gemm_kernels.gemm_tn(a, b, out, variant=v | 0x200)) if K >= 256 else float("nan")
        us_qns = t(lambda: gemm_kernels.gemm_tn(a, b, out, variant=v | 0x300)) if K >= 256 else float("nan")
        print("    variant %d: max|err| %.2e  %.1f us (%.0f TF/s) | no store %.1f | K/4 %.1f | K/4 no store %.1f" % (
            v, err, us, fl / us / 1e6, us_ns, us_q, us_qns) + ph, flush=True)

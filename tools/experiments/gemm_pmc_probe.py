"""A few launches of the plain and fused weight-gradient GEMM kernels at 2048 x 2048 x 256 for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.experiments import gemm_kernels      # sets PYSGMCMC_AMD_LIB to the experiments build
import torch
from pysgmcmc_amd import kernels
dev = torch.device("cuda:0")
M = N = 2048; K = 256; n_tail = 2048
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(K, M, device=dev, generator=g); b = torch.randn(K, N, device=dev, generator=g) * 0.01
n = M * N + n_tail
th = torch.randn(n, device=dev, generator=g) * 0.05; V = torch.zeros(n, device=dev); minv = torch.rand(n, device=dev, generator=g) + 0.5
gt = torch.randn(n_tail, device=dev, generator=g) * 0.1
out = torch.empty(M, N, device=dev)
st = kernels.StepStats(n, dev)
ref = torch.empty(M, N, device=dev)
for rep in range(4):
    for v in (0, 9, 12):             # direct-to-LDS (4 workgroups per CU), register-staged, direct-to-LDS with 6 per CU
        gemm_kernels.gemm_tn(a, b, out, variant=v)
    gemm_kernels.gemm_tn_sghmc(a, b, th, V, minv, gt, 0.01, 1e5, 0.05, seed=1, step=rep, stats=st)
    torch.mm(a.t(), b, out=ref)
torch.cuda.synchronize()

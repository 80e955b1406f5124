"""Launch each hot-path kernel a few times at a given size, for rocprofv3 --pmc passes (dev tool).

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_probe.py 10002434
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/pmc_probe.py 10002434

Calibration launches with known byte counts in the SAME access pattern (16 B/lane coalesced):
  philox_normal fill  : 0 B read, 4n B written
  moments_update      : 12n B read, 8n B written
"""
import sys
import torch
sys.path.insert(0, ".")
from pysgmcmc_amd import kernels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_002_434
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda s=1.0: torch.randn(n, device=dev, generator=g) * s
theta, V, grad = mk(0.02), torch.zeros(n, device=dev), mk(0.1)
tau, gg, vh = torch.ones(n, device=dev), torch.ones(n, device=dev), torch.ones(n, device=dev)
minv = torch.rand(n, device=dev, generator=g) * 1.5 + 0.5
mean, m2, out = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.empty(n, device=dev)
# flush: touch 1 GiB so nothing of the working set is left in the 256 MiB Infinity Cache between kernels
junk = torch.empty(1 << 28, device=dev)
flush = lambda: junk.fill_(1.0)
st = kernels.StepStats(n, dev)
for r in range(reps):
    flush(); kernels.philox_normal(out, 1, r)
    flush(); kernels.moments_update(theta, mean, m2, r + 1)
    flush(); kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=1, step=r)
    flush(); kernels.sghmc_step(theta, V, grad, tau, gg, vh, minv, None, 0.01, 1e5, 0.05, True, seed=1, step=r)
    flush(); kernels.sgld_step(theta, grad, None, None, None, minv, None, 0.01, 1.0, 1e5, False, seed=1, step=r)
    flush(); kernels.sgld_step(theta, grad, tau, gg, vh, minv, None, 0.01, 1.0, 1e5, True, seed=1, step=r)
    flush(); kernels.rsghmc_step(theta, V, grad, 0.001, 1.0, 1.0, 1.0, 0.0, seed=1, step=r)
    # the variants the samplers launch in the pipeline: fused step statistics (STATS = 1: all of the operator's, 2: sum theta^2
    # only), the fused Welford moments (MOM), and the burn-in step that skips the minv store
    tsq = dict(theta_sq_only=True)
    for o in (None, tsq):
        flush(); kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=1, step=r, stats=st, opts=o)
        flush(); kernels.sgld_step(theta, grad, None, None, None, minv, None, 0.01, 1.0, 1e5, False, seed=1, step=r, stats=st, opts=o)
        flush(); kernels.rsghmc_step(theta, V, grad, 0.001, 1.0, 1.0, 1.0, 0.0, seed=1, step=r, stats=st, opts=o)
    flush(); kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=1, step=r, stats=st,
                                opts=dict(theta_sq_only=True, moments=(mean, m2, r + 2)))
    flush(); kernels.sghmc_step(theta, V, grad, tau, gg, vh, minv, None, 0.01, 1e5, 0.05, True, seed=1, step=r, stats=st, opts=tsq)
torch.cuda.synchronize()
print("probe done n=%d reps=%d" % (n, reps))

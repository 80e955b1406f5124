"""Back-to-back cost of the fused step statistics (STATS = true variants) next to the plain kernels (dev tool)."""
import sys
import torch
sys.path.insert(0, ".")
from pysgmcmc_amd import kernels

dev = torch.device("cuda:0")
for n in (10_002_434, 49_826_818):
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda s: torch.randn(n, device=dev, generator=g) * s
    theta, V, grad = mk(0.02), torch.zeros(n, device=dev), mk(0.1)
    minv = torch.rand(n, device=dev, generator=g) * 1.5 + 0.5
    st = kernels.StepStats(n, dev)
    calls = {
        "sghmc_frozen": lambda i, s, o: kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=1, step=i, stats=s, opts=o),
        "sgld_frozen": lambda i, s, o: kernels.sgld_step(theta, grad, None, None, None, minv, None, 1e-3, 1.0, 1e5, False, seed=1, step=i, stats=s, opts=o),
        "rsghmc": lambda i, s, o: kernels.rsghmc_step(theta, V, grad, 1e-3, 1.0, 1.0, 1.0, 0.0, seed=1, step=i, stats=s, opts=o),
    }
    for name, call0 in calls.items():
        for label, s, o in (("plain", None, None), ("stats", st, None), ("tsq", st, dict(theta_sq_only=True))):
            call = lambda i, s, call0=call0, o=o: call0(i, s, o)
            for i in range(10):
                call(i, s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(200):
                call(i, s)
            e1.record()
            torch.cuda.synchronize()
            print("n=%d %-13s %-5s %.2f us" % (n, name, label, e0.elapsed_time(e1) / 200 * 1e3))
            theta.normal_(0, 0.02, generator=g); V.zero_()

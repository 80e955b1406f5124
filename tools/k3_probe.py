"""where does the relativistic step (K3) lose against the frozen SGHMC step (K1) at the same size?
Cold launches (1 GiB flush before each) of K1 frozen, K2 frozen and K3 at n parameters under a few launch geometries, kernel
timestamps; run it under ``rocprofv3 --pmc`` (tools/gpu/k3_counters.sh) for the SQ counters of the same launches."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from pysgmcmc_amd import kernels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 49_826_818
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda s=1.0: torch.randn(n, device=dev, generator=g) * s
theta, V, grad = mk(0.02), torch.zeros(n, device=dev), mk(0.1)
minv = torch.rand(n, device=dev, generator=g) * 1.5 + 0.5
junk = torch.empty(1 << 28, device=dev)
calls = {
    "K1 sghmc frozen (24 B)": lambda i, L: kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=1, step=i, launch=L),
    "K2 sgld frozen (16 B)": lambda i, L: kernels.sgld_step(theta, grad, None, None, None, minv, None, 1e-3, 1.0, 1e5, False, seed=1, step=i, launch=L),
    "K3 rsghmc (20 B)": lambda i, L: kernels.rsghmc_step(theta, V, grad, 1e-3, 1.0, 1.0, 1.0, 0.0, seed=1, step=i, launch=L),
}
bpp = {"K1 sghmc frozen (24 B)": 24, "K2 sgld frozen (16 B)": 16, "K3 rsghmc (20 B)": 20}
geoms = [("default", {}), ("bt=128", dict(block_threads=128)), ("bt=256", dict(block_threads=256)),
         ("bt=256 nt=0", dict(block_threads=256, nontemporal=0)), ("bt=128 nt=1", dict(block_threads=128, nontemporal=1)),
         ("qpt=2 bt=128", dict(block_threads=128, quads_per_thread=2)), ("qpt=2 bt=256", dict(block_threads=256, quads_per_thread=2))]
for name, call in calls.items():
    for label, geom in (geoms if name.startswith("K3") else geoms[:1]):
        for cold in (True, False):
            us = []
            for i in range(reps):
                if cold:
                    junk.fill_(1.0)
                ev = kernels.KernelEvents()
                call(i, kernels.LaunchConfig(events=ev, **geom))
                torch.cuda.synchronize()
                us.append(ev.elapsed_us())
                V.zero_()
            us = np.array(us[1:])
            print("%-24s %-14s %-5s n=%d: %7.2f us  (min %.2f)  %.0f GB/s = %.3f of 8 TB/s" % (
                name, label, "cold" if cold else "warm", n, us.mean(), us.min(), bpp[name] * n / us.mean() / 1e3,
                bpp[name] * n / us.mean() / 1e3 / 8000), flush=True)

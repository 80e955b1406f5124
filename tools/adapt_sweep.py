"""Launch geometry of the BURN-IN (ADAPT) kernels at 49 826 818 parameters (2.4 GB per SGHMC launch: 6 arrays read, 6
written) -- VERDICT r02 item 5: quads per lane, nt on/off, block size, and the step that does not store minv (44 vs
48 B/param). Kernel timestamps (hipExtLaunchKernel events), back to back and cold (1 GiB flush before every launch)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from pysgmcmc_amd import kernels

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 49_826_818
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda s: torch.randn(n, device=dev, generator=g) * s
theta, V, grad = mk(0.02), torch.zeros(n, device=dev), mk(0.1)
tau, gg, vh = (torch.ones(n, device=dev) for _ in range(3))
minv = torch.ones(n, device=dev)
junk = torch.empty(1 << 28, device=dev)


def measure(kind, cfg_kw, opts, cold, reps=30):
    kev = [kernels.KernelEvents() for _ in range(reps)]
    for i in range(reps + 3):
        if cold:
            junk.fill_(1.0)
        L = kernels.LaunchConfig(events=kev[i - 3] if i >= 3 else None, **cfg_kw)
        if kind == "sghmc":
            kernels.sghmc_step(theta, V, grad, tau, gg, vh, minv, None, 0.01, 1e5, 0.05, True, seed=1, step=i, launch=L, opts=opts)
        else:
            kernels.sgld_step(theta, grad, tau, gg, vh, minv, None, 1e-3, 1.0, 1e5, True, seed=1, step=i, launch=L, opts=opts)
    torch.cuda.synchronize()
    us = np.array([k.elapsed_us() for k in kev])
    theta.normal_(0, 0.02, generator=g); V.zero_(); tau.fill_(1); gg.fill_(1); vh.fill_(1)
    return float(us.mean()), float(np.median(us))


for kind, bpp in (("sghmc", 48), ("sgld", 40)):
    print("== %s burn-in step, n = %d (%d B/param algorithmic, %d with the minv store skipped)" % (kind, n, bpp, bpp - 4))
    for cold in (False, True):
        for bt in (128, 256):
            for qpt in (1, 2, 4):
                for nt in (0, 1):
                    for skip in (False, True):
                        if skip and (qpt != 1):
                            continue
                        mean, med = measure(kind, dict(block_threads=bt, quads_per_thread=qpt, nontemporal=nt),
                                            dict(skip_minv_store=True) if skip else None, cold)
                        b = (bpp - 4 if skip else bpp) * n
                        print("%-5s bt=%3d qpt=%d nt=%d skip_minv=%d  %7.1f us (median %7.1f)  %6.0f GB/s  frac %.3f of 8 TB/s" % (
                            "cold" if cold else "b2b", bt, qpt, nt, skip, mean, med, b / mean / 1e3, b / mean / 1e3 / 8000), flush=True)

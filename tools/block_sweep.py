"""Block-size sweep of the single-pass update kernels (dev tool)."""
import sys, torch
sys.path.insert(0, ".")
from pysgmcmc_amd import kernels
dev = torch.device("cuda:0")
for n in (10_002_434, 50_000_000):
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda: torch.randn(n, device=dev, generator=g)
    theta, V, grad = mk() * 0.02, torch.zeros(n, device=dev), mk() * 0.1
    tau, gg, vh = (torch.ones(n, device=dev) for _ in range(3))
    minv = torch.rand(n, device=dev, generator=g) + 0.5
    stt = kernels.StepStats(n, dev)
    st = [0]
    def frozen(): st[0] += 1; kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=1, step=st[0])
    def frozen_stats(): st[0] += 1; kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=1, step=st[0], stats=stt)
    def adapt(): st[0] += 1; kernels.sghmc_step(theta, V, grad, tau, gg, vh, minv, None, 0.01, 1e5, 0.05, True, seed=1, step=st[0])
    for rnd in range(2):
        for name, fn, bpp in (("frozen", frozen, 24), ("frozen+stats", frozen_stats, 24), ("adapt", adapt, 48)):
            for bt in (64, 128, 192, 256):
                kernels.set_launch_config(bt, 1, 1 << 20, 2)
                for _ in range(10): fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(100): fn()
                b.record(); torch.cuda.synchronize()
                us = a.elapsed_time(b) / 100 * 1e3
                print("n=%d %-13s bt=%3d : %7.1f us %6.0f GB/s" % (n, name, bt, us, bpp * n / us / 1e3))

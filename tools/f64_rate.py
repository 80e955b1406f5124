"""f64 kernel rates (dev tool): the reference's default dtype is float64."""
import sys, torch
sys.path.insert(0, ".")
from pysgmcmc_amd import kernels
n = 10_002_434
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for dt, bpp in ((torch.float32, 4), (torch.float64, 8)):
    mk = lambda: torch.randn(n, device=dev, generator=g, dtype=dt)
    theta, V, grad, xi = mk() * 0.02, torch.zeros(n, device=dev, dtype=dt), mk() * 0.1, mk()
    tau, gg, vh = (torch.ones(n, device=dev, dtype=dt) for _ in range(3))
    minv = torch.rand(n, device=dev, generator=g, dtype=dt) + 0.5
    def timeit(fn, iters=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3
    st = [0]
    def frozen(): st[0] += 1; kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=1, step=st[0])
    def frozen_inj(): kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, xi=xi)
    def adapt(): st[0] += 1; kernels.sghmc_step(theta, V, grad, tau, gg, vh, minv, None, 0.01, 1e5, 0.05, True, seed=1, step=st[0])
    def fill(): st[0] += 1; kernels.philox_normal(xi, 1, st[0])
    for name, fn, arrays in (("frozen philox", frozen, 6), ("frozen injected", frozen_inj, 7), ("adapt philox", adapt, 12), ("normal fill", fill, 1)):
        t = timeit(fn)
        print("%s %-16s %8.1f us  %6.0f GB/s" % (str(dt).split(".")[1], name, t, arrays * bpp * n / t / 1e3))

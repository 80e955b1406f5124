"""Sweep launch geometry of the fused update kernels on the GPU box (dev tool).

usage: python tools/tune_launch.py [n_params] [f32|f64] > gpurun_out/tune.txt
Times each configuration with torch.cuda.Event over `iters` launches on the
current stream; prints achieved algorithmic GB/s.
"""
import sys
import itertools
import torch
sys.path.insert(0, ".")
from pysgmcmc_amd import kernels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_002_434
dt = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
scale = 2 if dt == torch.float64 else 1                      # bytes per element / 4
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda: torch.randn(n, device=dev, generator=g, dtype=dt)
theta, V, grad = mk() * 0.02, torch.zeros(n, device=dev, dtype=dt), mk() * 0.1
tau, gg, vh = torch.ones(n, device=dev, dtype=dt), torch.ones(n, device=dev, dtype=dt), torch.ones(n, device=dev, dtype=dt)
minv = torch.rand(n, device=dev, generator=g, dtype=dt) * 1.5 + 0.5
xi = mk()


def timeit(fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


step = [0]
def frozen():
    step[0] += 1
    kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=1, step=step[0])
def adapt():
    step[0] += 1
    kernels.sghmc_step(theta, V, grad, tau, gg, vh, minv, None, 0.01, 1e5, 0.05, True, seed=1, step=step[0])
def frozen_inj():
    kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, xi=xi)
def sgld_frozen():
    step[0] += 1
    kernels.sgld_step(theta, grad, None, None, None, minv, None, 0.01, 1.0, 1e5, False, seed=1, step=step[0])
def rsghmc():
    step[0] += 1
    kernels.rsghmc_step(theta, V, grad, 0.001, 1.0, 1.0, 1.0, 0.0, seed=1, step=step[0])
def copy():
    V.copy_(theta)

print("n =", n, dt)
t = timeit(copy)
print("torch copy_ (%d B/param): %.1f us  %.0f GB/s" % (8 * scale, t, 8 * scale * n / t / 1e3))
cases = [("sghmc_frozen", frozen, 24), ("sghmc_adapt", adapt, 48), ("sghmc_frozen_injected", frozen_inj, 28),
         ("sgld_frozen", sgld_frozen, 16), ("rsghmc", rsghmc, 20)]
for name, fn, bpp in cases:
    best = None
    for bt, qpt, mb, nt in itertools.product([128, 256] if scale == 2 else [256], [1, 2, 4] if scale == 1 else [1, 2],
                                             [1024, 2048, 4096, 8192, 1 << 20] if scale == 1 else [8192, 1 << 20], [0, 1]):
        kernels.set_launch_config(bt, qpt, mb, nt)
        t = timeit(fn, iters=60, warm=5)
        gbs = bpp * scale * n / t / 1e3
        print("%-22s bt=%d qpt=%d max_blocks=%-7d nt=%d : %8.1f us  %7.0f GB/s  (%.1f%% of 8 TB/s)" % (
            name, bt, qpt, mb, nt, t, gbs, gbs / 80.0))
        if best is None or t < best[0]:
            best = (t, bt, qpt, mb, nt)
    print("BEST %s: %.1f us bt=%d qpt=%d max_blocks=%d nt=%d" % ((name,) + best))
    sys.stdout.flush()

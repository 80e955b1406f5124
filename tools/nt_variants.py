"""A/B of non-temporal hint placement (loads only / stores only / both / none) for the update kernels (dev tool).
Each variant is a separate build of the same ABI (-DSGMCMC_NT_LOADS=0 / -DSGMCMC_NT_STORES=0), selected with
PYSGMCMC_AMD_LIB; interleaved rounds."""
import os, subprocess, sys
CHILD = r'''
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
    st = [0]
    def frozen(): st[0] += 1; kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=1, step=st[0])
    def adapt(): st[0] += 1; kernels.sghmc_step(theta, V, grad, tau, gg, vh, minv, None, 0.01, 1e5, 0.05, True, seed=1, step=st[0])
    for name, fn, bpp in (("frozen", frozen, 24), ("adapt", adapt, 48)):
        for nt in (0, 1):
            kernels.set_launch_config(256, 1, 1 << 20, nt)
            for _ in range(10): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(100): fn()
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / 100 * 1e3
            print("%-11s n=%d %-6s nt=%d : %7.1f us %6.0f GB/s" % (sys.argv[1], n, name, nt, us, bpp * n / us / 1e3))
'''
for rnd in range(2):
    for label, lib in (("both", None), ("loads-only", "build/lib_ntloads.so"), ("stores-only", "build/lib_ntstores.so")):
        env = dict(os.environ)
        if lib:
            env["PYSGMCMC_AMD_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-c", CHILD, label], env=env, capture_output=True, text=True)
        print(out.stdout, end="")
        if out.returncode:
            print(out.stderr[-500:])

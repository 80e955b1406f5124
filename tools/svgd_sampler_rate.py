"""next(SVGDSampler) rate in the reference's toy regime (50 particles x 2 parameters, Gaussian target):
per-particle cost function (auto-batched with torch.func.vmap), explicitly batched cost, and the kernels alone."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysgmcmc_amd.samplers import SVGDSampler  # noqa: E402

dev = torch.device("cuda:0")
x0 = np.random.RandomState(0).normal(size=(50, 2))


def run(cost, label, force_loop=False, graph=False):
    s = SVGDSampler(particles=[torch.tensor(r, device=dev) for r in x0], cost_fun=cost, dtype=torch.float32)
    s.sample_format = "view"
    if force_loop:
        s._vmapped = False
    s.use_hip_graph = graph
    for _ in range(20):
        next(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        next(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-46s %8.1f us per step (%.0f steps/s)" % (label, dt * 1e6, 1 / dt))


per_particle = lambda p: 0.5 * (p ** 2).sum()
batched = lambda P: 0.5 * (P ** 2).sum(dim=1)
batched.batched = True
run(per_particle, "per-particle cost, particle-by-particle loop", force_loop=True)
run(per_particle, "per-particle cost, auto-batched (vmap)")
run(batched, "batched cost function")
try:
    run(batched, "batched cost function, cost pipeline in a hipGraph", graph=True)
    run(batched, "batched cost function, whole step in a hipGraph", graph="full")
except Exception as e:                                   # noqa: BLE001
    print("hipGraph mode failed:", repr(e)[:300])

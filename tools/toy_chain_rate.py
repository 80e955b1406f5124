"""next(sampler) rate on the reference's toy target (banana, 2 scalar parameters): eager / cost graph / whole-step graph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysgmcmc_amd.diagnostics.objective_functions import banana_log_likelihood  # noqa: E402
from pysgmcmc_amd.sampling import Sampler  # noqa: E402

dev = "cuda:0"
cost = lambda params: -banana_log_likelihood(params)
for method in (Sampler.SGHMC, Sampler.SGLD, Sampler.RelativisticSGHMC):
    for graph in (False, True, "full"):
        params = [torch.tensor(0.0, device=dev), torch.tensor(6.0, device=dev)]
        s = Sampler.get_sampler(method, params=params, cost_fun=cost, dtype=torch.float32, seed=1)
        s.sample_format = "view"
        s.use_hip_graph = graph
        for _ in range(50):
            next(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 2000
        for _ in range(n):
            next(s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("%-18s use_hip_graph=%-5s : %7.1f us per step (%.0f samples/s)" % (method.value, graph, dt * 1e6, 1 / dt))

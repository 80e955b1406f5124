"""The reference's toy target (diagnostics/objective_functions.py: banana) with every sampler of the package,
constructed through the reference's factory `Sampler.get_sampler`. SGLD / SGHMC / relativistic SGHMC draw a
chain; SVGD moves a set of particles."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from itertools import islice

import numpy as np
import torch

from pysgmcmc_amd.diagnostics.objective_functions import banana_log_likelihood
from pysgmcmc_amd.sampling import Sampler

dev = "cuda:0"
cost = lambda params: -banana_log_likelihood(params)

for method, kw, n in ((Sampler.SGHMC, dict(burn_in_steps=1000), 5000),
                      (Sampler.SGLD, dict(burn_in_steps=1000), 5000),
                      (Sampler.RelativisticSGHMC, dict(), 5000)):
    params = [torch.tensor(0.0, device=dev), torch.tensor(6.0, device=dev)]
    sampler = Sampler.get_sampler(method, params=params, cost_fun=cost, dtype=torch.float32, seed=1, **kw)
    sampler.sample_format = "device"
    sampler.use_hip_graph = "full"                       # replay the whole step from one hipGraph
    draws = torch.stack([torch.stack(s) for s, _ in islice(sampler, n)])[n // 2:].cpu().numpy()
    print("%-18s mean (%6.2f, %6.2f)  std (%5.2f, %5.2f)" % (method.value, *draws.mean(0), *draws.std(0)))

x0 = np.random.RandomState(0).normal(size=(50, 2)) + np.array([0.0, 6.0])
svgd = Sampler.get_sampler(Sampler.SVGD, particles=[torch.tensor(r, device=dev) for r in x0],
                           cost_fun=lambda p: -banana_log_likelihood([p[0], p[1]]), dtype=torch.float32)
svgd.sample_format = "device"
for particles, costs in islice(svgd, 2000):
    pass
P = torch.stack(particles).cpu().numpy()
print("%-18s mean (%6.2f, %6.2f)  std (%5.2f, %5.2f)   (50 particles, 2000 steps)" % ("SVGD", *P.mean(0), *P.std(0)))

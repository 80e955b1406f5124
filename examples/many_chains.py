"""256 SGHMC chains of the reference's default BNN advanced together (one workgroup per chain), then the
Gelman-Rubin statistic across them -- the job `pysgmcmc/diagnostics/sample_chains.py` does chain after chain."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import time

import numpy as np
import torch

from pysgmcmc_amd.diagnostics.sampler_diagnostics import gelman_rubin_from_chains
from pysgmcmc_amd.samplers.fused_chains import FusedBNNChains

rng = np.random.RandomState(1)
X = rng.rand(100, 1)
y = np.sinc(X * 10 - 5).sum(axis=1)
chains = FusedBNNChains.for_dataset(X, y, n_chains=256, burn_in_steps=1000, seed=7)
t0 = time.perf_counter()
chains.steps(5000)                                       # burn-in and mixing
snaps = chains.collect(50, every=100)                    # [256 chains, 50 snapshots, 5252 parameters]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
# R-hat of the network's prediction at x = 0.5 is more telling than of single weights (weight-space symmetries)
rhat = gelman_rubin_from_chains(snaps[:, :, -1:])        # the log-variance parameter
print("%d chains x %d steps in %.2f s = %.2f M samples/s; R-hat of the noise log-variance: %.3f"
      % (chains.n_chains, chains.n_iterations, dt, chains.n_chains * chains.n_iterations / dt / 1e6, float(rhat)))

"""The reference's ESS-vs-stepsize experiment (docs/source/experiments/compute_ess.py:176-246) for one stepsize per target,
in seconds instead of hours: RelativisticSGHMCSampler on gmm2 / gmm3 / banana, 20 consecutive segments x 10 000 kept samples,
every 10th of 2e6 steps, 3 chains (seeds) at once -- the samplers are built through the public API and advanced by ONE
kernel launch per segment (BuiltinTargetChains). Prints next to the values the reference's repository holds
(docs/source/notebooks/data/effective_sample_sizes/Relativistic_SGHMC.json, mean of its 5 runs)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysgmcmc_amd.diagnostics.objective_functions import (  # noqa: E402
    banana_log_likelihood, gmm2_log_likelihood, gmm3_log_likelihood, to_negative_log_likelihood)
from pysgmcmc_amd.diagnostics.sampler_diagnostics import effective_n  # noqa: E402
from pysgmcmc_amd.samplers import RelativisticSGHMCSampler  # noqa: E402
from pysgmcmc_amd.samplers.builtin_target_chains import BuiltinTargetChains  # noqa: E402
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule  # noqa: E402

dev = torch.device("cuda:0")
ref = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_outputs.json")))
curves = ref["ess_relativistic_sghmc"]["curves"]
for target, fn, start, eps in (("gmm2", gmm2_log_likelihood, [0.0], "1.51"), ("gmm3", gmm3_log_likelihood, [0.0], "2.01"),
                               ("banana", banana_log_likelihood, [0.0, 6.0], "1.51")):
    t0 = time.time()
    samplers = [RelativisticSGHMCSampler(params=[torch.tensor(v, dtype=torch.float32, device=dev) for v in start],
                                         cost_fun=to_negative_log_likelihood(fn), stepsize_schedule=ConstantStepsizeSchedule(float(eps)),
                                         session=dev, dtype=torch.float32, seed=seed) for seed in (1, 2, 3)]
    runner = BuiltinTargetChains(samplers)
    segments = torch.stack([runner.run(99_991, keep_every=10) for _ in range(20)])     # (segment, kept, chain, dim)
    ess = [np.mean([effective_n(segments[:, :, c, k]) for k in range(segments.shape[3])]) for c in range(len(samplers))]
    print("%-6s stepsize %s: ESS of 200 000 kept samples = %s  (reference: %.0f)   [%.1f s for 3 x 2e6 steps + the ESS estimate]" % (
        target, eps, ", ".join("%.0f" % e for e in ess), np.mean(curves[target][eps]), time.time() - t0))

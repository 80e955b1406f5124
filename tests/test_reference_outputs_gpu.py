"""The HIP RelativisticSGHMCSampler against the reference's own ESS data (tests/golden/reference_outputs.json) through
the public API, with the protocol of docs/source/experiments/compute_ess.py at a reduced length.

Chain of evidence: reference outputs == oracle at the FULL protocol (tests/test_reference_outputs.py, CPU);
oracle == HIP kernels bit for bit per step (tests/test_hip_parity.py); here: HIP sampler == oracle == reference in the
statistic itself. ESS is proportional to the number of samples when it is a small fraction of them, so the reduced
run (10 segments x 1 000 kept samples, every 10th of 1e5 steps) is compared (a) with the oracle on the SAME reduced
protocol (several seeds) and (b) with the reference's ESS per kept sample. Tolerance: 20 % (ESS of 1e4 samples has a
run-to-run spread of ~7 %).
"""
import json
import os
from itertools import islice

import numpy as np
import pytest
import torch

from test_reference_outputs import REF, reference_protocol_ess

pytestmark = pytest.mark.gpu


def _hip_protocol_ess(gpu, oracle, target, eps, seed, n_chains, samples_per_chain, keep_every):
    from pysgmcmc_amd.diagnostics.objective_functions import (
        banana_log_likelihood, gmm2_log_likelihood, gmm3_log_likelihood, to_negative_log_likelihood)
    from pysgmcmc_amd.samplers import RelativisticSGHMCSampler
    from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
    fn = {"banana": banana_log_likelihood, "gmm2": gmm2_log_likelihood, "gmm3": gmm3_log_likelihood}[target]
    start = [0.0, 6.0] if target == "banana" else [0.0]
    params = [torch.tensor(v, dtype=torch.float32, device=gpu) for v in start]        # compute_ess.py:214-224
    s = RelativisticSGHMCSampler(stepsize_schedule=ConstantStepsizeSchedule(eps), params=params,
                                 cost_fun=to_negative_log_likelihood(fn), session=gpu, dtype=torch.float32, seed=seed)
    s.sample_format = "view"                       # kept samples are copied by torch.stack below, right when they are yielded
    s.use_hip_graph = "full"                       # the toy step is launch-bound; the whole step replays from one graph
    chains = []
    for _ in range(n_chains):                      # consecutive segments of ONE sampler, compute_ess.py:176-182,232-240
        seg = [torch.stack([v.reshape(()) for v in (smp if isinstance(smp, list) else [smp])])
               for smp, _ in islice(s, 0, samples_per_chain * keep_every, keep_every)]
        chains.append(torch.stack(seg).cpu().numpy())
    x = np.stack(chains).astype(np.float64)        # (m, n, dim)
    assert np.isfinite(x).all()
    return float(np.mean([oracle.effective_n(x[:, :, k]) for k in range(x.shape[2])]))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("target,eps", [("gmm2", "1.51"), ("gmm3", "2.01")])
def test_hip_relativistic_sampler_matches_reference_ess(gpu, oracle, target, eps):
    m, n, keep = 10, 1000, 10
    got = np.array([_hip_protocol_ess(gpu, oracle, target, float(eps), seed=7 + s, n_chains=m, samples_per_chain=n,
                                      keep_every=keep) for s in range(2)])
    same_protocol = np.array([reference_protocol_ess(oracle, target, float(eps), seed=300 + s, n_chains=m,
                                                     samples_per_chain=n, keep_every=keep) for s in range(6)])
    ref_full = np.mean(REF["ess_relativistic_sghmc"]["curves"][target][eps])          # of 200 000 kept samples
    ref_scaled = ref_full * (m * n) / 200000.0
    assert abs(got.mean() / same_protocol.mean() - 1) < 0.20, (got, same_protocol)
    assert abs(got.mean() / ref_scaled - 1) < 0.20, (got, ref_scaled)

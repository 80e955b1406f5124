"""The HIP RelativisticSGHMCSampler against the reference's own ESS data (tests/golden/reference_outputs.json) at the
FULL protocol of docs/source/experiments/compute_ess.py:176-246 -- 20 consecutive segments x 10 000 kept samples, every
10th of 2e6 steps -- on gmm2 @ 1.51, gmm3 @ 2.01 and the banana @ 1.51, 3 seeds each, with the CPU test's bar (mean within
7 % of the reference's 5 runs, |z| < 4.5). The samplers are built through the public API with scalar parameters (the
reference's configuration) and stepped by the n-steps-per-launch toy path (``BuiltinTargetChains``: the update operators
and Philox stream of kernel K3), both with the default initial-momentum draw and with ``strict_reference_quirks = True``
(one host draw per parameter tensor, relativistic_sghmc.py:108-113).

Chain of evidence: reference outputs == oracle at the full protocol (tests/test_reference_outputs.py, CPU);
oracle == HIP kernels bit for bit per step (tests/test_hip_parity.py); toy path == public sampler API
(tests/test_builtin_target_chains_gpu.py); here: HIP == reference in the statistic itself, at full strength.
"""
import numpy as np
import pytest
import torch

from test_reference_outputs import REF

pytestmark = pytest.mark.gpu


def _device_protocol_ess(gpu, oracle, target, eps, seeds, strict, n_chains=20, samples_per_chain=10000, keep_every=10):
    from pysgmcmc_amd.diagnostics.objective_functions import (
        banana_log_likelihood, gmm2_log_likelihood, gmm3_log_likelihood, to_negative_log_likelihood)
    from pysgmcmc_amd.samplers import RelativisticSGHMCSampler
    from pysgmcmc_amd.samplers.builtin_target_chains import BuiltinTargetChains
    from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
    fn = {"banana": banana_log_likelihood, "gmm2": gmm2_log_likelihood, "gmm3": gmm3_log_likelihood}[target]
    start = [0.0, 6.0] if target == "banana" else [0.0]                                 # compute_ess.py:214-224
    samplers = []
    for seed in seeds:
        params = [torch.tensor(v, dtype=torch.float32, device=gpu) for v in start]
        s = RelativisticSGHMCSampler(stepsize_schedule=ConstantStepsizeSchedule(eps), params=params,
                                     cost_fun=to_negative_log_likelihood(fn), session=gpu, dtype=torch.float32, seed=seed)
        s.strict_reference_quirks = strict
        samplers.append(s)
    runner = BuiltinTargetChains(samplers)
    steps = (samples_per_chain - 1) * keep_every + 1                                    # what islice(sampler, 0, n * keep, keep) consumes
    segs = [runner.run(steps, keep_every) for _ in range(n_chains)]                     # consecutive segments of ONE sampler each
    x = torch.stack(segs).double().cpu().numpy()                                        # (segment, kept, seed, dim)
    assert np.isfinite(x).all() and x.shape[:2] == (n_chains, samples_per_chain)
    assert all(s.n_iterations == n_chains * steps for s in samplers)
    return np.array([np.mean([oracle.effective_n(x[:, :, j, k]) for k in range(x.shape[3])]) for j in range(len(seeds))])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("strict", [False, True])
@pytest.mark.parametrize("target,eps", [("gmm2", "1.51"), ("gmm3", "2.01"), ("banana", "1.51")])
def test_hip_relativistic_sampler_reproduces_the_reference_ess_at_the_full_protocol(gpu, oracle, target, eps, strict):
    ref = np.array(REF["ess_relativistic_sghmc"]["curves"][target][eps])
    assert len(ref) == 5
    got = _device_protocol_ess(gpu, oracle, target, float(eps), seeds=[2000 + s for s in range(3)], strict=strict)
    ratio = got.mean() / ref.mean()
    z = (got.mean() - ref.mean()) / np.sqrt(got.var(ddof=1) / len(got) + ref.var(ddof=1) / len(ref))
    assert 0.93 < ratio < 1.07, (target, eps, strict, ref, got)
    assert abs(z) < 4.5, (target, eps, strict, ref, got, z)

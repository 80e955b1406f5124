"""Several independent BNN chains sharing ONE GPU, stepped concurrently (pysgmcmc_amd.samplers.ConcurrentChains).

    python examples/concurrent_chains_one_gpu.py [n_chains]

The reference runs an ensemble's chains one after the other (pysgmcmc/diagnostics/sample_chains.py:369-382). Here every chain
has its own HIP stream and hipGraph and one host thread enqueues them round-robin: the second chain's kernels fill the parts of
the chip the first one's leave idle. Prints the ensemble's samples/s next to the same chains stepped one after the other, and
the Gelman-Rubin statistic of ALL parameters across the chains (Welford moments folded into every 10th update launch,
RhatExchange over the local chains: no collective, no host copy of any sample).
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysgmcmc_amd.data_batches import Placeholder, generate_batches  # noqa: E402
from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments, RhatExchange  # noqa: E402
from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params  # noqa: E402
from pysgmcmc_amd.samplers import ConcurrentChains, SGHMCSampler  # noqa: E402
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule  # noqa: E402

n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
X, y = torch.randn(50_000, 256, device=dev, generator=g), torch.randn(50_000, device=dev, generator=g)


def make_chain(k):
    xp, yp = Placeholder(dtype=torch.float32, device=dev), Placeholder(dtype=torch.float32, device=dev)
    params = init_mlp_params(256, hidden=(1024, 1024), seed=100 + k, dtype=torch.float32, device=dev)   # 1.3 M parameters
    s = SGHMCSampler(params=params, cost_fun=BNNCost(xp, yp, batch_size=256, n_examples=50_000),
                     batch_generator=generate_batches(X, y, xp, yp, batch_size=256, seed=k),
                     stepsize_schedule=ConstantStepsizeSchedule(0.01), burn_in_steps=100, scale_grad=50_000.0,
                     session=dev, dtype=torch.float32, seed=1234 + k)
    s.sample_format = "view"            # samples stay in HBM (the numpy format would synchronise every step)
    s.use_hip_graph = True
    return s


STEPS = 1000
rates = {}
for label in ("one after the other", "concurrently"):
    chains = [make_chain(k) for k in range(n_chains)]
    group = ConcurrentChains(chains)
    group.run(150)                      # burn-in + graph capture
    group.synchronize()
    moments = [ChainMoments(s.arena.n, dev) for s in chains]
    for s, m in zip(chains, moments):
        s.attach_moments(m, every=10)   # theta' of every 10th step is folded into the chain's moments inside the update launch
    t0 = time.perf_counter()
    if label == "concurrently":
        group.run(STEPS)
        group.synchronize()
    else:
        for s in chains:
            for _ in range(STEPS):
                next(s)
        torch.cuda.synchronize()
    rates[label] = n_chains * STEPS / (time.perf_counter() - t0)
    summary = None
    if n_chains > 1:
        group.join()
        exchange = RhatExchange(chains[0].arena.n, dev, mode="allreduce")
        exchange.start(moments)         # several local chains: their packs are added, nothing crosses a link
        summary = {k: round(v, 4) for k, v in exchange.finish(with_summary=True)[1].items()}
    print("%d chains %-20s %8.0f samples/s   R-hat over %d parameters x %d kept samples: %s" % (
        n_chains, label + ":", rates[label], chains[0].arena.n, moments[0].count, summary))
print("ensemble speed-up from sharing the GPU: %.2fx" % (rates["concurrently"] / rates["one after the other"]))

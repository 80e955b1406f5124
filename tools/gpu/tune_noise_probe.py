"""Run-to-run spread of a public-API 10 M-parameter chain and of TunableOp's picks (debug aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, pysgmcmc_amd
mode = sys.argv[1] if len(sys.argv) > 1 else "caller"
if mode == "caller":
    pysgmcmc_amd.configure_for_device_bound_chains(tuning_ms=int(os.environ.get("TUNE_MS", "30")), tuning_iters=int(os.environ.get("TUNE_ITERS", "20")))
    if "TUNE_ROT" in os.environ:
        import torch.cuda.tunable as _t
        _t.set_rotating_buffer_size(int(os.environ["TUNE_ROT"]))
from pysgmcmc_amd.data_batches import Placeholder, generate_batches
from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
from pysgmcmc_amd.samplers import SGHMCSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
import torch.cuda.tunable as tunable
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
X, y = rng.randn(100000, 784).astype(np.float32), rng.randn(100000).astype(np.float32)
xp, yp = Placeholder(dtype=torch.float32, device=dev), Placeholder(dtype=torch.float32, device=dev)
params = init_mlp_params(784, hidden=(2048, 2048, 2048), seed=0, dtype=torch.float32, device=dev)
s = SGHMCSampler(params=params, cost_fun=BNNCost(xp, yp, batch_size=256, n_examples=100000),
                 batch_generator=generate_batches(X, y, xp, yp, batch_size=256, seed=0), stepsize_schedule=ConstantStepsizeSchedule(1e-3),
                 burn_in_steps=8, mdecay=0.05, scale_grad=1e5, session=dev, dtype=torch.float32, seed=1)
s.sample_format = "view"
for _ in range(60): next(s)
best = 0.0
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): next(s)
    torch.cuda.synchronize(); best = max(best, 300 / (time.perf_counter() - t0))
res = tunable.get_results()
picks = ["%s:%s" % (r[1].split("_")[0] + r[1][-14:], r[2][-28:]) for r in res]
print("RATE %.0f  tuning=%s  picks: %s" % (best, s.cost_fun.gemm_tuning_applied, " | ".join(p[-10:] for p in sorted(picks))))

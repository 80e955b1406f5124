"""Steps/s of BASELINE.json configs[1] (SGHMC on the 3x50 tanh sinc BNN, 5 252 params, fp32) in the three
stepping modes, plus the CPU oracle's rate for the same update (dev tool)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from itertools import islice
from pysgmcmc_amd.data_batches import Placeholder, generate_batches
from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
from pysgmcmc_amd.samplers import SGHMCSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

dev = torch.device("cuda:0")
rng = np.random.RandomState(1)
X = rng.rand(100, 1); y = np.sinc(X * 10 - 5).sum(axis=1)

def chain(mode, dtype=torch.float32):
    xp, yp = Placeholder(dtype=dtype, device=dev), Placeholder(dtype=dtype, device=dev)
    s = SGHMCSampler(params=init_mlp_params(1, seed=3, dtype=dtype, device=dev),
                     cost_fun=BNNCost(xp, yp, batch_size=20, n_examples=100),
                     batch_generator=generate_batches(X, y, xp, yp, 20, seed=1),
                     stepsize_schedule=ConstantStepsizeSchedule(0.01), burn_in_steps=100, scale_grad=100.0,
                     session=dev, dtype=dtype, seed=1)
    s.sample_format = "view"
    s.use_hip_graph = mode
    return s

for dtype in (torch.float32, torch.float64):
    for mode in (False, True, "full"):
        s = chain(mode, dtype)
        list(islice(s, 300))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 3000
        for _ in islice(s, n): pass
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%s use_hip_graph=%-5s : %8.0f samples/s  (%.1f us/step)" % (str(dtype).split(".")[1], mode, n / dt, dt / n * 1e6))

    s = chain(False, dtype)
    s.fused_bnn_steps(300)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): s.fused_bnn_steps(100)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s fused kernel, 100 steps/launch : %8.0f samples/s  (%.1f us/step)" % (str(dtype).split(".")[1], 3000 / dt, dt / 3000 * 1e6))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000): s.fused_bnn_steps(1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s fused kernel, 1 step/launch    : %8.0f samples/s  (%.1f us/step)" % (str(dtype).split(".")[1], 1000 / dt, dt / 1000 * 1e6))

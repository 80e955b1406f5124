"""Fused small-BNN kernel: step time vs number of concurrent chains (one workgroup each) (dev tool)."""
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from pysgmcmc_amd import kernels
dev = torch.device("cuda:0")
sizes = [1, 50, 50, 50, 1]
P = sum(sizes[l] * sizes[l + 1] + sizes[l + 1] for l in range(4)) + 1
stride = ((P + 63) // 64) * 64
g = torch.Generator(device=dev).manual_seed(0)
X = torch.rand(100, 1, device=dev, generator=g); y = torch.sin(X[:, 0] * 6)
n_steps = 200
for dt in (torch.float32,):
    for nc in (1, 4, 32, 256, 1024):
        rows = {k: torch.zeros(nc * stride, device=dev, dtype=dt) for k in ("V", "grad")}
        rows.update({k: torch.ones(nc * stride, device=dev, dtype=dt) for k in ("tau", "g", "v_hat", "minv")})
        rows["theta"] = (torch.randn(nc * stride, device=dev, generator=g) * 0.2).to(dt)
        starts = torch.randint(0, 81, (nc * n_steps,), device=dev, generator=g, dtype=torch.int32)
        costs = torch.empty(nc * n_steps, device=dev, dtype=dt)
        def run(first):
            kernels.bnn_fused_sghmc_steps(rows["theta"], rows["V"], rows["grad"], rows["tau"], rows["g"], rows["v_hat"],
                                          rows["minv"], sizes, X.to(dt), y.to(dt), starts, 20, 20, 100, 1.0, 1e-6, 0.01,
                                          0.01, 100.0, 0.05, first, n_steps, 50, 7, costs, n_chains=nc, chain_stride=stride)
        run(0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(5): run(200 * (r + 1))
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / (5 * n_steps)
        print("%s chains=%5d : %6.1f us per step per chain-group -> %10.0f samples/s aggregate" % (str(dt).split(".")[1], nc, dtm * 1e6, nc / dtm))

# the same through the sampler-level group (FusedBNNChains): chains built like BayesianNeuralNetwork's default set-up
from pysgmcmc_amd.samplers.fused_chains import FusedBNNChains  # noqa: E402
import time  # noqa: E402

rng = np.random.RandomState(1)
Xs = rng.rand(100, 1)
ys = np.sinc(Xs * 10 - 5).sum(axis=1)
for m in (1, 64, 256):
    grp = FusedBNNChains.for_dataset(Xs, ys, m, seed=3, device="cuda:0")
    grp.steps(100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        grp.steps(100)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 1000
    print("FusedBNNChains %4d chains: %6.1f us per step of all chains -> %10.0f samples/s" % (m, dt * 1e6, m / dt))

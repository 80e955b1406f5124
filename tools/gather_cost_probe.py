"""Upper bound of what folding the per-step minibatch window gather into another launch can save: the 10 M-parameter chain
stepped with and without the gather launch (timing only: without it the chain keeps seeing one window)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pysgmcmc_amd import kernels
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning

dev = torch.device("cuda:0")
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)
real = kernels.window_gather
for label, fn in (("with the gather launch", real), ("without", lambda *a, **k: None), ("with", real), ("without", lambda *a, **k: None)):
    s = bench.build_chain(dev, 0, "bnn10m-sghmc", burn_in=8)
    s.sample_format = "view"
    s.use_hip_graph = True
    s.collect_stats = "theta_sq"
    for _ in range(150):
        next(s)
    kernels.window_gather = fn
    for _ in range(20):
        next(s)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(400):
            next(s)
        e1.record()
        torch.cuda.synchronize()
        res.append(round(e0.elapsed_time(e1) / 400 * 1e3, 1))
    kernels.window_gather = real
    print("%-24s device us/step %s" % (label, res), flush=True)
    del s
    torch.cuda.empty_cache()

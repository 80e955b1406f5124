"""Does the first layer's bias gradient from the [x | 1]^T delta product (BNNCost.bias_gradient_from_product) also pay on the 49.8 M-parameter
net of configs[4] (512 inputs: 513 rows start a new macro-tile row)? Device us per step of the bnn50m-sgld chain with and without:
788 -> 780 us (the saved column-sum pass outweighs the extra tile row)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning
dev = torch.device("cuda:0")
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)
for label, flag in (("ones-row off", False), ("ones-row on", True), ("off", False), ("on", True)):
    s = bench.build_chain(dev, 0, "bnn50m-sgld", burn_in=8)
    s.sample_format, s.use_hip_graph, s.collect_stats = "view", True, "theta_sq"
    s.cost_fun.bias_gradient_from_product = flag
    for _ in range(60):
        next(s)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            next(s)
        e1.record()
        torch.cuda.synchronize()
        res.append(round(e0.elapsed_time(e1) / 100 * 1e3, 1))
    print("%-14s device us/step: %s" % (label, res), flush=True)
    del s
    torch.cuda.empty_cache()

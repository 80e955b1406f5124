"""Step time of the 10 M-parameter bench chain with the hidden layers' forward products as ONE launch each (BNNCost.fused_dense:
kernels.bnn_dense_tanh, product + bias + tanh [+ the output unit's dot product]) against library GEMM + activation launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning
dev = torch.device("cuda:0")
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)
for label, fused in (("library mm + bias_tanh / rowdot", False), ("fused dense layers (default)", True), ("library again", False), ("fused again", True)):
    s = bench.build_chain(dev, 0, os.environ.get("PROBE_WORKLOAD", "bnn10m-sghmc"), burn_in=8)
    s.sample_format, s.use_hip_graph, s.collect_stats = "view", True, "theta_sq"
    s.cost_fun.fused_dense = fused
    for _ in range(150):
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
    print("%-40s device us/step: %s   theta finite: %s" % (label, res, bool(torch.isfinite(s.arena.row("theta")).all())), flush=True)
    del s
    torch.cuda.empty_cache()

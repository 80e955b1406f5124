"""Step time of the 10 M-parameter bench chain: cost graph + one update launch (the default) against the update fused into
the weight-gradient GEMMs (sampler.fuse_update_into_gemm), with the library's and with the hand-written gW products."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.experiments import gemm_kernels      # sets PYSGMCMC_AMD_LIB to the experiments build
from tools.experiments.stepping import upgrade
import torch
import bench

dev = torch.device("cuda:0")
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)
for label, fused, gw in (("library GEMMs + one K1 launch (default)", False, "blas"), ("hand-written gW GEMMs + one K1 launch", False, "mfma"),
                         ("update fused into the gW GEMMs", True, "blas")):
    s = bench.build_chain(dev, 0, os.environ.get("PROBE_WORKLOAD", "bnn10m-sghmc"), burn_in=8)
    s = upgrade(s)
    s.sample_format = "view"
    s.use_hip_graph = True
    s.collect_stats = "theta_sq"
    s.cost_fun.gw_gemm = gw
    s.fuse_update_into_gemm = fused
    for _ in range(150):
        next(s)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        for _ in range(200):
            next(s)
        host = time.perf_counter() - t0
        e1.record()
        torch.cuda.synchronize()
        res.append((round(e0.elapsed_time(e1) / 200 * 1e3, 1), round(host / 200 * 1e6, 1)))
    print("%-46s device us/step, (host enqueue us/step): %s" % (label, res), flush=True)
    assert torch.isfinite(s.arena.row("theta")).all()
    del s
    torch.cuda.empty_cache()

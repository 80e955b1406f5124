"""Step time of the 10 M-parameter bench chain under the hipGraph modes: cost graph + direct update (True), the whole
step in one graph ("full"), and the update forked layer by layer onto a side stream (inside the graph, or between graph
segments), with the full-occupancy update kernel and with small persistent grids that leave the CUs to the GEMMs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.experiments import gemm_kernels      # sets PYSGMCMC_AMD_LIB to the experiments build
from tools.experiments.stepping import upgrade
import torch
import bench
from pysgmcmc_amd import kernels

dev = torch.device("cuda:0")
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)
cases = [("cost graph + direct update", True, False, None), ("full graph", "full", False, None),
         ("full graph + in-graph overlap", "full", True, None)]
for mb in (128, 256, 512, 1024):
    for qpt in (2, 4):
        cases.append(("full + overlap, grid cap %d, %d quads/lane" % (mb, qpt), "full", True, dict(max_blocks=mb, quads_per_thread=qpt, block_threads=256)))
cases.append(("segmented graphs + side stream", True, True, None))
only = os.environ.get("PROBE_ONLY")
for label, graph, overlap, geom in cases:
    if only and only not in label:
        continue
    s = bench.build_chain(dev, 0, "bnn10m-sghmc", burn_in=8)
    s = upgrade(s)
    s.sample_format = "view"
    s.use_hip_graph = graph
    s.overlap_update = overlap
    s.collect_stats = "theta_sq"
    if geom:
        s.launch = kernels.LaunchConfig(**geom)
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

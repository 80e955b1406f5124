"""Independent 10 M-parameter SGHMC chains sharing ONE GPU, each on its own stream with its own hipGraph, stepped round-robin by
one host thread: aggregate samples/s against the single chain (every kernel of a step leaves part of the chip idle -- the M = 256
GEMMs run the matrix pipe at ~62 % -- so a second chain's launches can fill the gaps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning

dev = torch.device("cuda:0")
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)
STEPS = 300
MODE = {"full": "full"}.get(os.environ.get("PROBE_GRAPH", ""), True)    # PROBE_GRAPH=full: the update inside the graph (less host work per step)
print("use_hip_graph =", MODE)
for n_chains in (1, 2, 3, 4):
    chains, streams = [], []
    for c in range(n_chains):
        s = bench.build_chain(dev, c, "bnn10m-sghmc", burn_in=8)
        s.sample_format = "view"
        s.use_hip_graph = MODE
        s.collect_stats = "theta_sq"
        chains.append(s)
        streams.append(torch.cuda.Stream(device=dev))
    for s, st in zip(chains, streams):
        with torch.cuda.stream(st):
            for _ in range(60):
                next(s)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(STEPS):
            for s, st in zip(chains, streams):
                with torch.cuda.stream(st):
                    next(s)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        res.append((round(n_chains * STEPS / wall, 1), round(host / STEPS * 1e6, 1)))
    print("%d chain(s) on one GPU: aggregate samples/s (host enqueue us per round): %s" % (n_chains, res), flush=True)
    for s in chains:
        assert torch.isfinite(s.arena.row("theta")).all()
    del chains, streams
    torch.cuda.empty_cache()

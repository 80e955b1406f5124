"""What bench.py's own instrumentation costs per step of the 10 M-parameter chain: the bare loop of next(sampler) against the
loop with (a) the per-launch kernel timer, (b) host flow control, (c) the fused Welford moments + trace every 10 steps, and the
"full" graph mode (update inside the graph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pysgmcmc_amd.profiling import UpdateKernelTimer
from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments

dev = torch.device("cuda:0")
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)
STEPS = 400
for label, graph, timer_every, depth, mom in (("bare loop", True, 0, 0, 0), ("kernel timer on every launch", True, 1, 0, 0),
                                              ("kernel timer on every 8th launch", True, 8, 0, 0),
                                              ("flow control (depth 64) on torch events", True, 0, 64, 0),
                                              ("timer every launch + flow control on its events", True, 1, 64, 0),
                                              ("moments + trace every 10 steps", True, 0, 0, 10),
                                              ("everything (bench.py)", True, 1, 64, 10),
                                              ("full graph mode, bare loop", "full", 0, 0, 0)):
    s = bench.build_chain(dev, 0, "bnn10m-sghmc", burn_in=8)
    s.sample_format = "view"
    s.use_hip_graph = graph
    s.collect_stats = "theta_sq"
    n = s.arena.n
    moments = ChainMoments(n, dev)
    coords = torch.tensor([0, n // 2, n - 1], device=dev)
    trace = torch.zeros(STEPS, 4, device=dev)
    timer = UpdateKernelTimer(device=dev)
    timer.sample_every = max(timer_every, 1)
    s.kernel_timer = timer
    for _ in range(150):
        next(s)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        timer.kevents, timer.tags = [], []
        timer.reserve(STEPS)
        timer.enabled = bool(timer_every)
        ends, kept = [], 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        for i in range(STEPS):
            s.attach_moments(moments if mom else None, mom or 1)
            _, cost = next(s)
            if mom and s.n_iterations % mom == 0:
                trace[kept, 0:1].copy_(cost.reshape(1))
                torch.index_select(s.arena.row("theta"), 0, coords, out=trace[kept, 1:4])
                kept += 1
            if depth:
                if timer_every == 1:
                    ends.append(timer.kevents[-1])
                else:
                    ev = torch.cuda.Event()
                    ev.record()
                    ends.append(ev)
                if i >= depth:
                    ends[i - depth].synchronize()
        host = time.perf_counter() - t0
        e1.record()
        torch.cuda.synchronize()
        timer.enabled = False
        res.append((round(e0.elapsed_time(e1) / STEPS * 1e3, 1), round(host / STEPS * 1e6, 1)))
    print("%-52s device us/step (host loop us/step): %s" % (label, res), flush=True)
    del s, moments, trace
    torch.cuda.empty_cache()

"""``--workload svgd16-10m``: the SVGD update path in the contract's format."""
import json
import sys
import time

import numpy as np
import torch

from benchlib.baselines import svgd_cpu_baseline
from benchlib.common import HBM_PEAK_GBS
from benchlib.workloads import WORKLOADS


def run_svgd(args, dev, rank, world, dist):
    """`--workload svgd16-10m`: one step = sgmcmc_svgd_step_f32 (kernel matrix + update, 4 launches) on
    n particles x 10 002 434 parameters with fixed synthetic gradients. Particles never leave HBM."""
    from pysgmcmc_amd import kernels
    spec = WORKLOADS[args.workload]
    layers = spec["layers"]
    n = spec["particles"]
    dim = sum(a * b + b for a, b in zip(layers, layers[1:] + (1,))) + 1
    ld = (dim + 63) // 64 * 64                                  # the sampler's row pitch
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    x = torch.randn(n * ld, device=dev, generator=g) * (1.0 / dim ** 0.5)
    grad = torch.randn(n * ld, device=dev, generator=g) * 0.1
    hist = torch.zeros_like(x)
    ws = kernels.svgd_workspace(n, x)
    step = lambda: kernels.svgd_step(x, grad, hist, n, dim, 1e-3, 0.9, 1e-6, ws, ld=ld, repulsion_sign=-1)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    pairs = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        pairs.append((e0, e1))
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(x).all()
    if rank == 0:
        us = float(np.mean([a.elapsed_time(b) for a, b in pairs])) * 1e3
        alg_bytes = 24 * n * dim                                # S1 reads X (4 B), S4 R{X,G,H} W{X,H} (20 B) per element
        achieved = alg_bytes / (us * 1e-6) / 1e9
        line = {
            "metric": "SVGD update-steps/sec + HBM GB/s (%% roofline), %d particles x BNN 10M params" % n,
            "value": round(world * args.steps / elapsed, 2), "unit": "update-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: SVGD step (pairwise distances, median bandwidth, kernel matrix, K[G|X] + AdaGrad "
                                   "update) on %d particles x %d parameters, row pitch %d, fixed synthetic gradients; "
                                   "1 particle set per GPU" % (args.workload, n, dim, ld),
                       "particles": n, "params": dim, "chains": world},
            "roofline": {"bound": "hbm", "kernel": "sgmcmc_svgd_step_f32 (svgd_gram_mfma16_kernel + svgd_update_mfma16_kernel; "
                                                   "per-kernel times in profiles/r01_svgd_kernel_stats.md)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # PMC passes of profiles/r01_svgd_pmc_traffic.md: S1 4.00 B, S4 12.00 + 8.00 B per element
                         "traffic": int(24.0 * n * dim), "traffic_source": "profiles/r01_svgd_pmc_traffic.md",
                         "algorithmic_bytes_per_launch": alg_bytes, "us_per_launch_mean": round(us, 2),
                         "launches_timed": len(pairs),
                         "timing": "hipEvent pair around every step (4 launches) of the timed region"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = svgd_cpu_baseline(n, dim, args.cpu_seconds)
        print(json.dumps(line))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()



"""SVGD step (sgmcmc_svgd_step_*) timing: kernel-matrix launches S1-S3 and the streaming update S4 for a few
(particles x parameters) shapes; algorithmic bytes = 4 B x n x d read by S1 + 20 B x n x d for S4."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pysgmcmc_amd import kernels

dev = torch.device("cuda:0")
shapes = [(50, 2), (16, 5252), (8, 10_002_434), (16, 10_002_434), (32, 10_002_434), (64, 10_002_434), (128, 1_000_000),
          (20, 50_000_000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for dt in ((torch.float64,) if os.environ.get("SVGD_F64") else (torch.float32,)):
    for n, d in shapes:
        ld = d if os.environ.get("SVGD_DENSE_PITCH") else (d + 63) // 64 * 64    # the sampler pads rows to 64 elements
        x = (torch.randn(n * ld, device=dev, dtype=dt) * (1.0 / d ** 0.5)).contiguous()
        g = torch.randn(n * ld, device=dev, dtype=dt) * 0.01
        h = torch.zeros_like(x)
        ws = kernels.svgd_workspace(n, x)
        for _ in range(3):
            kernels.svgd_step(x, g, h, n, d, 1e-3, 0.9, 1e-6, ws, ld=ld, repulsion_sign=-1)
        torch.cuda.synchronize()
        reps = 20 if n * d > 1e7 else 200
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(reps):
            kernels.svgd_kernel(x, n, d, ws, ld=ld, kernel_gradients=False)
        e1.record()
        for _ in range(reps):
            kernels.svgd_step(x, g, h, n, d, 1e-3, 0.9, 1e-6, ws, ld=ld, repulsion_sign=-1)
        e2.record()
        torch.cuda.synchronize()
        t_k = e0.elapsed_time(e1) / reps * 1e3
        t_s = e1.elapsed_time(e2) / reps * 1e3
        el = n * d * x.element_size()
        print("%s n=%4d d=%9d : kernel matrix %9.1f us (%6.0f GB/s of X), full step %9.1f us (%6.0f GB/s of 6 passes), "
              "update alone ~%9.1f us (%6.0f GB/s of 5 passes)"
              % (str(dt).split(".")[1], n, d, t_k, el / t_k * 1e-3, t_s, 6 * el / t_s * 1e-3, t_s - t_k,
                 5 * el / max(t_s - t_k, 1e-3) * 1e-3), flush=True)
        del x, g, h

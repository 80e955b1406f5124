"""Round 5: the fused layer launches on more tiles than compute units (configs[4]: 256 x 4864 x 4864 = 608 tiles,
two rounds of full tiles + 192 half tiles) against the library product + activation launch, isolated and in the 49.8 M-parameter chain's step.
Run on the GPU box: python tools/dense_rounds_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pysgmcmc_amd  # noqa: E402

pysgmcmc_amd.configure_for_device_bound_chains()
import torch  # noqa: E402

from benchlib.workloads import build_chain  # noqa: E402
from pysgmcmc_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")


def t_us(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, K, N in ((256, 4864, 4864), (256, 512, 4864), (512, 2048, 2048), (256, 2048, 2048)):
    g = torch.Generator(device=dev).manual_seed(0)
    h = torch.tanh(torch.randn(M, K, device=dev, generator=g))
    W = torch.randn(K, N, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g) * 0.1
    out, out2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    lib_mm = t_us(lambda: torch.mm(h, W, out=out2))
    lib = t_us(lambda: (torch.mm(h, W, out=out2), kernels.bias_tanh(out2, b)))
    fused = t_us(lambda: kernels.bnn_dense_tanh(h, W, b, out))
    d, act, o = torch.randn(M, N, device=dev, generator=g), torch.tanh(torch.randn(M, K, device=dev, generator=g)), torch.empty(M, K, device=dev)
    Wt = W.t().contiguous()                                       # backward reads W [N_out = K][K_in = N] along its rows
    parts = torch.empty(M // 32, K, device=dev)
    cs = torch.empty(K, device=dev)
    lib_b = t_us(lambda: (torch.mm(d, W.t(), out=o), kernels.tanh_backward_colsum(o, act, cs)))
    fused_b = t_us(lambda: kernels.bnn_dense_tanh_backward(d, W, act, o, colsum_parts=parts)) if K % 64 == 0 else float("nan")
    print("%4d x %4d x %4d  forward: library mm %.1f us, mm + bias_tanh %.1f, fused %.1f   | backward (delta W^T, %d x %d x %d): library mm + tanh' + colsum %.1f, fused %.1f"
          % (M, K, N, lib_mm, lib, fused, M, N, K, lib_b, fused_b), flush=True)

for workload in ("bnn50m-sgld", "bnn50m-rsghmc"):
    for fused in (False, True, False, True):
        s = build_chain(dev, 0, workload, burn_in=8)
        s.sample_format, s.use_hip_graph, s.collect_stats = "view", True, "theta_sq"
        s.cost_fun.fused_layers = fused
        for _ in range(40):
            next(s)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(60):
                next(s)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 60 * 1e6)
        plan = s.cost_fun.plan_summary(s.params, s.arena.grad_views, torch.zeros(1))
        print("%s fused_layers=%s: %.1f us per step   forward %s" % (workload, fused, best, plan["forward"]), flush=True)
        del s
        torch.cuda.empty_cache()

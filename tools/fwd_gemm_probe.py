"""Forward products of the 10 M-parameter BNN (batch 256): torch.addmm (bias in the GEMM epilogue, what the cost path uses)
against torch.mm (bias left to the tanh kernel), both with TunableOp; microseconds from a hipGraph of 20 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pysgmcmc_amd import kernels
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning

dev = torch.device("cuda:0")
enable_gemm_tuning(True, max_duration_ms=int(os.environ.get("TUNE_MS", "30")), max_iterations=int(os.environ.get("TUNE_ITERS", "20")))


def graph_us(fn, reps=20, loops=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(loops):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * loops) * 1e3


with torch.no_grad():
    for (B, K, N) in ((256, 784, 2048), (256, 2048, 2048)):
        h = torch.randn(B, K, device=dev)
        W = torch.randn(K, N, device=dev) * 0.02
        b = torch.randn(N, device=dev)
        out = torch.empty(B, N, device=dev)
        d = torch.randn(B, N, device=dev)
        dprev = torch.empty(B, K, device=dev)
        gW = torch.empty(K, N, device=dev)
        print("B=%d K=%d N=%d: addmm(b, h, W) %.1f us | mm(h, W) %.1f us | delta: mm(d, W^T) %.1f us | gW: mm(h^T, d) %.1f us | tanh_ %.1f us | bias_tanh %.1f us" % (
            B, K, N, graph_us(lambda: torch.addmm(b, h, W, out=out)), graph_us(lambda: torch.mm(h, W, out=out)),
            graph_us(lambda: torch.mm(d, W.t(), out=dprev)), graph_us(lambda: torch.mm(h.t(), d, out=gW)),
            graph_us(lambda: torch.tanh_(out)), graph_us(lambda: kernels.bias_tanh(out, b))), flush=True)

"""Feasibility of getting the bias gradient from the weight-gradient GEMM itself (activations with a column of ones, so that
[h | 1]^T delta = [gW ; gb] lands on the arena's [W | b] slice): the library product with 2049 rows against 2048 (TunableOp on)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning
dev = torch.device("cuda:0")
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)

def graph_us(fn, reps=20, loops=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(loops): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * loops) * 1e3

B = 256
for fan_in, fan_out in ((2048, 2048), (784, 2048)):
    hbuf = torch.tanh(torch.randn(B, fan_in + 4, device=dev)); hbuf[:, fan_in] = 1.0
    d = torch.randn(B, fan_out, device=dev) * 0.01
    out = torch.empty(fan_in + 1, fan_out, device=dev)
    h, haug = hbuf[:, :fan_in], hbuf[:, :fan_in + 1]
    hc = h.contiguous()
    t0 = graph_us(lambda: torch.mm(hc.t(), d, out=out[:fan_in]))
    t1 = graph_us(lambda: torch.mm(h.t(), d, out=out[:fan_in]))
    t2 = graph_us(lambda: torch.mm(haug.t(), d, out=out))
    ok = torch.allclose(out[fan_in], d.sum(dim=0), rtol=1e-4, atol=1e-5)
    print("fan_in %d: gW dense h %.2f us | pitched h (ld = fan_in + 4) %.2f us | [h | 1] with %d rows %.2f us   (bias row ok: %s)" % (
        fan_in, t0, t1, fan_in + 1, t2, ok), flush=True)

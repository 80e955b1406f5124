"""Where the 7.7 us of the fused loss head go (sgmcmc_bnn_head_last_layer_backward_f32 at 256 x 2048): the launch with the output unit's
mean as 32 partial dot products per row (what the fused forward layer leaves), as a plain vector, the backward part alone
(sgmcmc_bnn_last_layer_backward_f32) and the tanh' + column-sum launch of the same shape; device us per launch from hipGraphs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pysgmcmc_amd import kernels

dev = torch.device("cuda:0")
B, N = 256, 2048
g = torch.Generator(device=dev).manual_seed(0)
h = torch.tanh(torch.randn(B, N, device=dev, generator=g))
w = torch.randn(N, device=dev, generator=g) / N ** 0.5
parts = torch.randn(32, B, device=dev, generator=g) * 0.1
vec = parts.sum(0).contiguous()
y = torch.randn(B, device=dev, generator=g)
log_var = torch.full((1,), -3.0, device=dev)
tsq = torch.rand(16, dtype=torch.float64, device=dev)
last_bias, bias_prev = torch.zeros(1, device=dev), torch.zeros(N, device=dev)
cost, gs, gb, mse = (torch.zeros(1, device=dev) for _ in range(4))
delta, colsum, gw = torch.empty(B, N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
dvec = torch.randn(B, device=dev, generator=g)


def head(mean):
    kernels.bnn_head_last_layer_backward(mean, y, log_var, tsq, last_bias, 256.0, 60000.0, 1e7, 1.0, 1e-6, 0.01, w, h, bias_prev, 0.0,
                                         cost, gs, gb, mse, delta, colsum, gw, fold_prior_grad=True)


def graph_us(fn, reps=20, loops=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(loops):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(round(e0.elapsed_time(e1) / (reps * loops) * 1e3, 2))
    return res


print("head, mean as 32 partial dot products :", graph_us(lambda: head(parts)))
print("head, mean as a vector                :", graph_us(lambda: head(vec)))
print("last-layer backward alone             :", graph_us(lambda: kernels.bnn_last_layer_backward(dvec, w, h, delta, colsum, gw)))
print("tanh' + column sums of the same shape :", graph_us(lambda: kernels.tanh_backward_colsum(delta, h, colsum)))
print("bias_tanh (read + write 2 MB)         :", graph_us(lambda: kernels.bias_tanh(delta, bias_prev)))

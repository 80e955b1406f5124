"""The register-direct experiment kernel (tools/gpu/bnn_dense_reg.hip: operands from global memory straight into registers, no
LDS and no barriers in the K loop) against the product's LDS-ring launches (kernels.bnn_dense_tanh / bnn_dense_tanh_backward) and the
library products: correctness against fp64, then device microseconds per launch from hipGraphs of 20 back-to-back repetitions.
Build the kernel with ``make -C tools/gpu`` first."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pysgmcmc_amd import kernels
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning

_so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu", "libbnn_dense_reg_probe.so"))
_so.bnn_dense_reg_forward_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
_so.bnn_dense_reg_backward_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 8 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)


def reg_forward(h, W, b, out, P):
    rc = _so.bnn_dense_reg_forward_f32(h.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), h.shape[0], W.shape[1], h.shape[1],
                                       h.stride(0), W.stride(0), out.stride(0), P, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def reg_backward(d, W, act, out, P):
    rc = _so.bnn_dense_reg_backward_f32(d.data_ptr(), W.data_ptr(), act.data_ptr(), out.data_ptr(), d.shape[0], W.shape[0], d.shape[1],
                                        d.stride(0), W.stride(0), act.stride(0), out.stride(0), P, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


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
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(loops):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(round(e0.elapsed_time(e1) / (reps * loops) * 1e3, 2))
    return res


PS = [int(p) for p in os.environ.get("PROBE_P", "4,6,8,12").split(",")]
with torch.no_grad():
    g = torch.Generator(device=dev).manual_seed(0)
    for M, K, N in ((256, 2048, 2048), (256, 784, 2048), (64, 80, 128)):
        h = torch.tanh(torch.randn(M, K, device=dev, generator=g))
        W = torch.randn(K, N, device=dev, generator=g) / K ** 0.5
        b = torch.randn(N, device=dev, generator=g) * 0.1
        ref = torch.tanh(h.double() @ W.double() + b.double())
        out = torch.empty(M, N, device=dev)
        for P in PS:
            out.fill_(7.0)
            reg_forward(h, W, b, out, P)
            print("forward  %4d x %4d x %4d  P = %2d  max |reg - fp64| %.3g" % (M, K, N, P, float((out.double() - ref).abs().max())), flush=True)
        # backward: delta [M][K2] W2 [N][K2]
        d = torch.randn(M, N, device=dev, generator=g)
        W2 = torch.randn(K if K % 64 == 0 else 832, N, device=dev, generator=g) / N ** 0.5      # [fan-in rows][fan-out = contraction]
        act = torch.tanh(torch.randn(M, W2.shape[0], device=dev, generator=g))
        refb = (d.double() @ W2.double().t()) * (1 - act.double() ** 2)
        outb = torch.empty(M, W2.shape[0], device=dev)
        for P in PS:
            outb.fill_(7.0)
            reg_backward(d, W2, act, outb, P)
            print("backward %4d x %4d x %4d  P = %2d  max |reg - fp64| %.3g (scale %.3g)" % (M, N, W2.shape[0], P, float((outb.double() - refb).abs().max()), float(refb.abs().max())), flush=True)
        if M != 256:
            continue
        out2 = torch.empty(M, N, device=dev)
        print("  forward  us: library mm + bias_tanh      ", graph_us(lambda: (torch.mm(h, W, out=out2), kernels.bias_tanh(out2, b))))
        print("  forward  us: product LDS-ring launch     ", graph_us(lambda: kernels.bnn_dense_tanh(h, W, b, out2)))
        for P in PS:
            print("  forward  us: register-direct, P = %2d     " % P, graph_us(lambda: reg_forward(h, W, b, out, P)))
        if K % 64 == 0:
            print("  backward us: library mm + tanh_backward  ", graph_us(lambda: (torch.mm(d, W2.t(), out=outb), kernels.tanh_backward(outb, act))))
            print("  backward us: product LDS-ring launch     ", graph_us(lambda: kernels.bnn_dense_tanh_backward(d, W2, act, outb)))
            for P in PS:
                print("  backward us: register-direct, P = %2d     " % P, graph_us(lambda: reg_backward(d, W2, act, outb, P)))

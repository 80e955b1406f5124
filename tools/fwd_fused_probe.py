"""Gate measurement of VERDICT r03 item 1(a): ONE forward layer of the 10 M-parameter BNN at batch 256,
``out = tanh(h W + b)``, as the library GEMM + ``bias_tanh`` launch (today's cost path, TunableOp-picked solution) against
the experiment kernel ``tools/gpu/bnn_dense_tanh.hip`` (hand-written fp32 MFMA product with the activation as its epilogue;
build it with ``make -C tools/gpu``; result in ``profiles/r04_fwd_epilogue_probe.txt``).

Correctness against an fp64 product first; then microseconds from hipGraphs. Two regimes:
  * ``same``  : the same layer 20 times in one graph (weights stay in L2 / Infinity Cache);
  * ``chain`` : the three hidden layers of the net one after the other (784->2048->2048->2048, distinct weights,
                the last with the output unit's dot product), as the forward pass runs them.
``BNN_DENSE_TANH_PROBE=n`` selects a timing variant of the experiment kernel (1 no MFMAs, 2 no loads, 8 register-staged
operands, 10 ring of 6). Run under ``rocprofv3 --kernel-trace --stats`` for per-kernel durations."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pysgmcmc_amd import kernels
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning

_so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu", "libbnn_dense_tanh_probe.so")
_probe = ctypes.CDLL(_so)
_probe.bnn_dense_tanh_probe_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 3
_probe.bnn_dense_tanh_probe_f32.restype = ctypes.c_int


def bnn_dense_tanh(h, W, bias, out, w_next=None, dot_parts=None):
    rc = _probe.bnn_dense_tanh_probe_f32(h.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), h.shape[0], W.shape[1],
                                         h.shape[1], h.stride(0), W.stride(0), out.stride(0),
                                         w_next.data_ptr() if w_next is not None else None,
                                         dot_parts.data_ptr() if dot_parts is not None else None,
                                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0, "bnn_dense_tanh_probe_f32 failed: %d" % rc
    return out


dev = torch.device("cuda:0")
enable_gemm_tuning(True, max_duration_ms=int(os.environ.get("TUNE_MS", "30")), max_iterations=int(os.environ.get("TUNE_ITERS", "20")))
torch.manual_seed(0)


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
    B = 256
    sizes = [784, 2048, 2048, 2048]
    Ws = [torch.randn(k, n, device=dev) / k ** 0.5 for k, n in zip(sizes[:-1], sizes[1:])]
    bs = [torch.randn(n, device=dev) * 0.1 for n in sizes[1:]]
    w_out = torch.randn(2048, device=dev) / 2048 ** 0.5
    X = torch.randn(B, 784, device=dev)
    hs_lib = [torch.empty(B, n, device=dev) for n in sizes[1:]]
    hs_mfma = [torch.empty(B, n, device=dev) for n in sizes[1:]]
    dot_parts = torch.empty(2048 // 64, B, device=dev)
    out_lib = torch.empty(B, device=dev)

    # ---- correctness
    h = X
    for l in range(3):
        ref = torch.tanh(h.double() @ Ws[l].double() + bs[l].double())
        torch.mm(h, Ws[l], out=hs_lib[l])
        kernels.bias_tanh(hs_lib[l], bs[l])
        bnn_dense_tanh(h, Ws[l], bs[l], hs_mfma[l], w_next=w_out if l == 2 else None,
                               dot_parts=dot_parts if l == 2 else None)
        e_lib = (hs_lib[l].double() - ref).abs().max().item()
        e_mfma = (hs_mfma[l].double() - ref).abs().max().item()
        print("layer %d (K=%d): max |err| vs fp64  library %.3e   mfma+epilogue %.3e   (max |lib - mfma| %.3e)" % (
            l, sizes[l], e_lib, e_mfma, (hs_lib[l] - hs_mfma[l]).abs().max().item()))
        assert e_mfma < 5e-6, "fused forward layer is wrong"
        h = hs_mfma[l]
    dot_ref = hs_mfma[2].double() @ w_out.double()
    dot = dot_parts.double().sum(dim=0)
    print("output-unit dot product: max |err| %.3e" % (dot - dot_ref).abs().max().item())
    assert (dot - dot_ref).abs().max().item() < 1e-5

    # ---- same layer back to back
    for l in (0, 1):
        h_in = X if l == 0 else hs_lib[0]

        def lib_layer(l=l, h_in=h_in):
            torch.mm(h_in, Ws[l], out=hs_lib[l])
            kernels.bias_tanh(hs_lib[l], bs[l])

        t_mm = graph_us(lambda l=l, h_in=h_in: torch.mm(h_in, Ws[l], out=hs_lib[l]))
        t_lib = graph_us(lib_layer)
        t_mfma = graph_us(lambda l=l, h_in=h_in: bnn_dense_tanh(h_in, Ws[l], bs[l], hs_mfma[l]))
        print("same  K=%4d: library mm %.2f us, mm + bias_tanh %.2f us, mfma + epilogue %.2f us  (gain %.2f us)" % (
            sizes[l], t_mm, t_lib, t_mfma, t_lib - t_mfma), flush=True)

    # ---- the forward chain of the net
    def lib_chain():
        h = X
        for l in range(3):
            torch.mm(h, Ws[l], out=hs_lib[l])
            if l < 2:
                kernels.bias_tanh(hs_lib[l], bs[l])
            else:
                kernels.tanh_rowdot(hs_lib[l], w_out, out_lib, bias=bs[l])
            h = hs_lib[l]

    def mfma_chain():
        h = X
        for l in range(3):
            bnn_dense_tanh(h, Ws[l], bs[l], hs_mfma[l], w_next=w_out if l == 2 else None,
                                   dot_parts=dot_parts if l == 2 else None)
            h = hs_mfma[l]

    t_lib = graph_us(lib_chain, reps=5, loops=60)
    t_mfma = graph_us(mfma_chain, reps=5, loops=60)
    print("chain 784->2048->2048->2048: library (3 mm + 2 bias_tanh + bias_tanh_rowdot) %.2f us, mfma + epilogues (3 launches) "
          "%.2f us  (gain %.2f us per forward pass)" % (t_lib, t_mfma, t_lib - t_mfma), flush=True)

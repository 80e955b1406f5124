"""Backward step through a hidden layer: library GEMM (delta W^T) + tanh_backward_colsum against ONE launch
(kernels.bnn_dense_tanh_backward: tanh' and the per-row-tile column sums as the epilogue, the sums of the previous launch added
up on the side), isolated (hipGraph of 20 back-to-back repetitions, device time per repetition), then the
10 M-parameter chain's step time with the two backward launches fused and not (BNNCost.fused_dense_backward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pysgmcmc_amd import kernels
from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning
dev = torch.device("cuda:0")
enable_gemm_tuning(True, max_duration_ms=30, max_iterations=20)
M, K, N = 256, 2048, 2048
g = torch.Generator(device=dev).manual_seed(1)
delta = torch.randn(M, K, device=dev, generator=g)
W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
act = torch.tanh(torch.randn(M, N, device=dev, generator=g))
out, cs = torch.empty(M, N, device=dev), torch.empty(N, device=dev)
parts, parts2 = torch.zeros(M // 32, N, device=dev), torch.zeros(M // 32, N, device=dev)


def lib_pair():
    torch.mm(delta, W.t(), out=out)
    kernels.tanh_backward_colsum(out, act, cs)


def fused():                                         # column sums per row tile + the previous launch's sums added up on the side
    kernels.bnn_dense_tanh_backward(delta, W, act, out, colsum_parts=parts, finish=(parts2, cs, None, 0.0))


def fused_plain():                                   # no column sums (the first layer's bias gradient comes from [x | 1]^T delta)
    kernels.bnn_dense_tanh_backward(delta, W, act, out)


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                fn()
        res = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                gr.replay()
            e1.record(st)
            st.synchronize()
            res.append(round(e0.elapsed_time(e1) / (10 * reps) * 1e3, 2))
    return res


lib_pair()
ref_out, ref_cs = out.clone(), cs.clone()
kernels.bnn_dense_tanh_backward(delta, W, act, out, colsum_parts=parts2)
fused()
print("max |fused - library| out %.3g  colsum %.3g (scale %.3g / %.3g)" % (
    float((out - ref_out).abs().max()), float((cs - ref_cs).abs().max()), float(ref_out.abs().max()), float(ref_cs.abs().max())))
print("isolated, us per backward step  library mm + tanh_backward_colsum:", timed(lib_pair))
print("isolated, us per backward step  bnn_dense_tanh_backward + sums    :", timed(fused))
print("isolated, us per backward step  bnn_dense_tanh_backward, no sums  :", timed(fused_plain))
print("isolated, us                    library mm alone                 :", timed(lambda: torch.mm(delta, W.t(), out=out)))
if os.environ.get("PROBE_SKIP_CHAIN"):
    sys.exit(0)
for label, fb in (("library mm + tanh_backward_colsum", False), ("fused backward steps (default)", True), ("library again", False), ("fused again", True)):
    s = bench.build_chain(dev, 0, os.environ.get("PROBE_WORKLOAD", "bnn10m-sghmc"), burn_in=8)
    s.sample_format, s.use_hip_graph, s.collect_stats = "view", True, "theta_sq"
    s.cost_fun.fused_dense_backward = s.cost_fun.bias_gradient_from_product = fb
    for _ in range(150):
        next(s)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(400):
            next(s)
        e1.record()
        torch.cuda.synchronize()
        res.append(round(e0.elapsed_time(e1) / 400 * 1e3, 1))
    print("%-40s device us/step: %s   theta finite: %s" % (label, res, bool(torch.isfinite(s.arena.row("theta")).all())), flush=True)
    del s
    torch.cuda.empty_cache()

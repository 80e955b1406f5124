"""Timing-only probes of the experiment kernel tools/gpu/bnn_dense_tanh.hip (results of the probe variants are wrong by design).
BNN_DENSE_TANH_PROBE values: 0 the kernel; 1 no MFMAs; 2 no loads (MFMAs on whatever LDS holds); 16 every ring stage preloaded with
real data, no loads in the loop; 48 = 16 + the loop's chunks streamed into an LDS stage nobody reads; 80 = 16 + streamed into
registers; 8 operands staged through registers; 10 ring of 6."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu", "libbnn_dense_tanh_probe.so"))
so.bnn_dense_tanh_probe_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
torch.manual_seed(0)

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

B, K, N = 256, 2048, 2048
for data in ("tanh", "randn", "zeros"):
    if data == "tanh":
        h = torch.tanh(torch.randn(B, K, device=dev)); W = torch.randn(K, N, device=dev) / K ** 0.5
    elif data == "randn":
        h = torch.randn(B, K, device=dev); W = torch.randn(K, N, device=dev) / K ** 0.5
    else:
        h = torch.zeros(B, K, device=dev); W = torch.zeros(K, N, device=dev)
    b = torch.randn(N, device=dev) * 0.1
    out = torch.empty(B, N, device=dev)
    call = lambda: so.bnn_dense_tanh_probe_f32(h.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), B, N, K, K, N, N, None, None,
                                               torch.cuda.current_stream().cuda_stream)
    res = []
    for probe in [int(x) for x in os.environ.get("PROBES", "0,2,16,48,80,1").split(",")]:
        os.environ["BNN_DENSE_TANH_PROBE"] = str(probe)
        res.append("%d: %.2f" % (probe, graph_us(call)))
    lib = graph_us(lambda: torch.mm(h, W, out=out))
    print("operands %-5s | library mm %.2f us | probes (us): %s" % (data, lib, "  ".join(res)), flush=True)

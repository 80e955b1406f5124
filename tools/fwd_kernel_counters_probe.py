"""A few launches of the product's fused forward layer (kernels.bnn_dense_tanh) and of the library pair it replaces, at the two
shapes of the 10 M-parameter net, for rocprofv3 --pmc passes (tools/gpu/fwd_kernel_counters.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pysgmcmc_amd import kernels
import torch.cuda.tunable as tunable
dev = torch.device("cuda:0")
# `tune`: let TunableOp pick the library solutions and write them to a file (un-profiled run); otherwise: use that file and do
# NOT tune, so that the profiled runs contain the picked kernels only
FILE = os.path.join("/tmp", "fwd_kernel_counters_tunableop.csv")
tunable.enable(True)
tunable.set_filename(FILE)
if len(sys.argv) > 1 and sys.argv[1] == "tune":
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(30)
    tunable.set_max_tuning_iterations(20)
else:
    tunable.tuning_enable(False)
    tunable.read_file(FILE)
torch.manual_seed(0)
B = 256
for K, N in ((2048, 2048), (784, 2048)):
    h = torch.tanh(torch.randn(B, K, device=dev)); W = torch.randn(K, N, device=dev) / K ** 0.5; b = torch.randn(N, device=dev) * 0.1
    out, out2 = torch.empty(B, N, device=dev), torch.empty(B, N, device=dev)
    for rep in range(8):
        kernels.bnn_dense_tanh(h, W, b, out)
        torch.mm(h, W, out=out2)
        kernels.bias_tanh(out2, b)
    torch.cuda.synchronize()
    print("K=%d max |fused - library| %.3e" % (K, (out - out2).abs().max().item()))

"""Does the spacing between the state rows of the arena matter? (HBM channel/bank interleave; dev tool)
All rows live in ONE allocation at a chosen row stride; time the burn-in (12 streams) and frozen (6 streams) kernels."""
import sys, torch
sys.path.insert(0, ".")
from pysgmcmc_amd import kernels
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
base = ((n + 63) // 64) * 64
MiB = 1 << 20
def round_up(x, m): return ((x + m - 1) // m) * m
cands = [("dense64", base),
         ("2MiB-multiple", round_up(base * 4, 2 * MiB) // 4),
         ("2MiB+256B", round_up(base * 4, 2 * MiB) // 4 + 64),
         ("2MiB+4KiB", round_up(base * 4, 2 * MiB) // 4 + 1024),
         ("2MiB+64KiB", round_up(base * 4, 2 * MiB) // 4 + 16384),
         ("2MiB+68KiB+256B", round_up(base * 4, 2 * MiB) // 4 + 17408 + 64),
         ("64MiB-multiple", round_up(base * 4, 64 * MiB) // 4),
         ("64MiB+1MiB+4KiB+256B", round_up(base * 4, 64 * MiB) // 4 + 262144 + 1024 + 64)]
for rnd in range(2):
    for label, stride in cands:
        buf = torch.zeros(7 * stride + 64, device=dev)
        rows = [buf[k * stride:k * stride + n] for k in range(7)]
        theta, V, grad, tau, g, vh, minv = rows
        theta.normal_(); grad.normal_(); tau.fill_(1); g.fill_(1); vh.fill_(1); minv.fill_(1)
        st = [0]
        def frozen(): st[0] += 1; kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=1, step=st[0])
        def adapt(): st[0] += 1; kernels.sghmc_step(theta, V, grad, tau, g, vh, minv, None, 0.01, 1e5, 0.05, True, seed=1, step=st[0])
        res = []
        for fn, bpp in ((frozen, 24), (adapt, 48)):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50): fn()
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / 50 * 1e3
            res.append("%7.1f us %5.0f GB/s" % (us, bpp * n / us / 1e3))
        print("n=%d %-24s stride=%11d B : frozen %s | burn-in %s" % (n, label, stride * 4, res[0], res[1]))
        del buf, rows, theta, V, grad, tau, g, vh, minv
        torch.cuda.empty_cache()

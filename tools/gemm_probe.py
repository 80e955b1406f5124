"""Time each GEMM of the 10 M-parameter BNN step in isolation and a few alternative formulations (dev tool)."""
import sys, torch
dev = torch.device("cuda:0")
B, D, H = 256, 784, 2048
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
X, W1, W2, W4 = r(B, D), r(D, H), r(H, H), r(H, 1)
b = r(H)
h, d = r(B, H), r(B, H)
d4 = r(B, 1)
outBH, outDH, outHH, outH1 = torch.empty(B, H, device=dev), torch.empty(D, H, device=dev), torch.empty(H, H, device=dev), torch.empty(H, 1, device=dev)
outHD = torch.empty(H, D, device=dev)
outHB = torch.empty(H, B, device=dev)
ones = torch.ones(B, device=dev)
outH = torch.empty(H, device=dev)

def t(fn, name, flops, bytes_):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): fn()
    c.record(); torch.cuda.synchronize()
    us = a.elapsed_time(c) / 50 * 1e3
    print("%-46s %7.1f us  %6.1f TF/s  %6.0f GB/s" % (name, us, flops / us / 1e6, bytes_ / us / 1e3))

f = lambda m, n, k: 2.0 * m * n * k
t(lambda: torch.addmm(b, X, W1, out=outBH), "fwd1  addmm(b, X[256x784], W1[784x2048])", f(B, H, D), 4 * (D * H + B * D + B * H))
t(lambda: torch.addmm(b, h, W2, out=outBH), "fwd2  addmm(b, h[256x2048], W2[2048x2048])", f(B, H, H), 4 * (H * H + 2 * B * H))
t(lambda: torch.mm(h, W4, out=torch.empty(B, 1, device=dev)), "fwd4  mm(h, W4[2048x1])", f(B, 1, H), 4 * (B * H))
t(lambda: torch.mm(X.t(), d, out=outDH), "dW1   mm(X^T[784x256], d[256x2048])", f(D, H, B), 4 * (D * H + B * D + B * H))
t(lambda: torch.mm(d.t(), X, out=outHD), "dW1'  mm(d^T[2048x256], X[256x784]) (transposed out)", f(D, H, B), 4 * (D * H + B * D + B * H))
t(lambda: torch.mm(h.t(), d, out=outHH), "dW2   mm(h^T[2048x256], d[256x2048])", f(H, H, B), 4 * (H * H + 2 * B * H))
t(lambda: torch.mm(h.t(), d4, out=outH1), "dW4   mm(h^T, d4[256x1])", f(H, 1, B), 4 * B * H)
t(lambda: torch.mm(d, W2.t(), out=outBH), "dX2   mm(d[256x2048], W2^T)", f(B, H, H), 4 * (H * H + 2 * B * H))
t(lambda: torch.mm(d4, W4.t(), out=outBH), "dX4   mm(d4[256x1], W4^T[1x2048])", f(B, H, 1), 4 * B * H)
t(lambda: torch.mv(d.t(), ones, out=outH), "db    mv(d^T, ones)", 2.0 * B * H, 4 * B * H)
t(lambda: torch.tanh_(outBH), "tanh_ [256x2048]", B * H, 8 * B * H)
# bigger batch for reference
for Bb in (1024, 4096):
    hb, db_ = r(Bb, H), r(Bb, H)
    ob = torch.empty(Bb, H, device=dev)
    t(lambda: torch.addmm(b, hb, W2, out=ob), "fwd2 at batch %d" % Bb, f(Bb, H, H), 4 * (H * H + 2 * Bb * H))
    t(lambda: torch.mm(hb.t(), db_, out=outHH), "dW2 at batch %d" % Bb, f(H, H, Bb), 4 * (H * H + 2 * Bb * H))
# two streams: dW2 concurrently with dX2
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): torch.mm(h.t(), d, out=outHH)
    with torch.cuda.stream(s2): torch.mm(d, W2.t(), out=outBH)
    cur.wait_stream(s1); cur.wait_stream(s2)
t(both, "dW2 || dX2 on two streams", f(H, H, B) + f(B, H, H), 0)

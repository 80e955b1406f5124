"""samples/s of the 10 M-parameter chain when every sample is copied to host numpy like the reference does
(sample_format = "numpy": 40 MB D2H per step) vs device views (dev tool)."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
for fmt in ("view", "numpy"):
    s = bench.build_chain(dev, 0)
    s.sample_format = fmt
    s.use_hip_graph = True
    for _ in range(20): next(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): next(s)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("sample_format=%-6s %8.1f samples/s (%.3f ms/step)" % (fmt, 100 / dt, dt * 10))

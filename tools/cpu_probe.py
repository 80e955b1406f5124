import os, time, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import sgmcmc_oracle as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
lib = O.load_c()
n = 10_002_434
rng = np.random.default_rng(0)
st = O.CState(rng.standard_normal(n, dtype=np.float32) * 0.02, np.float32)
st.minv[:] = rng.random(n, dtype=np.float32) + 0.5
grad = rng.standard_normal(n, dtype=np.float32) * 0.1
xi = rng.standard_normal(n, dtype=np.float32)
for th in (1, 4, 8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    lib.oracle_set_num_threads(th)
    for label, x in (("philox", None), ("injected", xi)):
        O.c_sghmc_step(st, grad, 0.01, 1e5, 0.05, False, x, seed=1, step=0)
        t0 = time.perf_counter(); k = 0
        while k < 20 and time.perf_counter() - t0 < 3: O.c_sghmc_step(st, grad, 0.01, 1e5, 0.05, False, x, seed=1, step=k); k += 1
        dt = (time.perf_counter() - t0) / k
        print("threads %3d %-8s %8.2f ms/step  %7.1f steps/s" % (th, label, dt * 1e3, 1 / dt))

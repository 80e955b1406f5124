"""CPU baselines of the bench workloads: the oracle (kind "port") timed on the GPU box's host cores. The only place besides
tests/ and smoke() that may touch ``oracle/`` -- as the thing timed NEXT to the product, never inside its timed region."""
import time

import numpy as np

from benchlib.common import BATCH, N_DATA, usable_cores
from benchlib.workloads import WORKLOADS


def cpu_baseline(n, budget_s):
    """The CPU port (kind "port") on the host cores. `value` = the COMPLETE step (numpy/BLAS BNN gradient + fused C
    update), samples/s like the GPU `value`. The update is the port a CPU user would run: one Philox call per quad and
    single-precision Box-Muller (oracle_baseline_sghmc_frozen_step_f32). The PARITY oracle's update -- which evaluates
    the f32 noise stream element by element through double-precision libm so that it matches the device stream to
    4e-6 -- is timed next to it and labelled as what it is: a checker, ~20x slower, not a baseline."""
    from oracle import sgmcmc_oracle as O
    lib = O.load_c()
    cores = usable_cores()
    lib.oracle_set_num_threads(cores)
    rng = np.random.default_rng(0)
    st = O.CState(rng.standard_normal(n, dtype=np.float32) * 0.02, np.float32)
    st.minv[:] = rng.random(n, dtype=np.float32) * 1.5 + 0.5
    grad = rng.standard_normal(n, dtype=np.float32) * 0.1
    fast = lambda state, g, step: O.baseline_sghmc_frozen_step(state, g, 0.01, float(N_DATA), 0.05, seed=1, step=step)
    fast(st, grad, 0)                                                                   # warm
    t0 = time.perf_counter()
    steps = 0
    while steps < 400 and (time.perf_counter() - t0) < budget_s / 4:
        fast(st, grad, steps + 1)
        steps += 1
    dt = time.perf_counter() - t0
    # the same on ONE core (B(1) next to B(all))
    lib.oracle_set_num_threads(1)
    t1c = time.perf_counter()
    osteps = 0
    while osteps < 20 and (time.perf_counter() - t1c) < budget_s / 6:
        fast(st, grad, 1000 + osteps)
        osteps += 1
    odt = time.perf_counter() - t1c
    lib.oracle_set_num_threads(cores)
    # the parity oracle's update with its checked noise stream (double-precision libm per element): NOT a baseline
    t1p = time.perf_counter()
    psteps = 0
    while psteps < 10 and (time.perf_counter() - t1p) < budget_s / 6:
        O.c_sghmc_step(st, grad, 0.01, float(N_DATA), 0.05, False, None, seed=1, step=2000 + psteps)
        psteps += 1
    pdt = time.perf_counter() - t1p
    # baseline A: op-by-op numpy mirror of the reference's unfused TF graph (injected noise drawn
    # by numpy, temporaries materialised, + the per-step copy-out of all parameters)
    ns = O.OpByOpState(st.theta, np.float32)
    frozen = st.minv.reshape(-1, 1)
    t1 = time.perf_counter()
    asteps = 0
    while asteps < 10 and (time.perf_counter() - t1) < budget_s / 4:
        xi = rng.standard_normal(n, dtype=np.float32)
        O.opbyop_sghmc_step(ns, grad, 0.01, float(N_DATA), 0.05, xi, frozen_minv=frozen)
        _ = ns.theta.copy()
        asteps += 1
    adt = time.perf_counter() - t1
    # the same fused update with pre-generated noise (no RNG work): the memory-bound CPU figure
    xi = rng.standard_normal(n, dtype=np.float32)
    t2 = time.perf_counter()
    isteps = 0
    while isteps < 200 and (time.perf_counter() - t2) < budget_s / 6:
        O.c_sghmc_step(st, grad, 0.01, float(N_DATA), 0.05, False, xi)
        isteps += 1
    idt = time.perf_counter() - t2
    # the FULL step on the CPU (same unit as `value`): numpy/BLAS forward + analytic backward of the same BNN
    # on a window of the same synthetic data shape, then the fused C update with Philox noise
    layers = WORKLOADS["bnn10m-sghmc"]["layers"]
    sizes = list(layers) + [1]
    params = []
    for fi, fo in zip(sizes[:-1], sizes[1:]):
        params.append((rng.standard_normal((fi, fo), dtype=np.float32) / np.sqrt(fi)).astype(np.float32))
        params.append(np.zeros(fo, np.float32))
    params.append(np.full((1, 1), np.log(1e-3), np.float32))
    Xb = rng.standard_normal((BATCH, layers[0]), dtype=np.float32)
    Yb = rng.standard_normal((BATCH, 1), dtype=np.float32)
    fst = O.CState(np.concatenate([p.ravel() for p in params]), np.float32)
    fst.minv[:] = st.minv[:fst.n] if st.n >= fst.n else 1.0
    offs = np.cumsum([0] + [p.size for p in params])
    try:                                    # BLAS threads = usable cores (256 threads under a 16-CPU quota thrash)
        from threadpoolctl import threadpool_limits
        blas_limit = threadpool_limits(limits=cores)
    except Exception:
        blas_limit = None
    fsteps, t3 = 0, time.perf_counter()
    while fsteps < 62 and (time.perf_counter() - t3) < budget_s / 2:
        if fsteps == 2:
            t3 = time.perf_counter()        # two untimed warm-up steps
        views = [fst.theta[offs[k]:offs[k + 1]].reshape(params[k].shape) for k in range(len(params))]
        _, grads = O.bnn_cost_and_grad(views, Xb, Yb, BATCH, N_DATA)
        gflat = np.concatenate([g.ravel() for g in grads])
        fast(fst, gflat, fsteps)
        fsteps += 1
    fsteps = max(fsteps - 2, 0)
    fdt = time.perf_counter() - t3
    if blas_limit is not None:
        blas_limit.restore_original_limits()
    return {"value": round(fsteps / fdt, 3) if fsteps else None, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%d complete steps of the same workload (numpy/BLAS BNN forward + analytic backward at batch %d on "
                      "%d threads, then the fused C update of %d fp32 params with OpenMP on %d threads: one Philox call per "
                      "quad, single-precision Box-Muller, generated in the loop like the GPU kernel), %.1f s; TensorFlow is "
                      "not installable here, so this port stands in for the reference's TF-CPU sampler" % (
                          fsteps, BATCH, cores, n, cores, fdt),
            "update_only_steps_per_s": round(steps / dt, 3),
            "update_only_sample": "%d frozen SGHMC update steps (no BNN gradient), %.1f s" % (steps, dt),
            "update_only_one_core_steps_per_s": round(osteps / odt, 3) if osteps else None,
            "update_only_injected_noise_steps_per_s": round(isteps / idt, 3) if isteps else None,
            "parity_oracle_update_steps_per_s": round(psteps / pdt, 3) if psteps else None,
            "parity_oracle_note": "the parity oracle's update (f32 noise through double-precision libm, Philox recomputed per "
                                  "element so that it reproduces the device stream): a checker, not a baseline -- rounds 1-2 "
                                  "reported this figure as update_only_steps_per_s",
            "opbyop_numpy_update_steps_per_s": round(asteps / adt, 3) if asteps else None,
            "opbyop_note": "op-by-op numpy mirror of the reference's unfused TF graph (temporaries materialised, "
                           "+ the per-step copy-out of all parameters): the closest proxy of TF-CPU's update"}


def svgd_cpu_baseline(n_particles, dim, budget_s):
    """The numpy restatement of pysgmcmc/samplers/svgd.py (oracle/, kind "port") on a column sample of the
    same workload; its cost is linear in the number of columns, so the rate is scaled to the full width."""
    from oracle import sgmcmc_oracle as O
    d_s = min(dim, 200_000)
    rng = np.random.default_rng(0)
    X = (rng.normal(size=(n_particles, d_s)) / np.sqrt(dim)).astype(np.float32)
    G = (rng.normal(size=(n_particles, d_s)) * 0.1).astype(np.float32)
    H = np.zeros_like(X)
    cores = usable_cores()
    O.svgd_step(X, G, H, 1e-3, 0.9, 1e-6, -1.0)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < budget_s and reps < 200:
        O.svgd_step(X, G, H, 1e-3, 0.9, 1e-6, -1.0)
        reps += 1
    dt = (time.perf_counter() - t0) / max(reps, 1)
    return {"value": round(1.0 / (dt * dim / d_s), 3), "unit": "update-steps/s", "cores": cores, "kind": "port",
            "sample": "numpy op-by-op restatement (oracle/sgmcmc_oracle.py svgd_step), %d particles x %d of the %d "
                      "columns, %d steps; seconds per step scaled by %d / %d (the cost is linear in the columns); "
                      "numpy/BLAS threads as configured on the host" % (n_particles, d_s, dim, reps, dim, d_s)}



"""``--workload sinc-bnn``: BASELINE.json configs[1] -- SGHMC on the reference's own BNN test case
(``pysgmcmc/tests/bayesian_neural_network/test_train_predict.py:20-48``): 3 x 50 tanh net on 100 points of
``y = sinc(10 x - 5)``, batch 20, the BNN's sampler defaults (``models/bayesian_neural_network.py:151-155,451-457``:
eps = 0.01, mdecay = 0.05, scale_grad = N = 100, burn_in_steps = 1000), 5 252 parameters, fp32.

A step of this size is launch-bound, not bandwidth-bound (the whole state is 147 KB): ``value`` is the rate of the product's
default path for it -- the fused whole-step kernel K8 (``sgmcmc_bnn_fused_sghmc_steps_f32``: window gather, net, NLL, gradients,
update, burn-in switch, up to 100 steps per launch in ONE workgroup) -- and ``modes`` reports the same chain stepped through
``next(sampler)`` eagerly, with the cost pipeline replayed from a hipGraph, and with the whole step in one graph."""
import json
import sys
import gc
import time

import numpy as np
import torch

from benchlib.common import HBM_PEAK_GBS, usable_cores

N_POINTS, BATCH_SINC, BURN_IN, CHUNK = 100, 20, 1000, 100


def sinc_data(seed):
    rng = np.random.RandomState(seed)                          # test_train_predict.py:20-25
    X = rng.rand(N_POINTS, 1)
    return X, np.sinc(X * 10 - 5).sum(axis=1)


def build_sinc_chain(dev, rank, mode=False):
    from pysgmcmc_amd.data_batches import Placeholder, generate_batches
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
    from pysgmcmc_amd.samplers import SGHMCSampler
    from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
    X, y = sinc_data(1)
    xp, yp = Placeholder(dtype=torch.float32, device=dev), Placeholder(dtype=torch.float32, device=dev)
    s = SGHMCSampler(params=init_mlp_params(1, seed=3 + rank, dtype=torch.float32, device=dev),
                     cost_fun=BNNCost(xp, yp, batch_size=BATCH_SINC, n_examples=N_POINTS),
                     batch_generator=generate_batches(X, y, xp, yp, BATCH_SINC, seed=1 + rank),
                     stepsize_schedule=ConstantStepsizeSchedule(0.01), burn_in_steps=BURN_IN, mdecay=0.05,
                     scale_grad=float(N_POINTS), session=dev, dtype=torch.float32, seed=1234 + rank)
    s.sample_format = "view"
    s.collect_stats = "theta_sq"
    s.use_hip_graph = mode
    return s


def sinc_cpu_baseline(budget_s):
    """The same step on ONE host core (kind "port"): numpy forward + analytic backward of the 3 x 50 net at batch 20
    (oracle.bnn_cost_and_grad) and the fused C update of the 5 252 parameters (single-precision Box-Muller, one Philox call per
    quad). One thread: at 21 KB per array OpenMP and BLAS threading only add overhead."""
    from oracle import sgmcmc_oracle as O
    lib = O.load_c()
    lib.oracle_set_num_threads(1)
    rng = np.random.default_rng(0)
    sizes = [1, 50, 50, 50, 1]
    params = []
    for fi, fo in zip(sizes[:-1], sizes[1:]):
        params.append((rng.standard_normal((fi, fo)) / np.sqrt(fi)).astype(np.float32))
        params.append(np.zeros(fo, np.float32))
    params.append(np.full((1, 1), np.log(1e-3), np.float32))
    st = O.CState(np.concatenate([p.ravel() for p in params]), np.float32)
    offs = np.cumsum([0] + [p.size for p in params])
    X, y = sinc_data(1)
    X, y = X.astype(np.float32), y.astype(np.float32).reshape(-1, 1)
    starts = np.random.RandomState(1).randint(0, N_POINTS - BATCH_SINC + 1, size=4096)
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)
    except Exception:
        limit = None
    steps, t0 = 0, time.perf_counter()
    while steps < 20000 and (time.perf_counter() - t0) < budget_s:
        if steps == 20:
            t0 = time.perf_counter()                           # 20 untimed warm-up steps
        lo = int(starts[steps % 4096])
        views = [st.theta[offs[k]:offs[k + 1]].reshape(params[k].shape) for k in range(len(params))]
        _, grads = O.bnn_cost_and_grad(views, X[lo:lo + BATCH_SINC], y[lo:lo + BATCH_SINC], BATCH_SINC, N_POINTS)
        O.baseline_sghmc_frozen_step(st, np.concatenate([g.ravel() for g in grads]), 0.01, float(N_POINTS), 0.05, seed=1, step=steps)
        steps += 1
    dt = time.perf_counter() - t0
    steps = max(steps - 20, 0)
    if limit is not None:
        limit.restore_original_limits()
    lib.oracle_set_num_threads(usable_cores())
    return {"value": round(steps / dt, 1) if steps else None, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": "%d complete steps (numpy forward + analytic backward of the 3 x 50 net at batch 20, then the fused C update "
                      "of 5 252 fp32 parameters with in-loop Philox noise) on one thread, %.1f s; TensorFlow is not installable "
                      "here, so this port stands in for the reference's TF-CPU sampler" % (steps, dt)}


def _rate(step_fn, n, sync):
    step_fn(max(n // 10, 30))
    sync()
    t0 = time.perf_counter()
    step_fn(n)
    sync()
    return n / (time.perf_counter() - t0)


def run_sinc(args, dev, rank, world, dist):
    sync = torch.cuda.synchronize

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    # host hygiene as in benchlib/chain_run.py: a full garbage collection of a process that has torch loaded stops the host for
    # ~40 ms; collect now and freeze the survivors, and keep the collector off inside the timed region
    gc.collect()
    gc.freeze()
    s = build_sinc_chain(dev, rank)
    assert s.fused_bnn_available(), "the 3 x 50 net must fit the fused whole-step kernel"
    n = s.arena.n
    # prime (untimed): the chain's 1000 burn-in steps + 200 frozen ones, then --warmup steps
    done = 0
    while done < BURN_IN + 200:
        s.fused_bnn_steps(CHUNK)
        done += CHUNK
    assert not s._adapting
    if args.warmup:
        s.fused_bnn_steps(args.warmup)
    launches, left = [], args.steps
    while left > 0:
        launches.append(min(left, CHUNK))
        left -= launches[-1]
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in launches]
    gc.collect()
    gc.disable()                                               # (a pause of 40 ms is five times this region at --steps 400)
    fence()
    t0 = time.perf_counter()
    for k, (e0, e1) in zip(launches, pairs):
        e0.record()
        s.fused_bnn_steps(k)
        e1.record()
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    assert torch.isfinite(s.arena.row("theta")).all()
    if rank != 0:
        return
    launch_us = np.array([a.elapsed_time(b) for a, b in pairs]) * 1e3
    us_per_step = float(launch_us.sum() / args.steps)
    # the other stepping modes of the same chain (after the timed region, fresh chains, burn-in by the fused kernel first)
    modes = {"fused_steps_%d_per_launch" % CHUNK: None}
    fused = build_sinc_chain(dev, 0)
    fused.fused_bnn_steps(CHUNK * 12)
    modes["fused_steps_%d_per_launch" % CHUNK] = round(_rate(lambda k: [fused.fused_bnn_steps(CHUNK) for _ in range(k // CHUNK)], 3000, sync), 1)
    modes["fused_steps_1_per_launch"] = round(_rate(lambda k: [fused.fused_bnn_steps(1) for _ in range(k)], 1000, sync), 1)
    for label, mode in (("full_graph", "full"), ("hip_graph", True), ("eager", False)):
        c = build_sinc_chain(dev, 0, mode)
        c.fused_bnn_steps(CHUNK * 12)                          # through burn-in; the timed steps are frozen ones
        modes[label] = round(_rate(lambda k, c=c: [next(c) for _ in range(k)], 2000 if mode else 600, sync), 1)
    # the ensemble form of the same kernel: one workgroup per chain, one chain per CU (pysgmcmc_amd.samplers.FusedBNNChains; the
    # reference runs chains one after the other, diagnostics/sample_chains.py:369-382) -- chains x steps per second of ONE GPU
    from pysgmcmc_amd.samplers.fused_chains import FusedBNNChains
    X, y = sinc_data(1)
    many = {}
    for n_chains in (256,):
        group = FusedBNNChains.for_dataset(X, y, n_chains, batch_size=BATCH_SINC, seed=11, dtype=torch.float32, device=dev,
                                           stepsize=0.01, burn_in_steps=BURN_IN, mdecay=0.05)
        for _ in range(12):
            group.steps(CHUNK)                                 # through burn-in
        e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        sync()
        t0 = time.perf_counter()
        for e0, e1 in e:
            e0.record()
            group.steps(CHUNK)
            e1.record()
        sync()
        wall = time.perf_counter() - t0
        k_us = np.array([a.elapsed_time(b) for a, b in e]) * 1e3
        assert torch.isfinite(group.theta()).all()
        many["chains_%d" % n_chains] = {
            "chains": n_chains, "steps_per_launch": CHUNK, "samples_per_s": round(n_chains * CHUNK * len(e) / wall, 1),
            "us_per_launch_mean": round(float(k_us.mean()), 1), "us_per_step_of_all_chains": round(float(k_us.mean()) / CHUNK, 2),
            "samples_per_s_device_time": round(n_chains * CHUNK / (float(k_us.mean()) * 1e-6), 1),
            "note": "one launch advances %d independent chains by %d steps (one 512-lane workgroup per chain, one chain per CU); "
                    "samples_per_s is wall clock including the host side (window draws of every chain), not part of `value`" % (n_chains, CHUNK)}
        del group
    alg_bytes = 6 * 4 * n + BATCH_SINC * 2 * 4                 # R{theta, V, grad, minv} W{theta, V} + the window rows, if it went to HBM
    line = {
        "metric": "MCMC samples/sec, sinc BNN 3x50 (BASELINE configs[1])",
        "value": round(world * args.steps / elapsed, 1), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "sinc-bnn: SGHMC (frozen phase) on the reference's BNN test case, tanh MLP 1-50-50-50-1 + "
                               "log-variance = %d params, 100 points of sinc(10 x - 5), batch 20, eps 0.01, mdecay 0.05, "
                               "scale_grad 100, burn-in 1000 (done in the prime phase); whole steps in the fused kernel, "
                               "%d steps per launch; 1 chain per GPU" % (n, CHUNK),
                   "params": n, "batch": BATCH_SINC, "chains": world, "steps_per_launch": CHUNK},
        "modes_samples_per_s": modes,
        "many_chains_per_gpu": many,
        "roofline": {"bound": "hbm", "kernel": "bnn_fused_sghmc_kernel<float, 0> (ONE 512-lane workgroup per chain)",
                     "achieved": round(alg_bytes / (us_per_step * 1e-6) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg_bytes / (us_per_step * 1e-6) / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                     "algorithmic_bytes_per_step": alg_bytes, "us_per_step": round(us_per_step, 2),
                     "us_per_launch_mean": round(float(launch_us.mean()), 1), "launches_timed": len(launches),
                     "us_per_launch": [round(float(v), 1) for v in launch_us[:16]],
                     "note": "latency-bound, one workgroup: the chain's whole state (%d KB) lives in LDS / L2 for the length of a "
                             "launch, so the HBM roofline does not bind -- a step is ~25 barrier-separated phases of 8 waves "
                             "on one CU (DESIGN.md section 3, K8); the figure that matters is us_per_step. rocprofv3 launch "
                             "durations: profiles/r04_sinc_bnn_kernel_stats.csv" % (7 * 4 * n // 1024),
                     "timing": "hipEvent pair around every launch of the timed region"},
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = sinc_cpu_baseline(min(args.cpu_seconds, 8.0))
    print(json.dumps(line))
    sys.stdout.flush()

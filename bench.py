#!/usr/bin/env python3
"""Benchmark of the SG-MCMC update path on MI355X (contract: see DESIGN.md "Measurement").

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): SGHMC on a 4-layer tanh MLP BNN
784-2048-2048-2048-1 + scalar log-variance = 10 002 434 parameters, fp32, synthetic
data (N = 100 000 rows ~ N(0,1), minibatch windows of 256), post-burn-in (frozen
preconditioner) phase. One "step" = one complete ``next(sampler)``: feed the minibatch,
BNN forward + analytic backward writing gradients into the flat arena (rocBLAS GEMMs),
then the fused SGHMC update kernel (in-register Philox noise). Nothing is skipped.

N > 1: independent chains, one per GPU (weak scaling, no collective on the data path);
every --rhat-every steps the chains exchange Welford moments with one RCCL all-reduce
to compute R-hat (the only communication the path has).

Phases: (1) PRIME, always and untimed, independent of --warmup: the chain's burn-in (8 adapting
steps), then frozen steps with a Welford moments update, a trace append and (N > 1) one complete
R-hat exchange -- every code path the timed loop can take has run once (first use of a kernel
loads its code object: tens of ms); (2) --warmup untimed steps; (3) EXACTLY --steps timed steps
between barrier + synchronize fences, all in the frozen-preconditioner phase.

Output: ONE JSON line on rank 0. ``value`` = whole-job samples/s over the whole timed region
(``step_ms_median`` / ``step_ms_max`` from per-step HIP events expose one-offs). ``roofline`` = the
fused update kernel measured live with HIP events inside the timed region (10 M parameters: 240 MB
per launch, Infinity-Cache-assisted); ``roofline_hbm_resident`` = the same kernels on 49 826 818
parameters (1.2 GB per launch, cannot live in the 256 MiB Infinity Cache), measured after the
timed region. ``cpu_baseline`` (N = 1 only) = the same full step on the host cores (numpy/BLAS
BNN gradient + the fused C oracle update), unit samples/s like ``value``.
"""
import argparse
import gc
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md:35
# algorithmic bytes per parameter per launch, fp32
BYTES_PER_PARAM = {"sghmc_frozen": 24, "sghmc_adapt": 48, "sgld_frozen": 16, "sgld_adapt": 40, "rsghmc": 20}
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
FP32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 matrix-core peak, /opt/skills/guides/MI355X_MICROARCH.md
PRIME_BURN_IN = 8              # adapting steps of the chain, run in the prime phase (never timed)
PRIME_FROZEN = 4               # frozen steps of the prime phase with a moments update + trace append each
PRIME_STEADY = int(os.environ.get("BENCH_PRIME_STEADY", "124"))   # further frozen steps: ~30 ms of device work, after which
                               # the step time has settled (measured: 0.237 ms/step right after start-up, 0.218 after 100 steps)
N_HBM_RESIDENT = 49_826_818    # configs[4]'s parameter count: 1.2 GB per frozen SGHMC launch


UPDATE_KERNEL_SOURCES = ("pysgmcmc_amd/csrc/sgmcmc_stream.hpp", "pysgmcmc_amd/csrc/sgmcmc_device.hpp",
                         "pysgmcmc_amd/csrc/sgmcmc_sghmc.hip", "pysgmcmc_amd/csrc/sgmcmc_sgld.hip",
                         "pysgmcmc_amd/csrc/sgmcmc_rsghmc.hip", "pysgmcmc_amd/csrc/sgmcmc_kernels.hip",
                         "pysgmcmc_amd/csrc/Makefile", "include/sgmcmc_hip.h")


def kernel_source_hash():
    """sha256 over the sources the streaming update kernels K1-K5 are built from (kernel shape, operators, their host side,
    build flags, the C ABI header): identifies the build a PMC traffic table was collected with (tools/pmc_traffic.py
    stores it; there is no .git on the GPU box)."""
    h = hashlib.sha256()
    for rel in UPDATE_KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(rel.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(mode, n, variant=""):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.py: separate --pmc FETCH_SIZE /
    WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, calibrated on launches with
    known byte counts) -- but ONLY if that table was collected with the kernel sources this run uses (source hash
    recorded in the file); a stale table yields None rather than an old byte count next to fresh timings. Returns
    (bytes per launch or None, source string)."""
    name = os.path.relpath(PMC_TRAFFIC_FILE, ROOT)
    try:
        with open(PMC_TRAFFIC_FILE) as fh:
            doc = json.load(fh)
        have, want = doc.get("kernel_source_hash"), kernel_source_hash()
        if have != want:
            return None, "%s was collected with kernel sources %s, this build is %s: traffic not reported" % (name, have, want)
        sizes = doc["sizes"][str(n)]
        entry = sizes[mode + variant]         # "" plain, "_stats" every statistic, "_tsq" sum theta^2 only, "_tsq_mom" + fused moments
        return int(round(entry["bytes_per_param"] * n)), "%s (%s, kernel sources %s)" % (name, doc.get("collected", "?"), have)
    except (OSError, KeyError, ValueError):
        return None, "no PMC pass for n=%d in %s" % (n, name)
BATCH = 256
N_DATA = 100_000
# workloads: the default is BASELINE.json configs[2]; the 50 M ones are configs[4]'s two samplers
# (HBM-resident working sets, burn-in stepsize ramp) for profiles/, not the headline line.
WORKLOADS = {
    "bnn10m-sghmc": dict(sampler="sghmc", layers=(784, 2048, 2048, 2048)),        # 10 002 434 params
    "bnn50m-sgld": dict(sampler="sgld", layers=(512, 4864, 4864, 4864)),          # 49 826 818 params
    "bnn50m-rsghmc": dict(sampler="rsghmc", layers=(512, 4864, 4864, 4864)),
    # SVGD update path on 16 particles of the same 10 M-parameter model (synthetic gradients)
    "svgd16-10m": dict(sampler="svgd", layers=(784, 2048, 2048, 2048), particles=16),
}
LAYERS = WORKLOADS["bnn10m-sghmc"]["layers"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rhat-every", type=int, default=100, help="R-hat exchange cadence (steps), N > 1")
    ap.add_argument("--moments-every", type=int, default=10, help="Welford moments cadence (steps)")
    ap.add_argument("--rhat-mode", choices=["reduce_scatter", "allreduce"], default="reduce_scatter",
                    help="R-hat exchange: parameter-sharded reduce-scatter (half the xGMI traffic; default) or all-reduce")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="bnn10m-sghmc")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to exercise the N > 1 code path "
                         "on a box with fewer GPUs than ranks)")
    ap.add_argument("--all-ranks-on-gpu0", action="store_true", help="testing aid: every rank uses cuda:0")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="keep the BLAS heuristics instead of letting TunableOp pick the GEMM solutions in warm-up")
    ap.add_argument("--eager", action="store_true", help="step eagerly instead of replaying one hipGraph per step")
    ap.add_argument("--max-queue-depth", type=int, default=64,
                    help="steps the host may run ahead of the device in the timed loop (0 = unbounded)")
    ap.add_argument("--chains-per-gpu", type=int, default=1,
                    help="independent chains that SHARE each GPU (own stream + hipGraph each, pysgmcmc_amd.samplers."
                         "ConcurrentChains); the default 1 is BASELINE.json's sharding (one chain per GPU)")
    ap.add_argument("--time-every", type=int, default=0,
                    help="the update launches of every k-th timed step carry the HIP timestamp events (a timed launch costs ~8 us of "
                         "device time per step); 0 = steps // 5 clamped to [1, 7]")
    ap.add_argument("--overlap", choices=["on", "off"], default=os.environ.get("BENCH_OVERLAP", "off"),
                    help="off (default): one update launch after the backward pass. on: update the arena layer by layer on a "
                         "side stream under the remaining backward GEMMs -- bit-identical chain, measured SLOWER on MI355X "
                         "(profiles/r03_overlap_probe.txt); kept to reproduce that result")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="N > 1 started without a launcher: wall-clock limit of the ranks this process spawns (s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-update-only", action="store_true", help="skip the kernel-only loops (for rocprof runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU baseline leg")
    return ap.parse_args()


def build_chain(dev, rank, workload="bnn10m-sghmc", burn_in=8):
    from pysgmcmc_amd.data_batches import Placeholder, generate_batches
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
    from pysgmcmc_amd.samplers import RelativisticSGHMCSampler, SGHMCSampler, SGLDSampler
    from pysgmcmc_amd.stepsize_schedules import BurnInRampStepsizeSchedule, ConstantStepsizeSchedule

    spec = WORKLOADS[workload]
    layers = spec["layers"]
    g = torch.Generator(device=dev).manual_seed(0)             # same synthetic dataset on every rank
    X = torch.randn(N_DATA, layers[0], device=dev, generator=g)
    y = torch.randn(N_DATA, device=dev, generator=g)
    xp = Placeholder(dtype=torch.float32, device=dev, name="X_Minibatch")
    yp = Placeholder(dtype=torch.float32, device=dev, name="Y_Minibatch")
    params = init_mlp_params(layers[0], hidden=layers[1:], seed=1000 + rank, dtype=torch.float32, device=dev)
    cost = BNNCost(xp, yp, batch_size=BATCH, n_examples=N_DATA)
    common = dict(params=params, cost_fun=cost,
                  batch_generator=generate_batches(X, y, xp, yp, batch_size=BATCH, seed=rank),
                  session=dev, dtype=torch.float32, seed=1234 + rank)
    if spec["sampler"] == "sghmc":
        return SGHMCSampler(stepsize_schedule=ConstantStepsizeSchedule(0.01), mdecay=0.05,
                            scale_grad=float(N_DATA),
                            burn_in_steps=burn_in,             # adapted during warmup; timed steps are frozen
                            **common)
    if spec["sampler"] == "sgld":
        # configs[4]: preconditioned SGLD with a burn-in stepsize ramp (a StepsizeSchedule subclass)
        return SGLDSampler(stepsize_schedule=BurnInRampStepsizeSchedule(1e-4, 1e-3, burn_in_steps=burn_in),
                           A=1.0, scale_grad=float(N_DATA), burn_in_steps=burn_in, **common)
    return RelativisticSGHMCSampler(stepsize_schedule=BurnInRampStepsizeSchedule(1e-4, 1e-3, burn_in_steps=burn_in),
                                    mass=1.0, speed_of_light=1.0, D=1.0, Bhat=0.0, **common)


def update_only(sampler, iters=200):
    """Back-to-back launches of the fused kernel alone on the chain's own arrays (no gradient work)."""
    from pysgmcmc_amd import kernels
    a = sampler.arena
    out = {}
    for name, adapt in (("sghmc_frozen", False), ("sghmc_adapt", True)):
        state = a.state_dict()
        for _ in range(10):
            kernels.sghmc_step(a.row("theta"), a.row("V"), a.row("grad"), a.row("tau"), a.row("g"), a.row("v_hat"),
                               a.row("minv"), None, 0.01, float(N_DATA), 0.05, adapt, seed=1, step=0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            kernels.sghmc_step(a.row("theta"), a.row("V"), a.row("grad"), a.row("tau"), a.row("g"), a.row("v_hat"),
                               a.row("minv"), None, 0.01, float(N_DATA), 0.05, adapt, seed=1, step=i + 1)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        a.load_state_dict(state)
        out[name] = {"us_per_launch": round(us, 2), "steps_per_s": round(1e6 / us, 1),
                     "GBps": round(BYTES_PER_PARAM[name] * a.n / us / 1e3, 1)}
    return out


def chains_per_gpu_leg(dev, sampler, workload, rounds=200):
    """Ensemble throughput of ONE GPU outside `value` (which stays one chain per GPU, as BASELINE.json's north_star shards the
    ensemble): the timed chain alone, then together with a second independent chain of the same workload, each chain on its own
    stream with its own hipGraph (pysgmcmc_amd.samplers.ConcurrentChains), bare loops of `rounds` steps per chain."""
    from pysgmcmc_amd.samplers import ConcurrentChains
    sampler.attach_moments(None)
    sampler.kernel_timer = None
    other = build_chain(dev, 1, workload, burn_in=PRIME_BURN_IN)
    other.sample_format, other.use_hip_graph, other.collect_stats = "view", sampler.use_hip_graph, sampler.collect_stats
    out = {}
    for label, chains in (("one_chain", [sampler]), ("two_chains", [sampler, other])):
        group = ConcurrentChains(chains)
        group.run(PRIME_BURN_IN + 60)
        group.synchronize()
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            group.run(rounds)
            group.synchronize()
            best = max(best, len(chains) * rounds / (time.perf_counter() - t0))
        group.join()
        out[label + "_samples_per_s"] = round(best, 1)
    out["two_over_one"] = round(out["two_chains_samples_per_s"] / out["one_chain_samples_per_s"], 3)
    out["note"] = ("not part of `value`: independent chains share the GPU, one stream and one hipGraph each, stepped round-robin "
                   "by one host thread, no moments / timer; the second chain's launches fill the idle parts of the first one's "
                   "(launch ramps and tails of ~15 dependent launches per step, M = 256 GEMMs at ~62 % of the matrix pipe)")
    del other
    torch.cuda.empty_cache()
    return out


def hbm_resident_roofline(dev, n=N_HBM_RESIDENT, iters=40):
    """The update kernels on a working set that cannot live in the 256 MiB Infinity Cache (configs[4]'s
    49 826 818 parameters: 0.8-2.4 GB per launch). Every launch carries its own HIP event pair that receives the
    kernel's start/stop timestamps (plus a hipEventRecord bracket around it for comparison); state is synthetic
    (theta ~ N(0, 0.02^2), grad ~ N(0, 0.1^2), minv ~ U(0.5, 2)), in-register Philox noise."""
    from pysgmcmc_amd import kernels
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda sc: torch.randn(n, device=dev, generator=g) * sc
    theta, V, grad = mk(0.02), torch.zeros(n, device=dev), mk(0.1)
    minv = torch.rand(n, device=dev, generator=g) * 1.5 + 0.5
    tau, gg, vh = (torch.ones(n, device=dev) for _ in range(3))
    calls = {
        "sghmc_frozen": lambda i, L: kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, float(N_DATA),
                                                        0.05, False, seed=1, step=i, launch=L),
        "sghmc_adapt": lambda i, L: kernels.sghmc_step(theta, V, grad, tau, gg, vh, minv, None, 0.01, float(N_DATA),
                                                       0.05, True, seed=1, step=i, launch=L),
        "sgld_frozen": lambda i, L: kernels.sgld_step(theta, grad, None, None, None, minv, None, 1e-3, 1.0, float(N_DATA),
                                                      False, seed=1, step=i, launch=L),
        "sgld_adapt": lambda i, L: kernels.sgld_step(theta, grad, tau, gg, vh, minv, None, 1e-3, 1.0, float(N_DATA),
                                                     True, seed=1, step=i, launch=L),
        "rsghmc": lambda i, L: kernels.rsghmc_step(theta, V, grad, 1e-3, 1.0, 1.0, 1.0, 0.0, seed=1, step=i, launch=L),
    }
    out = {}
    for name, call in calls.items():
        for i in range(5):
            call(i, None)
        torch.cuda.synchronize()
        pairs, kevs = [], [kernels.KernelEvents() for _ in range(iters)]
        for i in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call(5 + i, kernels.LaunchConfig(events=kevs[i]))
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        us = np.array([k.elapsed_us() for k in kevs])                   # the kernels' own timestamps
        bracket = np.array([a.elapsed_time(b) for a, b in pairs]) * 1e3
        alg = BYTES_PER_PARAM[name] * n
        traffic, src = pmc_traffic(name, n)
        out[name] = {"us_per_launch_mean": round(float(us.mean()), 2), "us_per_launch_median": round(float(np.median(us)), 2),
                     "us_bracket_mean": round(float(bracket.mean()), 2),
                     "algorithmic_bytes_per_launch": alg, "achieved": round(alg / (us.mean() * 1e-6) / 1e9, 1),
                     "frac": round(alg / (us.mean() * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "launches_timed": iters}
        theta.normal_(0.0, 0.02, generator=g)              # keep the chain finite across 225 synthetic steps
        V.zero_()
    assert torch.isfinite(theta).all()
    head = out["sghmc_frozen"]
    return {"bound": "hbm", "kernel": "stream_quads_vec<SghmcOp<float,false,false>,1,true,0,false,false> (128-lane blocks, nt)",
            "achieved": head["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head["frac"],
            "traffic": head["traffic"], "traffic_source": pmc_traffic("sghmc_frozen", n)[1],
            "params": n, "working_set_note": "%.2f GB per frozen SGHMC launch: HBM-resident, cannot be served by the 256 MiB "
                                             "Infinity Cache" % (head["algorithmic_bytes_per_launch"] / 1e9),
            "timing": "kernel start/stop timestamps (hipExtLaunchKernel events) of every launch, back to back, after the "
                      "timed region (not part of `value`); us_bracket_mean = hipEventRecord pair around the call",
            "kernels": out}


def usable_cores():
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota). (The GPU box shows
    256 logical CPUs but a 16-CPU cgroup quota; 256 OpenMP threads there run 8x SLOWER than 16.)"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return cores


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
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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


def run_svgd(args, dev, rank, world, dist):
    """`--workload svgd16-10m`: one step = sgmcmc_svgd_step_f32 (kernel matrix + update, 4 launches) on
    n particles x 10 002 434 parameters with fixed synthetic gradients. Particles never leave HBM."""
    from pysgmcmc_amd import kernels
    spec = WORKLOADS[args.workload]
    layers = spec["layers"]
    n = spec["particles"]
    dim = sum(a * b + b for a, b in zip(layers, layers[1:] + (1,))) + 1
    ld = (dim + 63) // 64 * 64                                  # the sampler's row pitch
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    x = torch.randn(n * ld, device=dev, generator=g) * (1.0 / dim ** 0.5)
    grad = torch.randn(n * ld, device=dev, generator=g) * 0.1
    hist = torch.zeros_like(x)
    ws = kernels.svgd_workspace(n, x)
    step = lambda: kernels.svgd_step(x, grad, hist, n, dim, 1e-3, 0.9, 1e-6, ws, ld=ld, repulsion_sign=-1)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    pairs = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        pairs.append((e0, e1))
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(x).all()
    if rank == 0:
        us = float(np.mean([a.elapsed_time(b) for a, b in pairs])) * 1e3
        alg_bytes = 24 * n * dim                                # S1 reads X (4 B), S4 R{X,G,H} W{X,H} (20 B) per element
        achieved = alg_bytes / (us * 1e-6) / 1e9
        line = {
            "metric": "SVGD update-steps/sec + HBM GB/s (%% roofline), %d particles x BNN 10M params" % n,
            "value": round(world * args.steps / elapsed, 2), "unit": "update-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: SVGD step (pairwise distances, median bandwidth, kernel matrix, K[G|X] + AdaGrad "
                                   "update) on %d particles x %d parameters, row pitch %d, fixed synthetic gradients; "
                                   "1 particle set per GPU" % (args.workload, n, dim, ld),
                       "particles": n, "params": dim, "chains": world},
            "roofline": {"bound": "hbm", "kernel": "sgmcmc_svgd_step_f32 (svgd_gram_mfma16_kernel + svgd_update_mfma16_kernel; "
                                                   "per-kernel times in profiles/r01_svgd_kernel_stats.md)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # PMC passes of profiles/r01_svgd_pmc_traffic.md: S1 4.00 B, S4 12.00 + 8.00 B per element
                         "traffic": int(24.0 * n * dim), "traffic_source": "profiles/r01_svgd_pmc_traffic.md",
                         "algorithmic_bytes_per_launch": alg_bytes, "us_per_launch_mean": round(us, 2),
                         "launches_timed": len(pairs),
                         "timing": "hipEvent pair around every step (4 launches) of the timed region"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = svgd_cpu_baseline(n, dim, args.cpu_seconds)
        print(json.dumps(line))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(args):
    """``bench.py --gpus N`` (N > 1) started WITHOUT a launcher: this process -- which has not touched the GPU and never
    will -- starts N fresh ranks of this same script (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, rendezvous on 127.0.0.1), lets rank 0 write the ONE JSON line to the inherited stdout, and returns the
    first non-zero exit code (the other ranks are then stopped). A wall-clock limit (--launch-timeout) stops ranks that
    hang in a collective. The ``python -m torch.distributed.run ... bench.py --gpus N`` form keeps working: it sets
    WORLD_SIZE, so this function is not entered."""
    n = args.gpus
    probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    probe.bind(("127.0.0.1", 0))
    port = probe.getsockname()[1]
    probe.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:]      # this script (or one that borrows the launcher)
    ranks = []
    for r in range(n):
        ranks.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0"),
                                      stdout=None if r == 0 else sys.stderr, start_new_session=True))
    deadline = time.monotonic() + args.launch_timeout
    rc, why = 0, None
    try:
        while True:
            codes = [p.poll() for p in ranks]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc, why = bad[0][1], "rank %d exited with code %d" % bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                rc, why = 124, "ranks still running after --launch-timeout %.0f s" % args.launch_timeout
                break
            time.sleep(0.05)
    finally:
        for p in ranks:                                        # only the process groups started above, by their exact ids
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 9)
                except ProcessLookupError:
                    pass
        for p in ranks:
            p.wait()
    if why:
        print("bench: %s; stopped the other ranks" % why, file=sys.stderr)
    return rc if rc >= 0 else 128 - rc


@torch.no_grad()
def gemm_only_us(sampler, iters=60):
    """The eight fp32 GEMMs of one step (three forward, five backward; same operands, shapes and output buffers as the
    cost pipeline) replayed back to back from their own hipGraph: microseconds per step and their FLOP count."""
    cost, params, gv = sampler.cost_fun, sampler.params, sampler.arena.grad_views
    X = cost.x_placeholder.value
    ws = cost._buffers(params, X.shape[0])
    hs, ds = ws["h"], ws["d"]
    L = (len(params) - 1) // 2 - 1                              # index of the single-output layer

    def gemms():
        h, flops = X, 0
        for l in range(L):
            torch.mm(h, params[2 * l], out=hs[l])                  # the bias rides in the activation launch
            flops += 2 * h.shape[0] * h.shape[1] * params[2 * l].shape[1]
            h = hs[l]
        for l in range(L - 1, -1, -1):
            h_in = X if l == 0 else hs[l - 1]
            if l > 0:
                torch.mm(ds[l], params[2 * l].t(), out=ds[l - 1])
                flops += 2 * ds[l].shape[0] * ds[l].shape[1] * params[2 * l].shape[0]
            torch.mm(h_in.t(), ds[l], out=gv[2 * l])
            flops += 2 * h_in.shape[1] * h_in.shape[0] * ds[l].shape[1]
        return flops
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        flops = gemms()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        gemms()
    for _ in range(5):
        graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, flops


def cost_pipeline_us(sampler, iters=60):
    """The captured cost/gradient pipeline alone (every graph segment, no update launch): microseconds per step."""
    segments = sampler._graphs[("cost",)][0]
    for _ in range(5):
        for g, _sl in segments:
            g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for g, _sl in segments:
            g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def update_kernel_instance(op_name, adapt, big, sampler, moments=False):
    """Name of the stream_quads_vec instance the library launches for this sampler's update under the launch
    configuration in effect (sampler.launch, else the Python-side default, else the library's defaults)."""
    from pysgmcmc_amd import kernels
    cfg = dict(kernels.get_launch_config())
    if sampler.launch is not None:
        cfg.update({k: v for k, v in sampler.launch.as_dict().items() if v != (-1 if k == "nontemporal" else 0)})
    qpt = cfg["quads_per_thread"]
    nt = big if cfg["nontemporal"] == 2 else bool(cfg["nontemporal"])
    bt = cfg["block_threads"] if cfg["block_threads"] > 0 else (128 if big else 256)
    loop = qpt != 1 or (sampler.arena.n // 4 + bt - 1) // bt > cfg["max_blocks"]
    stats = {True: 1, "theta_sq": 2}.get(sampler.collect_stats, 0)
    if loop and stats == 2:
        stats = 1                                          # the looping variants reduce every statistic
    return "stream_quads_vec<%s<float,%s,false>,%d,%s,%d,%s,%s> (%d-lane blocks)" % (
        op_name, "true" if adapt else "false", qpt, "true" if nt else "false", stats, "true" if loop else "false",
        "true" if (moments and not loop) else "false", bt)


def launch_table(timer, n, bytes_per_param, moments_every):
    """Per-launch records of a timed region: (step, lo, hi, microseconds, algorithmic bytes). A launch of a moments
    step also carries the fused Welford update (+16 B per f32 parameter)."""
    rows = []
    for kev, tag in zip(timer.kevents, timer.tags):
        step, lo, hi = tag if tag is not None else (None, 0, n)
        with_mom = step is not None and moments_every and (step + 1) % moments_every == 0
        rows.append((step, lo, hi, kev.elapsed_us(), (bytes_per_param + (16 if with_mom else 0)) * (hi - lo), bool(with_mom)))
    return rows


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: spawn the ranks from here, BEFORE anything in this process touches the GPU
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists for the update path)")
    if os.environ.get("BENCH_TEST_FAIL_RANK") == str(rank) and world > 1:
        sys.exit(7)                                            # tests/: a rank that dies must take the job down
    dev = torch.device("cuda", 0 if args.all_ranks_on_gpu0 else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=600))     # RCCL over xGMI
        else:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))

    if WORKLOADS[args.workload]["sampler"] == "svgd":
        return run_svgd(args, dev, rank, world, dist)
    from pysgmcmc_amd import kernels
    if not args.no_gemm_tuning:
        from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning
        try:
            enable_gemm_tuning(True, max_duration_ms=int(os.environ.get("BENCH_TUNE_MS", "30")),
                               max_iterations=int(os.environ.get("BENCH_TUNE_ITERS", "20")))   # rocBLAS/hipBLASLt solution per shape, tuned in the prime phase
        except Exception as exc:                               # tuning is an optimisation, never a requirement
            print("bench: GEMM tuning unavailable (%s); using the BLAS heuristics" % exc, file=sys.stderr)
            args.no_gemm_tuning = True
    # burn-in (preconditioner adaptation) happens in the PRIME phase, so every warm-up and every timed step is
    # in the frozen phase whatever --warmup is
    sampler = build_chain(dev, rank, args.workload, burn_in=PRIME_BURN_IN)
    kind = WORKLOADS[args.workload]["sampler"]
    sampler.sample_format = "view"                             # no D2H copy of 40 MB per sample
    sampler.use_hip_graph = not args.eager
    sampler.overlap_update = args.overlap == "on" and not args.eager
    sampler.collect_stats = "theta_sq"                         # the BNN loss head is the only consumer of the fused statistics
    n = sampler.arena.n
    from pysgmcmc_amd.profiling import UpdateKernelTimer
    # per-launch kernel timestamps of the update kernel; BENCH_BRACKET=1 also records a hipEventRecord pair around each call
    timer = UpdateKernelTimer(bracket=os.environ.get("BENCH_BRACKET", "0") == "1", device=dev)
    sampler.kernel_timer = timer
    from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments, RhatExchange
    moments = ChainMoments(n, dev)
    # --chains-per-gpu K > 1: K - 1 more independent chains on this GPU (chain ids rank + world * c: distinct seeds, initial
    # weights and window streams across the whole job), stepped concurrently with the first one, each on its own stream
    K = max(int(args.chains_per_gpu), 1)
    chains, all_moments, group = [sampler], [moments], None
    if K > 1:
        if kind != "sghmc" or args.eager or sampler.overlap_update:
            raise SystemExit("--chains-per-gpu > 1 is implemented for the SGHMC workload in hipGraph mode without --overlap")
        from pysgmcmc_amd.samplers import ConcurrentChains
        for c in range(1, K):
            other = build_chain(dev, rank + world * c, args.workload, burn_in=PRIME_BURN_IN)
            other.sample_format, other.use_hip_graph, other.collect_stats = "view", sampler.use_hip_graph, sampler.collect_stats
            chains.append(other)
            all_moments.append(ChainMoments(n, dev))
        group = ConcurrentChains(chains)
    exchange = RhatExchange(n, dev, mode=args.rhat_mode) if world > 1 else None
    # R-hat cadence: --rhat-every steps (configs[3]: 100). A timed region shorter than that would contain no
    # collective at all, so one exchange is then placed mid-run: its cost is inside `value` at every N > 1.
    # (started after 2/3 of the steps, collected before the end, so it overlaps with sampling like the periodic ones).
    rhat_every = args.rhat_every if args.steps >= args.rhat_every else max((2 * args.steps + 2) // 3, 1)
    half = max(rhat_every // 2, 1)
    # thinned low-dimensional trace for ESS: [cost, theta[c0], theta[c1], theta[c2]] every moments_every steps,
    # appended on the device (no sync); gathered across chains AFTER the timed region
    coords = torch.tensor([0, n // 2, n - 1], device=dev)
    total_steps = PRIME_BURN_IN + PRIME_FROZEN + PRIME_STEADY + args.steps + args.warmup
    trace = torch.zeros(total_steps // max(args.moments_every, 1) + PRIME_FROZEN + 2, 4, device=dev)
    kept = [0]
    ex_events = []                                                     # (start, packed, finish-begin, finish-end) HIP events
    periodic_exchange = [False]                                        # the periodic R-hat exchange runs in the timed region only

    def one_step(i, every=None):
        every = args.moments_every if every is None else every
        # K4 rides in the update launch of every `every`-th step (sgmcmc_step_opts_t.moments_*): no separate pass over theta
        for chain, mom in zip(chains, all_moments):
            chain.attach_moments(mom if every else None, every or 1)
        if group is None:
            _, cost = next(sampler)
        else:
            cost = next(group)[0][1]                                   # every chain of this GPU, round-robin on their streams
        if every and sampler.n_iterations % every == 0:                # this step's update folded theta' into the moments
            with torch.cuda.stream(group.streams[0] if group is not None else torch.cuda.current_stream(dev)):
                trace[kept[0], 0:1].copy_(cost.reshape(1))             # (the thinned ESS trace follows the GPU's first chain)
                torch.index_select(sampler.arena.row("theta"), 0, coords, out=trace[kept[0], 1:4])
            kept[0] += 1
        if exchange is not None and periodic_exchange[0]:
            # the only exchange on the path: ONE all-reduce of 3P floats over RCCL/xGMI, issued
            # asynchronously and collected half a period later, so it overlaps with sampling.
            # finish() leaves the R-hat summary on the device: no host synchronisation in the loop.
            if (i + 1) % rhat_every == 0 and moments.count >= 2 and not exchange.pending:
                rhat_start()
            elif exchange.pending and (i + 1) % rhat_every == half % rhat_every:
                rhat_finish()

    def rhat_start():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        if group is not None:
            group.join()                                               # the pack reads every local chain's moments ...
        exchange.start(all_moments if K > 1 else moments)              # pack kernel(s) + async collective (RCCL stream)
        if group is not None:
            group.fork()                                               # ... before the chains update them again
        ev[1].record()
        ex_events.append(ev)

    def rhat_finish():
        ev = ex_events[-1]
        ev[2].record()
        exchange.finish()                                              # stream wait + finish kernel + K6 summary
        ev[3].record()
        ev.append("done")

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Host hygiene: a full (generation-2) Python garbage collection walks every object torch and numpy created at
    # import time and takes ~40 ms here -- it fired once per ~250 steps INSIDE long timed regions and starved the
    # device. Collect now, BEFORE the prime phase (a pause after it would let the device clocks drop again), and
    # freeze the survivors (gc stays enabled; later collections only see new objects).
    gc.collect()
    if os.environ.get("BENCH_NO_GC_FREEZE") != "1":
        gc.freeze()
    # ---- phase 1: PRIME (untimed, independent of --warmup): burn-in, then every other code path once
    for i in range(PRIME_BURN_IN):
        one_step(i, every=0)
    assert not getattr(sampler, "_adapting", False), "prime phase must leave the chain in the frozen phase"
    for i in range(PRIME_FROZEN):
        one_step(i, every=1)                                           # frozen step with the fused K4 + trace append
    for i in range(PRIME_STEADY):
        one_step(i, every=0)                                           # plain frozen steps until the device runs steadily
    if exchange is not None:
        try:
            rhat_start()
            rhat_finish()
            exchange.summary.as_dict()
        except RuntimeError as exc:                                    # e.g. a backend without reduce-scatter support
            if exchange.mode != "reduce_scatter":
                raise
            print("bench: reduce-scatter exchange failed (%s); falling back to the all-reduce exchange" % exc, file=sys.stderr)
            exchange = RhatExchange(n, dev, mode="allreduce")
            del ex_events[:]
            rhat_start()
            rhat_finish()
            exchange.summary.as_dict()
    prime_rhat_events = len(ex_events)
    fence()
    # ---- phase 2: --warmup untimed steps (the Welford moments keep accumulating from the prime phase on, so an R-hat
    # exchange is possible from the first timed step)
    kept[0] = 0
    for i in range(args.warmup):
        one_step(i)
    frozen_phase = not getattr(sampler, "_adapting", False)
    kept[0] = 0
    # ---- phase 3: the timed region
    launches_per_step = len(sampler._graphs[("cost",)][0]) if sampler.use_hip_graph else 1
    timer.reserve(args.steps * launches_per_step)
    time_every = args.time_every if args.time_every > 0 else max(1, min(7, args.steps // 5))
    timer.sample_every = time_every                                    # a step is timed iff its number % time_every == 0
    timer.enabled = True
    periodic_exchange[0] = True
    fence()
    t0 = time.perf_counter()
    host_stamps = [t0]
    depth = args.max_queue_depth
    step_end = []                                                      # (i, last update launch) of every step that was timed
    seen, synced = 0, 0
    for i in range(args.steps):
        one_step(i)
        if len(timer.kevents) > seen:
            seen = len(timer.kevents)
            step_end.append((i, timer.kevents[-1]))
            # host-side flow control: never run more than `depth` (+ time_every) steps ahead of the device (the HIP runtime
            # lets the host queue ~750 steps and then stalls host AND device for milliseconds while it recycles its pools)
            while depth and synced < len(step_end) and step_end[synced][0] <= i - depth:
                synced += 1
                if synced == len(step_end) or step_end[synced][0] > i - depth:
                    step_end[synced - 1][1].synchronize()
        host_stamps.append(time.perf_counter())                        # host-side enqueue time of each step (no sync)
    if exchange is not None and exchange.pending:                      # inside the timed region
        rhat_finish()
    fence()
    elapsed = time.perf_counter() - t0
    host_ms = np.diff(np.array(host_stamps)) * 1e3
    final_fence_ms = (t0 + elapsed - host_stamps[-1]) * 1e3
    timer.enabled = False
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(sampler.arena.row("theta")).all()
    ess = None
    if kept[0] >= 8:
        from pysgmcmc_amd.diagnostics.sampler_diagnostics import ess_across_ranks
        ess = ess_across_ranks(trace[:kept[0]].contiguous())          # all-gather of kept x 4 floats (untimed)

    line = None
    if rank == 0:
        mode = "rsghmc" if kind == "rsghmc" else "%s_%s" % (kind, "frozen" if frozen_phase else "adapt")
        op_name = {"sghmc": "SghmcOp", "sgld": "SgldOp", "rsghmc": "RsghmcOp"}[kind]
        br = timer.bracket_us()                     # hipEventRecord bracket around the call (BENCH_BRACKET=1), else empty
        b_us = float(br.mean()) if br.size else None
        ev_us = timer.empty_bracket_us()
        rows = launch_table(timer, n, BYTES_PER_PARAM[mode], args.moments_every)
        plain = [r for r in rows if not r[5]] or rows                  # launches without the fused Welford update
        k_us_sum = float(sum(r[3] for r in plain))
        alg_plain = float(sum(r[4] for r in plain))
        achieved = alg_plain / (k_us_sum * 1e-6) / 1e9
        steps_plain = len({r[0] for r in plain}) if plain[0][0] is not None else len(plain)
        k_us = k_us_sum / max(steps_plain, 1)                          # update time per step (sum of its launches)
        alg_bytes = BYTES_PER_PARAM[mode] * n
        big = alg_bytes > (640 << 20)
        traffic, traffic_src = pmc_traffic(mode, n, variant="_tsq")       # the pipeline launches the sum-theta^2-only variant
        # per-step device time: from the end of one step's last update launch to the end of the next one's
        step_ms = np.array([step_end[j][1].us_until(step_end[j + 1][1]) / (step_end[j + 1][0] - step_end[j][0])
                            for j in range(len(step_end) - 1)]) * 1e-3 if len(step_end) > 1 else None
        slices = None
        if launches_per_step > 1:
            slices = []
            for lo, hi in sorted({(r[1], r[2]) for r in plain}, reverse=True):
                sel = [r for r in plain if (r[1], r[2]) == (lo, hi)]
                us = float(np.mean([r[3] for r in sel]))
                slices.append({"elements": [lo, hi], "params": hi - lo, "us_per_launch_mean": round(us, 2),
                               "GBps": round(BYTES_PER_PARAM[mode] * (hi - lo) / us / 1e3, 1),
                               "stream": "main" if lo == 0 else "side (under the backward GEMMs)"})
        with_mom = [r for r in rows if r[5]]
        line = {
            "metric": "MCMC samples/sec + fused-update HBM GB/s (% roofline), BNN 10M params",
            "value": round(world * K * args.steps / elapsed, 2),
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "step_ms_median": round(float(np.median(step_ms)), 4) if step_ms is not None else None,
            "step_ms_max": round(float(step_ms.max()), 4) if step_ms is not None else None,
            # host side of the timed region: enqueue time per step (the device runs asynchronously behind it) and
            # the time the closing fence waited for the device to drain
            "host_enqueue_ms": {"first_step": round(float(host_ms[0]), 4), "median": round(float(np.median(host_ms)), 4),
                                "max": round(float(host_ms.max()), 4), "argmax": int(host_ms.argmax()),
                                "final_fence": round(final_fence_ms, 4)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s (%s) full next(sampler) step: BNN fwd+bwd + fused update; "
                                   "4-layer tanh MLP BNN %s-1, %d params, batch %d, %d chain(s) per GPU" % (
                                       args.workload, kind.upper(), mode, "-".join(map(str, WORKLOADS[args.workload]["layers"])),
                                       n, BATCH, K),
                       "params": n, "batch": BATCH, "chains": world * K, "chains_per_gpu": K,
                       "rhat_every": rhat_every if world > 1 else None,
                       "moments_every": args.moments_every, "moments": "fused into the update launch (K4 in K1)",
                       "hip_graph": bool(sampler.use_hip_graph),
                       "update_overlap": "layer slices on a side stream under the backward GEMMs (%d launches per step)" %
                                         launches_per_step if launches_per_step > 1 else "off: one launch after the backward pass",
                       "gemm_tuning": not args.no_gemm_tuning,
                       "prime_steps": {"burn_in": PRIME_BURN_IN, "frozen": PRIME_FROZEN + PRIME_STEADY},
                       "max_queue_depth": args.max_queue_depth, "time_every": time_every,
                       "launch": kernels.get_launch_config(), "kernel_source_hash": kernel_source_hash()},
            # template args: <Op<float, ADAPT, INJECT>, quads per lane, NT, STATS (2 = sum theta^2 only), LOOP, MOMENTS>, from
            # the launch configuration in effect (library defaults: 1 quad per lane, nt iff the launch streams > 640 MiB,
            # single-pass variant while the grid is uncapped)
            "roofline": {"bound": "hbm", "kernel": update_kernel_instance(op_name, not frozen_phase, big, sampler),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes if launches_per_step == 1 else None,
                         "algorithmic_bytes_per_step": alg_bytes,
                         "us_per_step_mean": round(k_us, 2),
                         "launches_timed": len(rows), "launches_in_the_rate": len(plain), "launches_per_step": launches_per_step,
                         "slices": slices,
                         "with_fused_moments": None if not with_mom else {
                             "launches": len(with_mom), "bytes_per_param": BYTES_PER_PARAM[mode] + 16,
                             "us_per_step_mean": round(sum(r[3] for r in with_mom) / len({r[0] for r in with_mom}), 2),
                             "GBps": round(sum(r[4] for r in with_mom) / sum(r[3] for r in with_mom) / 1e3, 1)},
                         # the conservative figure of round 1: hipEventRecord pair AROUND the call
                         "bracket": None if b_us is None else {
                             "us_per_launch_mean": round(b_us, 2), "us_per_launch_median": round(float(np.median(br)), 2),
                             "us_empty_event_pair": round(ev_us, 2)},
                         "cache_note": ("%.0f MB per step: HBM-resident (larger than the 256 MiB Infinity Cache)" if big else
                                        "%.0f MB per step fits the 256 MiB Infinity Cache: part of this rate is cache-"
                                        "assisted; the HBM-resident figure is `roofline_hbm_resident`") % (alg_bytes / 1e6),
                         "timing": "the update launches of every %d-th step of the timed region carry a HIP event pair that receives "
                                   "the kernel's own start/stop timestamps (hipExtLaunchKernel; the duration rocprofv3 reports) -- "
                                   "not every step, because a launch with events costs the step 8 us of device time "
                                   "(tools/bench_overhead_probe.py); achieved = algorithmic bytes of the timed launches / the sum of "
                                   "their durations; `roofline_unoverlapped` times EVERY launch of a second loop outside `value`" % time_every + (
                                       " -- the slices run CONCURRENTLY with the backward GEMMs, so this is the contended "
                                       "rate; `roofline_unoverlapped` is the same kernel alone in the pipeline"
                                       if launches_per_step > 1 else "") + (
                                       " -- with %d chains per GPU the timed launches (first chain) run CONCURRENTLY with the other "
                                       "chains' kernels: contended rate; `roofline_unoverlapped` is the kernel alone in its pipeline" % K
                                       if K > 1 else "")},
        }
        if exchange is not None:
            timed = ex_events[prime_rhat_events:]
            done = [ev for ev in timed if len(ev) == 5]
            line["rccl"] = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "mode": exchange.mode,
                            "exchanges_timed": len(done),
                            "payload_bytes": int(exchange.pack.numel() * exchange.pack.element_size())}
            if done:
                # start -> finish wall on the compute stream (includes the steps sampled in between), the pack launch,
                # and the tail the compute stream actually spends on the exchange when it collects it
                line["rccl"]["rhat_exchange_ms"] = {
                    "start_to_finish": round(float(np.mean([ev[0].elapsed_time(ev[3]) for ev in done])), 3),
                    "pack_and_issue": round(float(np.mean([ev[0].elapsed_time(ev[1]) for ev in done])), 3),
                    "wait_finish_summary": round(float(np.mean([ev[2].elapsed_time(ev[3]) for ev in done])), 3)}
            line["rccl"]["collective_alone_ms"] = None
            line["rhat"] = {k: round(v, 4) for k, v in exchange.summary.as_dict().items()} if exchange.exchanges else None
        if ess is not None:
            line["ess"] = {"kept_per_chain": kept[0], "cost": ess[0], "theta_coords": ess[1:]}
    if exchange is not None:
        # the collective alone: blocking collective of the same payload, after the timed region (all ranks)
        torch.cuda.synchronize()
        dist.barrier()
        ts = []
        native_rs = exchange.mode == "reduce_scatter" and exchange._native_rs
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if native_rs:
                dist.reduce_scatter_tensor(exchange.shard_sum, exchange.pack)
            else:
                dist.all_reduce(exchange.pack)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        if rank == 0:
            line["rccl"]["collective_alone_ms"] = round(float(np.median(ts[1:])), 3)
    if dist is not None:
        # the job ends HERE for every rank: rank 0's extra legs below run with no process group alive, so no rank sits
        # in a collective (or its watchdog) while they take their seconds
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    if group is not None:                                              # the legs below step the GPU's first chain alone
        group.join()
        torch.cuda.synchronize()
        del chains[1:], all_moments[1:]
        group = None
        torch.cuda.empty_cache()
    if kind == "sghmc" and sampler.use_hip_graph:
        # ---- un-overlapped figures + where the step time goes (identical code path at every N: SCALE N = 1 equals BENCH)
        legs = max(min(args.steps, 60), 20)
        sampler.attach_moments(None)
        if launches_per_step > 1:
            sampler.overlap_update = False
            sampler._graphs.clear()
            for _ in range(6):
                next(sampler)
        t2 = UpdateKernelTimer(reserve=legs, device=dev)
        sampler.kernel_timer = t2
        t2.enabled = True
        for _ in range(legs):
            next(sampler)
        torch.cuda.synchronize()
        t2.enabled = False
        sampler.kernel_timer = None
        u_us = t2.kernel_us()
        serial_step_us = float(np.median(t2.step_us()))
        un = BYTES_PER_PARAM[mode] * n / (float(u_us.mean()) * 1e-6) / 1e9
        line["roofline_unoverlapped"] = {
            "bound": "hbm", "achieved": round(un, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(un / HBM_PEAK_GBS, 4),
            "us_per_launch_mean": round(float(u_us.mean()), 2), "us_per_launch_median": round(float(np.median(u_us)), 2),
            "launches_timed": int(u_us.size), "step_ms_median": round(serial_step_us * 1e-3, 4),
            "note": "the same chain stepped with ONE update launch after the backward pass (no overlap), %d steps after the "
                    "timed region: the kernel alone in the pipeline, as rounds 1-2 reported `roofline`" % legs}
        g_us, g_flops = gemm_only_us(sampler)
        c_us = cost_pipeline_us(sampler)
        meas_us = float(np.median(step_ms)) * 1e3 if step_ms is not None else None
        line["step_breakdown_us"] = {
            "gemm": round(g_us, 1), "small_launches": round(c_us - g_us, 1), "update": round(float(u_us.mean()), 1),
            "serial_sum": round(c_us + float(u_us.mean()), 1), "serial_step_measured": round(serial_step_us, 1),
            "timed_region_step_median": None if meas_us is None else round(meas_us, 1),
            "hidden_by_overlap": round(serial_step_us - meas_us, 1) if (meas_us is not None and launches_per_step > 1) else None,
            "gemm_tflops": round(g_flops / (g_us * 1e-6) / 1e12, 1),
            "gemm_frac_of_fp32_mfma_peak": round(g_flops / (g_us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 3),
            "gemm_flop_per_step": int(g_flops),
            "note": "gemm = the step's eight fp32 library GEMMs replayed alone from a hipGraph; small_launches = the captured "
                    "cost pipeline alone minus gemm (bias + tanh, tanh-backward + bias gradient, loss head ...); "
                    "update = the fused update launched once after the backward pass. Differences of graph replays: rocprofv3's "
                    "per-kernel durations of the same step (profiles/r03_bench10m_kernel_stats.csv: GEMMs 136, seven small "
                    "launches 37, update 35 us) carry ~1.5 us of profiler overhead per kernel"}
    if not args.no_update_only and kind == "sghmc":
        if sampler.use_hip_graph and K == 1:
            line["chains_per_gpu"] = chains_per_gpu_leg(dev, sampler, args.workload)
        line["update_only"] = update_only(sampler)
        del moments, trace
        line["roofline_hbm_resident"] = hbm_resident_roofline(dev)
    if world == 1 and not args.no_cpu_baseline and kind == "sghmc":
        line["cpu_baseline"] = cpu_baseline(n, args.cpu_seconds)
    print(json.dumps(line))
    sys.stdout.flush()


if __name__ == "__main__":
    main()

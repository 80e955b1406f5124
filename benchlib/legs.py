"""Measurements of ``bench.py`` that run AFTER the timed region (rank 0, no process group alive): the update kernel alone,
the HBM-resident sizes, where the step time goes, two chains per GPU."""
import time

import numpy as np
import torch

from benchlib.common import BYTES_PER_PARAM, HBM_PEAK_GBS, N_DATA, N_HBM_RESIDENT, PRIME_BURN_IN, pmc_traffic
from benchlib.workloads import build_chain


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
                   "(launch ramps and tails of the 11 dependent launches per step, M = 256 products at ~0.76-0.79 of the matrix pipe inside a launch)")
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
        # burn-in steps that do not write minv (samplers' store_minv_every_step = False: only the LAST burn-in step stores it)
        "sghmc_adapt_no_minv_store": lambda i, L: kernels.sghmc_step(theta, V, grad, tau, gg, vh, minv, None, 0.01, float(N_DATA), 0.05, True,
                                                                    seed=1, step=i, launch=L, opts=dict(skip_minv_store=True)),
        "sgld_adapt_no_minv_store": lambda i, L: kernels.sgld_step(theta, grad, tau, gg, vh, minv, None, 1e-3, 1.0, float(N_DATA), True,
                                                                  seed=1, step=i, launch=L, opts=dict(skip_minv_store=True)),
    }
    # the reference's default dtype is float64 (pysgmcmc/samplers/base_classes.py:25): the frozen-phase kernels on doubles, same
    # parameter count (twice the bytes), own arrays
    d64 = {}

    def f64_arrays():
        if not d64:
            g64 = torch.Generator(device=dev).manual_seed(1)
            d64["theta"] = torch.randn(n, device=dev, generator=g64, dtype=torch.float64) * 0.02
            d64["V"] = torch.zeros(n, device=dev, dtype=torch.float64)
            d64["grad"] = torch.randn(n, device=dev, generator=g64, dtype=torch.float64) * 0.1
            d64["minv"] = torch.rand(n, device=dev, generator=g64, dtype=torch.float64) * 1.5 + 0.5
        return d64["theta"], d64["V"], d64["grad"], d64["minv"]
    calls64 = {
        "sghmc_frozen_f64": lambda i, L: (lambda t, v, gr, mi: kernels.sghmc_step(t, v, gr, None, None, None, mi, None, 0.01, float(N_DATA), 0.05,
                                                                              False, seed=1, step=i, launch=L))(*f64_arrays()),
        "sgld_frozen_f64": lambda i, L: (lambda t, v, gr, mi: kernels.sgld_step(t, gr, None, None, None, mi, None, 1e-3, 1.0, float(N_DATA), False,
                                                                            seed=1, step=i, launch=L))(*f64_arrays()),
        "rsghmc_f64": lambda i, L: (lambda t, v, gr, mi: kernels.rsghmc_step(t, v, gr, 1e-3, 1.0, 1.0, 1.0, 0.0, seed=1, step=i, launch=L))(*f64_arrays()),
    }
    bytes_per_param = dict(BYTES_PER_PARAM, sghmc_adapt_no_minv_store=44, sgld_adapt_no_minv_store=36,
                           sghmc_frozen_f64=48, sgld_frozen_f64=32, rsghmc_f64=40)
    out = {}
    for name, call in list(calls.items()) + list(calls64.items()):
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
        alg = bytes_per_param[name] * n
        traffic, src = pmc_traffic(name, n) if name in BYTES_PER_PARAM else (None, None)
        out[name] = {"us_per_launch_mean": round(float(us.mean()), 2), "us_per_launch_median": round(float(np.median(us)), 2),
                     "us_bracket_mean": round(float(bracket.mean()), 2),
                     "algorithmic_bytes_per_launch": alg, "achieved": round(alg / (us.mean() * 1e-6) / 1e9, 1),
                     "frac": round(alg / (us.mean() * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "launches_timed": iters}
        theta.normal_(0.0, 0.02, generator=g)              # keep the chain finite across 225 synthetic steps
        V.zero_()
        if d64:
            d64["theta"].normal_(0.0, 0.02)
            d64["V"].zero_()
    assert torch.isfinite(theta).all() and torch.isfinite(d64["theta"]).all()
    d64.clear()
    head = out["sghmc_frozen"]
    return {"bound": "hbm", "kernel": "stream_quads_vec<SghmcOp<float,false,false>,1,true,0,false,false> (128-lane blocks, nt)",
            "achieved": head["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head["frac"],
            "traffic": head["traffic"], "traffic_source": pmc_traffic("sghmc_frozen", n)[1],
            "params": n, "working_set_note": "%.2f GB per frozen SGHMC launch: HBM-resident, cannot be served by the 256 MiB "
                                             "Infinity Cache" % (head["algorithmic_bytes_per_launch"] / 1e9),
            "timing": "kernel start/stop timestamps (hipExtLaunchKernel events) of every launch, back to back, after the "
                      "timed region (not part of `value`); us_bracket_mean = hipEventRecord pair around the call",
            "kernels": out}


@torch.no_grad()
def gemm_only_us(sampler, iters=60):
    """The eight fp32 products of one step (three forward, five backward; same operands, shapes and output buffers as the cost
    pipeline) replayed back to back from their own hipGraph: microseconds per step and their FLOP count. Each is what the
    pipeline runs: the forward layers as the fused product + bias + tanh launch (kernels.bnn_dense_tanh) and the delta W^T products
    as the fused product + tanh' launch (kernels.bnn_dense_tanh_backward) where BNNCost's plan puts them there, else the library GEMM; the first layer's weight gradient with its extra bias row when the pipeline
    forms it that way. Returns (us, flops, forward layers on the fused launch, backward products on the fused launch)."""
    from pysgmcmc_amd import kernels
    cost, params, gv = sampler.cost_fun, sampler.params, sampler.arena.grad_views
    X = cost.x_placeholder.value
    ws = cost._buffers(params, X.shape[0])
    hs, ds = ws["h"], ws["d"]
    L = (len(params) - 1) // 2 - 1                              # index of the single-output layer
    n_fused = [0, 0]
    plan = cost._plan(params, gv, X, ws, True)                  # the plan the sampler's steps walk (statistics partials at hand)

    def gemms():
        h, flops, n_fused[0], n_fused[1] = X, 0, 0, 0
        for l in range(L):
            if plan.forward[l].startswith("dense_tanh"):
                kernels.bnn_dense_tanh(h, params[2 * l], params[2 * l + 1].view(-1), hs[l])
                n_fused[0] += 1
            else:
                torch.mm(h, params[2 * l], out=hs[l])              # the bias rides in the activation launch
            flops += 2 * h.shape[0] * h.shape[1] * params[2 * l].shape[1]
            h = hs[l]
        for l in range(L - 1, -1, -1):
            h_in = X if l == 0 else hs[l - 1]
            if l > 0:
                if plan.backward[l] == "dense_tanh_backward":
                    kernels.bnn_dense_tanh_backward(ds[l], params[2 * l], hs[l - 1], ds[l - 1])
                    n_fused[1] += 1
                else:
                    torch.mm(ds[l], params[2 * l].t(), out=ds[l - 1])
                flops += 2 * ds[l].shape[0] * ds[l].shape[1] * params[2 * l].shape[0]
            if plan.gw_batch is not None and plan.gw_batch[0] <= l <= plan.gw_batch[1]:
                lo, hi, s_h, s_d, s_g, _ = plan.gw_batch
                if l == lo:                                    # the group's weight gradients as ONE strided batched product
                    stack = lambda t, st: torch.as_strided(t, (hi - lo + 1,) + tuple(t.shape), (st,) + tuple(t.stride()))
                    torch.bmm(stack(hs[lo - 1], s_h).transpose(1, 2), stack(ds[lo], s_d), out=stack(gv[2 * lo], s_g))
            elif l == 0 and plan.ones_row:
                d_in, width = int(X.shape[1]), int(params[0].shape[1])
                torch.mm(plan.x_ones.t(), ds[0], out=torch.as_strided(gv[0], (d_in + 1, width), (width, 1)))
            else:
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
    return e0.elapsed_time(e1) / iters * 1e3, flops, n_fused[0], n_fused[1]


def cost_pipeline_us(sampler, iters=60):
    """The captured cost/gradient pipeline alone (every graph segment, no update launch): microseconds per step."""
    graph = sampler._graphs[("cost",)][0]
    for _ in range(5):
        graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def update_kernel_instance(op_name, adapt, big, sampler, moments=False, scalar="float"):
    """Name of the stream_quads_vec instance the library launches for this sampler's update under the launch
    configuration in effect (sampler.launch, else the Python-side default, else the library's defaults)."""
    from pysgmcmc_amd import kernels
    cfg = dict(kernels.get_launch_config())
    if sampler.launch is not None:
        cfg.update({k: v for k, v in sampler.launch.as_dict().items() if v != (-1 if k == "nontemporal" else 0)})
    qpt = cfg["quads_per_thread"]
    nt = big if cfg["nontemporal"] == 2 else bool(cfg["nontemporal"])
    if scalar == "double" and cfg["nontemporal"] == 2:
        nt = False                                         # f64 launches use plain accesses at every size (round 5)
    bt = cfg["block_threads"] if cfg["block_threads"] > 0 else (128 if big else 256)
    loop = qpt != 1 or (sampler.arena.n // 4 + bt - 1) // bt > cfg["max_blocks"]
    stats = {True: 1, "theta_sq": 2}.get(sampler.collect_stats, 0)
    if loop and stats == 2:
        stats = 1                                          # the looping variants reduce every statistic
    return "stream_quads_vec<%s<%s,%s,false>,%d,%s,%d,%s,%s> (%d-lane blocks)" % (
        op_name, scalar, "true" if adapt else "false", qpt, "true" if nt else "false", stats, "true" if loop else "false",
        "true" if (moments and not loop) else "false", bt)


def launch_table(timer, n, bytes_per_param, moments_every, moments_bytes=16):
    """Per-launch records of a timed region: (step, lo, hi, microseconds, algorithmic bytes). A launch of a moments
    step also carries the fused Welford update (+16 B per f32 parameter)."""
    rows = []
    for kev, tag in zip(timer.kevents, timer.tags):
        step, lo, hi = tag if tag is not None else (None, 0, n)
        with_mom = step is not None and moments_every and (step + 1) % moments_every == 0
        rows.append((step, lo, hi, kev.elapsed_us(), (bytes_per_param + (moments_bytes if with_mom else 0)) * (hi - lo), bool(with_mom)))
    return rows




def product_defaults_leg(args, timeout=600):
    """`value` again for the SAME chain with nothing set: a child process (the graph launch path is fixed when the HIP runtime
    initialises, so it cannot be switched back in this one) runs ``bench.py --product-defaults`` -- no
    ``pysgmcmc_amd.configure_for_device_bound_chains()``: the runtime's default graph launch; the GEMM solutions picked by
    ``BNNCost`` itself in the first evaluation of its plan (``auto_gemm_tuning``, the product default since round 6) -- with the
    same workload, steps and warm-up, while this process sits idle. What a user of the public API gets by default."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--workload", args.workload, "--moments-every", str(args.moments_every), "--product-defaults"]
    torch.cuda.synchronize()
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, check=True).stdout.decode()
        child = json.loads(out.strip().splitlines()[-1])
    except Exception as exc:                                   # the main line must not die of its extra figure
        return {"value": None, "error": repr(exc)[:300]}
    return {"value": child["value"], "unit": "samples/s", "ms_per_step": child["ms_per_step"],
            "gemm_tuning": child["config"]["gemm_tuning"], "hip_runtime_env": child["config"]["hip_runtime_env"],
            "note": "the same workload, steps and warm-up in a child process WITHOUT pysgmcmc_amd.configure_for_device_bound_chains(): "
                    "what the public API gives with nothing set (gemm_tuning 'auto': BNNCost tunes the first evaluation of its plan by itself; the "
                    "runtime's default graph launch); `value` is with that one documented call made first"}

"""The timed chain run of ``bench.py``: one chain per GPU (or ``--chains-per-gpu`` K) of a BNN workload stepped through
``next(sampler)``, with the fused Welford moments, the thinned ESS trace and -- for N > 1 -- the asynchronous R-hat exchange.

Phases (``ChainBench.prime`` / ``warmup`` / ``timed``): see the module docstring of ``bench.py``. ``headline()`` builds the
JSON line from what the timed region recorded; ``post_run_legs()`` adds the measurements taken after it."""
import gc
import os
import sys
import time

import numpy as np
import pysgmcmc_amd
import torch

from benchlib import legs
from benchlib.baselines import cpu_baseline
from benchlib.common import (BATCH, BYTES_PER_PARAM, FP32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, PRIME_BURN_IN, PRIME_FROZEN, PRIME_STEADY,
                             RHAT_EVERY_CONFIG3, kernel_source_hash, pmc_traffic)
from benchlib.workloads import WORKLOADS, build_chain


class ChainBench(object):
    ESS_MIN_KEPT = 50                                          # kept samples per chain below which the line reports no ESS

    def __init__(self, args, dev, rank, world, dist):
        from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments, RhatExchange
        from pysgmcmc_amd.profiling import UpdateKernelTimer
        self.args, self.dev, self.rank, self.world, self.dist = args, dev, rank, world, dist
        # (GEMM selection and the graph launch path: pysgmcmc_amd.configure_for_device_bound_chains(), called by bench.py's main()
        # before anything touched the GPU; args.device_bound_switch holds what took effect)
        # burn-in (preconditioner adaptation) happens in the PRIME phase, so every warm-up and every timed step is
        # in the frozen phase whatever --warmup is
        self.f64 = getattr(args, "dtype", "f32") == "f64"
        self.tdtype = torch.float64 if self.f64 else torch.float32
        self.esize = 2 if self.f64 else 1                      # element size in units of the f32 tables of benchlib/common.py
        self.sampler = sampler = build_chain(dev, rank, args.workload, burn_in=PRIME_BURN_IN, dtype=self.tdtype)
        self.kind = WORKLOADS[args.workload]["sampler"]
        sampler.sample_format = "view"                         # no D2H copy of 40 MB per sample
        sampler.use_hip_graph = not args.eager
        sampler.collect_stats = "theta_sq"                     # the BNN loss head is the only consumer of the fused statistics
        self.n = n = sampler.arena.n
        # per-launch kernel timestamps of the update kernel; BENCH_BRACKET=1 also records a hipEventRecord pair around each call
        self.timer = UpdateKernelTimer(bracket=os.environ.get("BENCH_BRACKET", "0") == "1", device=dev)
        sampler.kernel_timer = self.timer
        self.moments = ChainMoments(n, dev, dtype=self.tdtype)
        # --chains-per-gpu K > 1: K - 1 more independent chains on this GPU (chain ids rank + world * c: distinct seeds, initial
        # weights and window streams across the whole job), stepped concurrently with the first one, each on its own stream
        self.K = K = max(int(args.chains_per_gpu), 1)
        self.chains, self.all_moments, self.group = [sampler], [self.moments], None
        if K > 1:
            if self.kind != "sghmc" or args.eager:
                raise SystemExit("--chains-per-gpu > 1 is implemented for the SGHMC workload in hipGraph mode")
            from pysgmcmc_amd.samplers import ConcurrentChains
            for c in range(1, K):
                other = build_chain(dev, rank + world * c, args.workload, burn_in=PRIME_BURN_IN, dtype=self.tdtype)
                other.sample_format, other.use_hip_graph, other.collect_stats = "view", sampler.use_hip_graph, sampler.collect_stats
                self.chains.append(other)
                self.all_moments.append(ChainMoments(n, dev, dtype=self.tdtype))
            self.group = ConcurrentChains(self.chains)
        self.exchange = RhatExchange(n, dev, mode=args.rhat_mode) if world > 1 else None
        # R-hat cadence: --rhat-every steps (configs[3]: 100). A timed region shorter than that would contain no
        # collective at all, so one exchange is then placed mid-run: its cost is inside `value` at every N > 1
        # (started after 2/3 of the steps, collected before the end, so it overlaps with sampling like the periodic ones).
        self.rhat_every = args.rhat_every if args.steps >= args.rhat_every else max((2 * args.steps + 2) // 3, 1)
        self.half = max(self.rhat_every // 2, 1)
        # thinned low-dimensional trace for ESS: [cost, theta[c0], theta[c1], theta[c2]] every moments_every steps,
        # appended on the device (no sync); gathered across chains AFTER the timed region
        self.coords = torch.tensor([0, n // 2, n - 1], device=dev)
        total_steps = PRIME_BURN_IN + PRIME_FROZEN + PRIME_STEADY + args.steps + args.warmup
        self.trace = torch.zeros(total_steps // max(args.moments_every, 1) + PRIME_FROZEN + 2, 4, device=dev, dtype=self.tdtype)
        self.kept = 0
        self.ex_events = []                                    # (start, packed, finish-begin, finish-end) HIP events
        self.periodic_exchange = False                         # the periodic R-hat exchange runs in the timed region only
        self.prime_rhat_events = 0

    # ------------------------------------------------------------------ one step, the exchange
    def one_step(self, i, every=None):
        args, sampler, group = self.args, self.sampler, self.group
        every = args.moments_every if every is None else every
        # K4 rides in the update launch of every `every`-th step (sgmcmc_step_opts_t.moments_*): no separate pass over theta
        for chain, mom in zip(self.chains, self.all_moments):
            chain.attach_moments(mom if every else None, every or 1)
        if group is None:
            _, cost = next(sampler)
        else:
            cost = next(group)[0][1]                           # every chain of this GPU, round-robin on their streams
        if every and sampler.n_iterations % every == 0:        # this step's update folded theta' into the moments
            with torch.cuda.stream(group.streams[0] if group is not None else torch.cuda.current_stream(self.dev)):
                self.trace[self.kept, 0:1].copy_(cost.reshape(1))      # (the thinned ESS trace follows the GPU's first chain)
                torch.index_select(sampler.arena.row("theta"), 0, self.coords, out=self.trace[self.kept, 1:4])
            self.kept += 1
        ex = self.exchange
        if ex is not None and self.periodic_exchange:
            # the only exchange on the path: ONE collective of 3P floats over RCCL/xGMI, issued asynchronously and collected
            # half a period later, so it overlaps with sampling. finish() leaves the R-hat summary on the device: no host
            # synchronisation in the loop.
            if (i + 1) % self.rhat_every == 0 and self.moments.count >= 2 and not ex.pending:
                self.rhat_start()
            elif ex.pending and (i + 1) % self.rhat_every == self.half % self.rhat_every:
                self.rhat_finish()

    def rhat_start(self):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        if self.group is not None:
            self.group.join()                                  # the pack reads every local chain's moments ...
        self.exchange.start(self.all_moments if self.K > 1 else self.moments)   # pack kernel(s) + async collective (RCCL stream)
        if self.group is not None:
            self.group.fork()                                  # ... before the chains update them again
        ev[1].record()
        self.ex_events.append(ev)

    def rhat_finish(self):
        ev = self.ex_events[-1]
        ev[2].record()
        self.exchange.finish()                                 # stream wait + finish kernel + K6 summary
        ev[3].record()
        ev.append("done")

    def fence(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ phases
    def prime(self):
        """Phase 1 (untimed, independent of --warmup): burn-in, then every other code path once."""
        from pysgmcmc_amd.diagnostics.sampler_diagnostics import RhatExchange
        # Host hygiene: a full (generation-2) Python garbage collection walks every object torch and numpy created at
        # import time and takes ~40 ms here -- it fired once per ~250 steps INSIDE long timed regions and starved the
        # device. Collect now, BEFORE the prime phase (a pause after it would let the device clocks drop again), and
        # freeze the survivors (gc stays enabled; later collections only see new objects).
        gc.collect()
        if os.environ.get("BENCH_NO_GC_FREEZE") != "1":
            gc.freeze()
        for i in range(PRIME_BURN_IN):
            self.one_step(i, every=0)
        assert not getattr(self.sampler, "_adapting", False), "prime phase must leave the chain in the frozen phase"
        for i in range(PRIME_FROZEN):
            self.one_step(i, every=1)                          # frozen step with the fused K4 + trace append
        for i in range(PRIME_STEADY):
            self.one_step(i, every=0)                          # plain frozen steps until the device runs steadily
        if self.exchange is not None:
            try:
                self.rhat_start()
                self.rhat_finish()
                self.exchange.summary.as_dict()
            except RuntimeError as exc:                        # e.g. a backend without reduce-scatter support
                if self.exchange.mode != "reduce_scatter":
                    raise
                print("bench: reduce-scatter exchange failed (%s); falling back to the all-reduce exchange" % exc, file=sys.stderr)
                self.exchange = RhatExchange(self.n, self.dev, mode="allreduce")
                del self.ex_events[:]
                self.rhat_start()
                self.rhat_finish()
                self.exchange.summary.as_dict()
        self.prime_rhat_events = len(self.ex_events)
        self.fence()

    def warmup(self):
        """Phase 2: --warmup untimed steps (the Welford moments keep accumulating from the prime phase on, so an R-hat
        exchange is possible from the first timed step)."""
        self.kept = 0
        for i in range(self.args.warmup):
            self.one_step(i)
        self.frozen_phase = not getattr(self.sampler, "_adapting", False)
        self.kept = 0

    def timed(self):
        """Phase 3: EXACTLY --steps steps between barrier + synchronize fences; elapsed = max over ranks."""
        args, timer = self.args, self.timer
        timer.reserve(args.steps)
        self.time_every = args.time_every if args.time_every > 0 else max(1, min(7, args.steps // 5))
        timer.sample_every = self.time_every                   # a step is timed iff its number % time_every == 0
        timer.enabled = True
        self.periodic_exchange = True
        self.fence()
        t0 = time.perf_counter()
        host_stamps = [t0]
        depth = args.max_queue_depth
        self.step_end = step_end = []                          # (i, last update launch) of every step that was timed
        seen, synced = 0, 0
        for i in range(args.steps):
            self.one_step(i)
            if len(timer.kevents) > seen:
                seen = len(timer.kevents)
                step_end.append((i, timer.kevents[-1]))
                # host-side flow control: never run more than `depth` (+ time_every) steps ahead of the device (the HIP runtime
                # lets the host queue ~750 steps and then stalls host AND device for milliseconds while it recycles its pools)
                while depth and synced < len(step_end) and step_end[synced][0] <= i - depth:
                    synced += 1
                    if synced == len(step_end) or step_end[synced][0] > i - depth:
                        step_end[synced - 1][1].synchronize()
            host_stamps.append(time.perf_counter())            # host-side enqueue time of each step (no sync)
        if self.exchange is not None and self.exchange.pending:    # inside the timed region
            self.rhat_finish()
        self.fence()
        elapsed = time.perf_counter() - t0
        self.host_ms = np.diff(np.array(host_stamps)) * 1e3
        self.final_fence_ms = (t0 + elapsed - host_stamps[-1]) * 1e3
        timer.enabled = False
        # exposed exchange time of THIS rank's compute stream inside the timed region: pack launch + the tail it spends
        # collecting the collective (the collective itself runs on RCCL's stream under the sampling steps)
        done = [ev for ev in self.ex_events[self.prime_rhat_events:] if len(ev) == 5]
        self.exposed_ms = float(sum(ev[0].elapsed_time(ev[1]) + ev[2].elapsed_time(ev[3]) for ev in done))
        if self.dist is not None:
            t = torch.tensor([elapsed, self.exposed_ms], dtype=torch.float64, device=self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed, self.exposed_ms_max = float(t[0].item()), float(t[1].item())
        else:
            self.exposed_ms_max = self.exposed_ms
        self.elapsed = elapsed
        assert torch.isfinite(self.sampler.arena.row("theta")).all()
        self.ess = None
        if self.kept >= self.ESS_MIN_KEPT:
            from pysgmcmc_amd.diagnostics.sampler_diagnostics import ess_across_ranks
            self.ess = ess_across_ranks(self.trace[:self.kept].contiguous())     # all-gather of kept x 4 floats (untimed)

    # ------------------------------------------------------------------ the line
    def headline(self):
        """The JSON line (rank 0; None elsewhere) from what the timed region recorded."""
        if self.rank != 0:
            return None
        from pysgmcmc_amd import kernels
        args, sampler, timer, n, K, world = self.args, self.sampler, self.timer, self.n, self.K, self.world
        kind, elapsed, step_end = self.kind, self.elapsed, self.step_end
        self.mode = mode = "rsghmc" if kind == "rsghmc" else "%s_%s" % (kind, "frozen" if self.frozen_phase else "adapt")
        op_name = {"sghmc": "SghmcOp", "sgld": "SgldOp", "rsghmc": "RsghmcOp"}[kind]
        br = timer.bracket_us()                     # hipEventRecord bracket around the call (BENCH_BRACKET=1), else empty
        b_us = float(br.mean()) if br.size else None
        ev_us = timer.empty_bracket_us()
        bpp = BYTES_PER_PARAM[mode] * self.esize                # f64: every element touch is 8 bytes
        rows = legs.launch_table(timer, n, bpp, args.moments_every, moments_bytes=16 * self.esize)
        plain = [r for r in rows if not r[5]] or rows          # launches without the fused Welford update
        k_us_sum = float(sum(r[3] for r in plain))
        achieved = float(sum(r[4] for r in plain)) / (k_us_sum * 1e-6) / 1e9
        k_us = k_us_sum / max(len(plain), 1)
        alg_bytes = bpp * n
        big = alg_bytes > (640 << 20)
        traffic, traffic_src = pmc_traffic(mode, n, variant="_tsq") if not self.f64 else (None, "no PMC pass of the f64 instances")       # the pipeline launches the sum-theta^2-only variant
        # per-step device time: from the end of one timed step's update launch to the end of the next one's
        self.step_ms = step_ms = np.array([step_end[j][1].us_until(step_end[j + 1][1]) / (step_end[j + 1][0] - step_end[j][0])
                                           for j in range(len(step_end) - 1)]) * 1e-3 if len(step_end) > 1 else None
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
            "host_enqueue_ms": {"first_step": round(float(self.host_ms[0]), 4), "median": round(float(np.median(self.host_ms)), 4),
                                "max": round(float(self.host_ms.max()), 4), "argmax": int(self.host_ms.argmax()),
                                "final_fence": round(self.final_fence_ms, 4)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if self.f64 else "f32", "data": "synthetic",
            "config": {"workload": "%s: %s (%s) full next(sampler) step: BNN fwd+bwd + fused update; "
                                   "4-layer tanh MLP BNN %s-1, %d params, batch %d, %d chain(s) per GPU" % (
                                       args.workload, kind.upper(), mode, "-".join(map(str, WORKLOADS[args.workload]["layers"])),
                                       n, BATCH, K),
                       "params": n, "batch": BATCH, "chains": world * K, "chains_per_gpu": K,
                       "rhat_every": self.rhat_every if world > 1 else None,
                       "moments_every": args.moments_every, "moments": "fused into the update launch (K4 in K1)",
                       "product_arithmetic": self._product_arithmetic(),
                       "hip_graph": bool(sampler.use_hip_graph), "gemm_tuning": getattr(sampler.cost_fun, "gemm_tuning_applied", None) or (not args.no_gemm_tuning),
                       "prime_steps": {"burn_in": PRIME_BURN_IN, "frozen": PRIME_FROZEN + PRIME_STEADY},
                       "max_queue_depth": args.max_queue_depth, "time_every": self.time_every,
                       "launch": kernels.get_launch_config(), "kernel_source_hash": kernel_source_hash(),
                       "device_bound_switch": getattr(args, "device_bound_switch", None),
                       "hip_runtime_env": pysgmcmc_amd.runtime_env(),
                       "hip_runtime_env_effective": bool((getattr(args, "device_bound_switch", None) or {}).get("plain_graph_launch", False))},
            # template args: <Op<float, ADAPT, INJECT>, quads per lane, NT, STATS (2 = sum theta^2 only), LOOP, MOMENTS>, from
            # the launch configuration in effect (library defaults: 1 quad per lane, nt iff the launch streams > 640 MiB,
            # single-pass variant while the grid is uncapped)
            "roofline": {"bound": "hbm", "kernel": legs.update_kernel_instance(op_name, kind == "rsghmc" or not self.frozen_phase, big, sampler, scalar="double" if self.f64 else "float"),   # (2nd template arg of RsghmcOp = POW2: m = c = 1)
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "us_per_launch_mean": round(k_us, 2),
                         "launches_timed": len(rows), "launches_in_the_rate": len(plain),
                         "with_fused_moments": None if not with_mom else {
                             "launches": len(with_mom), "bytes_per_param": bpp + 16 * self.esize,
                             "us_per_launch_mean": round(sum(r[3] for r in with_mom) / len(with_mom), 2),
                             "GBps": round(sum(r[4] for r in with_mom) / sum(r[3] for r in with_mom) / 1e3, 1)},
                         # the conservative figure of round 1: hipEventRecord pair AROUND the call
                         "bracket": None if b_us is None else {
                             "us_per_launch_mean": round(b_us, 2), "us_per_launch_median": round(float(np.median(br)), 2),
                             "us_empty_event_pair": round(ev_us, 2)},
                         "cache_note": ("%.0f MB per launch: HBM-resident (larger than the 256 MiB Infinity Cache)" if big else
                                        "%.0f MB per launch fits the 256 MiB Infinity Cache: part of this rate is cache-"
                                        "assisted; the HBM-resident figure is `roofline_hbm_resident`") % (alg_bytes / 1e6),
                         "timing": "the update launch of every %d-th step of the timed region carries a HIP event pair that receives "
                                   "the kernel's own start/stop timestamps (hipExtLaunchKernel; the duration rocprofv3 reports) -- "
                                   "not every step, because a launch with events costs the step 8 us of device time "
                                   "(round 3); achieved = algorithmic bytes of the timed launches / the sum of "
                                   "their durations; `roofline_unoverlapped` times EVERY launch of a second loop outside `value`" % self.time_every + (
                                       " -- with %d chains per GPU the timed launches (first chain) run CONCURRENTLY with the other "
                                       "chains' kernels: contended rate; `roofline_unoverlapped` is the kernel alone in its pipeline" % K
                                       if K > 1 else "")},
        }
        if self.exchange is not None:
            self._rccl_fields(line)
        if self.ess is not None:
            line["ess"] = {"kept_per_chain": self.kept, "cost": self.ess[0], "theta_coords": self.ess[1:]}
        else:
            # pymc3's effective_n formula is unguarded (diagnostics/sampler_diagnostics.py): on a handful of samples it returns
            # anything, negative numbers included -- not reported below ESS_MIN_KEPT kept samples per chain
            line["ess"] = {"kept_per_chain": self.kept, "cost": None, "theta_coords": None,
                           "note": "fewer than %d kept samples per chain: not estimated" % self.ESS_MIN_KEPT}
        return line

    def _product_arithmetic(self):
        """What the matrix products of the step are computed in, from the cost plan in use."""
        cost = self.sampler.cost_fun
        plans = list(getattr(cost, "_plans", {}).values())
        if not plans:
            return None
        d = plans[-1].as_dict()
        fused = sum(1 for op in d["forward"] if op.startswith("dense_tanh"))
        return {"forward / backward layers": "%d of %d forward layers as fp32-MFMA launches with their activation, the rest library %s products"
                                             % (fused, len(d["forward"]) - 1, "f64" if self.f64 else "fp32"),
                "batched weight gradients": d.get("batched_weight_gradient_arithmetic", "library product per layer")}

    def _rccl_fields(self, line):
        """N > 1: what the one exchange of the path cost, and `value` with that cost taken out. With the driver's 20-step
        command ONE exchange of 3P floats sits inside a ~4 ms timed region (cadence 13 steps instead of configs[3]'s 100), so
        whatever part of it is exposed lands fully in `value`; `value_ex_exchange` is the same job at a cadence where the
        exchange amortises to nothing, and is by construction >= `value`."""
        args, dist, ex = self.args, self.dist, self.exchange
        done = [ev for ev in self.ex_events[self.prime_rhat_events:] if len(ev) == 5]
        sampling_s = max(self.elapsed - self.exposed_ms_max * 1e-3, 1e-9)
        line["value_ex_exchange"] = round(self.world * self.K * args.steps / sampling_s, 2)
        line["rccl"] = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "mode": ex.mode,
                        "exchanges_timed": len(done),
                        "payload_bytes": int(ex.pack.numel() * ex.pack.element_size()),
                        "cadence_steps_used": self.rhat_every, "cadence_steps_config3": RHAT_EVERY_CONFIG3,
                        # compute-stream time the exchanges of the timed region cost (pack + collect), max over ranks;
                        # the collective itself runs on RCCL's stream under the sampling steps
                        "exposed_ms": round(self.exposed_ms_max, 4),
                        "exposed_ms_rank0": round(self.exposed_ms, 4),
                        "exposed_frac_of_timed_region": round(self.exposed_ms_max * 1e-3 / self.elapsed, 4)}
        if done:
            # start -> finish wall on the compute stream (includes the steps sampled in between), the pack launch,
            # and the tail the compute stream actually spends on the exchange when it collects it
            line["rccl"]["rhat_exchange_ms"] = {
                "start_to_finish": round(float(np.mean([ev[0].elapsed_time(ev[3]) for ev in done])), 3),
                "pack_and_issue": round(float(np.mean([ev[0].elapsed_time(ev[1]) for ev in done])), 3),
                "wait_finish_summary": round(float(np.mean([ev[2].elapsed_time(ev[3]) for ev in done])), 3)}
        line["rccl"]["collective_alone_ms"] = None
        line["rhat"] = {k: round(v, 4) for k, v in ex.summary.as_dict().items()} if ex.exchanges else None

    # ------------------------------------------------------------------ after the timed region
    def leave_group(self, line):
        """All ranks: the collective alone (blocking, same payload), then the job ends for every rank -- rank 0's extra legs
        run with no process group alive, so no rank sits in a collective (or its watchdog) while they take their seconds."""
        dist, ex = self.dist, self.exchange
        if ex is not None:
            torch.cuda.synchronize()
            dist.barrier()
            ts = []
            native_rs = ex.mode == "reduce_scatter" and ex._native_rs
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if native_rs:
                    dist.reduce_scatter_tensor(ex.shard_sum, ex.pack)
                else:
                    dist.all_reduce(ex.pack)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            if line is not None:
                line["rccl"]["collective_alone_ms"] = round(float(np.median(ts[1:])), 3)
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            dist.destroy_process_group()

    def post_run_legs(self, line):
        """Rank 0, after the job: the update kernel alone in the pipeline, where the step time goes, two chains per GPU, the
        HBM-resident sizes, the CPU baseline (N = 1). Identical code path at every N: SCALE N = 1 equals BENCH."""
        from pysgmcmc_amd.profiling import UpdateKernelTimer
        args, sampler, n, mode = self.args, self.sampler, self.n, self.mode
        if self.group is not None:                             # the legs below step the GPU's first chain alone
            self.group.join()
            torch.cuda.synchronize()
            del self.chains[1:], self.all_moments[1:]
            self.group = None
            torch.cuda.empty_cache()
        if self.kind == "sghmc" and sampler.use_hip_graph:
            n_legs = max(min(args.steps, 60), 20)
            sampler.attach_moments(None)
            t2 = UpdateKernelTimer(reserve=n_legs, device=self.dev)
            sampler.kernel_timer = t2
            t2.enabled = True
            for _ in range(n_legs):
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
                "note": "the same chain, %d steps after the timed region with the timer on EVERY update launch and no moments "
                        "steps: the kernel alone in the pipeline, as rounds 1-2 reported `roofline`" % n_legs}
            # the contract's `roofline` comes from the timed region, where only every time_every-th launch can carry events (5 of the
            # driver's 20 steps): the every-launch figure of this loop stands beside it in the same object
            line["roofline"]["every_launch_loop"] = {k: line["roofline_unoverlapped"][k] for k in (
                "launches_timed", "us_per_launch_mean", "us_per_launch_median", "achieved", "frac")}
            g_us, g_flops, n_fused, n_fused_back = legs.gemm_only_us(sampler)
            c_us = legs.cost_pipeline_us(sampler)
            meas_us = float(np.median(self.step_ms)) * 1e3 if self.step_ms is not None else None
            line["step_breakdown_us"] = {
                "gemm": round(g_us, 1), "cost_pipeline": round(c_us, 1), "update": round(float(u_us.mean()), 1),
                "serial_sum": round(c_us + float(u_us.mean()), 1), "serial_step_measured": round(serial_step_us, 1),
                "timed_region_step_median": None if meas_us is None else round(meas_us, 1),
                "gemm_tflops": round(g_flops / (g_us * 1e-6) / 1e12, 1),
                "gemm_frac_of_fp32_mfma_peak": round(g_flops / (g_us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 3),
                "gemm_flop_per_step": int(g_flops),
                # the step against its own two rooflines: the products on the fp32 matrix pipe + the update on HBM, nothing else
                "ideal_us": round(g_flops / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e6 + BYTES_PER_PARAM[mode] * n / (HBM_PEAK_GBS * 1e9) * 1e6, 1),
                "frac_of_ideal": None if meas_us is None else round(
                    (g_flops / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e6 + BYTES_PER_PARAM[mode] * n / (HBM_PEAK_GBS * 1e9) * 1e6) / meas_us, 3),
                "forward_layers_on_the_fused_launch": n_fused,
                "backward_products_on_the_fused_launch": n_fused_back,
                "note": "gemm = the step's eight fp32 products replayed alone from a hipGraph, each as the pipeline runs it: "
                        "the three forward layers (%d of them as ONE launch each with bias + tanh as the product's epilogue, "
                        "sgmcmc_bnn_dense_tanh_f32), the two delta W^T products (%d of them with tanh' of the layer below as the "
                        "epilogue, sgmcmc_bnn_dense_tanh_backward_f32) and the library weight-gradient products (the two 2048 x 2048 ones as ONE "
                        "strided batched product, the first layer's with its bias gradient as a 785th row) -- activations and tanh' are then inside `gemm`; cost_pipeline = the "
                        "captured cost pipeline alone (gemm + the loss head + the launches of layers on library products; two graph "
                        "replays differ by less than their noise, so their difference is not reported: the per-dispatch durations "
                        "of the step are in profiles/r05_step_timeline.txt); "
                        % (n_fused, n_fused_back) +
                        "update = the fused update launched once after the backward pass. Differences of graph replays: rocprofv3's "
                        "per-kernel durations of the same step (profiles/r04_bench10m_kernel_stats.csv) carry ~1.5 us of profiler "
                        "overhead per kernel. The fp32 MFMA peak is quoted at "
                        "2.4 GHz; under this load the chip sustains ~2.0 GHz (profiles/r04_fwd_epilogue_probe.txt)"}
        if not args.no_update_only and self.kind == "sghmc":
            if sampler.use_hip_graph and self.K == 1:
                line["chains_per_gpu"] = legs.chains_per_gpu_leg(self.dev, sampler, args.workload)
            line["update_only"] = legs.update_only(sampler)
            self.moments = self.trace = None
            line["roofline_hbm_resident"] = legs.hbm_resident_roofline(self.dev)
        if self.world == 1 and not args.no_cpu_baseline and self.kind == "sghmc":
            line["cpu_baseline"] = cpu_baseline(n, args.cpu_seconds)
        if self.world == 1 and self.K == 1 and sampler.use_hip_graph and not getattr(args, "no_product_defaults", False):
            line["value_product_defaults"] = legs.product_defaults_leg(args)
        return line

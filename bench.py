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

Other workloads (``--workload``): ``bnn50m-sgld`` / ``bnn50m-rsghmc`` (configs[4]'s samplers, HBM-resident), ``sinc-bnn``
(configs[1], the reference's own 3 x 50 BNN test case: fused whole-step kernel vs hipGraph vs eager), ``svgd16-10m``.
The pieces live in ``benchlib/`` (launcher, workloads, the timed chain run, post-run legs, CPU baselines).

Output: ONE JSON line on rank 0. ``value`` = whole-job samples/s over the whole timed region
(``step_ms_median`` / ``step_ms_max`` from per-step HIP events expose one-offs). ``roofline`` = the
fused update kernel measured live with HIP events inside the timed region (10 M parameters: 240 MB
per launch, Infinity-Cache-assisted); ``roofline_hbm_resident`` = the same kernels on 49 826 818
parameters (1.2 GB per launch, cannot live in the 256 MiB Infinity Cache), measured after the
timed region. ``cpu_baseline`` (N = 1 only) = the same full step on the host cores (numpy/BLAS
BNN gradient + the fused C oracle update), unit samples/s like ``value``. N > 1 lines also carry ``value_ex_exchange``
(samples/s with the exposed time of the R-hat exchange taken out of the timed region) and ``rccl.exposed_ms``.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

import pysgmcmc_amd  # noqa: E402

from benchlib.common import kernel_source_hash, pmc_traffic, usable_cores  # noqa: E402,F401  (tools/ import these from here)
from benchlib.launcher import self_launch  # noqa: E402,F401
from benchlib.workloads import LAYERS, WORKLOADS, build_chain  # noqa: E402,F401


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
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="N > 1 started without a launcher: wall-clock limit of the ranks this process spawns (s)")
    ap.add_argument("--dtype", default="f32", choices=("f32", "f64"),
                    help="arithmetic type of the chain: f64 is the reference's default dtype (pysgmcmc/samplers/base_classes.py:25); the f64 "
                         "line carries `value`, the step's device time and the update kernel's roofline (16 B per f64 element touch) -- "
                         "no legs after the timed region")
    ap.add_argument("--product-defaults", action="store_true",
                    help="step the same chain with NOTHING set (no pysgmcmc_amd.configure_for_device_bound_chains()): the rate a "
                         "user of the public API gets by default; the main line runs this as a child process (value_product_defaults)")
    ap.add_argument("--no-product-defaults", action="store_true", help="skip the value_product_defaults child run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-update-only", action="store_true", help="skip the kernel-only loops (for rocprof runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU baseline leg")
    return ap.parse_args()


def main():
    args = parse()
    if args.dtype != "f32" and WORKLOADS[args.workload]["sampler"] not in ("sghmc", "sgld", "rsghmc"):
        raise SystemExit("--dtype f64 is implemented for the BNN chain workloads (bnn10m-sghmc, bnn50m-sgld, bnn50m-rsghmc)")
    args.device_bound_switch = None
    if WORKLOADS[args.workload]["sampler"] in ("sghmc", "sgld", "rsghmc") and not args.eager and not args.product_defaults:
        # device-bound hipGraph steps (10 M / 49.8 M parameters): the package's one documented switch -- TunableOp GEMM selection in
        # the warm-up step + the runtime's plain graph launch path -- before any HIP call here or in the ranks spawned below (they
        # inherit the environment). What took effect goes into the line (config.device_bound_switch): under rocprofv3 the runtime
        # is initialised before Python runs and the second half is NOT in effect whatever the environment says.
        args.device_bound_switch = pysgmcmc_amd.configure_for_device_bound_chains(
            gemm_tuning=not args.no_gemm_tuning, tuning_ms=int(os.environ.get("BENCH_TUNE_MS", "30")),
            tuning_iters=int(os.environ.get("BENCH_TUNE_ITERS", "20")))
        args.no_gemm_tuning = not args.device_bound_switch["gemm_tuning"]
        if not args.device_bound_switch["plain_graph_launch"]:
            print("bench: the HIP runtime was initialised before bench.py ran: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is NOT in effect", file=sys.stderr)
    # (--product-defaults: NOTHING is set here -- BNNCost picks the GEMM solutions of a device-bound plan in its first evaluation by
    # itself, models/bayesian_neural_network.py `auto_gemm_tuning`; the runtime's default graph launch stays)
    if args.no_gemm_tuning:
        os.environ["PYSGMCMC_AMD_AUTO_GEMM_TUNING"] = "0"      # --no-gemm-tuning means the library's heuristics, in every rank
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
    kind = WORKLOADS[args.workload]["sampler"]
    if kind == "svgd":
        from benchlib.svgd import run_svgd
        return run_svgd(args, dev, rank, world, dist)
    if kind == "sinc":
        from benchlib.sinc import run_sinc
        return run_sinc(args, dev, rank, world, dist)
    import json
    from benchlib.chain_run import ChainBench
    run = ChainBench(args, dev, rank, world, dist)
    run.prime()                                                # phase 1: burn-in + every code path once (untimed)
    run.warmup()                                               # phase 2: --warmup untimed steps
    run.timed()                                                # phase 3: exactly --steps steps between fences
    line = run.headline()                                      # rank 0: the JSON line from the timed region
    if args.product_defaults:                                  # the child leg of value_product_defaults: the rate only
        print(json.dumps({"value": line["value"], "ms_per_step": line["ms_per_step"], "config": line["config"]}))
        return
    run.leave_group(line)                                      # N > 1: the collective alone, then the job ends for every rank
    if rank != 0:
        return
    if args.dtype == "f32":
        run.post_run_legs(line)                                # kernel alone, step breakdown, HBM-resident sizes, CPU baseline
    print(json.dumps(line))
    sys.stdout.flush()


if __name__ == "__main__":
    main()

"""SURVEY 8(f) row 3 and 8(e) on the device.

(f3) ``effective_sample_sizes`` / ``gelman_rubin`` / ``PYSGMCMCTrace.from_sampler`` / ``pymc3_multitrace``
(``pysgmcmc/diagnostics/sample_chains.py:14-384``, ``sampler_diagnostics.py:47-194``) driven by HIP samplers on the
reference's toy targets; results must equal the oracle's formulas evaluated on the very samples the chains produced.

(e) the N > 1 path with the REAL kernels: 2 ranks on ``cuda:0`` over gloo (RCCL refuses two ranks on one GPU; the
transport is the only stand-in): K1 chains, K4 Welford moments, ``rhat_pack`` -> all-reduce -> ``rhat_finish``
(+ device-side summary), the non-blocking ``RhatExchange``, the ESS all-gather -- checked against
``oracle.gelman_rubin`` / ``oracle.effective_n`` on the chains' samples, in f32 and f64.
"""
import json
import os
import socket
import subprocess
import time
import sys
from itertools import islice

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _targets():
    from pysgmcmc_amd.diagnostics.objective_functions import (
        banana_log_likelihood, gmm1_log_likelihood, to_negative_log_likelihood)
    return {
        "banana": (to_negative_log_likelihood(banana_log_likelihood),
                   lambda dev: [torch.tensor(0.0, device=dev), torch.tensor(6.0, device=dev)]),
        "gmm1": (to_negative_log_likelihood(gmm1_log_likelihood), lambda dev: [torch.tensor(0.0, device=dev)]),
    }


def _factory(kind, target, gpu, counter):
    """``get_sampler(session=...)`` of the reference's diagnostics entry points: a fresh chain per call, another
    seed per chain (``sampler_diagnostics.py:47-60``)."""
    from pysgmcmc_amd.samplers import RelativisticSGHMCSampler, SGHMCSampler, SGLDSampler
    from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
    cost, make = _targets()[target]
    ctor = {"sghmc": SGHMCSampler, "sgld": SGLDSampler, "rsghmc": RelativisticSGHMCSampler}[kind]

    def get_sampler(session=None):
        counter.append(1)
        kw = {} if kind == "rsghmc" else {"burn_in_steps": 20}
        eps = 0.01 if kind == "rsghmc" else 0.05
        return ctor(params=make(gpu), cost_fun=cost, stepsize_schedule=ConstantStepsizeSchedule(eps),
                    session=gpu, dtype=torch.float32, seed=100 + len(counter), **kw)
    return get_sampler


@pytest.mark.parametrize("kind", ["sghmc", "sgld", "rsghmc"])
@pytest.mark.parametrize("target", ["banana", "gmm1"])
def test_reference_diagnostics_entry_points_with_hip_samplers(gpu, oracle, kind, target):
    from pysgmcmc_amd.diagnostics.sample_chains import MultiTrace, PYSGMCMCTrace, pymc3_multitrace
    from pysgmcmc_amd.diagnostics.sampler_diagnostics import effective_sample_sizes, gelman_rubin
    n_vars = 2 if target == "banana" else 1
    # --- PYSGMCMCTrace.from_sampler on the device chain (sample_chains.py:127-180)
    calls = []
    s = _factory(kind, target, gpu, calls)()
    trace = PYSGMCMCTrace.from_sampler(chain_id=3, sampler=s, n_samples=40)
    assert len(trace) == 40 and trace.chain == 3 and trace.n_vars == n_vars
    assert trace.varnames == [str(i) for i in range(n_vars)]                       # enumerated names, :80-90
    assert s.n_iterations == 40
    last = [float(v) for v in s.arena.row("theta").cpu()]
    assert [float(v) for v in trace.samples[-1]] == last                           # the trace holds the chain's own samples
    assert np.array_equal(trace.get_values("0", burn=10, thin=3),
                          np.asarray([smp[0] for smp in trace.samples[10::3]]))
    assert set(trace.point(5)) == set(trace.varnames)
    # --- pymc3_multitrace: fresh chains one after another (sample_chains.py:338-384)
    calls = []
    mt = pymc3_multitrace(_factory(kind, target, gpu, calls), n_chains=3, samples_per_chain=120)
    assert isinstance(mt, MultiTrace) and mt.nchains == 3 and len(calls) == 3 and len(mt) == 120
    chains = np.stack([np.stack([np.asarray(v, np.float64) for v in mt.get_values(name, combine=False)])
                       for name in mt.varnames], axis=-1)                          # (m, n, vars)
    assert chains.shape == (3, 120, n_vars) and np.isfinite(chains).all()
    assert not np.array_equal(chains[0], chains[1])                                # another seed per chain
    # --- gelman_rubin / effective_sample_sizes (sampler_diagnostics.py:47-194): the factories are deterministic
    # (seed = 100 + call index), so the entry points see exactly the chains collected above
    calls = []
    rhat = gelman_rubin(_factory(kind, target, gpu, calls), n_chains=3, samples_per_chain=120)
    calls = []
    ess = effective_sample_sizes(_factory(kind, target, gpu, calls), n_chains=3, samples_per_chain=120)
    assert sorted(rhat) == sorted(ess) == sorted(mt.varnames)
    want_rhat = oracle.gelman_rubin(chains)
    for k, name in enumerate(mt.varnames):
        assert np.isclose(float(rhat[name]), want_rhat[k], rtol=1e-10), (name, rhat[name], want_rhat[k])
        assert int(ess[name]) == oracle.effective_n(chains[:, :, k]), (name, ess[name])
        assert float(rhat[name]) > 0.9 and 1 <= int(ess[name]) <= 3 * 120 * 3


# --------------------------------------------------------------------------------------------------------------
# 2 ranks on cuda:0 with the real kernels
# --------------------------------------------------------------------------------------------------------------

def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


WORKER = r"""
import os, sys, json
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
rank, world, dtname, out_dir = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1], sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from itertools import islice
from pysgmcmc_amd import _lib, kernels
from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments, RhatExchange, cross_chain_rhat, ess_across_ranks
from pysgmcmc_amd.samplers import SGHMCSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
assert kernels.sghmc_step.__module__ == "pysgmcmc_amd.kernels"          # the real ctypes front end, no shim
dt = torch.float32 if dtname == "float32" else torch.float64
n = 4099                                                                  # ragged: not a multiple of 4
x = torch.full((n,), 3.0 * (rank - 0.5 * (world - 1)) / max(world - 1, 1), dtype=dt, device=dev)   # over-dispersed starts
s = SGHMCSampler(params=[x], cost_fun=lambda p: 0.5 * (p[0] ** 2).sum(), burn_in_steps=50, session=dev,
                 stepsize_schedule=ConstantStepsizeSchedule(0.1), dtype=dt, seed=100 + rank)
s.sample_format = "view"
mom = ChainMoments(n, dev, dtype=dt)
kept, trace = [], []
for t, (sample, cost) in enumerate(islice(s, 650)):
    if t >= 50 and t % 3 == 0:
        mom.update(s.arena.row("theta"))                                  # K4 on the device
        kept.append(sample.detach().cpu().numpy().copy())
        trace.append([float(cost), float(sample[0]), float(sample[1])])
rhat, summ = cross_chain_rhat(mom)                                        # pack -> all-reduce -> finish (+ K6 summary)
assert rhat.is_cuda and rhat.dtype == dt
ex = RhatExchange(n, dev, dtype=dt)
ex.start(mom)
next(s)                                                                   # the chain moves on while the collective runs
rhat2, none = ex.finish()
assert none is None and torch.equal(rhat, rhat2) and ex.summary.as_dict() == summ and not ex.pending
# parameter-sharded exchange (reduce-scatter layout; over gloo the class all-reduces the sharded pack)
rs = RhatExchange(n, dev, dtype=dt, mode="reduce_scatter")
L = ((n + world - 1) // world + 3) // 4 * 4
assert rs.n_shards == world and rs.shard_len == L and rs.n_valid == min(L, n - rank * L)
rs.start(mom)
next(s)
shard, _ = rs.finish()
lo = rank * rs.shard_len
# (two ranks: a + b is order-independent, so the layouts agree bit for bit; with more ranks the transport adds the
# chains in a chunk-dependent order and the last bits may differ)
same = torch.equal if world == 2 else (lambda a, b: torch.allclose(a, b, rtol=1e-5, atol=0))
assert same(shard[:rs.n_valid], rhat[lo:lo + rs.n_valid]) and same(rs.gather(), rhat)
ssum = rs.summary.as_dict()
assert abs(ssum["mean"] - summ["mean"]) <= (1e-12 if world == 2 else 1e-6) * abs(summ["mean"])
assert ssum["max"] == summ["max"] if world == 2 else abs(ssum["max"] - summ["max"]) <= 1e-5 * summ["max"]
ess = ess_across_ranks(torch.tensor(trace, dtype=torch.float32, device=dev))
np.savez(os.path.join(out_dir, "rank%d.npz" % rank), kept=np.array(kept), rhat=rhat.cpu().numpy(),
         rhat_mean=summ["mean"], rhat_max=summ["max"], ess=np.array(ess), trace=np.array(trace),
         mean=mom.mean.cpu().numpy(), m2=mom.m2.cpu().numpy(), count=mom.count,
         lib=_lib.lib_path())
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dtname,world", [("float32", 2), ("float64", 2), ("float32", 8)])
def test_two_ranks_on_one_gpu_real_kernels_rhat_and_ess(gpu, oracle, tmp_path, dtname, world):
    """2 ranks (f32, f64) and the configs[3] rank count, 8 (f32), all on cuda:0."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), dtname, str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=500)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out[-3000:]
    r = [np.load(str(tmp_path / ("rank%d.npz" % k))) for k in range(world)]
    npdt = np.float32 if dtname == "float32" else np.float64
    chains = np.stack([r[k]["kept"] for k in range(world)])               # (m, 200, n)
    assert chains.shape == (world, 200, 4099) and not np.array_equal(chains[0], chains[1])
    # every rank holds the same R-hat; it equals the oracle's Gelman-Rubin on the very samples
    assert all(np.array_equal(r[0]["rhat"], r[k]["rhat"]) for k in range(1, world))
    want = oracle.gelman_rubin(chains)
    tol = 2e-3 if npdt == np.float32 else 1e-9                           # f32: Welford + sum-form B in single precision
    assert np.allclose(r[0]["rhat"], want, rtol=tol), np.abs(r[0]["rhat"] / want - 1).max()
    assert np.isclose(float(r[0]["rhat_max"]), float(r[0]["rhat"].max()), rtol=1e-6)
    assert np.isclose(float(r[0]["rhat_mean"]), float(r[0]["rhat"].astype(np.float64).mean()), rtol=1e-6)
    # bit-exact leg: K4 moments == the C oracle's Welford on the same samples, pack/finish == the C oracle's
    for k in range(world):
        mean, m2 = np.zeros(4099, npdt), np.zeros(4099, npdt)
        for c, smp in enumerate(r[k]["kept"]):
            oracle.c_moments_update(np.ascontiguousarray(smp.astype(npdt)), mean, m2, c + 1)
        assert np.array_equal(mean, r[k]["mean"]) and np.array_equal(m2, r[k]["m2"])
    if world == 2:      # (the sum of more than two packs depends on the transport's reduction order: f32 compared above)
        total = sum(oracle.c_rhat_pack(r[k]["mean"], r[k]["m2"], int(r[k]["count"])) for k in range(2))
        assert np.array_equal(oracle.c_rhat_finish(total.astype(npdt), 2, int(r[0]["count"])), r[0]["rhat"])
    # the sharded layout: chunk s of the pack = rows of parameter shard s; finishing the chunks gives the same R-hat
    L = ((4099 + world - 1) // world + 3) // 4 * 4
    tot = sum(oracle.c_rhat_pack(r[k]["mean"], r[k]["m2"], int(r[k]["count"]), world, L) for k in range(world)).astype(npdt)
    parts = [oracle.c_rhat_finish(np.ascontiguousarray(tot[s * 3 * L:(s + 1) * 3 * L]), world, int(r[0]["count"]),
                                  n=min(L, 4099 - s * L), ld=L) for s in range(world)]
    if world == 2:
        assert np.array_equal(np.concatenate(parts), r[0]["rhat"])
    else:
        assert np.allclose(np.concatenate(parts), r[0]["rhat"], rtol=1e-5)
    # ESS of cost and two coordinates from the all-gathered thinned traces
    traces = np.stack([r[k]["trace"] for k in range(world)])              # (m, 200, 3)
    assert all(np.array_equal(r[0]["ess"], r[k]["ess"]) for k in range(1, world))
    for k in range(3):
        assert int(r[0]["ess"][k]) == oracle.effective_n(traces[:, :, k].astype(np.float32).astype(np.float64))
    assert str(r[0]["lib"]).endswith("libsgmcmc_hip.so")


def _bench_n2(extra):
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra + [
           "--backend", "gloo", "--all-ranks-on-gpu0", "--no-update-only"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_PRIME_STEADY="60")    # (several ranks share one GPU here: a short prime phase)
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                              # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_n2_path_on_one_gpu(gpu):
    """``bench.py --gpus 2`` as the driver launches it (torch.distributed.run, one rank per process), with both ranks on
    cuda:0 over gloo: the line carries the rank count, the exchange timings and an R-hat summary, and N = 2 runs the
    same step code as N = 1."""
    d = _bench_n2(["--steps", "24", "--warmup", "4", "--rhat-every", "8", "--moments-every", "2", "--time-every", "1"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["unit"] == "samples/s" and d["steps"] == 24
    assert d["value"] > 0 and np.isclose(d["value"], 2 * 24 / (d["ms_per_step"] * 24 / 1e3), rtol=1e-3)
    rc = d["rccl"]
    assert rc["ranks"] == 2 and rc["exchanges_timed"] == 3 and 0 <= rc["payload_bytes"] - 3 * 4 * d["config"]["params"] < 1024
    assert rc["rhat_exchange_ms"]["start_to_finish"] > 0 and rc["collective_alone_ms"] > 0
    assert rc["mode"] == "reduce_scatter"                               # the default: parameter-sharded exchange
    assert d["rhat"]["max"] >= d["rhat"]["mean"] > 0
    assert d["roofline"]["launches_timed"] == 24 and 0 < d["roofline"]["frac"] < 1.2
    # every 2nd step's update launch also carries the Welford moments (K4 fused): those launches are timed apart
    assert d["roofline"]["launches_in_the_rate"] == 12 and d["roofline"]["with_fused_moments"]["launches"] == 12
    assert "cpu_baseline" not in d


@pytest.mark.timeout(900)
def test_bench_with_two_chains_per_gpu(gpu):
    """``--chains-per-gpu 2``: every rank steps two independent chains concurrently (own stream + hipGraph each), `value`
    counts all of them, and the one R-hat exchange covers ranks x 2 chains (local packs added before the collective)."""
    d = _bench_n2(["--steps", "20", "--warmup", "5", "--chains-per-gpu", "2"])
    assert d["n_gpus"] == 2 and d["config"]["chains"] == 4 and d["config"]["chains_per_gpu"] == 2
    assert np.isclose(d["value"], 4 * 20 / (d["ms_per_step"] * 20 / 1e3), rtol=1e-3)
    assert d["rccl"]["exchanges_timed"] == 1 and d["rhat"] is not None and d["rhat"]["max"] >= d["rhat"]["mean"] > 0
    assert d["roofline"]["launches_timed"] == 5 and "2 chain(s) per GPU" in d["config"]["workload"]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--chains-per-gpu", "2",
                          "--no-update-only", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    one = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert one["n_gpus"] == 1 and one["config"]["chains"] == 2 and one["value"] > 0 and "rccl" not in one


@pytest.mark.timeout(900)
def test_bench_n2_driver_arguments_contain_one_exchange(gpu):
    """With the driver's arguments (20 steps, 5 warm-up, default cadences) the timed region is shorter than the R-hat
    period of configs[3]; exactly ONE exchange is then placed inside it (started after 2/3 of the steps, collected
    before the end), so the collective's cost is in ``value`` at every N > 1."""
    d = _bench_n2(["--steps", "20", "--warmup", "5"])
    assert d["steps"] == 20 and d["warmup"] == 5 and d["config"]["rhat_every"] == 14 and d["config"]["moments_every"] == 10
    assert d["rccl"]["exchanges_timed"] == 1 and d["rccl"]["rhat_exchange_ms"]["start_to_finish"] > 0
    assert d["rccl"]["cadence_steps_used"] == 14 and d["rccl"]["cadence_steps_config3"] == 100
    assert d["value_ex_exchange"] >= d["value"] and d["rccl"]["exposed_ms"] > 0
    # the update launches of every 4th step carry timestamp events (20 // 5; a timed launch costs the step ~8 us)
    assert d["rhat"] is not None and d["config"]["time_every"] == 4 and d["roofline"]["launches_timed"] == 5


def _bench_self_launched(n, extra, timeout=800):
    """``python3 bench.py --gpus N ...`` with NO launcher, as the driver starts the 1-GPU bench: the process spawns its
    own ranks."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + extra + [
        "--backend", "gloo", "--all-ranks-on-gpu0", "--no-update-only"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["BENCH_PRIME_STEADY"] = "60"                             # (several ranks share one GPU here: a short prime phase)
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    return res, [l for l in res.stdout.splitlines() if l.startswith("{")]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n", [2, 8])
def test_bench_starts_its_own_ranks_without_a_launcher(gpu, n):
    """``python3 bench.py --gpus N`` without torch.distributed.run must not exit non-zero -- the parent
    (which never touches the GPU) spawns N fresh ranks, rank 0 prints the ONE JSON line, exit code 0. All ranks on
    cuda:0 over gloo here (one GPU on the box); 8 ranks = configs[3]'s shape."""
    res, lines = _bench_self_launched(n, ["--steps", "6", "--warmup", "2"])
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["rccl"]["ranks"] == n and d["rccl"]["exchanges_timed"] == 1 and d["value"] > 0
    assert d["config"]["chains"] == n and d["rhat"] is not None
    assert "roofline_unoverlapped" in d and "step_breakdown_us" in d      # rank 0's extra legs ran after the group was torn down
    # the exchange's exposed time and `value` with it taken out ride next to `value`
    rc = d["rccl"]
    assert rc["exposed_ms"] >= rc["exposed_ms_rank0"] > 0 and 0 < rc["exposed_frac_of_timed_region"] < 1
    assert d["value_ex_exchange"] >= d["value"] > 0
    assert np.isclose(d["value_ex_exchange"], n * 6 / (d["ms_per_step"] * 6e-3 - rc["exposed_ms"] * 1e-3), rtol=2e-2)
    assert rc["cadence_steps_used"] == d["config"]["rhat_every"] == 4 and rc["cadence_steps_config3"] == 100


@pytest.mark.timeout(600)
def test_self_launched_bench_propagates_a_failing_rank_and_stops_the_others(gpu):
    """A rank that dies takes the job down with its exit code instead of leaving the other ranks in a collective: the
    workload name is valid for the parent's argument parser but rank 1 is told to fail (BENCH_TEST_FAIL_RANK)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--backend", "gloo",
           "--all-ranks-on-gpu0", "--launch-timeout", "300"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_TEST_FAIL_RANK="1")
    t0 = time.time()
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=500)
    assert res.returncode == 7, (res.returncode, res.stderr[-2000:])
    assert "rank 1 exited with code 7" in res.stderr and not [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 400


RCCL_WORKER = r"""
import os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # "nccl" IS RCCL on ROCm
from pysgmcmc_amd import kernels
from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments
from pysgmcmc_amd.data_batches import Placeholder, generate_batches
from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
from pysgmcmc_amd.samplers import SGHMCSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
g = torch.Generator(device=dev).manual_seed(0)
X, y = torch.randn(4096, 64, device=dev, generator=g), torch.randn(4096, device=dev, generator=g)

def chain(graph):
    xp, yp = Placeholder(dtype=torch.float32, device=dev), Placeholder(dtype=torch.float32, device=dev)
    params = init_mlp_params(64, hidden=(512, 512), seed=3, dtype=torch.float32, device=dev)
    s = SGHMCSampler(params=params, cost_fun=BNNCost(xp, yp, batch_size=128, n_examples=4096),
                     batch_generator=generate_batches(X, y, xp, yp, batch_size=128, seed=1), burn_in_steps=4,
                     stepsize_schedule=ConstantStepsizeSchedule(0.01), scale_grad=4096.0, session=dev,
                     dtype=torch.float32, seed=9)
    s.sample_format = "view"
    s.use_hip_graph = graph
    return s

# reference run without any collective, eager
ref = chain(False)
for _ in range(40):
    next(ref)
want = ref.arena.row("theta").clone()
# the same chain, cost pipeline captured into a hipGraph WHILE the RCCL communicator (and its watchdog thread) is alive,
# with an asynchronous 3P-float all-reduce in flight during the steps
s = chain(True)
n = s.arena.n
mom = ChainMoments(n, dev)
pack = torch.empty(3 * n, device=dev)
works = []
for t in range(40):
    next(s)
    mom.update(s.arena.row("theta"))
    if t in (9, 25):
        kernels.rhat_pack(mom.mean, mom.m2, mom.count, pack)
        snap = pack.clone()
        works.append((dist.all_reduce(pack, async_op=True), snap))           # RCCL stream, overlaps the next steps
    if t in (14, 30):
        w, snap = works.pop()
        w.wait()                                                              # stream-level dependency, no host sync
        assert torch.equal(pack, snap)                                        # SUM over one rank = identity
        # finish on a 2-chain total built from this chain twice: R-hat = sqrt((n-1)/n) exactly where W > 0
        total = pack * 2
        rhat = torch.empty(n, device=dev)
        out4, ws = torch.zeros(4, dtype=torch.float64, device=dev), kernels.summary_workspace(dev)
        kernels.rhat_finish(total, n, 2, mom.count, rhat, out4, ws)
        c = float(mom.count)
        assert torch.allclose(rhat, torch.full_like(rhat, ((c - 1) / c) ** 0.5), rtol=1e-3)
        # RCCL reduce-scatter / all-gather entry points on the same buffers (one rank: identity)
        out = torch.empty_like(pack)
        dist.reduce_scatter_tensor(out, pack)
        assert torch.equal(out, pack)
        full = torch.empty_like(rhat)
        dist.all_gather_into_tensor(full, rhat)
        assert torch.equal(full, rhat)
assert not works
assert torch.equal(s.arena.row("theta"), want), "graph-stepped chain next to RCCL traffic differs from the eager chain"
dist.barrier()
dist.destroy_process_group()
print("rccl-ok", torch.cuda.get_device_name(0))
"""


@pytest.mark.timeout(600)
def test_rccl_communicator_next_to_hipgraph_stepping(gpu, tmp_path):
    """One rank, backend "nccl" (= RCCL): the communicator and its watchdog thread are alive while the sampler captures
    and replays its hipGraph, and asynchronous all-reduces of the 3P-float R-hat payload are in flight during the steps
    (the configs[3] pattern). The chain must equal the eager chain bit for bit and the exchange must return the payload."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0 and "rccl-ok" in res.stdout, (res.stdout[-2000:], res.stderr[-3000:])

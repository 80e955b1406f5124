"""``bench.py``'s non-default workloads in the contract's format (the default 10 M-parameter line is exercised by
``test_diagnostics_gpu.py`` at N = 2 / 8 and by the driver at N = 1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, timeout=600):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, capture_output=True, text=True,
                         timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_sinc_bnn_workload_is_configs_1(gpu):
    """BASELINE configs[1] (SGHMC on the reference's 3 x 50 sinc BNN, its own test case
    ``tests/bayesian_neural_network/test_train_predict.py:20-48``) as a bench line: the fused whole-step kernel's rate as
    `value`, the hipGraph / eager rates of the same chain next to it, the same step on a host core as `cpu_baseline`."""
    d = _bench(["--workload", "sinc-bnn", "--steps", "400", "--warmup", "50", "--cpu-seconds", "3"])
    assert d["unit"] == "samples/s" and d["n_gpus"] == 1 and d["steps"] == 400 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["config"]["params"] == 5252 and d["config"]["batch"] == 20 and "sinc" in d["config"]["workload"]
    m = d["modes_samples_per_s"]
    assert set(m) == {"fused_steps_100_per_launch", "fused_steps_1_per_launch", "full_graph", "hip_graph", "eager"}
    assert m["fused_steps_100_per_launch"] > m["hip_graph"] > m["eager"] > 100
    assert d["value"] > 5000 and abs(d["value"] - 1e3 / d["ms_per_step"]) <= 0.02 * d["value"]
    mc = d["many_chains_per_gpu"]["chains_256"]                  # K8 with one workgroup per CU: chains x steps per second
    assert mc["chains"] == 256 and mc["samples_per_s"] > 50 * d["value"] and mc["samples_per_s_device_time"] >= mc["samples_per_s"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["frac"] < 0.01 and "latency-bound, one workgroup" in r["note"] and r["launches_timed"] == 4
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "samples/s" and 10 < c["value"] < d["value"]


@pytest.mark.timeout(900)
def test_default_workload_line_at_n1(gpu):
    """The headline line (configs[2]) at N = 1 with the driver's arguments: contract fields, the roofline of the timed launches,
    and no N > 1 fields."""
    d = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-update-only", "--no-cpu-baseline"])
    assert d["metric"].startswith("MCMC samples/sec") and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["params"] == 10002434 and d["config"]["batch"] == 256 and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 0.01 * d["value"] and d["value"] > 1000
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["launches_timed"] == 5 and 0.3 < r["frac"] < 1.2
    assert r["algorithmic_bytes_per_launch"] == 24 * 10002434 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    assert "rccl" not in d and "value_ex_exchange" not in d
    # hygiene: the every-launch figure stands beside the 5-launch one; no ESS from a handful of kept samples
    assert r["every_launch_loop"]["launches_timed"] >= 20 and 0.3 < r["every_launch_loop"]["frac"] < 1.2
    assert d["ess"]["kept_per_chain"] < 50 and d["ess"]["cost"] is None and d["ess"]["theta_coords"] is None
    assert "step_breakdown_us" in d and d["step_breakdown_us"]["gemm"] > 0
    assert "small_launches" not in d["step_breakdown_us"] and d["step_breakdown_us"]["cost_pipeline"] >= d["step_breakdown_us"]["gemm"] * 0.9
    # the step against its own rooflines (products at the fp32 matrix peak + the update at the HBM peak): 92.4 + 30.0 us
    assert abs(d["step_breakdown_us"]["ideal_us"] - 122.4) < 0.2 and 0.4 < d["step_breakdown_us"]["frac_of_ideal"] < 1.0
    # `value` is measured with the package's ONE documented switch; the line says what took effect and carries the rate with nothing set
    sw = d["config"]["device_bound_switch"]
    assert sw == {"gemm_tuning": True, "plain_graph_launch": True} and d["config"]["hip_runtime_env_effective"] is True
    pd = d["value_product_defaults"]
    # nothing set in the child: BNNCost tunes the first evaluation of its device-bound plan by itself (round 6), the graph launch
    # path is the runtime's default: 0.97-1.0 of `value` on a quiet box; the two 20-step regions of this
    # command differ by up to 5 % run to run on their own, hence the wider band here
    assert pd["gemm_tuning"] == "auto" and pd["hip_runtime_env"] == {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": None}
    assert d["config"]["gemm_tuning"] == "caller"
    assert 0.88 * d["value"] < pd["value"] < 1.12 * d["value"], (pd["value"], d["value"])       # (r05: 0.75 at 49.8 M, 0.94 here)


@pytest.mark.timeout(900)
def test_public_api_chain_after_the_device_bound_switch_steps_at_the_bench_rate(gpu):
    """a chain built through the public API (samplers + BNNCost + generate_batches, as
    pysgmcmc/models/bayesian_neural_network.py:464-466,510-512 builds its sampler) after ONE call of
    ``pysgmcmc_amd.configure_for_device_bound_chains()`` steps at what bench.py reports as `value` to the process-to-process spread (both in fresh
    processes, 300 steps each, best of two)."""
    code = """
import time, numpy as np, torch, pysgmcmc_amd
took = pysgmcmc_amd.configure_for_device_bound_chains()
assert took == {"gemm_tuning": True, "plain_graph_launch": True}, took
from pysgmcmc_amd.data_batches import Placeholder, generate_batches
from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
from pysgmcmc_amd.samplers import SGHMCSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
X, y = rng.randn(100000, 784).astype(np.float32), rng.randn(100000).astype(np.float32)
xp, yp = Placeholder(dtype=torch.float32, device=dev), Placeholder(dtype=torch.float32, device=dev)
params = init_mlp_params(784, hidden=(2048, 2048, 2048), seed=0, dtype=torch.float32, device=dev)
s = SGHMCSampler(params=params, cost_fun=BNNCost(xp, yp, batch_size=256, n_examples=100000),
                 batch_generator=generate_batches(X, y, xp, yp, batch_size=256, seed=0), stepsize_schedule=ConstantStepsizeSchedule(1e-3),
                 burn_in_steps=8, mdecay=0.05, scale_grad=1e5, session=dev, dtype=torch.float32, seed=1)
s.sample_format = "view"                      # the reference copies every sample to the host; the roofline path does not
for _ in range(60): next(s)
best = 0.0
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): next(s)
    torch.cuda.synchronize(); best = max(best, 300 / (time.perf_counter() - t0))
print("RATE", best)
"""
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    rate = float([l for l in res.stdout.splitlines() if l.startswith("RATE")][-1].split()[1])
    best = max(_bench(["--gpus", "1", "--steps", "300", "--warmup", "20", "--no-update-only", "--no-cpu-baseline",
                       "--no-product-defaults"])["value"] for _ in range(2))
    # (two processes on one box differ by +-4 % with IDENTICAL GEMM picks -- tools/gpu/tune_noise_probe.py, round 6 -- hence 0.90)
    assert rate >= 0.90 * best, (rate, best)


@pytest.mark.timeout(900)
def test_bare_bnn_train_steps_near_the_bench_rate_with_nothing_called_first(gpu):
    """a bare ``BayesianNeuralNetwork(...).train()`` of the 10 M-parameter net -- NOTHING called first, no
    environment variable -- leaves a chain that steps at bench.py's `value` to the process-to-process spread (8 %): ``BNNCost`` picks the GEMM solutions of its
    device-bound plan by itself (``auto_gemm_tuning``, after warm evaluations); what is left is the runtime's default graph launch
    path (~2 %). The caller mirrored: pysgmcmc/models/bayesian_neural_network.py:464-468,510-512. Measured on the sampler ``train()``
    built and stepped (600 iterations), over 300 further steps, best of three (two ``train()`` calls timed against each other
    measure construction, capture and logging noise instead: +-50 %)."""
    code = """
import time, numpy as np, torch
from pysgmcmc_amd.models import BayesianNeuralNetwork
from pysgmcmc_amd.sampling import Sampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
import torch.cuda.tunable as tunable
assert not tunable.is_enabled() or not tunable.tuning_is_enabled()
rng = np.random.RandomState(0)
X, y = rng.randn(20000, 784).astype(np.float32), rng.randn(20000).astype(np.float32)
bnn = BayesianNeuralNetwork(sampling_method=Sampler.SGHMC, batch_size=256, stepsize_schedule=ConstantStepsizeSchedule(1e-3),
                            n_nets=100, n_iters=600, burn_in_steps=8, sample_steps=1000000, normalize_input=False,
                            normalize_output=False, seed=1, dtype=torch.float32, hidden=(2048, 2048, 2048))
bnn.train(X, y)
assert bnn.cost.gemm_tuning_applied == "auto", bnn.cost.gemm_tuning_applied
assert tunable.is_enabled() and not tunable.tuning_is_enabled()          # look-ups only after the plan's tuning evaluation
assert sum(p.numel() for p in bnn.network_params) == 10002434 and bnn.sampler.use_hip_graph is True and bnn.sampler.n_iterations == 600
best = 0.0
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): next(bnn.sampler)
    torch.cuda.synchronize(); best = max(best, 300 / (time.perf_counter() - t0))
print("RATE", best)
"""
    env = {k: v for k, v in os.environ.items() if k not in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "PYSGMCMC_AMD_AUTO_GEMM_TUNING",
                                                             "PYTORCH_TUNABLEOP_ENABLED", "PYTORCH_TUNABLEOP_TUNING")}
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=700, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    rate = float([l for l in res.stdout.splitlines() if l.startswith("RATE")][-1].split()[1])
    best = max(_bench(["--gpus", "1", "--steps", "300", "--warmup", "20", "--no-update-only", "--no-cpu-baseline",
                       "--no-product-defaults"])["value"] for _ in range(2))
    assert rate >= 0.90 * best, (rate, best)       # (process-to-process spread on one box: +-4 % each)

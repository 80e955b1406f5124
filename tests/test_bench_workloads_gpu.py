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
    assert "step_breakdown_us" in d and d["step_breakdown_us"]["gemm"] > 0

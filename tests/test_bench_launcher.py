"""bench.py's own rank launcher (``--gpus N`` without torch.distributed.run), as far as it can be checked without a GPU:
the parent spawns N fresh ranks with the rendezvous environment, and a failing rank's exit code comes back promptly."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_spawns_ranks_and_propagates_their_exit_code():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the GPU suite runs the real thing (tests/test_diagnostics_gpu.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--launch-timeout", "120"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    # no GPU here: every rank stops with "bench.py needs a HIP device" (exit code 1); the parent reports which rank and returns 1
    assert res.returncode == 1, (res.returncode, res.stderr[-1500:])
    assert res.stderr.count("bench.py needs a HIP device") >= 1 and "exited with code 1" in res.stderr
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 200


def test_launcher_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "--gpus 2 but WORLD_SIZE=3" in res.stderr

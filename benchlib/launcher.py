"""``bench.py --gpus N`` without a launcher: the parent (which never touches the GPU) starts the ranks itself."""
import os
import socket
import subprocess
import sys
import time


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



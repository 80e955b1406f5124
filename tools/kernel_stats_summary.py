"""Trim a rocprofv3 `--kernel-trace --stats` kernel_stats.csv to what is judged (dev tool):
every kernel of libsgmcmc_hip.so (anonymous-namespace kernels) plus the TOP other kernels by total time.
TunableOp's tuning launches (hundreds of GEMM candidates, at::cuda::flush_icache_kernel) are dropped.

  python3 tools/kernel_stats_summary.py gpurun_out/r02/prof_bench10m/b_kernel_stats.csv profiles/r02_bench10m_kernel_stats.csv [top] [min_calls]
"""
import csv
import sys

csv.field_size_limit(1 << 30)
src, dst = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rows = list(csv.DictReader(open(src)))
ours = [r for r in rows if "(anonymous namespace)::" in r["Name"] and "at::native" not in r["Name"]]
min_calls = int(sys.argv[4]) if len(sys.argv) > 4 else 100      # tuning candidates run ~12 times, step kernels >= steps
others = [r for r in rows if r not in ours and "flush_icache" not in r["Name"] and int(r["Calls"]) >= min_calls]
others.sort(key=lambda r: -float(r["TotalDurationNs"]))
keep = ours + others[:top]
keep.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(dst, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in keep:
        name = r["Name"].replace("(anonymous namespace)::", "")
        name = name.split(">(")[0] + ">" if name.startswith("void stream_quads") else name[:150]
        w.writerow([name, r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]), r["MinNs"], r["MaxNs"],
                    "%.1f" % float(r["StdDev"])])
print("%s: %d of %d rows kept" % (dst, len(keep), len(rows)))

# Round 6: the batched weight gradients of the step on the library's fp32 products (off) and on the bf16 planes (on), same box:
# `value` of plain runs, then per-kernel durations under rocprofv3 (through gpurun: bash tools/gpu/r06_gw_planes.sh)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gw_planes; rm -rf $O; mkdir -p $O
for wl in bnn50m-sgld bnn50m-rsghmc bnn10m-sghmc; do
  for mode in off on off on; do
    BENCH_GW_PLANES=$mode python3 bench.py --workload $wl --steps 200 --warmup 20 --no-update-only --no-cpu-baseline --no-product-defaults > $O/$wl.$mode.json 2> $O/$wl.$mode.err
    python3 -c "
import json; d=json.load(open('$O/$wl.$mode.json')); print('$wl  batched gW on bf16 planes = %-3s: %8.1f samples/s  %7.1f us per step (median %7.1f)' % ('$mode', d['value'], d['ms_per_step'] * 1e3, d['step_ms_median'] * 1e3))"
  done
done
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for mode in off on; do
  BENCH_GW_PLANES=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$mode -o s -- python3 bench.py --workload bnn50m-sgld --steps 100 --warmup 10 --no-update-only --no-cpu-baseline --no-product-defaults > $O/prof_$mode.json 2> $O/prof_$mode.err
  echo "== bnn50m-sgld under rocprofv3, batched gW on bf16 planes = $mode: kernels of the step by total time"
  python3 - <<PY
import csv
csv.field_size_limit(1 << 30)
rows=[r for r in csv.DictReader(open("$O/prof_$mode/s_kernel_stats.csv")) if 100 <= int(r["Calls"]) <= 3000]
rows.sort(key=lambda r:-int(r["TotalDurationNs"]))
for r in rows[:12]:
    print("%-86s calls %6s avg %9.1f us" % (r["Name"].replace("(anonymous namespace)::", "")[:86], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete

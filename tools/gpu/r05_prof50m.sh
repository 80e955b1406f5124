# per-kernel durations of the 49.8 M-parameter SGLD chain's step with the hidden layers on the fused launches and on the library
# (through gpurun: bash tools/gpu/r05_prof50m.sh)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
O=gpurun_out/prof50m; rm -rf $O; mkdir -p $O
for mode in all library; do
  BENCH_FUSED_LAYERS=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o s -- python3 bench.py --workload bnn50m-sgld --steps 100 --warmup 10 --no-update-only --no-cpu-baseline --no-product-defaults > $O/$mode.json 2> $O/$mode.err
  echo "== $mode"; python3 -c "
import json,sys; d=json.load(open('$O/$mode.json')); print(d['value'], d['ms_per_step'])"

  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/$mode/s_kernel_stats.csv")))
rows.sort(key=lambda r:-int(r["TotalDurationNs"]))
for r in rows[:18]:
    print("%-70s calls %6s avg %9.1f us total %9.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, int(r["TotalDurationNs"])/1e6))
PY
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete

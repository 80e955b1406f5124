# K3 (relativistic step) against K1 / K2 frozen, timings + SQ counters. Through gpurun: bash tools/gpu/k3_counters.sh
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/k3; rm -rf $O; mkdir -p $O
for N in 49826818 10002434; do
python3 tools/k3_probe.py $N 6 2>&1 | grep -v amdgpu.ids > $O/timings_$N.txt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq_$N -o q -- python3 tools/k3_probe.py $N 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/sq2_$N -o q -- python3 tools/k3_probe.py $N 3 > /dev/null 2>&1
done
find $O -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, collections
for N in (49826818, 10002434):
    for d in ("sq", "sq2"):
        f = glob.glob("gpurun_out/k3/%s_%d/**/*counter_collection.csv" % (d, N), recursive=True)
        if not f: continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            k = r["Kernel_Name"]
            if "stream_quads" not in k: continue
            agg[k[:110] + " grid=" + r.get("Grid_Size", "?") + " wg=" + r.get("Workgroup_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("== n=%d (%s): per launch, mean over the launches of the run" % (N, d))
        for k, c in agg.items():
            print(k); print("    " + "  ".join("%s=%.4g" % (n_, sum(v) / len(v)) for n_, v in sorted(c.items())))
PY
cat $O/timings_*.txt

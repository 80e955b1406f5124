# round 6 gate record: SQ counters of the weight-gradient kernels (through gpurun: bash tools/gpu/gw_counters.sh), SHAPE=0 (default) = two batched 2048 x 2048 products, SHAPE=3 = two 4864 x 4864
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gw_counters; rm -rf $O; mkdir -p $O
tools/gpu/bnn_gw_bf16x3 ${SHAPE:-0} > $O/run.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- tools/gpu/bnn_gw_bf16x3 ${SHAPE:-0} > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o q -- tools/gpu/bnn_gw_bf16x3 ${SHAPE:-0} > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2 -o q -- tools/gpu/bnn_gw_bf16x3 ${SHAPE:-0} > /dev/null 2>&1
find $O -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, collections
for d in ("sq", "sq2"):
    f = glob.glob("gpurun_out/gw_counters/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f:
        print("no counter file for", d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    order = []
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "gw_bf16x3" not in k and "Cijk" not in k: continue
        k = k[:75]
        if k not in order: order.append(k)
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== %s: mean per launch (launches 2 .. 50 of each kernel)" % d)
    for k in order:
        c = agg[k]
        print(k); print("    " + "  ".join("%s=%.4g" % (n_, sum(v[2:50]) / max(1, len(v[2:50]))) for n_, v in sorted(c.items())) + "  (launches %d)" % len(next(iter(c.values()))))
PY
grep -E "gw_bf16x3|Cijk" $O/stats/s_kernel_stats.csv | cut -d, -f1-4 | sed -e 's/Cijk_[A-Za-z0-9_]*/Cijk(library)/' | cut -c1-200
cat $O/run.txt | grep -E "PERSIST|time per|probes"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete

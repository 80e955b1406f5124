# SQ counters of the product's fused forward layer against the library product (through gpurun: bash tools/gpu/fwd_kernel_counters.sh)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fwdk; rm -rf $O; mkdir -p $O
python3 tools/fwd_kernel_counters_probe.py tune > $O/tune.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 tools/fwd_kernel_counters_probe.py > $O/run.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o q -- python3 tools/fwd_kernel_counters_probe.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -o q -- python3 tools/fwd_kernel_counters_probe.py > /dev/null 2>&1
find $O -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, collections
keep = ("bnn_dense_tanh_kernel", "bias_tanh_kernel", "Cijk_")
for d in ("sq", "sq2"):
    f = glob.glob("gpurun_out/fwdk/%s/**/*counter_collection.csv" % d, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if not any(x in k for x in keep): continue
        agg[k[:60] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in [k for k in agg if "bnn_dense_tanh" in k]:                 # 8 launches at K = 2048, then 8 at K = 784 (dispatch order)
        c = agg.pop(k)
        agg[k + " K=2048"] = {n_: v[:len(v) // 2] for n_, v in c.items()}
        agg[k + " K=784"] = {n_: v[len(v) // 2:] for n_, v in c.items()}
    print("== %s: per launch, mean over the launches of the run (first two dropped)" % d)
    for k, c in agg.items():
        print(k); print("    " + "  ".join("%s=%.4g" % (n_, sum(v[2:]) / len(v[2:])) for n_, v in sorted(c.items())))
PY
grep "bnn_dense_tanh\|bias_tanh\|Cijk" $O/stats/s_kernel_stats.csv | cut -c1-200; cat $O/run.txt | grep "K="
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete

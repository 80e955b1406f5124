cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gemm_pmc; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a -o s -- python3 tools/experiments/gemm_pmc_probe.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o s -- python3 tools/experiments/gemm_pmc_probe.py > $O/t.log 2>&1
python3 - <<'PY'
import csv, collections, glob
csv.field_size_limit(1<<30)
f=glob.glob('gpurun_out/gemm_pmc/a/**/*counter_collection.csv', recursive=True)[0]
per=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name=r['Kernel_Name']
    if 'gemm_tn' in name or 'Cijk' in name:
        key=name[:90]
        per[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in per.items():
    print(k)
    print('   ', {c: round(sum(x)/len(x)) for c,x in v.items()})
f=glob.glob('gpurun_out/gemm_pmc/t/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'gemm_tn' in r['Name'] or 'Cijk' in r['Name']:
        print(r['Name'][:80], r['Calls'], r['AverageNs'])
PY

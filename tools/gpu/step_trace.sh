# Per-dispatch timeline of the 10 M-parameter chain's step (where the gaps between the 10 launches are). Through gpurun: bash tools/gpu/step_trace.sh
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
# what bench.py asks the HIP runtime for on this workload (pysgmcmc_amd.configure_for_device_bound_chains) -- exported here because under
# rocprofv3 the runtime initialises before python runs
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
export BENCH_PRIME_STEADY=124      # the step selection below counts update launches from the start of the run
# STEPTRACE_ARGS: further bench.py arguments (e.g. "--dtype f64": the f64 chain's step), STEPTRACE_OUT: output directory
O=${STEPTRACE_OUT:-gpurun_out/steptrace}; rm -rf $O; mkdir -p $O
export STEPTRACE_DIR=$O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-update-only --no-product-defaults $STEPTRACE_ARGS > $O/line.json 2> $O/err.txt
python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob(os.environ['STEPTRACE_DIR'] + '/t/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# keep the last 3000 dispatches (the timed region and the legs after it are at the end; the legs are few)
names = [r['Kernel_Name'] for r in rows]
# find steps: a step ENDS with its update launch (stream_quads_vec; since round 6 the window gather rides in it), so step k is
# everything after update k - 1 up to and including update k
ends = [i for i, n in enumerate(names) if 'stream_quads' in n]
idx = [e + 1 for e in ends]
# use steps from the middle of the timed region: updates number 200..260 (prime 136 + warmup 20 + timed 100)
sel = idx[180:240]
short = lambda n: ('K1' if 'stream_quads' in n else 'gather' if 'window_gather' in n else 'head' if 'head_last' in n else 'fwd' if 'kernelILi4ELb0' in n or '<4, false' in n else 'bwd' if 'kernelILi4ELb1' in n or '<4, true' in n else 'gemm' if n.startswith('Cijk') else n.replace('void (anonymous namespace)::', '').split('<')[0].split('(')[0][:28])
# launches per step: the commonest count (f32, fused layers: 9 = 3 forward, head, 2 backward, batched gW, gW_0, update [+ the next window])
per_step = collections.Counter(b - a for a, b in zip(sel[:-1], sel[1:])).most_common(1)[0][0]
gaps = collections.defaultdict(list); durs = collections.defaultdict(list); steps = []
for a, b in zip(sel[:-1], sel[1:]):
    seq = rows[a:b]
    if len(seq) != per_step: continue
    steps.append((int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e3)
    nxt = rows[b]
    for k, r in enumerate(seq):
        nm = '%02d %s' % (k, short(r['Kernel_Name']))
        durs[nm].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        after = seq[k + 1] if k + 1 < len(seq) else nxt
        gaps[nm].append((int(after['Start_Timestamp']) - int(r['End_Timestamp'])) / 1e3)
import statistics as st
print('launches per step', per_step, '| steps analysed', len(steps), '| median first-start -> last-end us', round(st.median(steps), 1))
tot_d = tot_g = 0
for nm in sorted(durs):
    d, g = st.median(durs[nm]), st.median(gaps[nm])
    tot_d += d; tot_g += g
    print('%-32s dur %7.2f us   gap to the next launch %6.2f us' % (nm, d, g))
print('sum of durations %.1f  sum of gaps %.1f' % (tot_d, tot_g))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete

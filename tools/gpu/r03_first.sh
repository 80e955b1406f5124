#!/bin/bash
# round 3, first GPU session: the refactored library (ABI v3) under the whole GPU suite, then bench with and without overlap
set -x
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_first.txt 2>&1
tail -5 gpurun_out/r03/pytest_gpu_first.txt
python bench.py --steps 200 --warmup 20 --overlap off --no-cpu-baseline > gpurun_out/r03/bench_overlap_off.json 2> gpurun_out/r03/bench_overlap_off.err
python bench.py --steps 200 --warmup 20 --overlap on --no-cpu-baseline > gpurun_out/r03/bench_overlap_on.json 2> gpurun_out/r03/bench_overlap_on.err
python bench.py --steps 20 --warmup 5 --overlap on --no-cpu-baseline --no-update-only > gpurun_out/r03/bench_overlap_on_20.json 2> gpurun_out/r03/bench_overlap_on_20.err
for f in gpurun_out/r03/*.err; do tail -n 3 $f; done
python - <<'PY'
import json
for f in ("bench_overlap_off", "bench_overlap_on", "bench_overlap_on_20"):
    try:
        d = json.loads(open("gpurun_out/r03/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["step_ms_median"], d["roofline"]["frac"], d["roofline"].get("slices"),
              d.get("roofline_unoverlapped", {}).get("frac"), d.get("step_breakdown_us"), d["host_enqueue_ms"])
    except Exception as e:
        print(f, "FAILED", e)
PY

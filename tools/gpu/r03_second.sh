#!/bin/bash
set -x
mkdir -p gpurun_out/r03
python -m pytest tests/test_step_opts_gpu.py tests/test_svgd_gpu.py tests/test_diagnostics_gpu.py tests/test_samplers_gpu.py -m gpu -q > gpurun_out/r03/pytest_gpu_second.txt 2>&1
tail -5 gpurun_out/r03/pytest_gpu_second.txt
python bench.py --steps 200 --warmup 20 --overlap on --no-cpu-baseline > gpurun_out/r03/bench_overlap_on.json 2> gpurun_out/r03/bench_overlap_on.err
python bench.py --steps 20 --warmup 5 --overlap on --no-cpu-baseline --no-update-only > gpurun_out/r03/bench_overlap_on_20.json 2> gpurun_out/r03/bench_overlap_on_20.err
python bench.py --steps 20 --warmup 5 --overlap off --no-cpu-baseline --no-update-only > gpurun_out/r03/bench_overlap_off_20.json 2> gpurun_out/r03/bench_overlap_off_20.err
for f in gpurun_out/r03/*.err; do tail -n 3 $f; done
python - <<'PY'
import json
for f in ("bench_overlap_on", "bench_overlap_on_20", "bench_overlap_off_20"):
    try:
        d = json.loads(open("gpurun_out/r03/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["step_ms_median"], d["roofline"]["frac"], d["roofline"].get("slices"),
              d.get("roofline_unoverlapped", {}).get("frac"), d.get("step_breakdown_us"), d["host_enqueue_ms"])
    except Exception as e:
        print(f, "FAILED", e)
PY
